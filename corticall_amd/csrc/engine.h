// TraversalEngine on the device: shared device-side primitives (vertex + neighbourhood through the probe
// rows' neighbour index, per-walk visited table, per-walk LinkStore, cursor step) used by the walk and
// cursor kernels.  Every function cites the Java it restates (J/ = public/java/src/uk/ac/ox/well/cortexjdk/).
//
// The hot loop never touches a k-mer: a vertex is (record index, orientation).  Its successors come from
// the neighbour index of its probe row (memoised findRecord), its orientation quirks (palindrome, the Q6
// hash collision) from two per-record flag bits computed at load time.  K-mers are materialised from the
// row key only where the reference needs the string: link-table lookups, vertices without a record,
// results.  One traversal step = one 64-byte row read + one visited-table access.
#pragma once
#include "graph.h"
#include "links.h"

namespace ldbg {

// the records in unitig order for this engine's colour masks (runs.h); uinfo == nullptr: no index
struct RunIndexView { const uint64_t* uinfo; const uint32_t* uo; const uint8_t* ubase; };

struct EngineView {
    GraphView g;
    uint32_t trav_mask, recruit_mask, join_mask;
    uint32_t trav_sel4, recruit_sel4;   // the same masks for colours 0..3 as byte selectors (0xFF per selected colour) over packed edge bytes
    int first_trav;
    int stopper, max_len, connect_all, strict_flip;
    int lean_rows;        // odd k and the packed row layout (row_is_packed): the lean step's row reads carry no layout tests
    int cursor_on;        // !ec.getLinks().isEmpty(): dfs drives the cursor (TraversalEngine.java:363, 379) even
                          // when none of the configured link sets belongs to a traversal sample
    uint32_t link_flag_mask;   // probe-row link-flag bits of the link sets merged into `links`
    LinksView links;      // the traversal's link sets merged into one table (links.h)
    RunIndexView runs;    // set for walk launches only (walk.cpp); the cursor and dfs kernels step k-mer by k-mer
};

// strand status codes (per seed and direction)
enum : uint32_t {
    ST_OK = 0,
    ST_NULLPTR = 1,        // the reference throws NullPointerException here
    ST_LINKSTORE_FULL = 2, // per-walk link store capacity exceeded -> host retries with a larger store
    ST_BRANCH_NULL = 3,    // dfs branch returned null
    ST_COPY_OVERFLOW = 4,  // more than 32767 copies of one vertex
    ST_POOL_FULL = 5,      // path block / table pool exhausted -> host splits the batch
    ST_STOPPER_CONFIG = 6, // the stopping rule needs a ROI graph that was not configured (CortexJDKException in the reference)
    ST_DEPTH_OVERFLOW = 7, // dfs recursion deeper than the frame stack -> host retries with a deeper one
    ST_TABLE_FULL = 8,     // a strand's visited table reached its maximum size
    ST_LOG_FULL = 9,       // a strand's dfs log outgrew its block table -> host retries with a longer one
    ST_RETRY_PLAIN = 10,   // the strand met a case the run steps (runstep.h) leave to the k-mer-by-k-mer code -> the host
                           // walks it again without the run index
    ST_MERGE_UNSUPPORTED = 11   // dfs.cpp: merged_vertices met a vertex without a record
};

// ---- path entry: one vertex of a branch, 8 bytes
//  bits 0..32  record index + 1 (0 = null CortexRecord)
//  bit  33     flip: vertex k-mer is the reverse complement of the record's (canonical) k-mer
//  bits 34..35 the base this vertex appended to the contig (last base going forward, first base going back)
//  bits 36..59 copyIndex, 24-bit two's complement (CortexVertex.copyIndex)
//  bit  60     the vertex is a Q6 vertex (reverse complemented but isFlipped() == false): its neighbours do not
//              overlap it by k-1 bases, so contigs through it are spelled k-mer by k-mer (toContig, literally)
LDBG_HOSTDEV uint64_t path_pack(int64_t idx, bool flip, unsigned base, int copy, bool quirk = false) {
    return (uint64_t)(idx + 1) | ((uint64_t)(flip ? 1 : 0) << 33) | ((uint64_t)(base & 3u) << 34) |
           (((uint64_t)(uint32_t)copy & 0xFFFFFFull) << 36) | ((uint64_t)(quirk ? 1 : 0) << 60);
}
LDBG_HOSTDEV bool path_quirk(uint64_t e) { return (e >> 60) & 1ull; }
LDBG_HOSTDEV int64_t path_idx(uint64_t e) { return (int64_t)(e & 0x1FFFFFFFFull) - 1; }
LDBG_HOSTDEV bool path_flip(uint64_t e) { return (e >> 33) & 1ull; }
LDBG_HOSTDEV unsigned path_base(uint64_t e) { return (unsigned)((e >> 34) & 3ull); }
LDBG_HOSTDEV int path_copy(uint64_t e) { int32_t v = (int32_t)((e >> 36) & 0xFFFFFFull); return (v << 8) >> 8; }

LDBG_HOSTDEV int popc4(uint32_t m) { return (int)((m & 1u) + ((m >> 1) & 1u) + ((m >> 2) & 1u) + ((m >> 3) & 1u)); }
LDBG_HOSTDEV unsigned lowbit4(uint32_t m) { return (m & 1u) ? 0u : ((m & 2u) ? 1u : ((m & 4u) ? 2u : 3u)); }

// ---- a vertex with its neighbourhood: CortexVertex + TraversalUtils.getAllNextKmers/getAllPrevKmers
// (J/utils/traversal/TraversalUtils.java:510-590) + TraversalEngine.getNextVertices/getPrevVertices
// (J/utils/traversal/TraversalEngine.java:147-239) folded into 4-bit base masks.
struct Node {
    int32_t idx;         // record index, -1 = null CortexRecord
    int32_t copy;        // CortexVertex.copyIndex
    uint32_t vslot;      // slot of this vertex in the walk's visited table (valid once located; idx >= 0)
    uint64_t vent;       // the table entry at vslot, kept in step with every write (node_sync): the hot loop never
                         // reads an entry back that it already holds
    uint8_t flip;        // vertex k-mer != canonical orientation (by comparison): part of the vertex identity
    uint8_t fj;          // CanonicalKmer.isFlipped() — by Arrays.hashCode inequality (quirk Q6)
    uint8_t npe;         // record missing while recruitment colours are set (Q14)
    uint8_t lflags;      // link-flag bits of the record's probe row
    uint8_t next_mask;   // bit b: successor o[1:]+b
    uint8_t prev_mask;   // bit b: predecessor b+o[:-1]
    uint8_t base;        // base that was appended to reach this vertex (travel direction)
    uint8_t e1;          // ent1 is valid
#ifdef LDBG_LEAN_PROFILE
    unsigned long long p1, p2;   // (experiment build only) clock after the row loads were issued / after they returned
#endif
    uint32_t ent1;       // neighbour-index entry of the vertex's only neighbour in the direction it was reached in (it comes
                         // with the row, so the next step starts without a load)
    uint64_t ui;         // run-index entry of the record (runs.h), 0 = none; read with the row where the engine has an index
};

// edges of the node's record -> neighbour masks; link flags; Java flip from the record's collision bit
// `edges4`: the edge bytes of colours 0..3 packed into one word (byte c = colour c), `more`: the row's edge bytes for c >= 4
// FEW: the graph has at most 4 colours (known where the call is compiled), `more` is not read
template <bool FEW = false>
LDBG_HOSTDEV void node_fill_bytes(const EngineView& e, Node& n, uint32_t edges4, const uint8_t* more, uint8_t fl) {
    const GraphView& g = e.g;
    n.npe = 0; n.fj = n.flip;
    n.lflags = fl & LDBG_ROW_LINK_BITS;
    // isFlipped() is false for a reverse-complemented k-mer whose two orientations hash alike (Q6)
    if (e.strict_flip && (fl & LDBG_ROW_HASH_COLLISION)) n.fj = 0;
    const bool fj = n.fj != 0;
    // Every colour's edge byte goes through the same nibble selection, so the union over the selected colours can be taken on
    // the bytes first: OR of the traversal (recruitment) colours' bytes, then one nibble split.
    // CortexRecord.getOutEdgesAsBytes: bit i <-> base i ; getInEdgesAsBytes: bit (3-i) <-> base i ; complement=true relabels
    // base b as 3-b (CortexRecord.java:214-275)
    uint32_t tb = edges4 & e.trav_sel4, rb = edges4 & e.recruit_sel4;
    tb |= tb >> 16; tb |= tb >> 8; rb |= rb >> 16; rb |= rb >> 8;
    if (!FEW)
        for (int col = 4; col < g.C; col++) {
            if ((e.trav_mask >> col) & 1u) tb |= more[col];
            if ((e.recruit_mask >> col) & 1u) rb |= more[col];
        }
    tb &= 0xffu; rb &= 0xffu;
    const uint32_t tfw = !fj ? tb & 0xf : tb >> 4, trn = !fj ? tb >> 4 : tb & 0xf;      // successor base = bit position
    const uint32_t rfw = !fj ? rb & 0xf : rb >> 4, rrn = !fj ? rb >> 4 : rb & 0xf;      // predecessor base = 3 - bit position
    const uint32_t tf = tfw, rf = rfw;
    const uint32_t tr = ((trn & 1u) << 3) | ((trn & 2u) << 1) | ((trn & 4u) >> 1) | ((trn & 8u) >> 3);
    const uint32_t rr = ((rrn & 1u) << 3) | ((rrn & 2u) << 1) | ((rrn & 4u) >> 1) | ((rrn & 8u) >> 3);
    n.next_mask = (uint8_t)(tf ? tf : rf);    // recruitment colours only where the traversal colours give nothing
    n.prev_mask = (uint8_t)(tr ? tr : rr);
}
LDBG_HOSTDEV void node_fill(const EngineView& e, Node& n) {
    const GraphView& g = e.g;
    n.ui = e.runs.uinfo && n.idx >= 0 ? e.runs.uinfo[n.idx] : 0ull;
    if (n.idx >= 0) {
        const uint8_t* row = graph_row(g, n.idx);
        const uint8_t* ed = row + g.edges_off;
        uint32_t edges4 = 0;
        for (int col = 0; col < g.C && col < 4; col++) edges4 |= (uint32_t)ed[col] << (8 * col);
        node_fill_bytes(e, n, edges4, ed, row[g.flags_off]);
    } else {
        n.npe = e.recruit_mask != 0 ? 1 : 0; n.lflags = 0; n.fj = n.flip;
        n.next_mask = n.prev_mask = 0;
    }
}

// neighbour `base` of vertex p in travel direction `fwd`, through the neighbour index of p's probe row
// (memoised findRecord, graph.h).  p must have a record.
LDBG_HOSTDEV void node_child(const EngineView& e, const Node& p, bool fwd, unsigned base, Node& n) {
    const GraphView& g = e.g;
    // p's orientation for neighbour generation is its Java flip: o = fj ? rc(canon) : canon.
    //   fwd, !fj: next(c, b)            -> succ[b]         fwd, fj: next(rc(c), b) = rc(prev(c, 3-b)) -> pred[3-b], toggled
    //   rev, !fj: prev(c, b)            -> pred[b]         rev, fj: prev(rc(c), b) = rc(next(c, 3-b)) -> succ[3-b], toggled
    const bool fj = p.fj != 0;
    const unsigned j = fwd ? (!fj ? base : 4u + (3u - base)) : (!fj ? 4u + base : (3u - base));
    const uint32_t ent = graph_nbr(g, p.idx, (int)j);
    n.idx = (int32_t)(ent & 0x7FFFFFFFu) - 1;
    n.copy = 0; n.vslot = 0; n.vent = 0; n.base = (uint8_t)base; n.e1 = 0; n.ent1 = 0;
    bool flip = (((ent >> 31) & 1u) != 0) != fj;
    if (n.idx >= 0 && (graph_row(g, n.idx)[g.flags_off] & LDBG_ROW_PALINDROME)) flip = false;   // rc(x) == x
    n.flip = flip ? 1 : 0;      // (for a vertex without a record the flag is not part of any comparison)
    node_fill(e, n);
}

// ---- k-mers, materialised only where the reference needs the string
template <int W>
LDBG_HOSTDEV Kmer<W> node_kmer(const EngineView& e, const Node& n) {           // requires n.idx >= 0
    Kmer<W> c = graph_key<W>(e.g, n.idx);
    return n.flip ? kmer_revcomp<W>(c, e.g.k) : c;
}
template <int W>
LDBG_HOSTDEV Kmer<W> node_o(const EngineView& e, const Node& n) {              // orientation neighbours are built from
    Kmer<W> c = graph_key<W>(e.g, n.idx);
    return n.fj ? kmer_revcomp<W>(c, e.g.k) : c;
}
template <int W>
LDBG_HOSTDEV Kmer<W> child_kmer(const EngineView& e, const Node& p, bool fwd, unsigned base) {
    Kmer<W> o = node_o<W>(e, p);
    return fwd ? kmer_next<W>(o, e.g.k, base) : kmer_prev<W>(o, e.g.k, base);
}
// findRecord(sk) for an arbitrary k-mer (seeds): radix index + probe rows
template <int W>
LDBG_HOSTDEV void node_find(const EngineView& e, const Kmer<W>& sk, Node& n) {
    bool fc;
    Kmer<W> c = kmer_canonical<W>(sk, e.g.k, &fc);
    n.idx = (int32_t)graph_find_canonical<W>(e.g, c);
    n.copy = 0; n.vslot = 0; n.vent = 0; n.base = 0; n.e1 = 0; n.ent1 = 0;
    n.flip = fc ? 1 : 0;
    node_fill(e, n);
    if (n.idx < 0 && e.strict_flip && fc)   // no record to carry the collision bit: hash the strings (rare path)
        n.fj = kmer_java_hash<W>(c, e.g.k) != kmer_java_hash<W>(sk, e.g.k) ? 1 : 0;
}
LDBG_HOSTDEV void node_null(const EngineView& e, Node& n) {   // not a k-mer (non-ACGT): findRecord misses (Q4)
    n.idx = -1; n.copy = 0; n.vslot = 0; n.vent = 0; n.e1 = 0; n.ent1 = 0; n.flip = 0; n.fj = 0; n.lflags = 0; n.base = 0;
    n.next_mask = n.prev_mask = 0; n.ui = 0;
    n.npe = e.recruit_mask != 0 ? 1 : 0;
}

// ---- per-walk visited table (HashSet<CortexVertex> visited, TraversalEngine.java:360-425, plus the
// cursor's `seen` set :27,262-265): open addressing over 8-byte entries in HBM.  A table is carved out of a
// zeroed pool when a strand starts (4096 entries) and regrown x4 when half full, so memory follows the
// walk lengths.  A vertex is located once (when it is first looked up as a neighbour); later updates go
// straight to its slot.
//  entry: bits 0..33 key = ((record index + 1) << 1) | flip (never 0) ; bits 34..47 seen epoch ; bits 48..62 copies visited
//  `seen` is cleared by every seek() (TraversalEngine.java:333): a k-mer is in the cursor's seen set when its entry
//  carries the cursor's current epoch (1..16383; a dfs branch = a new epoch, dfs.cpp sweeps the table on wrap-around)
struct VisitedTable {
    uint64_t* tab;
    uint32_t mask;   // capacity - 1
    uint32_t used;   // claimed slots
};
#define LDBG_VT_KEY_MASK 0x3FFFFFFFFull
LDBG_HOSTDEV uint32_t vt_hash(uint64_t key) {      // keys are 34 bits; the table index is taken from the LOW bits of this
    uint32_t x = (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x9E3779B1u);
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;      // 32-bit finaliser: every input bit reaches the low bits
    return x;
}
LDBG_HOSTDEV uint64_t vt_key(int64_t idx, bool flip) { return ((uint64_t)(idx + 1) << 1) | (flip ? 1ull : 0ull); }
// slot of (idx, flip); claims a free slot (count 0, not seen) if the vertex is not in the table yet
#ifdef LDBG_HOSTSIM
struct LsDebug { uint64_t adds = 0, newkeys = 0, choices = 0, scan = 0, maxn = 0, steps = 0, sum_n = 0, runs_a = 0, runs_b = 0, run_vertices = 0, retries = 0, repeats = 0; };
inline LsDebug& ls_debug() { static LsDebug d; return d; }
#endif
// Linear probing, four slots per round: the entries at h .. h+3 are read together (one trip to memory; they share a cache
// line or two), and the first of them that is free or holds the key decides.  A wavefront waits for the longest probe chain
// among its 64 strands at every step, so what counts is the number of ROUNDS, not of entries read.
struct VtPeek { uint64_t e[4]; };
LDBG_HOSTDEV VtPeek vt_peek(const VisitedTable& t, uint32_t h) {
    VtPeek p;
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) p.e[q] = LDBG_GLOBAL(const uint64_t, t.tab)[(h + q) & t.mask];
    return p;
}
// probe from slot h, whose round `pk` has already been read; returns the slot and its entry
LDBG_HOSTDEV uint32_t vt_probe_from(VisitedTable& t, uint64_t key, uint32_t h, VtPeek pk, uint64_t* ent) {
    while (true) {
#ifdef LDBG_HOSTSIM
        ls_debug().scan++;
#endif
        uint32_t q = 4;
        uint64_t ev = 0;
#pragma unroll
        for (int i = 3; i >= 0; i--) {
            const bool stop = pk.e[i] == 0 || (pk.e[i] & LDBG_VT_KEY_MASK) == key;
            q = stop ? (uint32_t)i : q;
            ev = stop ? pk.e[i] : ev;
        }
        if (q < 4) {
            const uint32_t slot = (h + q) & t.mask;
            if (ev == 0) { LDBG_GLOBAL(uint64_t, t.tab)[slot] = key; t.used++; ev = key; }
            *ent = ev;
            return slot;
        }
        h = (h + 4) & t.mask;
        pk = vt_peek(t, h);
    }
}
LDBG_HOSTDEV uint32_t vt_locate(VisitedTable& t, int64_t idx, bool flip, uint64_t* ent) {
    const uint64_t key = vt_key(idx, flip);
    const uint32_t h = vt_hash(key) & t.mask;
    return vt_probe_from(t, key, h, vt_peek(t, h), ent);
}
LDBG_HOSTDEV int vt_count_e(uint64_t e) { return (int)((e >> 48) & 0x7FFFull); }
#define LDBG_VT_EPOCH_MAX 0x3FFFu
LDBG_HOSTDEV bool vt_seen_e(uint64_t e, uint32_t epoch) { return (uint32_t)((e >> 34) & LDBG_VT_EPOCH_MAX) == epoch; }
LDBG_HOSTDEV uint64_t vt_with_seen(uint64_t e, uint32_t epoch) { return (e & ~((uint64_t)LDBG_VT_EPOCH_MAX << 34)) | ((uint64_t)epoch << 34); }
LDBG_HOSTDEV uint64_t vt_with_count(uint64_t e, int c) { return (e & ~(0x7FFFull << 48)) | ((uint64_t)(c & 0x7FFF) << 48); }
LDBG_HOSTDEV void node_locate(VisitedTable& t, Node& n) { if (n.idx >= 0) n.vslot = vt_locate(t, n.idx, n.flip != 0, &n.vent); }
LDBG_HOSTDEV int node_count(const Node& n) { return n.idx >= 0 ? vt_count_e(n.vent) : 0; }
// a write to the table: through the node that holds the slot ...
LDBG_HOSTDEV void node_store(VisitedTable& t, Node& n, uint64_t val) { LDBG_GLOBAL(uint64_t, t.tab)[n.vslot] = val; n.vent = val; }
// ... and into every other live node that refers to the same vertex (a walk can stand on a k-mer and look at it)
LDBG_HOSTDEV void node_sync(Node& n, const Node& written) { if (n.idx >= 0 && written.idx >= 0 && n.vslot == written.vslot) n.vent = written.vent; }
// neighbour `base` of p, with its table slot.  The first probe of the table is issued before the neighbour's row is
// read (for odd k the orientation, hence the key, is known from p's neighbour index alone), so the two accesses overlap.
LDBG_HOSTDEV uint32_t node_child_entry(const EngineView& e, const Node& p, bool fwd, unsigned base) {
    const bool fj = p.fj != 0;
    const unsigned j = fwd ? (!fj ? base : 4u + (3u - base)) : (!fj ? 4u + base : (3u - base));
    return graph_nbr(e.g, p.idx, (int)j);
}
LDBG_HOSTDEV unsigned nbr_slot(bool fj, bool fwd, unsigned base) {
    return fwd ? (!fj ? base : 4u + (3u - base)) : (!fj ? 4u + base : (3u - base));
}
// LEAN: the caller has established that the row layout is the packed one (row_is_packed), k is odd and the entry names a
// record, so none of that is tested here (the lean step, lscoop.h, whose run time is its instruction count)
LDBG_HOSTDEV bool row_is_packed(const GraphView& g) { return g.C <= 3 && (g.edges_off & 3) == 0 && (g.nbr_off & 15) == 0 && (g.stride & 15) == 0; }
template <bool LEAN = false>
LDBG_HOSTDEV void node_from_entry(const EngineView& e, VisitedTable& t, const Node& p, uint32_t ent, unsigned base, bool fwd, Node& n) {
    const GraphView& g = e.g;
    const bool fj = p.fj != 0;
    n.idx = (int32_t)(ent & 0x7FFFFFFFu) - 1;
    n.copy = 0; n.vslot = 0; n.vent = 0; n.base = (uint8_t)base; n.e1 = 0; n.ent1 = 0;
    bool flip = (((ent >> 31) & 1u) != 0) != fj;
    const bool rec = LEAN || n.idx >= 0;
    const bool early = LEAN || (rec && (g.k & 1));       // odd k: no palindromes
    uint64_t key = 0;
    VtPeek e0 = {{0, 0, 0, 0}};
    uint32_t h = 0;
    if (early) { key = vt_key(n.idx, flip); h = vt_hash(key) & t.mask; e0 = vt_peek(t, h); }
    // The row is read in three wide loads (the table lays it out for that: W x u64 key | C edge bytes | flag byte | ... | 8 x u32
    // neighbour index at a 16-byte boundary): one word with the edge bytes and the flag byte (C <= 3), and the whole neighbour
    // index, so whichever entry the next step needs is already here.
    uint32_t nb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t ef = 0;
    const bool packed = LEAN || row_is_packed(g);
    uint64_t ui = 0;
    if (rec && e.runs.uinfo) ui = LDBG_GLOBAL(const uint64_t, e.runs.uinfo)[n.idx];
    if (rec) {
        const uint8_t* row = graph_row(g, n.idx);
        if (packed) {
            struct alignas(16) U4 { uint32_t x, y, z, w; };
            const U4 a = *(const U4*)(row + g.nbr_off), b = *(const U4*)(row + g.nbr_off + 16);
            ef = *(const uint32_t*)(row + g.edges_off);
            nb[0] = a.x; nb[1] = a.y; nb[2] = a.z; nb[3] = a.w; nb[4] = b.x; nb[5] = b.y; nb[6] = b.z; nb[7] = b.w;
        } else {
            const uint32_t* nbp = (const uint32_t*)(row + g.nbr_off);
#pragma unroll
            for (int q = 0; q < 8; q++) nb[q] = nbp[q];
        }
    }
    if (!LEAN && rec && !(g.k & 1) && (graph_row(g, n.idx)[g.flags_off] & LDBG_ROW_PALINDROME)) flip = false;   // rc(x) == x
#ifdef LDBG_LEAN_PROFILE
    if (LEAN) {
        asm volatile("" ::: "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        n.p1 = t1; n.p2 = t2;
    }
#endif
    n.flip = flip ? 1 : 0;
    if (LEAN) node_fill_bytes<true>(e, n, ef & ((1u << (8 * g.C)) - 1u), nullptr, (uint8_t)(ef >> (8 * g.C)));
    else if (rec && packed) node_fill_bytes(e, n, ef & ((1u << (8 * g.C)) - 1u), nullptr, (uint8_t)(ef >> (8 * g.C)));
    else node_fill(e, n);
    n.ui = ui;
    if (rec) {
        // (selected whatever the mask, so that the neighbour index is read together with the edge bytes, not after them)
        const uint32_t m = fwd ? n.next_mask : n.prev_mask;
        const unsigned j = nbr_slot(n.fj != 0, fwd, lowbit4(m));
        uint32_t v = nb[0];
#pragma unroll
        for (unsigned q = 1; q < 8; q++) v = j == q ? nb[q] : v;
        const bool one = popc4(m) == 1;
        n.ent1 = one ? v : 0u; n.e1 = one ? 1 : 0;
        if (!early) { key = vt_key(n.idx, flip); h = vt_hash(key) & t.mask; e0 = vt_peek(t, h); }
        n.vslot = vt_probe_from(t, key, h, e0, &n.vent);
    }
}
// node_from_entry<true> in two halves, so that the rows of SEVERAL vertices are in flight together (the run step reads two fringe
// vertices and a table entry that do not depend on each other: three trips to memory one after the other, or one).
// issue: every load the vertex needs — its row (neighbour index, edge and flag bytes), its run-index entry, the first probe round
// of its table slot.  finish: the rest.  A probe round read before ANOTHER vertex claimed a slot may be stale: `claimed` names
// the slots claimed since (0xFFFFFFFF = none); a round that covers one of them is read again.
struct NodeLoad {
    uint32_t nb[8];
    uint32_t ef, h;
    uint64_t ui, key;
    VtPeek e0;
    int32_t idx;
    bool flip;
};
LDBG_HOSTDEV void node_issue_lean(const EngineView& e, const VisitedTable& t, bool parent_fj, uint32_t ent, NodeLoad& L) {
    const GraphView& g = e.g;
    L.idx = (int32_t)(ent & 0x7FFFFFFFu) - 1;
    L.flip = (((ent >> 31) & 1u) != 0) != parent_fj;
    L.key = vt_key(L.idx, L.flip);
    L.h = vt_hash(L.key) & t.mask;
    L.e0 = vt_peek(t, L.h);
    L.ui = e.runs.uinfo ? LDBG_GLOBAL(const uint64_t, e.runs.uinfo)[L.idx] : 0ull;
    const uint8_t* row = graph_row(g, L.idx);
    struct alignas(16) U4 { uint32_t x, y, z, w; };
    const U4 a = *(const U4*)(row + g.nbr_off), b = *(const U4*)(row + g.nbr_off + 16);
    L.ef = *(const uint32_t*)(row + g.edges_off);
    L.nb[0] = a.x; L.nb[1] = a.y; L.nb[2] = a.z; L.nb[3] = a.w; L.nb[4] = b.x; L.nb[5] = b.y; L.nb[6] = b.z; L.nb[7] = b.w;
}
LDBG_HOSTDEV bool vt_round_covers(const VisitedTable& t, uint32_t h, uint32_t slot) { return slot != 0xFFFFFFFFu && ((slot - h) & t.mask) < 4u; }
LDBG_HOSTDEV void node_finish_lean(const EngineView& e, VisitedTable& t, const NodeLoad& L, unsigned base, bool fwd, Node& n, uint32_t claimed0 = 0xFFFFFFFFu,
                                   uint32_t claimed1 = 0xFFFFFFFFu) {
    const GraphView& g = e.g;
    n.idx = L.idx;
    n.copy = 0; n.vslot = 0; n.vent = 0; n.base = (uint8_t)base; n.e1 = 0; n.ent1 = 0;
    n.flip = L.flip ? 1 : 0;
    node_fill_bytes<true>(e, n, L.ef & ((1u << (8 * g.C)) - 1u), nullptr, (uint8_t)(L.ef >> (8 * g.C)));
    n.ui = L.ui;
    const uint32_t m = fwd ? n.next_mask : n.prev_mask;
    const unsigned j = nbr_slot(n.fj != 0, fwd, lowbit4(m));
    uint32_t v = L.nb[0];
#pragma unroll
    for (unsigned q = 1; q < 8; q++) v = j == q ? L.nb[q] : v;
    const bool one = popc4(m) == 1;
    n.ent1 = one ? v : 0u; n.e1 = one ? 1 : 0;
    const bool stale = vt_round_covers(t, L.h, claimed0) || vt_round_covers(t, L.h, claimed1);
    n.vslot = vt_probe_from(t, L.key, L.h, stale ? vt_peek(t, L.h) : L.e0, &n.vent);
}
LDBG_HOSTDEV void node_child_located(const EngineView& e, VisitedTable& t, const Node& p, bool fwd, unsigned base, Node& n) {
    node_from_entry(e, t, p, node_child_entry(e, p, fwd, base), base, fwd, n);
}

// ---- per-walk LinkStore (J/utils/traversal/LinkStore.java), elements kept in insertion order.
// Ages are kept relative: every incrementAges() call ages ALL elements (:37-43), so an element stores the
// value of the store's age counter at insertion and age = counter - birth; incrementAges and numNewPaths
// (:45-56) are O(1).  Each element caches its junction record's fields so the junction logic reads one
// element + one base byte per link.
struct LsElem {
    uint32_t str_off;  // junction string in LinksView.bases (one offset per junction record: identifies it)
    uint32_t birth;    // store.age at insertion
    int32_t hash;      // java.lang.String.hashCode of the (possibly complemented) junction string
    uint32_t key_seq;  // insertion sequence number of this element's key in the Java HashMap
    uint16_t len;
    uint16_t pos;
    uint8_t comp;      // junction string is used complemented (LinkStore.java:25)
    uint8_t nxn;       // junction bases cached in nx (> 0 whenever pos < len)
    uint16_t nx;       // the next nxn <= 8 junction bases from `pos` on (2 bits each, complemented already): the
                       // junction logic reads the pool of junction strings once per 8 positions
};
static_assert(sizeof(LsElem) == 24, "LsElem is moved as six words");
// element copies through pointers of a known address space (six word moves; the compiler pairs them up)
#ifdef LDBG_HOSTSIM
// LDBG_HOSTSIM_LDS_CHECK=1: the simulated LDS remembers which elements the running wavefront has written; reading one it has not is
// reading what an earlier workgroup left there (LDS is never cleared on the device either) — abort with a backtrace
struct LdsShadow { const char* base = nullptr; size_t bytes = 0; std::vector<uint8_t> written; bool on = false; };
inline LdsShadow& lds_shadow() { static LdsShadow s; return s; }
inline void lds_shadow_begin(const void* base, size_t bytes) {
    LdsShadow& s = lds_shadow();
    static const bool want = getenv("LDBG_HOSTSIM_LDS_CHECK") != nullptr;
    s.on = want;
    if (!want) return;
    s.base = (const char*)base; s.bytes = bytes; s.written.assign(bytes / 24 + 1, 0);
}
inline void lds_shadow_touch(const void* q, bool write) {
    LdsShadow& s = lds_shadow();
    if (!s.on || (const char*)q < s.base || (const char*)q >= s.base + s.bytes) return;
    const size_t i = (size_t)((const char*)q - s.base) / 24;
    if (write) { s.written[i] = 1; return; }
    if (!s.written[i]) {
        fprintf(stderr, "[hostsim] link-store element %zu of the simulated LDS is read before this wavefront wrote it\n", i);
        void* bt[48]; const int nb = backtrace(bt, 48); backtrace_symbols_fd(bt, nb, 2);
        abort();
    }
}
#endif
template <typename P> LDBG_HOSTDEV LsElem ls_elem_in(P q) {
#ifdef LDBG_HOSTSIM
    lds_shadow_touch((const void*)q, false);
#endif
    uint32_t w[6];
#pragma unroll
    for (int i = 0; i < 6; i++) w[i] = q[i];
    LsElem x;
    __builtin_memcpy(&x, w, 24);
    return x;
}
template <typename P> LDBG_HOSTDEV void ls_elem_out(P q, const LsElem& x) {
#ifdef LDBG_HOSTSIM
    lds_shadow_touch((const void*)q, true);
#endif
    uint32_t w[6];
    __builtin_memcpy(w, &x, 24);
#pragma unroll
    for (int i = 0; i < 6; i++) q[i] = w[i];
}
struct LinkStoreDev {
    LsElem* fast;       // the first `fast_cap` elements live here (LDS in the walk kernel), element i at fast[i * fast_stride]
    LsElem* el;         // the rest spill to HBM: element i at el[i - fast_cap]
    uint32_t fast_cap, fast_stride;
    uint32_t cap;       // total capacity
    uint32_t n;
    uint32_t java_cap;  // table size of the emulated java.util.HashMap (0 = not allocated)
    uint32_t nkeys;
    uint32_t next_seq;
    uint32_t age;       // number of incrementAges() calls
    uint32_t n_new;     // elements with age 0
    bool overflow;
};
// (on the device `fast` is only ever LDS and `el` HBM: said explicitly, or the accesses become flat_ ones that wait for both memories)
LDBG_HOSTDEV LsElem ls_get(const LinkStoreDev& s, uint32_t i) {
    if (i < s.fast_cap) return ls_elem_in(LDBG_LDS(const uint32_t, s.fast + i * s.fast_stride));
    return ls_elem_in(LDBG_GLOBAL(const uint32_t, s.el + (i - s.fast_cap)));
}
LDBG_HOSTDEV void ls_set(LinkStoreDev& s, uint32_t i, const LsElem& x) {
    if (i < s.fast_cap) ls_elem_out(LDBG_LDS(uint32_t, s.fast + i * s.fast_stride), x);
    else ls_elem_out(LDBG_GLOBAL(uint32_t, s.el + (i - s.fast_cap)), x);
}
LDBG_HOSTDEV void ls_clear(LinkStoreDev& s) { s.n = 0; s.java_cap = 0; s.nkeys = 0; s.next_seq = 0; s.age = 0; s.n_new = 0; s.overflow = false; }
LDBG_HOSTDEV unsigned ls_char(const LinksView& L, const LsElem& x, uint32_t i) {
    unsigned b = L.bases[x.str_off + i];
    return x.comp ? 3u - b : b;
}
// refill the cache of upcoming bases from position x.pos
LDBG_HOSTDEV void ls_fill_nx(const LinksView& L, LsElem& x) {
    uint32_t nx = 0, cnt = 0;
    for (uint32_t i = 0; i < 8; i++) {
        if ((uint32_t)x.pos + i >= x.len) break;
        nx |= (uint32_t)ls_char(L, x, (uint32_t)x.pos + i) << (2 * i);
        cnt++;
    }
    x.nx = (uint16_t)nx; x.nxn = (uint8_t)cnt;
}
// the cache of a new element (pos 0) comes with the junction record
LDBG_HOSTDEV void ls_first_nx(const JuncRec& jr, LsElem& x) {
    const uint32_t cnt = jr.len < 8u ? jr.len : 8u;
    const uint32_t mask = (1u << (2 * cnt)) - 1u;
    const uint32_t nx = jr.is_fw >> 16;
    x.nx = (uint16_t)((x.comp ? ~nx : nx) & mask);      // complement of base b is 3 - b
    x.nxn = (uint8_t)cnt;
}
LDBG_HOSTDEV unsigned ls_cur(const LsElem& x) { return x.nx & 3u; }          // junctions.charAt(pos)
LDBG_HOSTDEV void ls_advance(const LinksView& L, LsElem& x) {               // pos++ (pos + 1 < len holds)
    x.pos++;
    x.nx >>= 2; x.nxn--;
    if (x.nxn == 0) ls_fill_nx(L, x);
}
LDBG_HOSTDEV bool ls_same_string(const LinksView& L, const LsElem& a, const LsElem& b) {
    if (a.str_off == b.str_off && a.comp == b.comp) return true;
    if (a.len != b.len || a.hash != b.hash) return false;
    for (uint32_t i = 0; i < a.len; i++) if (ls_char(L, a, i) != ls_char(L, b, i)) return false;
    return true;
}
// is any live element (other than index `skip`) filed under HashMap key `key_seq`?
LDBG_HOSTDEV bool ls_key_alive(const LinkStoreDev& s, uint32_t key_seq, uint32_t n) {
    for (uint32_t i = n; i-- > 0;) if (ls_get(s, i).key_seq == key_seq) return true;
    return false;
}
// LinkStore.add :17-35 for merged link record m.  `query_flipped`: the cursor k-mer is the reverse complement
// of the canonical key; JuncRec.is_fw is stored as "link goes forward when the query is the canonical k-mer".
// s.nkeys = HashMap.size() is maintained incrementally (new key: +1 here; last element of a key expiring: -1 in
// ls_next_choice); the table doubles when a put() pushes it past 3/4 of the table size (HashMap.resize).
LDBG_HOSTDEV void ls_add(const LinksView& L, LinkStoreDev& s, uint32_t jlo, uint32_t jhi, bool query_flipped, bool fwd) {
    for (uint32_t j = jlo; j < jhi; j++) {
        const JuncRec jr = L.junc[j];
        bool lgf = ((jr.is_fw & 1u) != 0) != query_flipped;    // recordOrientationMatchesKmer == cjr.isForward() :24
        if (lgf != fwd) continue;
        LsElem x;
        x.str_off = jr.str_off; x.birth = s.age; x.hash = lgf ? jr.hash_asis : jr.hash_comp;
        x.len = (uint16_t)jr.len; x.pos = 0; x.comp = lgf ? 0 : 1; x.key_seq = 0;
        ls_first_nx(jr, x);
        // an element with the same junction string?  newest first: a walk circling a repeat re-adds the links it
        // added one revolution ago, so the match sits near the end of the (insertion-ordered) array
        bool have = false;
        for (uint32_t i = s.n; i-- > 0;)
            { const LsElem y = ls_get(s, i); if (ls_same_string(L, y, x)) { x.key_seq = y.key_seq; have = true; break; } }
        if (!have) {
            x.key_seq = s.next_seq++;
            s.nkeys++;
            if (s.java_cap == 0) s.java_cap = 16;
            if (s.nkeys > s.java_cap * 3 / 4) s.java_cap *= 2;
        }
        if (s.n >= s.cap || jr.len >= 65535u || s.n >= 0x7FFFu) { s.overflow = true; return; }
        ls_set(s, s.n++, x);
        s.n_new++;
#ifdef LDBG_HOSTSIM
        ls_debug().adds++; if (!have) ls_debug().newkeys++; if (s.n > ls_debug().maxn) ls_debug().maxn = s.n;
#endif
    }
}
LDBG_HOSTDEV void ls_increment_ages(LinkStoreDev& s) { s.age++; s.n_new = 0; }
LDBG_HOSTDEV int ls_num_new(const LinkStoreDev& s) { return (int)s.n_new; }
// LinkStore.getNextJunctionChoice :122-144 (+ getOldestLink :92-119, incrementPositionsAndExpire :58-90).
// Elements are in insertion order and births never decrease along the array, so the oldest links are a prefix.
LDBG_HOSTDEV bool ls_next_choice(const LinksView& L, LinkStoreDev& s, unsigned* choice) {
    if (s.n == 0) return false;
#ifdef LDBG_HOSTSIM
    ls_debug().choices++;
#endif
    const uint32_t minbirth = ls_get(s, 0).birth;   // oldest = largest age = smallest birth
    // first oldest element in java.util.HashMap iteration order: (bucket, key insertion order, list order)
    bool agree = true;
    unsigned ch0 = 0;
    uint32_t best_b = 0, best_seq = 0;
    for (uint32_t i = 0; i < s.n; i++) {
        const LsElem x = ls_get(s, i);
        if (x.birth != minbirth) break;
        unsigned c = ls_cur(x);
        uint32_t h = (uint32_t)x.hash;
        uint32_t b = (h ^ (h >> 16)) & (s.java_cap - 1);
        if (i == 0) { ch0 = c; best_b = b; best_seq = x.key_seq; }
        else {
            if (c != ch0) agree = false;
            if (b < best_b || (b == best_b && x.key_seq < best_seq)) { best_b = b; best_seq = x.key_seq; }
        }
    }
    if (!agree) return false;
    unsigned ch = ch0;
    for (uint32_t i = s.n; i-- > 0;)     // last element of that key's list wins (:129-133)
        { const LsElem y = ls_get(s, i); if (y.key_seq == best_seq) { ch = ls_cur(y); break; } }
    // incrementPositionsAndExpire(choice): four elements at a time so that their loads overlap
    uint32_t w = 0, n_new = 0;
    uint32_t dead[8];
    uint32_t n_dead = 0;
    bool many_dead = false;
    for (uint32_t i0 = 0; i0 < s.n; i0 += 4) {
        LsElem x[4];
        unsigned c[4];
        const uint32_t cnt = s.n - i0 < 4 ? s.n - i0 : 4;
        for (uint32_t q = 0; q < 4; q++) if (q < cnt) x[q] = ls_get(s, i0 + q);
        for (uint32_t q = 0; q < 4; q++) if (q < cnt) c[q] = ls_cur(x[q]);
        for (uint32_t q = 0; q < 4; q++) {
            if (q >= cnt) break;
            if ((uint32_t)x[q].pos + 1 >= x[q].len || c[q] != ch) {
                if (n_dead < 8) dead[n_dead++] = x[q].key_seq; else many_dead = true;
                continue;
            }
            ls_advance(L, x[q]);
            n_new += x[q].birth == s.age;
            ls_set(s, w++, x[q]);
        }
    }
    s.n = w;
    s.n_new = n_new;
    // keys whose last element expired leave the HashMap (:84-88)
    if (many_dead) {
        uint32_t nk = 0;
        for (uint32_t i = 0; i < s.n; i++) {
            bool first = true;
            for (uint32_t j = 0; j < i; j++) if (ls_get(s, j).key_seq == ls_get(s, i).key_seq) { first = false; break; }
            nk += first;
        }
        s.nkeys = nk;
    } else {
        for (uint32_t d = 0; d < n_dead; d++) {
            bool dup = false;
            for (uint32_t q = 0; q < d; q++) dup |= dead[q] == dead[d];
            if (!dup && !ls_key_alive(s, dead[d], s.n)) s.nkeys--;
        }
    }
    *choice = ch;
    return true;
}

// ---- cursor (TraversalEngine.seek / next / previous, TraversalEngine.java:241-339, 518-597)
struct Cursor {
    Node cur;
    Node nxt;           // the vertex hasNext()/hasPrevious() refers to, looked up one step ahead
    bool has;
    bool first;         // specificLinksFiles == null: the next step re-seeks and initialises the link store
    uint32_t status;
    uint32_t epoch;     // current generation of the `seen` set (never 0)
};

// initializeLinkStore / updateLinkStore (:548-597): links of vertex v, if its record carries any
template <int W>
LDBG_HOSTDEV void cursor_add_links(const EngineView& e, LinkStoreDev& s, const Node& v, bool fwd) {
    if (!(v.lflags & e.link_flag_mask)) return;
    const uint64_t m = e.links.rec_of[v.idx];
    if (m != ~0ull) ls_add(e.links, s, (uint32_t)m, (uint32_t)m + (uint32_t)(m >> 32), v.flip != 0, fwd);
}
// seek(sk): cursor on v, unique neighbour in direction `fwd` looked up (TraversalEngine.java:321-335)
LDBG_HOSTDEV void cursor_seek(const EngineView& e, Cursor& cu, LinkStoreDev& s, VisitedTable& vt, const Node& v, bool fwd) {
    cu.cur = v;
    cu.first = true;
    cu.status = ST_OK;
    ls_clear(s);
    uint32_t m = fwd ? v.next_mask : v.prev_mask;
    cu.has = popc4(m) == 1;
    if (cu.has) node_child_located(e, vt, v, fwd, lowbit4(m), cu.nxt);
}
// getAdjacentKmer (:518-546): the junction choice applied to the cursor k-mer must be one of the neighbours.
// The candidate is built from the cursor's own string, the neighbours from o (they differ under Q6 only).
template <int W>
LDBG_HOSTDEV int cursor_choice_base(const EngineView& e, const Node& t, uint32_t m, bool fwd, unsigned ch) {
    if (!(t.flip && !t.fj)) return ((m >> ch) & 1u) ? (int)ch : -1;
    Kmer<W> sk = node_kmer<W>(e, t);
    Kmer<W> cand = fwd ? kmer_next<W>(sk, e.g.k, ch) : kmer_prev<W>(sk, e.g.k, ch);
    for (unsigned b = 0; b < 4; b++)
        if (((m >> b) & 1u) && kmer_eq<W>(child_kmer<W>(e, t, fwd, b), cand)) return (int)b;
    return -1;
}
// what the walk kernel's wave-cooperative phases (lscoop.h) have already done for this step
struct StepPre {
    bool links_done;     // initializeLinkStore / updateLinkStore
    bool choice_done;    // getNextJunctionChoice, with its result
    bool choice_ok;
    unsigned ch;
    bool has_child;
    uint16_t n_added;    // link-store elements the cooperative phase added for this lane (step-kind counters of the walk kernel)
    Node child;          // the single successor of the vertex stepped onto, located ahead of the link-store phases
};
// next()/previous() (TraversalEngine.java:241-319); requires cu.has.  Returns the vertex stepped onto.
// PRE: the link-store work of the step was done ahead by the caller (walk kernel) and arrives in *pre; the one-lane
// LinkStore code is then not even compiled into the kernel.
template <int W, bool PRE = false>
LDBG_HOSTDEV Node cursor_step(const EngineView& e, Cursor& cu, LinkStoreDev& s, VisitedTable& vt, bool fwd, const StepPre* pre = nullptr, uint32_t* marks = nullptr) {
    const bool links_done = PRE;
    if (cu.first) {
        cu.first = false;                              // seek(cur) recomputes the same state; then
        if constexpr (!PRE) cursor_add_links<W>(e, s, cu.cur, fwd);     // initializeLinkStore :548-568
    }
    if constexpr (!PRE) cursor_add_links<W>(e, s, cu.nxt, fwd);         // updateLinkStore :570-597
    (void)links_done;
    Node t = cu.nxt;
    cu.cur = t;
    if (t.npe) cu.status = ST_NULLPTR;
    const uint32_t m = fwd ? t.next_mask : t.prev_mask;
    bool has = false;
    const int pc = popc4(m);
    if (pc == 1) {
        Node x;
        if (PRE && pre->has_child) x = pre->child;
        else node_child_located(e, vt, t, fwd, lowbit4(m), x);
        const uint64_t ex = x.idx >= 0 ? x.vent : 0ull;
        if (!vt_seen_e(ex, cu.epoch) || s.n > 0) {      // :262
            if (x.idx >= 0 && !vt_seen_e(ex, cu.epoch)) {                // seen.add(nextKmer)
                node_store(vt, x, vt_with_seen(ex, cu.epoch));
                node_sync(cu.cur, x);
                if (marks) ++*marks;
            }
            cu.nxt = x;
            has = true;
        }
    } else if (pc > 1) {
        unsigned ch = 0;
        bool ok;
        if constexpr (PRE) { ok = pre->choice_ok; ch = pre->ch; }
        else ok = ls_next_choice(e.links, s, &ch);
        if (ok) {
            const int mb = cursor_choice_base<W>(e, t, m, fwd, ch);
            if (mb >= 0) { node_child_located(e, vt, t, fwd, (unsigned)mb, cu.nxt); has = true; }
        }
        ls_increment_ages(s);                           // :271
    }
    cu.has = has;
#ifdef LDBG_HOSTSIM
    ls_debug().steps++; ls_debug().sum_n += s.n;
#endif
    if (ls_num_new(s) > 0) ls_increment_ages(s);        // :274-276 (Q12)
    if (s.overflow) cu.status = ST_LINKSTORE_FULL;
    return t;
}

}  // namespace ldbg
