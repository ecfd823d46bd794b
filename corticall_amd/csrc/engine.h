// TraversalEngine on the device: shared device-side primitives (vertex lookup with its neighbourhood,
// per-walk visited table, per-walk LinkStore, cursor step) used by the walk and cursor kernels.
// Every function cites the Java it restates (J/ = public/java/src/uk/ac/ox/well/cortexjdk/).
//
// Memory discipline of one traversal step (DESIGN.md §Walk kernel): the dependent chain is
//   radix-index load -> probe row (key + edges + link flags in one sector) -> visited-table slot
// Everything else (link table search, link store) only runs where the row's link flags say so.
#pragma once
#include "graph.h"
#include "links.h"

namespace ldbg {

struct EngineView {
    GraphView g;
    uint32_t trav_mask, recruit_mask, join_mask;
    int first_trav;
    int stopper, max_len, connect_all, strict_flip;
    int cursor_on;        // !ec.getLinks().isEmpty(): dfs drives the cursor (TraversalEngine.java:363, 379) even
                          // when none of the configured link sets belongs to a traversal sample
    uint32_t link_flag_mask;   // probe-row link-flag bits of the link sets merged into `links`
    LinksView links;      // the traversal's link sets merged into one table (links.h)
};

// strand status codes (per seed and direction)
enum : uint32_t {
    ST_OK = 0,
    ST_NULLPTR = 1,        // the reference throws NullPointerException here
    ST_LINKSTORE_FULL = 2, // per-walk link store capacity exceeded -> host retries with a larger store
    ST_BRANCH_NULL = 3,    // dfs branch returned null
    ST_COPY_OVERFLOW = 4,  // more than 32767 copies of one vertex
    ST_POOL_FULL = 5       // path block pool exhausted -> host splits the batch
};

// ---- path entry: one vertex of a branch, 8 bytes
//  bits 0..32  record index + 1 (0 = null CortexRecord)
//  bit  33     flip: vertex k-mer is the reverse complement of the record's (canonical) k-mer
//  bits 34..35 the base this vertex appended to the contig (last base going forward, first base going back)
//  bits 36..59 copyIndex, 24-bit two's complement (CortexVertex.copyIndex)
LDBG_HOSTDEV uint64_t path_pack(int64_t idx, bool flip, unsigned base, int copy) {
    return (uint64_t)(idx + 1) | ((uint64_t)(flip ? 1 : 0) << 33) | ((uint64_t)(base & 3u) << 34) |
           (((uint64_t)(uint32_t)copy & 0xFFFFFFull) << 36);
}
LDBG_HOSTDEV int64_t path_idx(uint64_t e) { return (int64_t)(e & 0x1FFFFFFFFull) - 1; }
LDBG_HOSTDEV bool path_flip(uint64_t e) { return (e >> 33) & 1ull; }
LDBG_HOSTDEV unsigned path_base(uint64_t e) { return (unsigned)((e >> 34) & 3ull); }
LDBG_HOSTDEV int path_copy(uint64_t e) { int32_t v = (int32_t)((e >> 36) & 0xFFFFFFull); return (v << 8) >> 8; }

LDBG_HOSTDEV int popc4(uint32_t m) { return (int)((m & 1u) + ((m >> 1) & 1u) + ((m >> 2) & 1u) + ((m >> 3) & 1u)); }
LDBG_HOSTDEV unsigned lowbit4(uint32_t m) { return (m & 1u) ? 0u : ((m & 2u) ? 1u : ((m & 4u) ? 2u : 3u)); }

// ---- a vertex with its neighbourhood: CortexVertex + TraversalUtils.getAllNextKmers/getAllPrevKmers
// (J/utils/traversal/TraversalUtils.java:510-590) + TraversalEngine.getNextVertices/getPrevVertices
// (J/utils/traversal/TraversalEngine.java:147-239) folded into 4-bit base masks.
template <int W>
struct Node {
    Kmer<W> sk;          // the vertex's k-mer, in the orientation it was reached
    int64_t idx;         // record index, -1 = null CortexRecord
    int32_t copy;        // CortexVertex.copyIndex
    uint32_t vslot;      // slot of this vertex in the walk's visited table (valid once located; idx >= 0)
    uint8_t flip;        // sk != canonical orientation (by comparison): part of the vertex identity
    uint8_t fj;          // CanonicalKmer.isFlipped() — by Arrays.hashCode inequality (quirk Q6)
    uint8_t npe;         // record missing while recruitment colours are set (Q14)
    uint8_t lflags;      // link-flag byte of the record's probe row
    uint8_t next_mask;   // bit b: successor o[1:]+b
    uint8_t prev_mask;   // bit b: predecessor b+o[:-1]
};
// orientation the neighbours are built from: the record's k-mer, or its reverse complement when
// isFlipped() (TraversalUtils.java:514, 539).  Equals sk except under a Q6 hash collision.
template <int W>
LDBG_HOSTDEV Kmer<W> node_o(const Node<W>& n, int k) {
    return (n.flip && !n.fj) ? kmer_revcomp<W>(n.sk, k) : n.sk;
}
template <int W>
LDBG_HOSTDEV Kmer<W> node_neighbour(const Node<W>& n, int k, bool fwd, unsigned base) {
    Kmer<W> o = node_o<W>(n, k);
    return fwd ? kmer_next<W>(o, k, base) : kmer_prev<W>(o, k, base);
}
template <int W>
LDBG_HOSTDEV void node_null(const EngineView& e, const Kmer<W>& sk, Node<W>& n) {   // not a k-mer / no record
    n.sk = sk; n.idx = -1; n.copy = 0; n.vslot = 0; n.flip = 0; n.fj = 0; n.lflags = 0;
    n.next_mask = n.prev_mask = 0;
    n.npe = e.recruit_mask != 0 ? 1 : 0;
}
// CanonicalKmer.isFlipped(): Arrays.hashCode(canonical) != Arrays.hashCode(supplied) (CanonicalKmer.java:16,23,33).
// Differs from "the k-mer was reverse complemented" only on a 32-bit hash collision (Q6); the hashes are first
// compared modulo 32 (a few popcounts), the 2k multiply-adds run for 1 flipped k-mer in 32.
template <int W>
LDBG_HOSTDEV bool java_flipped(const EngineView& e, const Kmer<W>& sk, bool flip_cmp) {
    if (!flip_cmp || !e.strict_flip) return flip_cmp;
    uint32_t hs, hr;
    kmer_java_hash_mod32<W>(sk, e.g.k, &hs, &hr);
    if (hs != hr) return true;
    return kmer_java_hash<W>(kmer_revcomp<W>(sk, e.g.k), e.g.k) != kmer_java_hash<W>(sk, e.g.k);
}
template <int W> LDBG_HOSTDEV void node_fill_masks(const EngineView& e, Node<W>& n);

// findRecord(sk) + neighbourhood: one radix-index load, then probe rows; the matching row yields edges and
// link flags from the same sector as its key.
template <int W>
LDBG_HOSTDEV void node_find(const EngineView& e, const Kmer<W>& sk, Node<W>& n) {
    const GraphView& g = e.g;
    bool fc;
    Kmer<W> c = kmer_canonical<W>(sk, g.k, &fc);
    n.sk = sk; n.copy = 0; n.vslot = 0; n.flip = fc ? 1 : 0; n.npe = 0; n.lflags = 0;
    const bool fj = java_flipped<W>(e, sk, fc);
    n.fj = fj ? 1 : 0;
    const int64_t idx = graph_find_canonical<W>(g, c);
    n.idx = idx;
    node_fill_masks<W>(e, n);
}
// edges of the node's record -> neighbour masks (+ link flags); one row read
template <int W>
LDBG_HOSTDEV void node_fill_masks(const EngineView& e, Node<W>& n) {
    const GraphView& g = e.g;
    const int64_t idx = n.idx;
    const bool fj = n.fj != 0;
    n.npe = 0; n.lflags = 0;
    uint32_t tf = 0, tr = 0, rf = 0, rr = 0;
    if (idx >= 0) {
        const uint8_t* row = graph_row(g, idx);
        const uint8_t* ed = row + g.edges_off;
        n.lflags = row[g.flags_off];
        for (int col = 0; col < g.C; col++) {
            uint32_t ebyte = ed[col];
            uint32_t lo = ebyte & 0xf, hi = ebyte >> 4;
            // CortexRecord.getOutEdgesAsBytes: bit i <-> base i ; getInEdgesAsBytes: bit (3-i) <-> base i ;
            // complement=true relabels base b as 3-b (CortexRecord.java:214-275)
            uint32_t fwd = !fj ? lo : hi;                      // successor base = bit position
            uint32_t revn = !fj ? hi : lo;                     // predecessor base = 3 - bit position
            uint32_t rev = ((revn & 1u) << 3) | ((revn & 2u) << 1) | ((revn & 4u) >> 1) | ((revn & 8u) >> 3);
            if ((e.trav_mask >> col) & 1u) { tf |= fwd; tr |= rev; }
            if ((e.recruit_mask >> col) & 1u) { rf |= fwd; rr |= rev; }
        }
    } else if (e.recruit_mask != 0) {
        n.npe = 1;
    }
    n.next_mask = (uint8_t)(tf ? tf : rf);    // recruitment colours only where the traversal colours give nothing
    n.prev_mask = (uint8_t)(tr ? tr : rr);
}

// neighbour `base` of vertex p in travel direction `fwd`, through the neighbour index of p's probe row
// (memoised findRecord, graph.h) — no search, no canonicalisation on the walk's critical path
template <int W>
LDBG_HOSTDEV void node_child(const EngineView& e, const Node<W>& p, bool fwd, unsigned base, Node<W>& n) {
    const GraphView& g = e.g;
    const Kmer<W> sk = node_neighbour<W>(p, g.k, fwd, base);
    if (!g.nbr_on || p.idx < 0 || !(g.k & 1)) { node_find<W>(e, sk, n); return; }   // even k: palindromes need the compare
    // p's orientation for neighbour generation is its Java flip: o = fj ? rc(canon) : canon.
    //   fwd, !fj: next(c, b)            -> succ[b]         fwd, fj: next(rc(c), b) = rc(prev(c, 3-b)) -> pred[3-b], toggled
    //   rev, !fj: prev(c, b)            -> pred[b]         rev, fj: prev(rc(c), b) = rc(next(c, 3-b)) -> succ[3-b], toggled
    const bool fj = p.fj != 0;
    const unsigned j = fwd ? (!fj ? base : 4u + (3u - base)) : (!fj ? 4u + base : (3u - base));
    const uint32_t ent = graph_nbr(g, p.idx, (int)j);
    n.sk = sk; n.copy = 0; n.vslot = 0;
    n.idx = (int64_t)(ent & 0x7FFFFFFFu) - 1;
    const bool flip = (((ent >> 31) & 1u) != 0) != fj;
    n.flip = (n.idx >= 0 && flip) ? 1 : 0;
    if (n.idx < 0) { bool fc; kmer_canonical<W>(sk, g.k, &fc); n.flip = fc ? 1 : 0; }   // null record: identity by k-mer
    n.fj = java_flipped<W>(e, sk, n.flip != 0) ? 1 : 0;
    node_fill_masks<W>(e, n);
}

// ---- per-walk visited table (HashSet<CortexVertex> visited, TraversalEngine.java:360-425, plus the
// cursor's `seen` set :27,262-265): open addressing over 8-byte entries in HBM.  A table is carved out of a
// zeroed pool when a strand starts (4096 entries) and regrown x4 when half full, so memory follows the
// walk lengths.  A vertex is located once (when it is first looked up as a neighbour); later updates go
// straight to its slot.
//  entry: bits 0..33 key = ((record index + 1) << 1) | flip (never 0) ; bits 48..62 copies visited ; bit 63 seen
struct VisitedTable {
    uint64_t* tab;
    uint32_t mask;   // capacity - 1
    uint32_t used;   // claimed slots
};
#define LDBG_VT_KEY_MASK 0x3FFFFFFFFull
LDBG_HOSTDEV uint32_t vt_hash(uint64_t key) {
    uint64_t x = key * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(x >> 32);
}
LDBG_HOSTDEV uint64_t vt_key(int64_t idx, bool flip) { return ((uint64_t)(idx + 1) << 1) | (flip ? 1ull : 0ull); }
// slot of the key starting the probe at h with the first entry already loaded; claims a free slot if absent
LDBG_HOSTDEV uint32_t vt_locate_from(VisitedTable& t, uint64_t key, uint32_t h, uint64_t e) {
    while (true) {
        if (e == 0) { t.tab[h] = key; t.used++; return h; }
        if ((e & LDBG_VT_KEY_MASK) == key) return h;
        h = (h + 1) & t.mask;
        e = t.tab[h];
    }
}
LDBG_HOSTDEV uint32_t vt_locate(VisitedTable& t, int64_t idx, bool flip) {
    const uint64_t key = vt_key(idx, flip);
    const uint32_t h = vt_hash(key) & t.mask;
    return vt_locate_from(t, key, h, t.tab[h]);
}
LDBG_HOSTDEV int vt_count_e(uint64_t e) { return (int)((e >> 48) & 0x7FFFull); }
LDBG_HOSTDEV bool vt_seen_e(uint64_t e) { return (e >> 63) != 0; }
LDBG_HOSTDEV uint64_t vt_with_count(uint64_t e, int c) { return (e & ~(0x7FFFull << 48)) | ((uint64_t)(c & 0x7FFF) << 48); }
template <int W>
LDBG_HOSTDEV void node_locate(VisitedTable& t, Node<W>& n) { if (n.idx >= 0) n.vslot = vt_locate(t, n.idx, n.flip != 0); }
template <int W>
LDBG_HOSTDEV int node_count(const VisitedTable& t, const Node<W>& n) { return n.idx >= 0 ? vt_count_e(t.tab[n.vslot]) : 0; }
// neighbour + its visited-table slot: the table probe is issued before the row is read so the two loads overlap
template <int W>
LDBG_HOSTDEV void node_child_located(const EngineView& e, VisitedTable& t, const Node<W>& p, bool fwd, unsigned base, Node<W>& n) {
    node_child<W>(e, p, fwd, base, n);
    node_locate<W>(t, n);
}

// ---- per-walk LinkStore (J/utils/traversal/LinkStore.java), elements kept in insertion order
struct LsElem {
    uint32_t jrec;     // index into LinksView.junc
    uint32_t age;
    uint32_t key_seq;  // insertion sequence number of this element's key in the Java HashMap
    uint16_t pos;
    uint16_t comp;     // junction string is used complemented (LinkStore.java:25)
};
struct LinkStoreDev {
    LsElem* el;
    uint32_t cap;       // capacity of el
    uint32_t n;
    uint32_t java_cap;  // table size of the emulated java.util.HashMap (0 = not allocated)
    uint32_t nkeys;
    uint32_t next_seq;
    bool overflow;
};
LDBG_HOSTDEV void ls_clear(LinkStoreDev& s) { s.n = 0; s.java_cap = 0; s.nkeys = 0; s.next_seq = 0; s.overflow = false; }
LDBG_HOSTDEV unsigned ls_char(const LinksView& L, const LsElem& x, uint32_t i) {
    unsigned b = L.bases[L.junc[x.jrec].str_off + i];
    return x.comp ? 3u - b : b;
}
LDBG_HOSTDEV uint32_t ls_len(const LinksView& L, const LsElem& x) { return L.junc[x.jrec].len; }
LDBG_HOSTDEV int32_t ls_hash(const LinksView& L, const LsElem& x) { return x.comp ? L.junc[x.jrec].hash_comp : L.junc[x.jrec].hash_asis; }
LDBG_HOSTDEV bool ls_same_string(const LinksView& L, const LsElem& a, const LsElem& b) {
    if (a.jrec == b.jrec && a.comp == b.comp) return true;
    uint32_t la = ls_len(L, a);
    if (la != ls_len(L, b) || ls_hash(L, a) != ls_hash(L, b)) return false;
    for (uint32_t i = 0; i < la; i++) if (ls_char(L, a, i) != ls_char(L, b, i)) return false;
    return true;
}
// LinkStore.add :17-35 for merged link record m.  `query_flipped`: the cursor k-mer is the reverse complement
// of the canonical key; JuncRec.is_fw is stored as "link goes forward when the query is the canonical k-mer".
LDBG_HOSTDEV void ls_add(const LinksView& L, LinkStoreDev& s, int64_t m, bool query_flipped, bool fwd) {
    for (uint32_t j = L.off[m]; j < L.off[m + 1]; j++) {
        bool lgf = (L.junc[j].is_fw != 0) != query_flipped;    // recordOrientationMatchesKmer == cjr.isForward() :24
        if (lgf != fwd) continue;
        LsElem x;
        x.jrec = j; x.age = 0; x.pos = 0; x.comp = lgf ? 0 : 1; x.key_seq = 0;
        bool have = false;
        for (uint32_t i = 0; i < s.n; i++)
            if (ls_same_string(L, s.el[i], x)) { x.key_seq = s.el[i].key_seq; have = true; break; }
        if (!have) {
            x.key_seq = s.next_seq++;
            s.nkeys++;
            if (s.java_cap == 0) s.java_cap = 16;
            if (s.nkeys > s.java_cap * 3 / 4) s.java_cap *= 2;   // HashMap.resize
        }
        if (s.n >= s.cap) { s.overflow = true; return; }
        s.el[s.n++] = x;
    }
}
LDBG_HOSTDEV void ls_increment_ages(LinkStoreDev& s) { for (uint32_t i = 0; i < s.n; i++) s.el[i].age++; }
LDBG_HOSTDEV int ls_num_new(const LinkStoreDev& s) { int c = 0; for (uint32_t i = 0; i < s.n; i++) c += s.el[i].age == 0; return c; }
// LinkStore.getNextJunctionChoice :122-144 (+ getOldestLink :92-119, incrementPositionsAndExpire :58-90)
LDBG_HOSTDEV bool ls_next_choice(const LinksView& L, LinkStoreDev& s, unsigned* choice) {
    if (s.n == 0) return false;
    uint32_t maxage = 0;
    for (uint32_t i = 0; i < s.n; i++) if (s.el[i].age > maxage) maxage = s.el[i].age;
    // first oldest element in java.util.HashMap iteration order: (bucket, key insertion order, list order)
    bool have = false, agree = true;
    unsigned ch0 = 0;
    uint32_t best_b = 0, best_seq = 0;
    for (uint32_t i = 0; i < s.n; i++) {
        const LsElem& x = s.el[i];
        if (x.age != maxage) continue;
        unsigned c = ls_char(L, x, x.pos);
        uint32_t h = (uint32_t)ls_hash(L, x);
        uint32_t b = (h ^ (h >> 16)) & (s.java_cap - 1);
        if (!have) { have = true; ch0 = c; best_b = b; best_seq = x.key_seq; }
        else {
            if (c != ch0) agree = false;
            if (b < best_b || (b == best_b && x.key_seq < best_seq)) { best_b = b; best_seq = x.key_seq; }
        }
    }
    if (!have || !agree) return false;
    unsigned ch = ch0;
    for (uint32_t i = 0; i < s.n; i++)   // last element of that key's list wins (:129-133)
        if (s.el[i].key_seq == best_seq) ch = ls_char(L, s.el[i], s.el[i].pos);
    // incrementPositionsAndExpire(choice)
    uint32_t w = 0;
    for (uint32_t i = 0; i < s.n; i++) {
        LsElem x = s.el[i];
        if ((uint32_t)x.pos + 1 >= ls_len(L, x) || ls_char(L, x, x.pos) != ch) continue;
        x.pos++;
        s.el[w++] = x;
    }
    s.n = w;
    uint32_t nk = 0;
    for (uint32_t i = 0; i < s.n; i++) {
        bool first = true;
        for (uint32_t j = 0; j < i; j++) if (s.el[j].key_seq == s.el[i].key_seq) { first = false; break; }
        nk += first;
    }
    s.nkeys = nk;
    *choice = ch;
    return true;
}

// ---- cursor (TraversalEngine.seek / next / previous, TraversalEngine.java:241-339, 518-597)
template <int W>
struct Cursor {
    Node<W> cur;
    Node<W> nxt;        // the vertex hasNext()/hasPrevious() refers to, looked up one step ahead
    bool has;
    bool first;         // specificLinksFiles == null: the next step re-seeks and initialises the link store
    uint32_t status;
};

// initializeLinkStore / updateLinkStore (:548-597): links of vertex v, if its record carries any
template <int W>
LDBG_HOSTDEV void cursor_add_links(const EngineView& e, LinkStoreDev& s, const Node<W>& v, bool fwd) {
    if (!(v.lflags & e.link_flag_mask)) return;
    Kmer<W> c = v.flip ? kmer_revcomp<W>(v.sk, e.g.k) : v.sk;
    int64_t m = links_find<W>(e.links, e.g.k, c);
    if (m >= 0) ls_add(e.links, s, m, v.flip != 0, fwd);
}
// seek(sk): cursor on v, unique neighbour in direction `fwd` looked up (TraversalEngine.java:321-335)
template <int W>
LDBG_HOSTDEV void cursor_seek(const EngineView& e, Cursor<W>& cu, LinkStoreDev& s, VisitedTable& vt, const Node<W>& v, bool fwd) {
    cu.cur = v;
    cu.first = true;
    cu.status = ST_OK;
    ls_clear(s);
    uint32_t m = fwd ? v.next_mask : v.prev_mask;
    cu.has = popc4(m) == 1;
    if (cu.has) {
        node_child_located<W>(e, vt, v, fwd, lowbit4(m), cu.nxt);
    }
}
// next()/previous() (TraversalEngine.java:241-319); requires cu.has.  Returns the vertex stepped onto.
template <int W>
LDBG_HOSTDEV Node<W> cursor_step(const EngineView& e, Cursor<W>& cu, LinkStoreDev& s, VisitedTable& vt, bool fwd) {
    if (cu.first) {
        cu.first = false;                              // seek(cur) recomputes the same state; then
        cursor_add_links<W>(e, s, cu.cur, fwd);        // initializeLinkStore :548-568
    }
    cursor_add_links<W>(e, s, cu.nxt, fwd);            // updateLinkStore :570-597
    Node<W> t = cu.nxt;
    cu.cur = t;
    if (t.npe) cu.status = ST_NULLPTR;
    const uint32_t m = fwd ? t.next_mask : t.prev_mask;
    bool has = false;
    const int pc = popc4(m);
    if (pc == 1) {
        Node<W> x;
        node_child_located<W>(e, vt, t, fwd, lowbit4(m), x);
        uint64_t ex = x.idx >= 0 ? vt.tab[x.vslot] : 0ull;
        if (!vt_seen_e(ex) || s.n > 0) {                // :262
            cu.nxt = x;
            has = true;
            if (x.idx >= 0) vt.tab[x.vslot] = ex | (1ull << 63);   // seen.add(nextKmer)
        }
    } else if (pc > 1) {
        unsigned ch;
        if (ls_next_choice(e.links, s, &ch)) {          // getAdjacentKmer :518-546
            Kmer<W> cand = fwd ? kmer_next<W>(t.sk, e.g.k, ch) : kmer_prev<W>(t.sk, e.g.k, ch);
            int mb = -1;
            for (unsigned b = 0; b < 4; b++)
                if (((m >> b) & 1u) && kmer_eq<W>(node_neighbour<W>(t, e.g.k, fwd, b), cand)) mb = (int)b;
            if (mb >= 0) { node_child_located<W>(e, vt, t, fwd, (unsigned)mb, cu.nxt); has = true; }
        }
        ls_increment_ages(s);                           // :271
    }
    cu.has = has;
    if (ls_num_new(s) > 0) ls_increment_ages(s);        // :274-276 (Q12)
    if (s.overflow) cu.status = ST_LINKSTORE_FULL;
    return t;
}

}  // namespace ldbg
