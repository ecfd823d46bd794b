// TraversalEngine on the device: shared device-side primitives (neighbourhood of an oriented k-mer,
// per-walk visited table, per-walk LinkStore, cursor step) used by the walk and DFS kernels.
// Every function cites the Java it restates (J/ = public/java/src/uk/ac/ox/well/cortexjdk/).
#pragma once
#include "graph.h"
#include "links.h"

namespace ldbg {

#define LDBG_MAX_LINKS 4

struct EngineView {
    GraphView g;
    uint32_t trav_mask, recruit_mask, join_mask;
    int first_trav;
    int stopper, max_len, connect_all, strict_flip;
    int cursor_on;   // !ec.getLinks().isEmpty(): dfs drives the cursor (TraversalEngine.java:363, 379) even when
                     // none of the configured link sets belongs to a traversal sample (nlinks == 0)
    int nlinks;
    LinksView links[LDBG_MAX_LINKS];
};

// strand status codes (per seed and direction)
enum : uint32_t {
    ST_OK = 0,
    ST_NULLPTR = 1,        // the reference throws NullPointerException here
    ST_LINKSTORE_FULL = 2, // per-walk link store capacity exceeded -> host retries with a larger store
    ST_BRANCH_NULL = 3,    // dfs branch returned null
    ST_COPY_OVERFLOW = 4,  // more than 32767 copies of one vertex
    ST_PATH_FULL = 5
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

// ---- neighbourhood of an oriented k-mer: TraversalUtils.getAllNextKmers/getAllPrevKmers
// (J/utils/traversal/TraversalUtils.java:510-590) + TraversalEngine.getNextVertices/getPrevVertices
// (J/utils/traversal/TraversalEngine.java:147-239), as 4-bit base masks.
template <int W>
struct Adj {
    int64_t idx;        // record of the k-mer, -1 = null
    bool flip;          // k-mer != canonical orientation (by comparison): vertex identity
    Kmer<W> o;          // orientation the neighbours are built from (record k-mer, or its revcomp when
                        // CanonicalKmer.isFlipped() — hash-based, quirk Q6)
    uint32_t next_mask; // bit b set: successor o[1:]+b
    uint32_t prev_mask; // bit b set: predecessor b+o[:-1]
    bool npe;           // record missing while recruitment colours are set (Q14)
};

template <int W>
LDBG_HOSTDEV void adj_from_idx(const EngineView& e, const Kmer<W>& sk, const Kmer<W>& canon, bool flip_cmp, int64_t idx, Adj<W>& a) {
    const GraphView& g = e.g;
    bool fj = flip_cmp;
    if (e.strict_flip && flip_cmp) fj = kmer_java_hash<W>(canon, g.k) != kmer_java_hash<W>(sk, g.k);
    a.idx = idx;
    a.flip = flip_cmp;
    a.o = fj ? sk : canon;
    a.npe = false;
    uint32_t tf = 0, tr = 0, rf = 0, rr = 0;
    if (idx >= 0) {
        const uint8_t* ed = graph_row(g, idx) + g.edges_off;
        for (int c = 0; c < g.C; c++) {
            uint32_t ebyte = ed[c];
            uint32_t lo = ebyte & 0xf, hi = ebyte >> 4;
            // CortexRecord.getOutEdgesAsBytes: bit i <-> base i ; getInEdgesAsBytes: bit (3-i) <-> base i ;
            // complement=true relabels base b as 3-b (CortexRecord.java:214-275)
            uint32_t fwd = !fj ? lo : hi;                      // successor base = bit position
            uint32_t revn = !fj ? hi : lo;                     // predecessor base = 3 - bit position
            uint32_t rev = ((revn & 1u) << 3) | ((revn & 2u) << 1) | ((revn & 4u) >> 1) | ((revn & 8u) >> 3);
            if ((e.trav_mask >> c) & 1u) { tf |= fwd; tr |= rev; }
            if ((e.recruit_mask >> c) & 1u) { rf |= fwd; rr |= rev; }
        }
    } else if (e.recruit_mask != 0) {
        a.npe = true;
    }
    a.next_mask = tf ? tf : rf;
    a.prev_mask = tr ? tr : rr;
}

template <int W>
LDBG_HOSTDEV void adj_lookup(const EngineView& e, const Kmer<W>& sk, Adj<W>& a) {
    bool fc;
    Kmer<W> c = kmer_canonical<W>(sk, e.g.k, &fc);
    int64_t idx = graph_find_canonical<W>(e.g, c);
    adj_from_idx<W>(e, sk, c, fc, idx, a);
}

// a vertex reference carried between iterations
template <int W>
struct VRef {
    Kmer<W> sk;
    int64_t idx;
    bool flip;
    int copy;
};

template <int W>
LDBG_HOSTDEV VRef<W> vref_find(const EngineView& e, const Kmer<W>& sk) {
    VRef<W> v;
    v.sk = sk;
    bool fc;
    Kmer<W> c = kmer_canonical<W>(sk, e.g.k, &fc);
    v.idx = graph_find_canonical<W>(e.g, c);
    v.flip = fc;
    v.copy = 0;
    return v;
}
template <int W>
LDBG_HOSTDEV void adj_of(const EngineView& e, const VRef<W>& v, Adj<W>& a) {
    bool fc;
    Kmer<W> c = kmer_canonical<W>(v.sk, e.g.k, &fc);
    adj_from_idx<W>(e, v.sk, c, fc, v.idx, a);
}
template <int W>
LDBG_HOSTDEV Kmer<W> neighbour(const Adj<W>& a, int k, bool fwd, unsigned base) {
    return fwd ? kmer_next<W>(a.o, k, base) : kmer_prev<W>(a.o, k, base);
}
LDBG_HOSTDEV int popc4(uint32_t m) { return (int)((m & 1u) + ((m >> 1) & 1u) + ((m >> 2) & 1u) + ((m >> 3) & 1u)); }
LDBG_HOSTDEV unsigned lowbit4(uint32_t m) { return (m & 1u) ? 0u : ((m & 2u) ? 1u : ((m & 4u) ? 2u : 3u)); }

// ---- per-walk visited table (HashSet<CortexVertex> visited, TraversalEngine.java:360-425, plus the
// cursor's `seen` set :27,262-265): open addressing over 8-byte entries in HBM, generation-tagged so
// a slot is reused by the next walk without clearing.
//  bits 0..32 key = (record index << 1) | flip ; bits 33..47 generation ; bits 48..62 copies visited ; bit 63 seen
struct VisitedTable {
    uint64_t* tab;
    uint32_t mask;   // capacity - 1
    uint32_t gen;    // 1..32767
};
LDBG_HOSTDEV uint32_t vt_hash(uint64_t key) {
    uint64_t x = key * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(x >> 32);
}
LDBG_HOSTDEV uint64_t vt_key(int64_t idx, bool flip) { return ((uint64_t)idx << 1) | (flip ? 1ull : 0ull); }
// returns slot position of the key, or of the first free slot
LDBG_HOSTDEV uint32_t vt_slot(const VisitedTable& t, uint64_t key, bool* found) {
    uint32_t h = vt_hash(key) & t.mask;
    while (true) {
        uint64_t e = t.tab[h];
        if (((e >> 33) & 0x7FFFull) != t.gen) { *found = false; return h; }
        if ((e & 0x1FFFFFFFFull) == key) { *found = true; return h; }
        h = (h + 1) & t.mask;
    }
}
LDBG_HOSTDEV int vt_count(const VisitedTable& t, int64_t idx, bool flip) {
    if (idx < 0) return 0;
    bool f;
    uint32_t s = vt_slot(t, vt_key(idx, flip), &f);
    return f ? (int)((t.tab[s] >> 48) & 0x7FFFull) : 0;
}
LDBG_HOSTDEV bool vt_seen(const VisitedTable& t, int64_t idx, bool flip) {
    if (idx < 0) return false;
    bool f;
    uint32_t s = vt_slot(t, vt_key(idx, flip), &f);
    return f ? (t.tab[s] >> 63) != 0 : false;
}
LDBG_HOSTDEV void vt_update(VisitedTable& t, int64_t idx, bool flip, int new_count, bool set_seen) {
    if (idx < 0) return;
    uint64_t key = vt_key(idx, flip);
    bool f;
    uint32_t s = vt_slot(t, key, &f);
    uint64_t e = f ? t.tab[s] : (key | ((uint64_t)t.gen << 33));
    if (new_count >= 0) e = (e & ~(0x7FFFull << 48)) | ((uint64_t)(new_count & 0x7FFF) << 48);
    if (set_seen) e |= 1ull << 63;
    t.tab[s] = e;
}

// ---- per-walk LinkStore (J/utils/traversal/LinkStore.java), elements kept in insertion order
struct LsElem {
    uint32_t jrec;     // index into LinksView.junc of link set `set`
    uint32_t age;
    uint32_t key_seq;  // insertion sequence number of this element's key in the Java HashMap
    uint16_t pos;
    uint8_t set;
    uint8_t comp;      // junction string is used complemented (LinkStore.java:25)
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
LDBG_HOSTDEV unsigned ls_char(const EngineView& e, const LsElem& x, uint32_t i) {
    unsigned b = e.links[x.set].bases[e.links[x.set].junc[x.jrec].str_off + i];
    return x.comp ? 3u - b : b;
}
LDBG_HOSTDEV uint32_t ls_len(const EngineView& e, const LsElem& x) { return e.links[x.set].junc[x.jrec].len; }
LDBG_HOSTDEV int32_t ls_hash(const EngineView& e, const LsElem& x) {
    const JuncRec& j = e.links[x.set].junc[x.jrec];
    return x.comp ? j.hash_comp : j.hash_asis;
}
LDBG_HOSTDEV bool ls_same_string(const EngineView& e, const LsElem& a, const LsElem& b) {
    if (a.set == b.set && a.jrec == b.jrec && a.comp == b.comp) return true;
    uint32_t la = ls_len(e, a);
    if (la != ls_len(e, b) || ls_hash(e, a) != ls_hash(e, b)) return false;
    for (uint32_t i = 0; i < la; i++) if (ls_char(e, a, i) != ls_char(e, b, i)) return false;
    return true;
}
// LinkStore.add :17-35 for link record m of set `set`; `matches` = record k-mer string equals the cursor k-mer
LDBG_HOSTDEV void ls_add(const EngineView& e, LinkStoreDev& s, int set, int64_t m, bool matches, bool fwd) {
    const LinksView& L = e.links[set];
    for (uint32_t j = L.off[m]; j < L.off[m + 1]; j++) {
        bool is_fw = L.junc[j].is_fw != 0;
        bool lgf = matches == is_fw;
        if (lgf != fwd) continue;
        LsElem x;
        x.jrec = j; x.age = 0; x.pos = 0; x.set = (uint8_t)set; x.comp = lgf ? 0 : 1; x.key_seq = 0;
        bool have = false;
        for (uint32_t i = 0; i < s.n; i++)
            if (ls_same_string(e, s.el[i], x)) { x.key_seq = s.el[i].key_seq; have = true; break; }
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
LDBG_HOSTDEV bool ls_next_choice(const EngineView& e, LinkStoreDev& s, unsigned* choice) {
    if (s.n == 0) return false;
    uint32_t maxage = 0;
    for (uint32_t i = 0; i < s.n; i++) if (s.el[i].age > maxage) maxage = s.el[i].age;
    // first oldest element in java.util.HashMap iteration order: (bucket, key insertion order, list order)
    bool have = false, agree = true;
    unsigned ch0 = 0;
    uint32_t best_b = 0, best_seq = 0, best_i = 0;
    for (uint32_t i = 0; i < s.n; i++) {
        const LsElem& x = s.el[i];
        if (x.age != maxage) continue;
        unsigned c = ls_char(e, x, x.pos);
        uint32_t h = (uint32_t)ls_hash(e, x);
        uint32_t b = (h ^ (h >> 16)) & (s.java_cap - 1);
        if (!have) { have = true; ch0 = c; best_b = b; best_seq = x.key_seq; best_i = i; }
        else {
            if (c != ch0) agree = false;
            if (b < best_b || (b == best_b && x.key_seq < best_seq)) { best_b = b; best_seq = x.key_seq; best_i = i; }
        }
    }
    if (!have || !agree) return false;
    (void)best_i;
    unsigned ch = ch0;
    for (uint32_t i = 0; i < s.n; i++)   // last element of that key's list wins (:129-133)
        if (s.el[i].key_seq == best_seq) ch = ls_char(e, s.el[i], s.el[i].pos);
    // incrementPositionsAndExpire(choice)
    uint32_t w = 0;
    for (uint32_t i = 0; i < s.n; i++) {
        LsElem x = s.el[i];
        if ((uint32_t)x.pos + 1 >= ls_len(e, x) || ls_char(e, x, x.pos) != ch) continue;
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
    VRef<W> cur;
    Adj<W> acur;        // neighbourhood of cur
    VRef<W> nxt;        // the k-mer hasNext()/hasPrevious() refers to, looked up one step ahead
    bool has;
    bool first;         // specificLinksFiles == null: the next step re-seeks and initialises the link store
    uint32_t status;
};

template <int W>
LDBG_HOSTDEV void cursor_add_links(const EngineView& e, LinkStoreDev& s, const VRef<W>& v, bool fwd) {
    bool fc;
    Kmer<W> c = kmer_canonical<W>(v.sk, e.g.k, &fc);
    for (int L = 0; L < e.nlinks; L++) {
        int64_t m = links_find<W>(e.links[L], e.g.k, c);
        if (m >= 0) {
            // recordOrientationMatchesKmer (LinkStore.java:18): the record's k-mer string is canon or rc(canon)
            bool rec_is_canon = e.links[L].kcanon[m] != 0;
            bool matches = rec_is_canon ? !fc : fc;
            ls_add(e, s, L, m, matches, fwd);
        }
    }
}
// seek(sk): cursor on v, unique neighbour in direction `fwd` looked up (TraversalEngine.java:321-335)
template <int W>
LDBG_HOSTDEV void cursor_seek(const EngineView& e, Cursor<W>& cu, LinkStoreDev& s, const VRef<W>& v, const Adj<W>& a, bool fwd) {
    cu.cur = v;
    cu.acur = a;
    cu.first = true;
    cu.status = ST_OK;
    ls_clear(s);
    uint32_t m = fwd ? a.next_mask : a.prev_mask;
    cu.has = popc4(m) == 1;
    if (cu.has) cu.nxt = vref_find<W>(e, neighbour<W>(a, e.g.k, fwd, lowbit4(m)));
}
// next()/previous() (TraversalEngine.java:241-319); requires cu.has.  Returns the vertex stepped onto.
template <int W>
LDBG_HOSTDEV VRef<W> cursor_step(const EngineView& e, Cursor<W>& cu, LinkStoreDev& s, VisitedTable& vt, bool fwd) {
    if (cu.first) {
        cu.first = false;                              // seek(cur) recomputes the same state; then
        cursor_add_links<W>(e, s, cu.cur, fwd);        // initializeLinkStore :548-568
    }
    cursor_add_links<W>(e, s, cu.nxt, fwd);            // updateLinkStore :570-597
    VRef<W> t = cu.nxt;
    cu.cur = t;
    adj_of<W>(e, t, cu.acur);
    if (cu.acur.npe) cu.status = ST_NULLPTR;
    uint32_t m = fwd ? cu.acur.next_mask : cu.acur.prev_mask;
    bool has = false;
    int pc = popc4(m);
    if (pc == 1) {
        VRef<W> x = vref_find<W>(e, neighbour<W>(cu.acur, e.g.k, fwd, lowbit4(m)));
        if (!vt_seen(vt, x.idx, x.flip) || s.n > 0) {   // :262
            cu.nxt = x;
            has = true;
            vt_update(vt, x.idx, x.flip, -1, true);     // seen.add(nextKmer)
        }
    } else if (pc > 1) {
        unsigned ch;
        if (ls_next_choice(e, s, &ch)) {                // getAdjacentKmer :518-546
            Kmer<W> cand = fwd ? kmer_next<W>(t.sk, e.g.k, ch) : kmer_prev<W>(t.sk, e.g.k, ch);
            bool member = false;
            for (unsigned b = 0; b < 4; b++)
                if ((m >> b) & 1u) member |= kmer_eq<W>(neighbour<W>(cu.acur, e.g.k, fwd, b), cand);
            if (member) { cu.nxt = vref_find<W>(e, cand); has = true; }
        }
        ls_increment_ages(s);                           // :271
    }
    cu.has = has;
    if (ls_num_new(s) > 0) ls_increment_ages(s);        // :274-276 (Q12)
    if (s.overflow) cu.status = ST_LINKSTORE_FULL;
    return t;
}

}  // namespace ldbg
