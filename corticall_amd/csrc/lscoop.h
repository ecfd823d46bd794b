// Wave-cooperative LinkStore operations for the walk kernel.
//
// A lane that adds links or takes a junction choice scans its whole link store (LinkStore.java:17-35, 92-144).  Done by
// the lane alone, that scan is a chain of dependent loads while the other 63 lanes of the wavefront wait for it — and
// with 64 strands per wavefront some lane is at a junction in most iterations.  Here the WHOLE wavefront carries out one
// lane's (the owner's) operation: lane h handles elements h, h+64, ...; agreement, minimum and "last of the key" are
// ballots and butterfly reductions; expiry is a ballot-prefix compaction.  The semantics are those of the one-lane
// versions in engine.h (ls_add / ls_next_choice), which the cursor and dfs kernels keep using; both go through the same
// parity cases (tests/parity_cases.py).
#pragma once
#include "engine.h"
#include "strand.h"

namespace ldbg {

// where the link stores of this wavefront's lanes live
struct LsWave {
    LsElem* fast;          // element i of lane L at fast[i * stride + L]   (LDS)
    uint32_t stride;       // lanes per wavefront
    uint32_t fast_cap;
    LsElem* el;            // element i >= fast_cap of lane L at el[L * ecap + (i - fast_cap)]   (HBM)
    uint32_t ecap;
};
LDBG_DEV LsElem lsw_get(const LsWave& v, int L, uint32_t i) {
    if (i < v.fast_cap) return ls_elem_in(LDBG_LDS(const uint32_t, v.fast + (i * v.stride + (uint32_t)L)));
    return ls_elem_in(LDBG_GLOBAL(const uint32_t, v.el + ((size_t)L * v.ecap + (i - v.fast_cap))));
}
LDBG_DEV void lsw_set(const LsWave& v, int L, uint32_t i, const LsElem& x) {
    if (i < v.fast_cap) ls_elem_out(LDBG_LDS(uint32_t, v.fast + (i * v.stride + (uint32_t)L)), x);
    else ls_elem_out(LDBG_GLOBAL(uint32_t, v.el + ((size_t)L * v.ecap + (i - v.fast_cap))), x);
}

// the owner's store header, identical on every lane while the wavefront works on it
struct LsHdr { uint32_t n, java_cap, nkeys, next_seq, age, n_new, cap; bool overflow; };
LDBG_DEV LsHdr lsw_header(const LinkStoreDev& s, int L) {
    LsHdr h;
    h.n = wave_bcast_u32(s.n, L); h.java_cap = wave_bcast_u32(s.java_cap, L); h.nkeys = wave_bcast_u32(s.nkeys, L);
    h.next_seq = wave_bcast_u32(s.next_seq, L); h.age = wave_bcast_u32(s.age, L); h.n_new = wave_bcast_u32(s.n_new, L);
    h.cap = wave_bcast_u32(s.cap, L); h.overflow = wave_bcast_u32(s.overflow ? 1u : 0u, L) != 0;
    return h;
}
LDBG_DEV void lsw_store_header(LinkStoreDev& s, const LsHdr& h) {
    s.n = h.n; s.java_cap = h.java_cap; s.nkeys = h.nkeys; s.next_seq = h.next_seq; s.age = h.age; s.n_new = h.n_new; s.overflow = h.overflow;
}

// The junction records an add needs are fetched by the WHOLE wavefront in one load before the cooperative phase: the records of all
// owners of this iteration are dealt out to consecutive lanes (owner after owner, `start` = the lane that holds an owner's first
// record); the add then reads them out of the lanes' registers.  (Fetched record by record inside the add — the owner lane had only
// the first two — every further record was a dependent trip to memory that 63 lanes waited for: 1.6 us per owner, half of a general
// step, profiles/r03_walk_general_split.log.)  Records that find no lane (more than a wavefront's worth in one iteration) are read
// from memory as before.
#define LDBG_NOT_GATHERED 0xFFFFFFFFu
struct AddPre { uint32_t jlo, jhi, start; };
LDBG_DEV JuncRec bcast_junc(const JuncRec& r, int L) {
    JuncRec o;
    o.str_off = wave_bcast_u32(r.str_off, L); o.len = wave_bcast_u32(r.len, L);
    o.hash_asis = (int32_t)wave_bcast_u32((uint32_t)r.hash_asis, L); o.hash_comp = (int32_t)wave_bcast_u32((uint32_t)r.hash_comp, L);
    o.is_fw = wave_bcast_u32(r.is_fw, L);
    return o;
}

// LinkStore.add (:17-35) of a merged link record (junction records [pre.jlo, pre.jhi)) into the owner's store
LDBG_DEV void coop_add(const LinksView& Lk, const LsWave& v, int L, LsHdr& h, const AddPre& pre, const JuncRec& gathered, bool query_flipped, bool fwd) {
    const uint32_t lane = (uint32_t)wave_lane(), WS = (uint32_t)wave_size();
    for (uint32_t j = pre.jlo; j < pre.jhi; j++) {
        const JuncRec jr = pre.start != LDBG_NOT_GATHERED ? bcast_junc(gathered, (int)(pre.start + (j - pre.jlo))) : Lk.junc[j];
        const bool lgf = ((jr.is_fw & 1u) != 0) != query_flipped;     // recordOrientationMatchesKmer == cjr.isForward() :24
        if (lgf != fwd) continue;
        LsElem x;
        x.str_off = jr.str_off; x.birth = h.age; x.hash = lgf ? jr.hash_asis : jr.hash_comp;
        x.len = (uint16_t)jr.len; x.pos = 0; x.comp = lgf ? 0 : 1; x.key_seq = 0;
        ls_first_nx(jr, x);
        // the newest element filed under the same junction string, if any
        uint64_t found = 0;
        if (h.n <= WS && WS > 1) {                        // one element per lane: the highest matching lane
            bool match = false;
            uint32_t ks = 0;
            if (lane < h.n) { const LsElem y = lsw_get(v, L, lane); match = ls_same_string(Lk, y, x); ks = y.key_seq; }
            const unsigned long long mb = wave_ballot(match);
            if (mb) found = (1ull << 32) | wave_bcast_u32(ks, 63 - __builtin_clzll(mb));
        } else
        for (uint32_t base = 0; base < h.n; base += WS) {           // several elements per lane: later rounds hold newer elements
            const uint32_t i = base + lane;
            bool match = false;
            uint32_t ks = 0;
            if (i < h.n) { const LsElem y = lsw_get(v, L, i); match = ls_same_string(Lk, y, x); ks = y.key_seq; }
            const unsigned long long mb = wave_ballot(match);
            if (mb) found = (1ull << 32) | wave_bcast_u32(ks, 63 - __builtin_clzll(mb));
        }
        if (found) x.key_seq = (uint32_t)found;
        else {
            x.key_seq = h.next_seq++;
            h.nkeys++;
            if (h.java_cap == 0) h.java_cap = 16;
            if (h.nkeys > h.java_cap * 3 / 4) h.java_cap *= 2;
        }
        if (h.n >= h.cap || jr.len >= 65535u || h.n >= 0x7FFFu) { h.overflow = true; return; }
        if (lane == 0) lsw_set(v, L, h.n, x);
        wave_fence();
        h.n++;
        h.n_new++;
#ifdef LDBG_HOSTSIM
        ls_debug().adds++; if (h.n > ls_debug().maxn) ls_debug().maxn = h.n;
#endif
    }
}

LDBG_DEV bool lsw_keeps(const LsElem& x, unsigned ch) { return !((uint32_t)x.pos + 1 >= x.len || ls_cur(x) != ch); }

// LinkStore.getNextJunctionChoice (:122-144) with getOldestLink (:92-119) and incrementPositionsAndExpire (:58-90)
LDBG_DEV bool coop_next_choice(const LinksView& Lk, const LsWave& v, int L, LsHdr& h, unsigned* choice) {
    if (h.n == 0) return false;
    const uint32_t lane = (uint32_t)wave_lane(), WS = (uint32_t)wave_size();
    if (h.n <= WS && WS > 1) {
        // the usual case — one element per lane: the element is read once and everything below happens in registers
        const bool valid = lane < h.n;
        LsElem x;
        x.birth = 0; x.key_seq = 0; x.hash = 0; x.nx = 0; x.pos = 0; x.len = 0;
        if (valid) x = lsw_get(v, L, lane);
        const uint32_t minbirth0 = wave_bcast_u32(x.birth, 0);
        const unsigned c = ls_cur(x);
        const unsigned c0 = wave_bcast_u32(c, 0);
        const bool old = valid && x.birth == minbirth0;
        if (wave_ballot(old && c != c0) != 0ull) return false;
        // first of the oldest links in HashMap iteration order: the oldest are few, walk their lanes
        const uint32_t hh = (uint32_t)x.hash;
        const uint64_t mine = ((uint64_t)((hh ^ (hh >> 16)) & (h.java_cap - 1)) << 32) | x.key_seq;
        uint64_t best1 = ~0ull;
        for (unsigned long long ob = wave_ballot(old); ob; ob &= ob - 1) {
            const uint64_t o = wave_bcast_u64(mine, __builtin_ctzll(ob));
            best1 = o < best1 ? o : best1;
        }
        const uint32_t seq1 = (uint32_t)best1;
        // the last element filed under that key = the highest lane holding it
        const unsigned long long kbm = wave_ballot(valid && x.key_seq == seq1);
        const unsigned ch1 = wave_bcast_u32(c, 63 - __builtin_clzll(kbm));
        const bool keep = valid && lsw_keeps(x, ch1);
        unsigned long long db = wave_ballot(valid && !keep);
        while (db) {
            const int dl = __builtin_ctzll(db);
            db &= db - 1;
            const uint32_t ks = wave_bcast_u32(x.key_seq, dl);
            const bool hit = valid && x.key_seq == ks && (int)lane != dl && ((int)lane < dl || keep);
            if (wave_ballot(hit) == 0ull) h.nkeys--;
        }
        if (keep) ls_advance(Lk, x);
        const unsigned long long kb = wave_ballot(keep);
        if (keep) lsw_set(v, L, (uint32_t)wave_count_below(kb), x);
        h.n_new = (uint32_t)__builtin_popcountll(wave_ballot(keep && x.birth == h.age));
        h.n = (uint32_t)__builtin_popcountll(kb);
        wave_fence();
        *choice = ch1;
        return true;
    }
    const LsElem first = lsw_get(v, L, 0);
    const uint32_t minbirth = first.birth;          // oldest = largest age = smallest birth; births never decrease along the array
    const unsigned ch0 = ls_cur(first);
    // the oldest links must agree on their next junction; the first of them in java.util.HashMap iteration order
    // (bucket, key insertion order) names the key whose last element supplies the choice
    bool disagree = false;
    uint64_t best = ~0ull;
    for (uint32_t base = 0; base < h.n; base += WS) {
        const uint32_t i = base + lane;
        bool old = false, differs = false;
        uint64_t mine = ~0ull;
        if (i < h.n) {
            const LsElem x = lsw_get(v, L, i);
            old = x.birth == minbirth;
            differs = old && ls_cur(x) != ch0;
            const uint32_t hh = (uint32_t)x.hash;
            mine = ((uint64_t)((hh ^ (hh >> 16)) & (h.java_cap - 1)) << 32) | x.key_seq;
        }
        if (wave_ballot(differs) != 0ull) disagree = true;
        for (unsigned long long ob = wave_ballot(old); ob; ob &= ob - 1) {      // the oldest links are few: walk their lanes
            const uint64_t o = wave_bcast_u64(mine, __builtin_ctzll(ob));
            best = o < best ? o : best;
        }
        if (wave_ballot(i < h.n && !old) != 0ull) break;       // past the prefix of oldest links
    }
    if (disagree) return false;
    const uint32_t best_seq = (uint32_t)best;
    uint64_t last = 0;
    for (uint32_t base = 0; base < h.n; base += WS) {           // last element of that key's list wins (:129-133)
        const uint32_t i = base + lane;
        bool mine = false;
        unsigned c = 0;
        if (i < h.n) { const LsElem x = lsw_get(v, L, i); mine = x.key_seq == best_seq; c = ls_cur(x); }
        const unsigned long long kb = wave_ballot(mine);
        if (kb) last = wave_bcast_u32(c, 63 - __builtin_clzll(kb));          // a later round holds later elements
    }
    const unsigned ch = (unsigned)(last & 3ull);
    // keys whose last element expires leave the HashMap (:84-88): a dead element takes its key along unless a surviving
    // element, or an earlier dead one (already counted), shares it
    for (uint32_t base = 0; base < h.n; base += WS) {
        const uint32_t i = base + lane;
        LsElem x;
        x.key_seq = 0;
        bool dead = false;
        if (i < h.n) { x = lsw_get(v, L, i); dead = !lsw_keeps(x, ch); }
        unsigned long long db = wave_ballot(dead);
        while (db) {
            const int dl = __builtin_ctzll(db);
            db &= db - 1;
            const uint32_t ks = wave_bcast_u32(x.key_seq, dl);
            const uint32_t di = base + (uint32_t)dl;
            bool held = false;
            for (uint32_t b2 = 0; b2 < h.n && !held; b2 += WS) {
                const uint32_t j = b2 + lane;
                bool hit = false;
                if (j < h.n) {
                    const LsElem y = lsw_get(v, L, j);
                    hit = y.key_seq == ks && (j < di || lsw_keeps(y, ch)) && j != di;
                }
                held = wave_ballot(hit) != 0ull;
            }
            if (!held) h.nkeys--;
        }
    }
    // incrementPositionsAndExpire(choice): survivors advance and close ranks
    uint32_t w = 0, n_new = 0;
    for (uint32_t base = 0; base < h.n; base += WS) {
        const uint32_t i = base + lane;
        LsElem x;
        bool keep = false;
        if (i < h.n) { x = lsw_get(v, L, i); keep = lsw_keeps(x, ch); }
        if (keep) ls_advance(Lk, x);
        const unsigned long long kb = wave_ballot(keep);
        if (keep) lsw_set(v, L, w + (uint32_t)wave_count_below(kb), x);
        n_new += (uint32_t)__builtin_popcountll(wave_ballot(keep && x.birth == h.age));
        w += (uint32_t)__builtin_popcountll(kb);
        wave_fence();
    }
    h.n = w;
    h.n_new = n_new;
    *choice = ch;
    return true;
}

// ---- FOUR owners at a time.  The cooperative operations above give one owner all 64 lanes, yet a link store rarely holds more
// than a dozen elements: 48 lanes idle while the other owners of the iteration wait their turn (4.9 owners of adds and 2.7 of
// choices per general iteration at C3: 8 of its 14 us, profiles/r03_walk_general_split.log).  Here the wavefront works as four GROUPS
// of 16 lanes, each group carrying out one owner's operation — same instruction stream, per-group data: ballots are cut into
// 16-bit group masks, broadcasts are lane permutes (ds_bpermute) from a per-group source lane, the owner's header travels to
// its group and back the same way.  An owner qualifies when its store (after the adds) fits one element per lane of a group
// and all its junction records were gathered; the others take the whole-wavefront path as before.  Semantics: those of
// coop_add / the one-element-per-lane branch of coop_next_choice, statement by statement.
#define LDBG_GS 16u          // the wider of the two group sizes (8 and 16 lanes: eight or four owners at a time)
LDBG_DEV uint32_t grp_shfl(uint32_t v, uint32_t src_lane) { return wave_shfl_u32(v, (int)(src_lane & 63u)); }
template <uint32_t GS> LDBG_DEV uint32_t grp_mask(unsigned long long ballot, uint32_t g) { return (uint32_t)(ballot >> (GS * g)) & ((1u << GS) - 1u); }
struct GrpTake { uint32_t myL; bool gv; int myq; };
// the next (up to) four owners of `todo` (wave-uniform): which owner this lane's group works for, and — for an owner lane of this
// batch — the group that works for it (myq, else -1)
template <uint32_t GS> LDBG_DEV GrpTake grp_take(unsigned long long& todo) {
    const uint32_t lane = (uint32_t)wave_lane(), g = lane / GS;
    GrpTake t;
    t.myL = 0; t.gv = false; t.myq = -1;
#pragma unroll
    for (uint32_t q = 0; q < 64u / GS; q++) {
        if (!todo) break;
        const int L = __builtin_ctzll(todo);
        todo &= todo - 1;
        if (g == q) { t.myL = (uint32_t)L; t.gv = true; }
        if ((int)lane == L) t.myq = (int)q;
    }
    return t;
}

template <uint32_t GS> LDBG_DEV void group_adds(const LinksView& Lk, const LsWave& v, LinkStoreDev& ls, unsigned long long todo, uint64_t m_cur, uint64_t m_nxt,
                         uint32_t start_cur, uint32_t start_nxt, uint32_t flags, const JuncRec& gathered, StepPre& pre) {
    const uint32_t lane = (uint32_t)wave_lane(), g = lane / GS, sub = lane % GS;
    while (todo) {
        const GrpTake t = grp_take<GS>(todo);
        uint32_t n = grp_shfl(ls.n, t.myL), java_cap = grp_shfl(ls.java_cap, t.myL), nkeys = grp_shfl(ls.nkeys, t.myL), next_seq = grp_shfl(ls.next_seq, t.myL);
        uint32_t n_new = grp_shfl(ls.n_new, t.myL);
        const uint32_t age = grp_shfl(ls.age, t.myL), cap = grp_shfl(ls.cap, t.myL);
        const uint32_t cnt_cur = grp_shfl(m_cur == ~0ull ? 0u : (uint32_t)(m_cur >> 32), t.myL), cnt_nxt = grp_shfl(m_nxt == ~0ull ? 0u : (uint32_t)(m_nxt >> 32), t.myL);
        const uint32_t sc = grp_shfl(start_cur, t.myL), sn = grp_shfl(start_nxt, t.myL), fl = grp_shfl(flags, t.myL);
        const uint32_t R = t.gv ? cnt_cur + cnt_nxt : 0u;
        uint32_t maxR = wave_bcast_u32(R, 0);
#pragma unroll
        for (uint32_t q = 1; q < 64u / GS; q++) { const uint32_t rq = wave_bcast_u32(R, (int)(q * GS)); maxR = maxR > rq ? maxR : rq; }
        LsElem y;
        y.str_off = 0; y.birth = 0; y.hash = 0; y.key_seq = 0; y.len = 0; y.pos = 0; y.comp = 0; y.nxn = 0; y.nx = 0;
        if (t.gv && sub < n) y = lsw_get(v, (int)t.myL, sub);
        bool ovf = false;
        const bool fwd = (fl & 4u) != 0;
        for (uint32_t r = 0; r < maxR; r++) {
            const bool in_cur = r < cnt_cur;
            const uint32_t src = in_cur ? sc + r : sn + (r - cnt_cur);
            JuncRec jr;
            jr.str_off = grp_shfl(gathered.str_off, src); jr.len = grp_shfl(gathered.len, src);
            jr.hash_asis = (int32_t)grp_shfl((uint32_t)gathered.hash_asis, src); jr.hash_comp = (int32_t)grp_shfl((uint32_t)gathered.hash_comp, src);
            jr.is_fw = grp_shfl(gathered.is_fw, src);
            const bool qf = in_cur ? (fl & 1u) != 0 : (fl & 2u) != 0;
            const bool lgf = ((jr.is_fw & 1u) != 0) != qf;                        // :24
            const bool use = t.gv && r < R && !ovf && lgf == fwd;
            LsElem x;
            x.str_off = jr.str_off; x.birth = age; x.hash = lgf ? jr.hash_asis : jr.hash_comp;
            x.len = (uint16_t)jr.len; x.pos = 0; x.comp = lgf ? 0 : 1; x.key_seq = 0;
            ls_first_nx(jr, x);
            const bool match = use && sub < n && ls_same_string(Lk, y, x);
            const uint32_t gm = grp_mask<GS>(wave_ballot(match), g);
            const uint32_t top = gm ? 31u - (uint32_t)__builtin_clz(gm) : 0u;        // the newest element filed under the same junction string
            const uint32_t ks = grp_shfl(y.key_seq, (g * GS) + top);
            if (use) {
                if (gm) x.key_seq = ks;
                else {
                    x.key_seq = next_seq++;
                    nkeys++;
                    if (java_cap == 0) java_cap = 16;
                    if (nkeys > java_cap * 3 / 4) java_cap *= 2;
                }
                if (n >= cap || jr.len >= 65535u || n >= 0x7FFFu) ovf = true;
                else {
                    if (sub == n) { y = x; lsw_set(v, (int)t.myL, n, x); }
                    n++; n_new++;
                }
            }
        }
        wave_fence();
        // the headers go back to their owners
        const uint32_t back = t.myq >= 0 ? (uint32_t)t.myq * GS : lane;
        const uint32_t rn = grp_shfl(n, back), rcap = grp_shfl(java_cap, back), rkeys = grp_shfl(nkeys, back), rseq = grp_shfl(next_seq, back), rnew = grp_shfl(n_new, back);
        const uint32_t rovf = grp_shfl(ovf ? 1u : 0u, back);
        if (t.myq >= 0) {
            pre.n_added = (uint16_t)(rn - ls.n);
            ls.n = rn; ls.java_cap = rcap; ls.nkeys = rkeys; ls.next_seq = rseq; ls.n_new = rnew; ls.overflow = ls.overflow || rovf != 0u;
        }
    }
}

// getNextJunctionChoice for (up to) four owners at once; every owner in `todo` has 1 <= n <= 16 elements
template <uint32_t GS> LDBG_DEV void group_choices(const LinksView& Lk, const LsWave& v, LinkStoreDev& ls, unsigned long long todo, StepPre& pre) {
    const uint32_t lane = (uint32_t)wave_lane(), g = lane / GS, sub = lane % GS;
    while (todo) {
        const GrpTake t = grp_take<GS>(todo);
        const uint32_t n = grp_shfl(ls.n, t.myL), java_cap = grp_shfl(ls.java_cap, t.myL), age = grp_shfl(ls.age, t.myL);
        uint32_t nkeys = grp_shfl(ls.nkeys, t.myL);
        const bool valid = t.gv && sub < n;
        LsElem x;
        x.str_off = 0; x.birth = 0; x.hash = 0; x.key_seq = 0; x.len = 0; x.pos = 0; x.comp = 0; x.nxn = 0; x.nx = 0;
        if (valid) x = lsw_get(v, (int)t.myL, sub);
        const uint32_t minbirth0 = grp_shfl(x.birth, g * GS);                    // births never decrease along the array: element 0 is among the oldest
        const unsigned c = ls_cur(x);
        const unsigned c0 = grp_shfl(c, g * GS);
        const bool old = valid && x.birth == minbirth0;
        const uint32_t differ = grp_mask<GS>(wave_ballot(old && c != c0), g);          // (every lane takes part in every ballot: no short circuits around them)
        const bool ok = t.gv && differ == 0u;                                    // the oldest links must agree (:92-119)
        // first of the oldest links in HashMap iteration order (bucket, key insertion order): minimum over the group
        const uint32_t hh = (uint32_t)x.hash;
        uint64_t best = old ? (((uint64_t)((hh ^ (hh >> 16)) & (java_cap - 1u)) << 32) | x.key_seq) : ~0ull;
#pragma unroll
        for (int m = (int)GS / 2; m > 0; m >>= 1) {
            const uint32_t lo = wave_shfl_u32((uint32_t)best, (int)(lane ^ (uint32_t)m)), hi = wave_shfl_u32((uint32_t)(best >> 32), (int)(lane ^ (uint32_t)m));     // (stays inside the group: m < GS)
            const uint64_t o = ((uint64_t)hi << 32) | lo;
            best = o < best ? o : best;
        }
        const uint32_t seq1 = (uint32_t)best;
        const uint32_t kbm = grp_mask<GS>(wave_ballot(valid && x.key_seq == seq1), g);       // the last element filed under that key supplies the choice (:129-133)
        const unsigned ch1 = grp_shfl(c, (g * GS) + (kbm ? 31u - (uint32_t)__builtin_clz(kbm) : 0u));
        const bool keep = valid && lsw_keeps(x, ch1);
        // keys whose last element expires leave the HashMap (:84-88)
        const unsigned long long deadb = wave_ballot(ok && valid && !keep);
        const uint32_t dead_g = grp_mask<GS>(deadb, g);
        uint32_t any = 0;
#pragma unroll
        for (uint32_t q = 0; q < 64u / GS; q++) any |= grp_mask<GS>(deadb, q);
        while (any) {
            const uint32_t sd = (uint32_t)__builtin_ctz(any);
            any &= any - 1u;
            const uint32_t ks = grp_shfl(x.key_seq, (g * GS) + sd);
            const bool hit = valid && x.key_seq == ks && sub != sd && (sub < sd || keep);
            const uint32_t hm = grp_mask<GS>(wave_ballot(hit), g);
            if (((dead_g >> sd) & 1u) && hm == 0u) nkeys--;
        }
        LsElem xa = x;
        if (ok && keep) ls_advance(Lk, xa);
        const uint32_t kb = grp_mask<GS>(wave_ballot(ok && keep), g);
        if (ok && keep) lsw_set(v, (int)t.myL, (uint32_t)__builtin_popcount(kb & ((1u << sub) - 1u)), xa);
        const uint32_t nn = grp_mask<GS>(wave_ballot(ok && keep && x.birth == age), g);
        wave_fence();
        const uint32_t back = t.myq >= 0 ? (uint32_t)t.myq * GS : lane;
        const uint32_t rok = grp_shfl(ok ? 1u : 0u, back), rch = grp_shfl(ch1, back), rkeys = grp_shfl(nkeys, back);
        const uint32_t rn = grp_shfl((uint32_t)__builtin_popcount(kb), back), rnew = grp_shfl((uint32_t)__builtin_popcount(nn), back);
        if (t.myq >= 0) {
            pre.choice_done = true; pre.choice_ok = rok != 0u; pre.ch = rch;
            if (rok) { ls.n = rn; ls.n_new = rnew; ls.nkeys = rkeys; }
        }
    }
}

// The link-store part of one cursor step (TraversalEngine.java:241-276) for every lane of the wavefront that is in cursor
// mode: per-lane prefetch, cooperative adds, cooperative junction choices.  `pre` then carries the results into
// cursor_step<W, true>.  Every lane of the wavefront must call this (cur_mode false where it does not apply).
template <int W>
LDBG_DEV void coop_step_prepare(const EngineView& e, StrandState& st, LinkStoreDev& ls, const LsWave& lw, bool cur_mode, StepPre& pre, unsigned long long* tdiag = nullptr) {
    // Two independent chains of dependent loads start here: (links) rec_of -> offsets -> junction records of the
    // vertex about to be stepped onto, and (graph) its neighbour pointer -> the next row + its table slot.  They are
    // issued stage by stage so that they overlap.
    const uint32_t nmask = cur_mode ? (st.fwd ? st.cu.nxt.next_mask : st.cu.nxt.prev_mask) : 0u;
    const bool one_child = cur_mode && popc4(nmask) == 1;
    const bool flagged = cur_mode && (st.cu.nxt.lflags & e.link_flag_mask);
    uint64_t m_cur = ~0ull, m_nxt = ~0ull;
    uint32_t child_ent = 0;
    // the range of a flagged vertex's junction records: with a run index it came with the vertex (runs.cpp: k_run_link_info), else one more load
    const bool in_ui = e.runs.uinfo != nullptr;
    if (flagged) m_nxt = in_ui ? (st.cu.nxt.ui ? st.cu.nxt.ui : ~0ull) : e.links.rec_of[st.cu.nxt.idx];
    if (one_child) child_ent = st.cu.nxt.e1 ? st.cu.nxt.ent1 : node_child_entry(e, st.cu.nxt, st.fwd, lowbit4(nmask));
    if (cur_mode && st.cu.first && (st.cu.cur.lflags & e.link_flag_mask)) m_cur = in_ui ? (st.cu.cur.ui ? st.cu.cur.ui : ~0ull) : e.links.rec_of[st.cu.cur.idx];
    pre.links_done = true; pre.choice_done = false; pre.choice_ok = false; pre.ch = 0; pre.n_added = 0;
    // deal the owners' junction records out to the lanes (cur's records first, then nxt's: the order of the adds), one load for all
    const unsigned long long owners = wave_ballot(m_cur != ~0ull || m_nxt != ~0ull);
    uint32_t start_cur = LDBG_NOT_GATHERED, start_nxt = LDBG_NOT_GATHERED, my_j = LDBG_NOT_GATHERED;
    {
        const uint32_t lane = (uint32_t)wave_lane(), WS = (uint32_t)wave_size();
        uint32_t base = 0;
        for (unsigned long long ob = owners; ob; ob &= ob - 1) {
            const int L = __builtin_ctzll(ob);
#pragma unroll
            for (int which = 0; which < 2; which++) {
                const uint64_t m = wave_bcast_u64(which == 0 ? m_cur : m_nxt, L);
                if (m == ~0ull) continue;
                const uint32_t jlo = (uint32_t)m, cnt = (uint32_t)(m >> 32);
                if (base + cnt > WS) continue;                       // (no lanes left: these records are read inside the add)
                if (lane >= base && lane < base + cnt) my_j = jlo + (lane - base);
                if ((int)lane == L) { if (which == 0) start_cur = base; else start_nxt = base; }
                base += cnt;
            }
        }
    }
    JuncRec gathered;
    gathered.str_off = 0; gathered.len = 0; gathered.hash_asis = 0; gathered.hash_comp = 0; gathered.is_fw = 0;
    if (my_j != LDBG_NOT_GATHERED) gathered = e.links.junc[my_j];
    // (the graph chain goes on while the records are on their way)
    pre.has_child = one_child;
    if (one_child) node_from_entry(e, st.vt, st.cu.nxt, child_ent, lowbit4(nmask), st.fwd, pre.child);
    unsigned long long need = owners;
#ifdef LDBG_WALK_DIAG
    if (tdiag) { tdiag[0] = __builtin_amdgcn_s_memrealtime(); tdiag[2] = (unsigned long long)__builtin_popcountll(need); }
#endif
    if (wave_size() == 64 && lw.fast_cap >= LDBG_GS) {
        // owners whose store, with everything this step may add, is one element per lane of a 16-lane group: four of them at a time
        const uint32_t r_cur = m_cur == ~0ull ? 0u : (uint32_t)(m_cur >> 32), r_nxt = m_nxt == ~0ull ? 0u : (uint32_t)(m_nxt >> 32);
        const bool mine = (m_cur != ~0ull || m_nxt != ~0ull) && !ls.overflow && ls.n + r_cur + r_nxt <= LDBG_GS &&
                          (m_cur == ~0ull || start_cur != LDBG_NOT_GATHERED) && (m_nxt == ~0ull || start_nxt != LDBG_NOT_GATHERED);
        // (a store of at most 8 elements — 94 % of the adds at C3 — takes an 8-lane group: eight owners at a time)
        const unsigned long long grouped8 = wave_ballot(mine && ls.n + r_cur + r_nxt <= 8u);
        const unsigned long long grouped = wave_ballot(mine);
#ifdef LDBG_WALK_DIAG
        if (tdiag) {      // [3] owners of adds that take the 16-lane group path | those whose store would fit an 8-lane group << 16 | sum of their store sizes << 32
            const unsigned long long g8 = wave_ballot(mine && ls.n + r_cur + r_nxt <= 8u);
            const uint32_t nsum = wave_incl_scan_u32((m_cur != ~0ull || m_nxt != ~0ull) ? ls.n + r_cur + r_nxt : 0u);
            tdiag[3] = (unsigned long long)__builtin_popcountll(grouped) | ((unsigned long long)__builtin_popcountll(g8) << 16) | ((unsigned long long)wave_bcast_u32(nsum, 63) << 32);
        }
#endif
        if (grouped) {
            const uint32_t flags = (st.cu.cur.flip ? 1u : 0u) | (st.cu.nxt.flip ? 2u : 0u) | (st.fwd ? 4u : 0u);
            if (grouped8) group_adds<8>(e.links, lw, ls, grouped8, m_cur, m_nxt, start_cur, start_nxt, flags, gathered, pre);
            if (grouped & ~grouped8) group_adds<16>(e.links, lw, ls, grouped & ~grouped8, m_cur, m_nxt, start_cur, start_nxt, flags, gathered, pre);
            need &= ~grouped;
        }
    }
    while (need) {
        const int L = __builtin_ctzll(need);
        need &= need - 1;
        LsHdr h = lsw_header(ls, L);
        const uint32_t flags = wave_bcast_u32((st.cu.cur.flip ? 1u : 0u) | (st.cu.nxt.flip ? 2u : 0u) | (st.fwd ? 4u : 0u) |
                                              (m_cur != ~0ull ? 8u : 0u) | (m_nxt != ~0ull ? 16u : 0u), L);
        if (flags & 8u) {
            const uint64_t m = wave_bcast_u64(m_cur, L);
            const AddPre ap{(uint32_t)m, (uint32_t)m + (uint32_t)(m >> 32), wave_bcast_u32(start_cur, L)};
            coop_add(e.links, lw, L, h, ap, gathered, (flags & 1u) != 0, (flags & 4u) != 0);
        }
        if ((flags & 16u) && !h.overflow) {
            const uint64_t m = wave_bcast_u64(m_nxt, L);
            const AddPre ap{(uint32_t)m, (uint32_t)m + (uint32_t)(m >> 32), wave_bcast_u32(start_nxt, L)};
            coop_add(e.links, lw, L, h, ap, gathered, (flags & 2u) != 0, (flags & 4u) != 0);
        }
        if (wave_lane() == L) { pre.n_added = (uint16_t)(h.n - ls.n); lsw_store_header(ls, h); }
    }
    need = wave_ballot(cur_mode && popc4(nmask) > 1);
#ifdef LDBG_WALK_DIAG
    if (tdiag) { tdiag[1] = __builtin_amdgcn_s_memrealtime(); tdiag[2] |= (unsigned long long)__builtin_popcountll(need) << 32; }
#endif
    if (wave_size() == 64 && lw.fast_cap >= LDBG_GS) {
        const unsigned long long grouped = wave_ballot(cur_mode && popc4(nmask) > 1 && ls.n >= 1u && ls.n <= LDBG_GS);
        const unsigned long long grouped8 = wave_ballot(cur_mode && popc4(nmask) > 1 && ls.n >= 1u && ls.n <= 8u);
        if (grouped8) group_choices<8>(e.links, lw, ls, grouped8, pre);
        if (grouped & ~grouped8) group_choices<16>(e.links, lw, ls, grouped & ~grouped8, pre);
        need &= ~grouped;
    }
    while (need) {                                    // junction choices (:266-272)
        const int L = __builtin_ctzll(need);
        need &= need - 1;
        LsHdr h = lsw_header(ls, L);
        unsigned ch = 0;
        const bool ok = coop_next_choice(e.links, lw, L, h, &ch);
        if (wave_lane() == L) { lsw_store_header(ls, h); pre.choice_done = true; pre.choice_ok = ok; pre.ch = ch; }
    }
}

// ---- the lean step.  Nearly all iterations of a link-guided walk or dfs branch are the same case: the cursor has a next
// vertex, that vertex has one successor with a record, no link annotations, no junction, no quirk.  The general step
// (coop_step_prepare + cursor_step + the kernel's own step) spends ~1000 instructions per iteration on its generality (PMC:
// profiles/r01_walk_instructions.log) and a walk cannot go faster than its own instruction stream; this is the same
// sequence of table reads and writes with the decisions taken out.  Any lane for which lean_cursor_ok() is false takes
// the general step in the same iteration.
LDBG_DEV bool lean_cursor_ok(const EngineView& e, const StrandState& st) {
    const Cursor& cu = st.cu;
    const Node& cv = st.cv;
    const Node& t = cu.nxt;
    if (!(st.status == ST_OK && e.cursor_on && cu.has && !cu.first && (e.g.k & 1))) return false;   // odd k: no palindromic k-mers
    const int acopy = cv.copy < 0 ? -cv.copy : cv.copy;
    return cv.idx >= 0 && t.idx >= 0 && !t.npe && cv.flip == cv.fj && t.flip == t.fj   // records present, no quirk-Q6 vertex
        && t.e1 && (t.ent1 & 0x7FFFFFFFu) != 0u && (t.ent1 & 0x7FFFFFFFu) != 0x7FFFFFFFu   // exactly one successor, it has a record (and its row is here: image.h)
        && !(t.lflags & e.link_flag_mask)                                           // no links to add
        && cv.vslot != t.vslot                                                      // not standing on the vertex it looks at
        && acopy >= vt_count_e(cv.vent) && acopy + 1 <= 32767;                      // cv not visited before, copies in range
}
// between two lean steps of a run: what a lean step can change
LDBG_DEV bool lean_cursor_again(const EngineView& e, const StrandState& st) {
    const Node& cv = st.cv;
    const Node& t = st.cu.nxt;
    const int acopy = cv.copy < 0 ? -cv.copy : cv.copy;
    return st.cu.has && !t.npe && t.flip == t.fj && t.e1 && (t.ent1 & 0x7FFFFFFFu) != 0u && (t.ent1 & 0x7FFFFFFFu) != 0x7FFFFFFFu
        && !(t.lflags & e.link_flag_mask) && cv.vslot != t.vslot && acopy >= vt_count_e(cv.vent) && acopy + 1 <= 32767;
}
// next()/previous() (TraversalEngine.java:241-279) onto cu.nxt with its only successor looked up one step ahead, then
// visited.add(cv) (:425).  Returns the vertex stepped onto with its copyIndex (:383-389); the caller connects it and advances.
template <int W>
LDBG_DEV Node lean_cursor_advance(const EngineView& e, StrandState& st, LinkStoreDev& ls, uint32_t* marks = nullptr) {
    const bool fwd = st.fwd;
    st.iters++;
    Node& cv = st.cv;
    Node av = st.cu.nxt;
    Node x;
    if (e.lean_rows) node_from_entry<true>(e, st.vt, av, av.ent1, lowbit4(fwd ? av.next_mask : av.prev_mask), fwd, x);
    else node_from_entry(e, st.vt, av, av.ent1, lowbit4(fwd ? av.next_mask : av.prev_mask), fwd, x);
    // lean_cursor_ok() made sure cv and av are different vertices; x can be either of them (a walk turning round on a 1- or
    // 2-cycle): only then do the cached table entries need patching (node_sync) — the order of reads and writes is the general step's
    const bool alias = x.vslot == cv.vslot || x.vslot == av.vslot;
    bool has = false;
    const bool seen = vt_seen_e(x.vent, st.cu.epoch);
    if (!seen || ls.n > 0) {                           // :262
        if (!seen) { node_store(st.vt, x, vt_with_seen(x.vent, st.cu.epoch)); if (marks) ++*marks; }
        has = true;
    }
    if (ls_num_new(ls) > 0) ls_increment_ages(ls);     // :274-276 (Q12)
    if (alias && has) { node_sync(cv, x); node_sync(av, x); }
    const int cnt = node_count(av);
    av.copy = fwd ? cnt : -cnt;
    const int acopy = cv.copy < 0 ? -cv.copy : cv.copy;
    node_store(st.vt, cv, vt_with_count(cv.vent, acopy + 1));
    if (alias && has) node_sync(x, cv);
    st.cu.has = has;
    if (has) st.cu.nxt = x;
    return av;
}

}  // namespace ldbg
