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

// what an owner lane fetched for its add before the wavefront turns to it (all owners fetch at once)
struct AddPre { uint32_t jlo, jhi; JuncRec r0, r1; };
LDBG_DEV AddPre add_prefetch(const LinksView& Lk, uint64_t m) {      // m = rec_of entry: first | count << 32
    AddPre p;
    p.jlo = (uint32_t)m; p.jhi = (uint32_t)m + (uint32_t)(m >> 32);
    p.r0 = Lk.junc[p.jlo];
    p.r1 = Lk.junc[p.jlo + 1 < p.jhi ? p.jlo + 1 : p.jlo];
    return p;
}
LDBG_DEV JuncRec bcast_junc(const JuncRec& r, int L) {
    JuncRec o;
    o.str_off = wave_bcast_u32(r.str_off, L); o.len = wave_bcast_u32(r.len, L);
    o.hash_asis = (int32_t)wave_bcast_u32((uint32_t)r.hash_asis, L); o.hash_comp = (int32_t)wave_bcast_u32((uint32_t)r.hash_comp, L);
    o.is_fw = wave_bcast_u32(r.is_fw, L);
    return o;
}
LDBG_DEV AddPre bcast_addpre(const AddPre& p, int L) {
    AddPre o;
    o.jlo = wave_bcast_u32(p.jlo, L); o.jhi = wave_bcast_u32(p.jhi, L);
    o.r0 = bcast_junc(p.r0, L); o.r1 = bcast_junc(p.r1, L);
    return o;
}

// LinkStore.add (:17-35) of a merged link record (junction records [pre.jlo, pre.jhi)) into the owner's store
LDBG_DEV void coop_add(const LinksView& Lk, const LsWave& v, int L, LsHdr& h, const AddPre& pre, bool query_flipped, bool fwd) {
    const uint32_t lane = (uint32_t)wave_lane(), WS = (uint32_t)wave_size();
    for (uint32_t j = pre.jlo; j < pre.jhi; j++) {
        const JuncRec jr = j == pre.jlo ? pre.r0 : (j == pre.jlo + 1 ? pre.r1 : Lk.junc[j]);
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

// The link-store part of one cursor step (TraversalEngine.java:241-276) for every lane of the wavefront that is in cursor
// mode: per-lane prefetch, cooperative adds, cooperative junction choices.  `pre` then carries the results into
// cursor_step<W, true>.  Every lane of the wavefront must call this (cur_mode false where it does not apply).
template <int W>
LDBG_DEV void coop_step_prepare(const EngineView& e, StrandState& st, LinkStoreDev& ls, const LsWave& lw, bool cur_mode, StepPre& pre) {
    // Two independent chains of dependent loads start here: (links) rec_of -> offsets -> junction records of the
    // vertex about to be stepped onto, and (graph) its neighbour pointer -> the next row + its table slot.  They are
    // issued stage by stage so that they overlap.
    const uint32_t nmask = cur_mode ? (st.fwd ? st.cu.nxt.next_mask : st.cu.nxt.prev_mask) : 0u;
    const bool one_child = cur_mode && popc4(nmask) == 1;
    const bool flagged = cur_mode && (st.cu.nxt.lflags & e.link_flag_mask);
    uint64_t m_cur = ~0ull, m_nxt = ~0ull;
    uint32_t child_ent = 0;
    if (flagged) m_nxt = e.links.rec_of[st.cu.nxt.idx];
    if (one_child) child_ent = st.cu.nxt.e1 ? st.cu.nxt.ent1 : node_child_entry(e, st.cu.nxt, st.fwd, lowbit4(nmask));
    if (cur_mode && st.cu.first && (st.cu.cur.lflags & e.link_flag_mask)) m_cur = e.links.rec_of[st.cu.cur.idx];
    AddPre ap_cur, ap_nxt;      // read only by the lanes that filled them (the flags below say which)
    if (m_nxt != ~0ull) { ap_nxt.jlo = (uint32_t)m_nxt; ap_nxt.jhi = (uint32_t)m_nxt + (uint32_t)(m_nxt >> 32); }
    pre.has_child = one_child;
    if (one_child) node_from_entry(e, st.vt, st.cu.nxt, child_ent, lowbit4(nmask), st.fwd, pre.child);
    if (m_nxt != ~0ull) {
        ap_nxt.r0 = e.links.junc[ap_nxt.jlo];
        ap_nxt.r1 = e.links.junc[ap_nxt.jlo + 1 < ap_nxt.jhi ? ap_nxt.jlo + 1 : ap_nxt.jlo];
    }
    if (m_cur != ~0ull) ap_cur = add_prefetch(e.links, m_cur);
    pre.links_done = true; pre.choice_done = false; pre.choice_ok = false; pre.ch = 0;
    unsigned long long need = wave_ballot(m_cur != ~0ull || m_nxt != ~0ull);
    while (need) {
        const int L = __builtin_ctzll(need);
        need &= need - 1;
        LsHdr h = lsw_header(ls, L);
        const uint32_t flags = wave_bcast_u32((st.cu.cur.flip ? 1u : 0u) | (st.cu.nxt.flip ? 2u : 0u) | (st.fwd ? 4u : 0u) |
                                              (m_cur != ~0ull ? 8u : 0u) | (m_nxt != ~0ull ? 16u : 0u), L);
        if (flags & 8u) coop_add(e.links, lw, L, h, bcast_addpre(ap_cur, L), (flags & 1u) != 0, (flags & 4u) != 0);
        if ((flags & 16u) && !h.overflow) coop_add(e.links, lw, L, h, bcast_addpre(ap_nxt, L), (flags & 2u) != 0, (flags & 4u) != 0);
        if (wave_lane() == L) lsw_store_header(ls, h);
    }
    need = wave_ballot(cur_mode && popc4(nmask) > 1);
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
