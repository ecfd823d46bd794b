// Run steps of the walk kernel: crossing an unbranched stretch of the graph (a piece of a chain of the run index, runs.h) in ONE
// step, with exactly the effects the loop at TraversalEngine.java:373-481 has when it is taken through the stretch k-mer by k-mer.
//
// A piece is a range of consecutive positions of the run index; seen in travel order its vertices are q_0 .. q_{n-1}.  The walk's
// own seed cuts the piece it lies in into [start, seed-1], [seed], [seed+1, end] (a walk STARTS at its seed, so for this walk the
// stretch before the seed and the one after it have different histories).  Per-vertex state (visited copies, the cursor's `seen`
// mark) is kept
//   * explicitly, in the strand's visited table under the vertex key, for the FRINGE q_0, q_1, q_{n-2}, q_{n-1} (and for every
//     vertex of a piece shorter than LDBG_RUN_MIN, and every vertex outside the chains);
//   * in ONE table entry under the piece's key for the INTERIOR q_2 .. q_{n-3}: the interior can only be entered from q_1 (each
//     q_i's only predecessor is q_{i-1}), and once entered it is crossed to the end or the walk ends inside it, so all interior
//     vertices always have the same number of visited copies and the same `seen` mark.
// The k-mer-by-k-mer code (lean and general steps) therefore never looks up an interior vertex: whenever the walk stands at
// (cv = q_0, next = q_1) with the cursor [mode A], or at cv = q_1 without it [mode B], the run step below is taken instead, and in
// the few situations it leaves alone the walk provably ends in that very iteration, or the strand is handed back to the host to be
// walked again without the index (ST_RETRY_PLAIN) — results never depend on which path produced them.
#pragma once
#include "lscoop.h"
#include "runs.h"
#include "strand.h"

namespace ldbg {

#define LDBG_RUN_MIN 8u          // pieces shorter than this are walked k-mer by k-mer

struct Piece {
    uint32_t S, E;      // positions of the (seed-cut) piece
    uint32_t q, n;      // travel index of the vertex, vertices in the piece
    bool plus, asc;     // the vertex is the chain-orientation member of its record; travel goes towards larger positions
};
// a cut: the vertex at position `cut` is a piece of its own (a walk's seed, the first vertex of a dfs branch)
LDBG_DEV void piece_cut(uint32_t& S, uint32_t& E, uint32_t pos, uint32_t cut) {
    if (cut >= S && cut <= E) {                         // (LDBG_RUN_NONE lies in no piece)
        if (pos < cut) E = cut - 1u;
        else if (pos > cut) S = cut + 1u;
        else S = E = pos;
    }
}
LDBG_DEV Piece piece_make(uint64_t ui, bool flip, bool fwd, uint32_t S, uint32_t E) {
    Piece p;
    const uint32_t pos = ui_pos(ui);
    p.S = S; p.E = E;
    p.plus = flip == ui_orient(ui);
    p.asc = p.plus == fwd;
    p.q = p.asc ? pos - p.S : p.E - pos;
    p.n = p.E - p.S + 1u;
    return p;
}
LDBG_DEV Piece piece_of(uint64_t ui, bool flip, bool fwd, uint32_t seed_pos) {
    const uint32_t pos = ui_pos(ui);
    uint32_t S = pos - ui_dstart(ui), E = pos + ui_dend(ui);
    piece_cut(S, E, pos, seed_pos);
    return piece_make(ui, flip, fwd, S, E);
}
// key of a piece's interior in the visited table: bit 33 is never set in a vertex key (engine.h: vt_key)
LDBG_DEV uint64_t piece_key(const Piece& p) { return (1ull << 33) | ((uint64_t)p.S << 1) | (p.plus ? 1ull : 0ull); }

// walk-only strand state
struct RunState {
    uint32_t seed_pos;       // position of the seed's record in the run index, LDBG_RUN_NONE if it has none
    uint32_t seen_marks;     // `seen` marks and table claims made so far (a revolution of a repeating walk makes none)
    uint32_t choices;        // junction choices taken so far
    uint32_t anchor_at;      // choice count at which the anchor snapshot was taken (0 = none)
    uint32_t anchor_gv, anchor_marks, anchor_n, anchor_cap;   // graph size, seen_marks, link-store size and HashMap table size at the anchor
    uint32_t period;         // vertices per revolution, once a first repetition of the anchor state has been seen (0 = none)
    uint64_t anchor_sig;     // signature of (vertices, link store) at the anchor
    uint64_t anchor_cv, anchor_t;   // table keys of cv and of the cursor's next vertex at the anchor
};

// what a lane keeps of its strand from one bulk-synchronous round to the next (walks over a sharded table's image, image.h)
struct StrandSave {
    StrandState st;
    RunState rs;
    uint32_t ls_n, ls_java_cap, ls_nkeys, ls_next_seq, ls_age, ls_n_new;
    uint8_t ls_overflow, active, begun, pad;
    LsElem fast[LDBG_LS_FAST];
};

// ---- walks that repeat themselves.  With links a walk may go round a tandem repeat for ever (the links it picks up on every
// revolution tell it to go round once more); the reference stops it at maxLength (:428), 75,000 vertices later.  Whatever the
// loop at :373-481 does next is a function of the current vertex, the cursor's next vertex, the LinkStore (elements with their
// positions and RELATIVE ages, which keys of its HashMap they share and the order in which the live keys were created, the
// table size) and of the visited / seen sets — and those two only through "has this k-mer been seen" and the copy counts.
// So: if that state recurs at a junction choice, and no vertex was seen for the first time in between (every k-mer of the
// revolution was already seen, so none of the `seen` tests can come out differently next time), the walk repeats that revolution
// until maxLength; only the copy counts move on, each by a fixed amount per revolution.  The state is compared in full
// (a snapshot of the link store per strand), after a cheap signature; two repetitions are required, the second one supplies
// the per-revolution increment of every copyIndex, and the remaining vertices are written as ONE descriptor (strand.h: REPEAT).
#define LDBG_SNAP_CAP 64u
#define LDBG_REPEAT_FROM 4u       // junction choices before the first anchor
struct LsSnap { uint32_t str_off, pos_comp, age, key_seq; };
LDBG_DEV uint64_t sig_mix(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }
LDBG_DEV uint64_t node_key(const Node& n) { return n.idx >= 0 ? vt_key(n.idx, n.flip != 0) : 0ull; }
LDBG_DEV uint64_t walk_signature(const StrandState& st, const LinkStoreDev& ls) {
    uint64_t h = sig_mix(node_key(st.cv) ^ (node_key(st.cu.nxt) << 30) ^ ((uint64_t)ls.n << 50) ^ ((uint64_t)ls.java_cap << 40));
    for (uint32_t i = 0; i < ls.n; i++) {
        const LsElem x = ls_get(ls, i);
        h = sig_mix(h ^ (uint64_t)x.str_off ^ ((uint64_t)x.pos << 32) ^ ((uint64_t)x.comp << 48));
        h = sig_mix(h ^ (uint64_t)(ls.age - x.birth));
    }
    return h;
}
LDBG_DEV void walk_snapshot(const StrandState& st, const LinkStoreDev& ls, LsSnap* snap) {
    for (uint32_t i = 0; i < ls.n; i++) {
        const LsElem x = ls_get(ls, i);
        LsSnap v;
        v.str_off = x.str_off; v.pos_comp = (uint32_t)x.pos | ((uint32_t)x.comp << 16); v.age = ls.age - x.birth; v.key_seq = x.key_seq;
        LDBG_GLOBAL(LsSnap, snap)[i] = v;
    }
}
// is the link store now the one of the snapshot?  (element by element; the HashMap keys up to an order-preserving renaming)
LDBG_DEV bool walk_same_store(const LinkStoreDev& ls, const LsSnap* snap) {
    for (uint32_t i = 0; i < ls.n; i++) {
        const LsElem x = ls_get(ls, i);
        const LsSnap v = LDBG_GLOBAL(const LsSnap, snap)[i];
        if (v.str_off != x.str_off || v.pos_comp != ((uint32_t)x.pos | ((uint32_t)x.comp << 16)) || v.age != ls.age - x.birth) return false;
        for (uint32_t j = 0; j < i; j++) {
            const uint32_t kj = ls_get(ls, j).key_seq, sj = LDBG_GLOBAL(const LsSnap, snap)[j].key_seq;
            const int now = x.key_seq < kj ? -1 : (x.key_seq > kj ? 1 : 0), then = v.key_seq < sj ? -1 : (v.key_seq > sj ? 1 : 0);
            if (now != then) return false;
        }
    }
    return true;
}
// Called after a general step that took a junction choice.  Returns true when the rest of the walk has been written as a
// REPEAT descriptor and the strand has ended (at maxLength, as it would have: :428, 470-472).
LDBG_DEV bool periodic_check(const WalkArgs& a, StrandState& st, const LinkStoreDev& ls, RunState& rs, LsSnap* snap) {
    rs.choices++;
    if (rs.choices < LDBG_REPEAT_FROM || ls.n > LDBG_SNAP_CAP || !st.cu.has || st.cv.idx < 0 || st.cu.nxt.idx < 0) return false;
    if (st.cv.copy == 0) return false;       // a state can only recur on a vertex the walk stands on for at least the second time
    const uint64_t sig = walk_signature(st, ls);
    const bool live = rs.anchor_at != 0u && rs.seen_marks == rs.anchor_marks;
    if (live && sig == rs.anchor_sig && ls.n == rs.anchor_n && ls.java_cap == rs.anchor_cap && node_key(st.cv) == rs.anchor_cv && node_key(st.cu.nxt) == rs.anchor_t &&
        st.gV > rs.anchor_gv && walk_same_store(ls, snap)) {
        const uint32_t P = st.gV - rs.anchor_gv;
        if (rs.period == P && st.gV >= 2u * P && st.gV <= (uint32_t)a.e.max_len) {
            const uint32_t R = (uint32_t)a.e.max_len + 1u - st.gV;        // vertices still to come; then one more iteration sees maxLength
            if (!path_append_pair(a, st.s, st.pw, pd_repeat_head(R), (uint64_t)(st.gV - 2u * P) | ((uint64_t)P << 32))) { st.status = ST_POOL_FULL; return true; }
            st.gV += R; st.iters += R + 1u;
#ifdef LDBG_HOSTSIM
            ls_debug().repeats++;
#endif
            return true;
        }
        rs.period = P;                                  // first repetition: the anchor moves here, the next one must take as long
        rs.anchor_gv = st.gV; rs.anchor_at = rs.choices;
        return false;                                   // (the state is the anchor's: snapshot, signature and keys stay)
    }
    if (!live || rs.choices >= 2u * rs.anchor_at) {      // a new anchor, at choice counts that double (Brent's cycle detection)
        rs.anchor_at = rs.choices; rs.anchor_gv = st.gV; rs.anchor_marks = rs.seen_marks; rs.anchor_n = ls.n; rs.anchor_cap = ls.java_cap; rs.period = 0u;
        rs.anchor_sig = sig; rs.anchor_cv = node_key(st.cv); rs.anchor_t = node_key(st.cu.nxt);
        walk_snapshot(st, ls, snap);
    }
    return false;
}

// mode A: the walk is at (cv = q_0, cursor's next = q_1) of a piece worth a run step, and the lean step's conditions hold
LDBG_DEV bool run_entry_a(const EngineView& e, const StrandState& st, const RunState& rs) {
    const Node& cv = st.cv;
    const Node& t = st.cu.nxt;
    if (!ui_valid(cv.ui) || !ui_valid(t.ui)) return false;
    const Piece pc = piece_of(cv.ui, cv.flip != 0, st.fwd, rs.seed_pos);
    if (pc.q != 0u || pc.n < LDBG_RUN_MIN) return false;
    const Piece pt = piece_of(t.ui, t.flip != 0, st.fwd, rs.seed_pos);
    return pt.S == pc.S && pt.E == pc.E && pt.plus == pc.plus && pt.q == 1u;
}
LDBG_DEV bool run_mode_a(const WalkArgs& a, const StrandState& st, const RunState& rs) {
    return lean_cursor_ok(a.e, st) && st.gV >= 2u && st.gV <= (uint32_t)a.e.max_len && run_entry_a(a.e, st, rs);
}
// mode B: no cursor (no links, or the cursor has run out: hasNext() is false for good); the walk stands on cv = q_1
LDBG_DEV bool run_mode_b(const WalkArgs& a, const StrandState& st, const RunState& rs) {
    const EngineView& e = a.e;
    const Node& cv = st.cv;
    if (st.status != ST_OK || (e.cursor_on && st.cu.has) || !(e.g.k & 1) || st.gV < 1u) return false;
    if (cv.idx < 0 || cv.npe || cv.flip != cv.fj || !ui_valid(cv.ui)) return false;
    const int acopy = cv.copy < 0 ? -cv.copy : cv.copy;
    if (!(acopy >= vt_count_e(cv.vent) && acopy + 1 <= 32767)) return false;
    const Piece pc = piece_of(cv.ui, cv.flip != 0, st.fwd, rs.seed_pos);
    return pc.q == 1u && pc.n >= LDBG_RUN_MIN;
}

// vertex at position `pos` of the run index as the walk meets it: flips inverted when it travels the mirror chain
LDBG_DEV void run_vertex(const EngineView& e, VisitedTable& vt, uint32_t pos, bool inv, bool fwd, Node& n) {
    const uint32_t u = LDBG_GLOBAL(const uint32_t, e.runs.uo)[pos];
    const unsigned bb = LDBG_GLOBAL(const uint8_t, e.runs.ubase)[pos];
    const unsigned first = !inv ? (bb & 3u) : 3u - ((bb >> 2) & 3u), last = !inv ? ((bb >> 2) & 3u) : 3u - (bb & 3u);
    Node parent;
    parent.fj = 0;
    const uint32_t ent = ((u & 0x7FFFFFFFu) + 1u) | ((((u >> 31) != 0u) != inv) ? 0x80000000u : 0u);
    node_from_entry(e, vt, parent, ent, fwd ? last : first, fwd, n);
}

LDBG_DEV bool run_emit(const WalkArgs& a, StrandState& st, uint32_t len, uint32_t acopy, const Piece& pc) {
    if (len == 0u) return true;
    // interior vertices q_2 .. : positions S+2.. ascending or E-2.. descending
    const uint32_t first = pc.asc ? pc.S + 2u : pc.E - 2u;
    return path_append_pair(a, st.s, st.pw, pd_run_head(len, acopy, pc.asc, !pc.plus), (uint64_t)first);
}

// One run step.  Returns true when the strand has ended (the caller finishes it).
template <int W>
LDBG_DEV bool run_step(const WalkArgs& a, StrandState& st, LinkStoreDev& ls, RunState& rs, bool mode_a) {
    const EngineView& e = a.e;
    const bool fwd = st.fwd;
    Node& cv = st.cv;
    const Piece pc = piece_of(cv.ui, cv.flip != 0, fwd, rs.seed_pos);
    const uint32_t n = pc.n, nB = n - 4u;
    const bool inv = !pc.plus;
    // the far fringe and the interior's entry
    Node y, z;
    const uint32_t used0 = st.vt.used;
    uint64_t eB = 0;
    const uint64_t kB = piece_key(pc);
    const uint32_t hB = vt_hash(kB) & st.vt.mask;
    uint32_t slotB;
    if (e.lean_rows) {
        // all three in flight together (engine.h: node_issue_lean): the two positions of the run index, then both rows with their table
        // rounds and the piece entry's round — two trips to memory where the vertex-by-vertex code makes five
        const uint32_t py = pc.asc ? pc.E - 1u : pc.S + 1u, pz = pc.asc ? pc.E : pc.S;
        const uint32_t uy = LDBG_GLOBAL(const uint32_t, e.runs.uo)[py], uz = LDBG_GLOBAL(const uint32_t, e.runs.uo)[pz];
        const unsigned by = LDBG_GLOBAL(const uint8_t, e.runs.ubase)[py], bz = LDBG_GLOBAL(const uint8_t, e.runs.ubase)[pz];
        const VtPeek pB = vt_peek(st.vt, hB);
        NodeLoad ly, lz;
        node_issue_lean(e, st.vt, false, ((uy & 0x7FFFFFFFu) + 1u) | ((((uy >> 31) != 0u) != inv) ? 0x80000000u : 0u), ly);
        node_issue_lean(e, st.vt, false, ((uz & 0x7FFFFFFFu) + 1u) | ((((uz >> 31) != 0u) != inv) ? 0x80000000u : 0u), lz);
        auto travel_base = [&](unsigned bb) { const unsigned first = !inv ? (bb & 3u) : 3u - ((bb >> 2) & 3u), last = !inv ? ((bb >> 2) & 3u) : 3u - (bb & 3u); return fwd ? last : first; };
        uint32_t u0 = st.vt.used;
        node_finish_lean(e, st.vt, ly, travel_base(by), fwd, y);
        const uint32_t cy = st.vt.used != u0 ? y.vslot : 0xFFFFFFFFu;        // a slot y has just claimed: rounds read before may not show it
        u0 = st.vt.used;
        node_finish_lean(e, st.vt, lz, travel_base(bz), fwd, z, cy);
        const uint32_t cz = st.vt.used != u0 ? z.vslot : 0xFFFFFFFFu;
        const bool staleB = vt_round_covers(st.vt, hB, cy) || vt_round_covers(st.vt, hB, cz);
        slotB = vt_probe_from(st.vt, kB, hB, staleB ? vt_peek(st.vt, hB) : pB, &eB);
    } else {
        run_vertex(e, st.vt, pc.asc ? pc.E - 1u : pc.S + 1u, inv, fwd, y);
        run_vertex(e, st.vt, pc.asc ? pc.E : pc.S, inv, fwd, z);
        slotB = vt_probe_from(st.vt, kB, hB, vt_peek(st.vt, hB), &eB);
    }
    rs.seen_marks += st.vt.used - used0;
    const int cntB = vt_count_e(eB), cntY = vt_count_e(y.vent);
    const int64_t allowed = (int64_t)e.max_len - (int64_t)st.gV + 1;          // iterations that can still append a vertex
    if (mode_a) {
        // steps i = 1 .. n-2: cursor onto t = q_i with x = q_{i+1} looked up, av = q_i, visited.add(q_{i-1})
        Node& t = st.cu.nxt;
        const uint32_t ep = st.cu.epoch;
        const int cntT = vt_count_e(t.vent);
        const bool seenB = vt_seen_e(eB, ep), seenY = vt_seen_e(y.vent, ep), seenZ = vt_seen_e(z.vent, ep);
        const uint32_t steps = n - 2u;
        const bool full = allowed >= (int64_t)steps;
        const uint32_t k = full ? steps : (uint32_t)allowed;                   // (allowed >= 1: lean_cursor_ok + gV <= maxLength)
        bool odd = false;
        if (ls.n == 0u) odd = seenB || (seenY && k >= n - 3u);                 // the cursor would run out inside the piece
        odd = odd || (k >= 2u && cntT + 1 > 32767) || (k >= 3u && cntB + 1 > 32767);
        if (odd) {
#ifdef LDBG_HOSTSIM
            ls_debug().retries++;
#endif
            st.status = ST_RETRY_PLAIN; return true;
        }
#ifdef LDBG_HOSTSIM
        ls_debug().runs_a++; ls_debug().run_vertices += k;
#endif
        if (ls_num_new(ls) > 0) ls_increment_ages(ls);                         // :274-276, first step; nothing is new afterwards
        st.iters += k;
        const int acv = cv.copy < 0 ? -cv.copy : cv.copy;
        Node tv = t;
        tv.copy = fwd ? cntT : -cntT;
        bool ok = path_append(a, st.s, st.pw, pack_vertex(tv));
        ok = ok && run_emit(a, st, k - 1u < nB ? k - 1u : nB, (uint32_t)cntB, pc);
        if (!full) {
            // maxLength falls inside the piece: one more iteration notices it and returns the graph (:428, 470-472)
            if (!ok) { st.status = ST_POOL_FULL; return true; }
            st.gV += k; st.iters += 1u;
            return true;
        }
        y.copy = fwd ? cntY : -cntY;
        ok = ok && path_append(a, st.s, st.pw, pack_vertex(y));
        if (!ok) { st.status = ST_POOL_FULL; return true; }
        st.gV += k;
        // visited.add(q_0 .. q_{n-3}); seen.add(q_2 .. q_{n-1})
        node_store(st.vt, cv, vt_with_count(cv.vent, acv + 1));
        node_store(st.vt, t, vt_with_count(t.vent, cntT + 1));
        uint64_t nb = vt_with_count(eB, cntB + 1);
        if (!seenB) { nb = vt_with_seen(nb, ep); rs.seen_marks++; }
        LDBG_GLOBAL(uint64_t, st.vt.tab)[slotB] = nb;
        if (!seenY) { node_store(st.vt, y, vt_with_seen(y.vent, ep)); rs.seen_marks++; }
        if (!seenZ) { node_store(st.vt, z, vt_with_seen(z.vent, ep)); rs.seen_marks++; }
        st.cu.has = !seenZ || ls.n > 0u;                                      // :262
        if (st.cu.has) st.cu.nxt = z;
        cv = y;
        st.cu.cur = cv;
        return false;
    }
    // mode B: iterations j = 1 .. n-3 on cv = q_j with the only neighbour x = q_{j+1}; x visited -> no adjacent vertex -> the
    // branch returns its graph (ContigStopper: adjacent edges != 1)
    const uint32_t want = n - 3u;
    uint32_t by_counts = want;
    if (cntB > 0) by_counts = 0u;
    else if (cntY > 0) by_counts = n - 4u;
    uint32_t k = by_counts;
    if (allowed < (int64_t)k) k = allowed > 0 ? (uint32_t)allowed : 0u;
    const bool ended = k < want;
#ifdef LDBG_HOSTSIM
    ls_debug().runs_b++; ls_debug().run_vertices += k;
#endif
    st.iters += k + (ended ? 1u : 0u);
    bool ok = run_emit(a, st, k < nB ? k : nB, 0u, pc);
    if (k == want) { y.copy = 0; ok = ok && path_append(a, st.s, st.pw, pack_vertex(y)); }
    if (!ok) { st.status = ST_POOL_FULL; return true; }
    st.gV += k;
    if (ended) return true;
    const int acv = cv.copy < 0 ? -cv.copy : cv.copy;
    node_store(st.vt, cv, vt_with_count(cv.vent, acv + 1));
    LDBG_GLOBAL(uint64_t, st.vt.tab)[slotB] = vt_with_count(eB, 1);
    cv = y;
    return false;
}

}  // namespace ldbg
