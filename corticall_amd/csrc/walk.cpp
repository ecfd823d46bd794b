// Batched contig walks: TraversalEngine.walk(seed) = toWalk(dfs(seed)) with ContigStopper
// (J/utils/traversal/TraversalEngine.java:64-110, 356-482; J/utils/stoppingrules/ContigStopper.java:12-19),
// with or without link annotations.  One strand walk (seed, direction) per lane; lanes pull strand
// walks from a queue until it is empty.
#include <algorithm>
#include <chrono>
#include <numeric>
#include <thread>

#include "lscoop.h"
#include "runs.h"
#include "runstep.h"
#include "shard.h"
#include "strand.h"

namespace ldbg {

// Launch geometry of the one-wavefront-per-strand (or per-seed) stream kernels, measured for k_contigs_rle at C3 (profiles/r03_rle_geometry.log):
// one-wavefront workgroups, one per seed (no grid-stride loop) 0.75 ms; 16,384 of them 0.83 ms; four-wavefront workgroups 0.79-0.87 ms, and
// 1.04 ms on a grid the chip holds at once (2,048 x 256).  The kernel is a chain of dependent trips per seed (PMC, profiles/r03a_pmc.log:
// 4,076 wavefronts in flight on average, 208 us each), so what helps is MORE short-lived wavefronts, not fewer long ones.
#define LDBG_STREAM_BLOCK 64
#define LDBG_STREAM_GRID 65536

// the walk kernel's strand_finish: strand_n counts VERTICES (runs are stored as descriptors), strand_c the stored entries
LDBG_DEV void walk_finish(const WalkArgs& a, StrandState& st) {
    strand_finish(a, st);
    const bool good = !st.branch_null && st.status == ST_OK;
    a.strand_n[st.s] = good ? st.gV : 0u;
    a.strand_c[st.s] = good ? st.pw.n : 0u;
}

// one iteration of the do-loop at TraversalEngine.java:373-481; returns true when the branch has ended
template <int W>
LDBG_DEV bool strand_step(const WalkArgs& a, StrandState& st, LinkStoreDev& ls, const StepPre& pre, RunState& rs) {
    const EngineView& e = a.e;
    const bool fwd = st.fwd;
    if (st.status != ST_OK) return true;     // pool exhausted while regrowing the table
    st.iters++;
    Node& cv = st.cv;
    const uint32_t m = fwd ? cv.next_mask : cv.prev_mask;
    int adj = 0;
    Node av = cv;
    if (e.cursor_on && st.cu.has) {                     // :379-407
        av = cursor_step<W, true>(e, st.cu, ls, st.vt, fwd, &pre, &rs.seen_marks);
        if (st.cu.status != ST_OK) { st.status = st.cu.status; return true; }
        if (st.cu.has) { node_sync(cv, st.cu.nxt); node_sync(av, st.cu.nxt); }   // the `seen` mark may sit in a slot they hold
        const int cnt = node_count(av);                 // first unused copyIndex
        av.copy = fwd ? cnt : -cnt;
        adj = 1;
    } else {
        for (unsigned b = 0; b < 4; b++) {
            if (!((m >> b) & 1u)) continue;
            Node x;
            node_child_located(e, st.vt, cv, fwd, b, x);
            if (node_count(x) > 0) continue;                        // avs.removeAll(seen) :416-422
            adj++;
            av = x;
        }
    }
    const int acopy = cv.copy < 0 ? -cv.copy : cv.copy;
    const uint64_t ecv = cv.idx >= 0 ? cv.vent : 0ull;
    const bool previously = acopy < vt_count_e(ecv);                    // :424
    if (!previously && cv.idx >= 0) {
        if (acopy + 1 > 32767) { st.status = ST_COPY_OVERFLOW; return true; }
        node_store(st.vt, cv, vt_with_count(ecv, acopy + 1));           // visited.add(cv) :425
        node_sync(av, cv);
        if (e.cursor_on) { node_sync(st.cu.cur, cv); if (st.cu.has) node_sync(st.cu.nxt, cv); }
    }
    const bool reached = st.gV > (uint32_t)e.max_len;                   // :428
    if (previously) { st.branch_null = true; return true; }             // :470-478, traversalSucceeded() still false
    if (adj != 1 || reached) return true;                               // ContigStopper succeeded -> return g
    if (st.gV == 0) {                                                   // connectVertex :494-516
        if (!path_append(a, st.s, st.pw, pack_vertex(cv))) { st.status = ST_POOL_FULL; return true; }
        st.gV = 1;
    }
    if (!path_append(a, st.s, st.pw, pack_vertex(av))) { st.status = ST_POOL_FULL; return true; }
    st.gV++;
    st.quirk |= (cv.flip && !cv.fj) || (av.flip && !av.fj);
    if (av.idx < 0) {   // a vertex without a record ends the branch; keep its k-mer for the result
        const Kmer<W> tk = child_kmer<W>(e, cv, fwd, av.base);
        uint64_t* out = a.term + st.s * W;
#pragma unroll
        for (int i = 0; i < W; i++) out[i] = tk.w[i];
    }
    cv = av;
    if (cv.npe) { st.status = ST_NULLPTR; return true; }
    return false;
}

// ---- the lean step (lscoop.h: lean_cursor_ok / lean_cursor_advance) as the walk kernel uses it: the cursor part, then
// connectVertex + advance (:432-440) with nothing that can end the strand or allocate
LDBG_DEV bool lean_ok(const WalkArgs& a, const StrandState& st) {
    return lean_cursor_ok(a.e, st) && st.gV >= 2 && st.gV <= (uint32_t)a.e.max_len       // not the first step, not at the maxLength cut
        && (st.pw.n & (LDBG_PATH_BLOCK - 1)) != 0u;                                       // room in the current path block
}
LDBG_DEV bool lean_again(const WalkArgs& a, const StrandState& st, const RunState& rs) {                     // between two lean steps of a run
    return lean_cursor_again(a.e, st) && st.gV <= (uint32_t)a.e.max_len && (st.pw.n & (LDBG_PATH_BLOCK - 1)) != 0u
        && !(a.e.runs.uinfo && run_entry_a(a.e, st, rs));      // the interior of a piece belongs to the run step (runstep.h)
}
template <int W>
LDBG_DEV void lean_step(const WalkArgs& a, StrandState& st, LinkStoreDev& ls, RunState& rs) {
#ifdef LDBG_LEAN_PROFILE
    const unsigned long long t0 = __builtin_readcyclecounter();
#endif
    const Node av = lean_cursor_advance<W>(a.e, st, ls, &rs.seen_marks);
#ifdef LDBG_LEAN_PROFILE
    if (a.st_gen && a.e.lean_rows) {     // [4s..]: issue, wait, rest (cycles), steps — read by hand from the diagnostics
        const unsigned long long t3 = __builtin_readcyclecounter();
        a.st_prof[4 * st.s] += st.cu.nxt.p1 - t0; a.st_prof[4 * st.s + 1] += st.cu.nxt.p2 - st.cu.nxt.p1;
        a.st_prof[4 * st.s + 2] += t3 - st.cu.nxt.p2; a.st_prof[4 * st.s + 3] += 1;
    }
#endif
    LDBG_GLOBAL(uint64_t, st.pw.cur)[st.pw.n & (LDBG_PATH_BLOCK - 1)] = pack_vertex(av);   // connectVertex(g, cv, {av}) :432-440
    st.pw.n++;
    st.gV++;
    st.cv = av;
}

// BS = lanes per workgroup (a full or partial wavefront).  Fewer lanes per wavefront = fewer strands whose link-store
// work the wavefront has to carry out one after the other, and more wavefronts per CU to hide each other's latency.
// IMG: the walk runs over the local image of a sharded table (image.h) — suspension, state kept from launch to launch.  A separate
// instantiation: the resident-table kernel does not pay for that code in registers (it cost 55 of them: 2 wavefronts per SIMD -> 1)
template <int W, int BS, bool IMG>
LDBG_WAVE_KERNEL_N(BS) void k_walk(WalkArgs a) {
    const int64_t slot = global_tid();
    if (slot >= a.n_slots) return;
#ifdef LDBG_WALK_DIAG
    if (a.wg_times && threadIdx.x == 0) a.wg_times[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    // step-kind counters of this lane over all its strands (summed per wavefront when it ends: ldbg_profile_get "walk_steps_*" — the
    // inputs of the byte model in DESIGN.md / bench.py), and this wavefront's loop iterations with a general part
    uint32_t kc_run = 0, kc_runv = 0, kc_lean = 0, kc_gen = 0, kc_add = 0, kc_choice = 0, wave_general = 0;
    // link store: the first LDBG_LS_FAST elements of every lane live in LDS ([element][lane]), the rest in HBM
#ifndef LDBG_HOSTSIM
    __shared__ LsElem lds_store[LDBG_LS_FAST * BS];
    LsElem* fast = lds_store + threadIdx.x;
    const uint32_t fast_stride = BS;
#else
    static LsElem lds_store[LDBG_LS_FAST * 64];          // (one simulated wavefront at a time: rt.h)
    if ((rt::poison() || getenv("LDBG_HOSTSIM_ZERO_LDS")) && wave_lane() == 0) memset((void*)lds_store, rt::poison() ? 0xAB : 0, sizeof lds_store);      // (lane 0 is the first fibre to run)
    if (wave_lane() == 0) lds_shadow_begin(lds_store, sizeof lds_store);
    LsElem* fast = lds_store + wave_lane();
    const uint32_t fast_stride = (uint32_t)wave_size();
#endif
    LinkStoreDev ls;
    ls.fast = fast; ls.fast_cap = LDBG_LS_FAST; ls.fast_stride = fast_stride;
    ls.el = a.ls + (size_t)slot * a.ecap;
    ls.cap = a.ecap + LDBG_LS_FAST;
    ls_clear(ls);
    LsWave lw;                                            // the link stores of this wavefront's lanes
    lw.fast = fast - wave_lane(); lw.stride = fast_stride; lw.fast_cap = LDBG_LS_FAST;
    lw.el = a.ls + (size_t)(slot - wave_lane()) * a.ecap; lw.ecap = a.ecap;
    StrandState st;
    st.vt.tab = nullptr; st.vt.mask = 0; st.vt.used = 0; st.status = ST_OK;
    RunState rs;
    rs.seed_pos = LDBG_RUN_NONE; rs.seen_marks = 0; rs.choices = 0; rs.anchor_at = 0; rs.anchor_gv = 0; rs.anchor_marks = 0; rs.anchor_n = 0; rs.anchor_cap = 0;
    rs.period = 0; rs.anchor_sig = 0; rs.anchor_cv = 0; rs.anchor_t = 0;
    const bool runs_on = a.e.runs.uinfo != nullptr;
    bool active = false, exhausted = false;
    // over an image (image.h): a strand that needs a row that has not been sent yet SUSPENDS for the rest of this launch; the
    // strand a lane was working on when the previous round ended is taken up again
    bool suspended = false, begun = true;
    if constexpr (IMG) {
        const StrandSave& sv = a.save[slot];
        if (sv.active) {
            st = sv.st; rs = sv.rs;
            ls.n = sv.ls_n; ls.java_cap = sv.ls_java_cap; ls.nkeys = sv.ls_nkeys; ls.next_seq = sv.ls_next_seq; ls.age = sv.ls_age; ls.n_new = sv.ls_n_new;
            ls.overflow = sv.ls_overflow != 0;
            for (uint32_t i = 0; i < LDBG_LS_FAST && i < sv.ls_n; i++) ls_set(ls, i, sv.fast[i]);
            active = true; begun = sv.begun != 0;
        }
    }
    uint32_t wave_iterations = 0;
    // all lanes of a wavefront stay in the loop until every one of them has run out of strands: the table
    // regrowth below is a wave-wide operation
    while (wave_ballot((active && !suspended) || (!active && !exhausted)) != 0ull) {
        if (IMG && a.yield_iters != 0u && wave_iterations >= a.yield_iters) break;      // (uniform: the round goes on with the next launch)
        wave_iterations++;
        if (!active && !exhausted) {
            const int64_t fi = (int64_t)atomic_add_u64(a.next_strand, 1ull);
            if (fi >= a.n_strands) exhausted = true;
            else {
                int64_t s;
                if (a.retry) s = (int64_t)a.retry[fi];
                else s = (int64_t)(((unsigned __int128)fi * (unsigned __int128)a.fetch_stride) % (unsigned __int128)a.n_strands);
                const bool fwd = (s & 1) != 0;
                if ((fwd && !a.run_fwd) || (!fwd && !a.run_rev)) {
                    a.strand_n[s] = 0; a.strand_c[s] = 0; a.status[s] = ST_BRANCH_NULL; a.iters[s] = 0; a.quirk[s] = 0;
                } else if (IMG) {
                    st.s = s; st.fwd = fwd; active = true; begun = false;       // begins below, once the rows around its seed are here
                } else {
                    active = strand_begin<W>(a, st, ls, s);
                    rs.seed_pos = st.cv.idx >= 0 && ui_valid(st.cv.ui) ? ui_pos(st.cv.ui) : LDBG_RUN_NONE;
                    rs.seen_marks = 0; rs.choices = 0; rs.anchor_at = 0; rs.period = 0;
                    if (!active) walk_finish(a, st);
                }
            }
        }
        if (IMG && active && !suspended) {
            if (!begun) {
                // the first iteration looks at the seed's neighbours (cursor_seek :321-335, or the branch loop itself :373-376)
                const int32_t sl = a.seed_valid[st.s >> 1] ? a.seed_slot[st.s >> 1] : -1;
                bool ready = true;
                if (sl >= 0) {
                    Kmer<W> sk;
                    const uint64_t* sw = a.seeds + (st.s >> 1) * W;
#pragma unroll
                    for (int i = 0; i < W; i++) sk.w[i] = sw[i];
                    Node sn;
                    seed_node<W>(a.e, sk, sl, sn);
                    ready = rows_ready(a.img, sn, st.fwd);
                }
                if (!ready) suspended = true;
                else {
                    begun = true;
                    active = strand_begin<W>(a, st, ls, st.s);
                    rs.seed_pos = LDBG_RUN_NONE; rs.seen_marks = 0; rs.choices = 0; rs.anchor_at = 0; rs.period = 0;
                    if (!active) walk_finish(a, st);
                }
            }
            if (active && begun && !suspended && st.status == ST_OK) {
                const bool ready = (a.e.cursor_on && st.cu.has) ? rows_ready(a.img, st.cu.nxt, st.fwd) : rows_ready(a.img, st.cv, st.fwd);
                if (!ready) suspended = true;
            }
        }
        const bool running = active && begun && !suspended;
#ifdef LDBG_WALK_DIAG
        unsigned long long tc0 = 0;
        if (a.wave_cat) tc0 = __builtin_amdgcn_s_memrealtime();
#endif
        wave_grow_tables(a, st, running);
#ifdef LDBG_WALK_DIAG
        unsigned long long tc1 = 0;
        if (a.wave_cat) tc1 = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- run step: a whole unbranched stretch at once (runstep.h)
        bool stepped = false;
        auto run_phase = [&](bool on) {
            if (!(runs_on && on)) return;
            const bool ma = run_mode_a(a, st, rs);
            const bool mb = !ma && run_mode_b(a, st, rs);
            if (ma || mb) {
                stepped = true;
                const uint32_t it0 = st.iters;
                const bool ended = run_step<W>(a, st, ls, rs, ma);
                kc_run++; kc_runv += st.iters - it0;
                if (ended) { walk_finish(a, st); active = false; }
            }
        };
        run_phase(running);
#ifdef LDBG_WALK_DIAG
        unsigned long long tc2 = 0;
        if (a.wave_cat) tc2 = __builtin_amdgcn_s_memrealtime();
#endif
        // The phases of an iteration fall through (resident table): a strand that has just crossed a stretch goes on with its lean steps,
        // and one that has moved and now stands before a junction or a link-flagged vertex takes its general step, all in THIS iteration.
        // Per unitig a strand then spends three iterations (choice; adds; run + lean + the next choice) where it spent five, and an
        // iteration costs the wavefront the same whichever of its lanes use which phase.  (Over an image a strand that has moved must have
        // its rows checked first: one phase per iteration there.)
        const bool lean = running && active && (!IMG || !stepped) && lean_ok(a, st) && !(runs_on && run_entry_a(a.e, st, rs));
        if (lean) {
            // a short run of lean steps without going round the outer loop (its ballots, refill and regrowth checks): every lean
            // step claims at most one table slot, and the regrowth check above leaves room for eight
            int r = 0;
            const uint32_t used0 = st.vt.used;
#pragma unroll 1
            do { lean_step<W>(a, st, ls, rs); } while (++r < a.lean_run && lean_again(a, st, rs));     // one copy of the step: the loop lives in the instruction cache
            kc_lean += (uint32_t)r;
            rs.seen_marks += st.vt.used - used0;
            st.cu.cur = st.cv;                         // the cursor stands on the walk's current vertex
        }
#ifdef LDBG_WALK_DIAG
        if (a.wave_cat && threadIdx.x == 0) {
            const unsigned long long tc3 = __builtin_amdgcn_s_memrealtime();
            unsigned long long* wc = a.wave_cat + 16 * blockIdx.x;
            wc[0] += 1; wc[1] += tc1 - tc0; wc[2] += tc2 - tc1; wc[3] += tc3 - tc2;
        }
#endif
        bool general = running && active;
        if (general && (lean || stepped)) {          // it has moved: does it need a general step now, or another run / lean step (next iteration)?
            if constexpr (IMG) general = false;
            else general = !(runs_on && (run_mode_a(a, st, rs) || run_mode_b(a, st, rs))) && !(lean_ok(a, st) && !(runs_on && run_entry_a(a.e, st, rs)));
        }
        const unsigned long long general_lanes = wave_ballot(general);
        if (general_lanes == 0ull) continue;          // the whole wavefront took a lean or a run step (or waits for rows)
        wave_general++;
#ifdef LDBG_WALK_DIAG
        unsigned long long tc4 = 0, tprep[4] = {0, 0, 0, 0};
        if (a.wave_cat) tc4 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long t_general = a.st_gen ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
        // ---- link-store work of this step, carried out by the whole wavefront for one lane at a time (lscoop.h)
        const bool cur_mode = general && st.status == ST_OK && a.e.cursor_on && st.cu.has;
        StepPre pre;
        const uint32_t used1 = st.vt.used;
#ifdef LDBG_WALK_DIAG
        coop_step_prepare<W>(a.e, st, ls, lw, cur_mode, pre, a.wave_cat ? tprep : nullptr);
        if (a.st_gen && general) {       // diagnostics: time this strand spends in general steps (prepare + cooperative phases + step)
            a.st_gen[2 * st.s] += __builtin_amdgcn_s_memrealtime() - t_general;
            a.st_gen[2 * st.s + 1] += 1;
        }
        unsigned long long tc5 = 0;
        if (a.wave_cat) tc5 = __builtin_amdgcn_s_memrealtime();
#else
        coop_step_prepare<W>(a.e, st, ls, lw, cur_mode, pre);
#endif
        kc_add += pre.n_added; kc_choice += pre.choice_done ? 1u : 0u;
        if (general) {
            kc_gen++;
            bool ended = strand_step<W>(a, st, ls, pre, rs);
            rs.seen_marks += st.vt.used - used1;
            if (!ended && a.snap && pre.choice_done && st.status == ST_OK) ended = periodic_check(a, st, ls, rs, a.snap + (size_t)slot * LDBG_SNAP_CAP);
            if (ended) { walk_finish(a, st); active = false; }
        }
#ifdef LDBG_WALK_DIAG
        if (a.wave_cat && threadIdx.x == 0) {       // general part: [5] all of it, [6] per-lane prefetch, [7] cooperative adds, [8] cooperative choices, [9] the step itself;
            unsigned long long* wc = a.wave_cat + 16 * blockIdx.x;      // [10] lanes with a general step, [11] owners of adds, [12] owners of choices
            const unsigned long long tc6 = __builtin_amdgcn_s_memrealtime();
            wc[4] += 1; wc[5] += tc6 - tc4; wc[6] += tprep[0] - tc4; wc[7] += tprep[1] - tprep[0]; wc[8] += tc5 - tprep[1]; wc[9] += tc6 - tc5;
            wc[10] += (unsigned long long)__builtin_popcountll(general_lanes); wc[11] += tprep[2] & 0xFFFFFFFFull; wc[12] += tprep[2] >> 32;
            wc[13] += tprep[3] & 0xFFFFull; wc[14] += (tprep[3] >> 16) & 0xFFFFull; wc[15] += tprep[3] >> 32;
        }
#endif
    }
    if constexpr (IMG) {
        StrandSave& sv = a.save[slot];
        sv.active = active ? 1 : 0;
        if (active) {
            sv.st = st; sv.rs = rs; sv.begun = begun ? 1 : 0;
            sv.ls_n = ls.n; sv.ls_java_cap = ls.java_cap; sv.ls_nkeys = ls.nkeys; sv.ls_next_seq = ls.next_seq; sv.ls_age = ls.age; sv.ls_n_new = ls.n_new;
            sv.ls_overflow = ls.overflow ? 1 : 0;
            for (uint32_t i = 0; i < LDBG_LS_FAST && i < ls.n; i++) sv.fast[i] = ls_get(ls, i);
            atomic_add_u64(a.unfinished, 1ull);
        }
    }
#ifdef LDBG_WALK_DIAG
    if (a.wg_times && threadIdx.x == 0) a.wg_times[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (a.kinds) {
        // one sum per counter and wavefront, one atomic each; and the wavefront's own iteration counts (the latency side of the model)
        const uint32_t kc[6] = {kc_run, kc_runv, kc_lean, kc_gen, kc_add, kc_choice};
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const uint32_t tot = wave_incl_scan_u32(kc[q]);
            if (wave_lane() == wave_size() - 1 && tot) atomic_add_u64(a.kinds + q, (unsigned long long)tot);
        }
        if (wave_lane() == 0) {
            atomic_add_u64(a.kinds + 6, (unsigned long long)wave_iterations);
            atomic_add_u64(a.kinds + 7, (unsigned long long)wave_general);
            unsigned long long* mx = a.kinds + 8;           // busiest wavefront: most iterations with a general part | its iterations
            const unsigned long long mine = ((unsigned long long)wave_general << 32) | (unsigned long long)wave_iterations;
#ifndef LDBG_HOSTSIM
            atomicMax(mx, mine);
#else
            if (mine > *mx) *mx = mine;
#endif
        }
    }
}

// ---- result assembly -------------------------------------------------------------------------
// the seeds of a batch: n x k ASCII bytes -> packed words + the Q4 validity byte (a string with a byte outside ACGT is no record's k-mer: its
// words are zero and findRecord misses, kmer.h: ascii_batch_to_words is the host form of the same rule)
LDBG_KERNEL void k_seed_words(const unsigned char* ascii, int64_t n, int k, int W, uint64_t* words, uint8_t* valid) {
    const int nw = (k + 31) / 32, lead = W - nw;         // words that carry bases; leading all-zero words
    for (int64_t q = global_tid(); q < n; q += global_nthreads()) {
        const unsigned char* c = ascii + q * k;
        uint64_t w[4] = {0ull, 0ull, 0ull, 0ull};
        bool bad = false;
        int i = 0;
        for (int wi = lead; wi < W; wi++) {
            const int cnt = wi == lead ? k - 32 * (nw - 1) : 32;
            uint64_t acc = 0;
            for (int j = 0; j < cnt; j++) {
                unsigned v = 0;
                switch (c[i++]) {                 // upper case only: a seed is looked up, not encoded (kmer.h)
                    case 'A': v = 0; break;
                    case 'C': v = 1; break;
                    case 'G': v = 2; break;
                    case 'T': v = 3; break;
                    default: bad = true; break;
                }
                acc = (acc << 2) | (uint64_t)v;
            }
            w[wi] = acc;
        }
        for (int wi = 0; wi < W; wi++) words[q * W + wi] = bad ? 0ull : w[wi];
        valid[q] = bad ? 0 : 1;
    }
}
// strands the run steps handed back (ST_RETRY_PLAIN): their numbers, for the second launch (any order: strands are independent)
LDBG_KERNEL void k_retry_list(const uint32_t* status, int64_t ns, uint32_t* list, unsigned long long* count) {
    for (int64_t s = global_tid(); s < ns; s += global_nthreads())
        if (status[s] == ST_RETRY_PLAIN) list[atomic_add_u64(count, 1ull)] = (uint32_t)s;
}
struct AsmArgs {
    EngineView e;
    int64_t n;
    int op_and, k;
    const uint64_t* seeds;
    const uint64_t* pool; const uint32_t* block_table; int max_blocks;
    const uint32_t* strand_n; const uint32_t* status; const uint32_t* iters; const uint8_t* quirk;
    int64_t* walk_len;                       // [n]
    uint8_t* seed_ok;                        // [n]
    uint32_t* contig_len;                    // [n] bytes of the contig (0: no walk)
    unsigned long long* flags;               // [0] strands that ran out of pool, [1] strands that ended in an error, [2] strands with a quirk-Q6 vertex, [3] loop iterations
};
LDBG_DEV unsigned long long wave_sum_u64(unsigned long long v) { for (int m = wave_size() >> 1; m > 0; m >>= 1) v += wave_shfl_xor_u64(v, m); return v; }
// toWalk's seed test (TraversalUtils.java:392-397) + OR/AND combination (TraversalEngine.java:85-99); what the host wants to know about
// the batch as a whole (did a strand run out of pool, end in an error, pass a quirk vertex; k-mers traversed) is counted here, so that the
// per-strand arrays stay on the device
LDBG_KERNEL void k_walk_lengths(AsmArgs a) {
    const int64_t WS = wave_size(), lane = wave_lane();
    const int64_t wave = global_tid() / WS, nwaves = (global_nthreads() + WS - 1) / WS;
    unsigned long long n_full = 0, n_err = 0, n_quirk = 0, n_iters = 0;
    for (int64_t base = wave * WS; base < a.n; base += nwaves * WS) {
        const int64_t i = base + lane;
        if (i >= a.n) continue;
        const uint32_t sr = a.status[2 * i], sf = a.status[2 * i + 1];
        uint32_t nr = a.strand_n[2 * i], nf = a.strand_n[2 * i + 1];
        bool null_r = sr == ST_BRANCH_NULL, null_f = sf == ST_BRANCH_NULL;
        bool err = (sr != ST_OK && !null_r) || (sf != ST_OK && !null_f);
        bool is_null = a.op_and ? (null_r || null_f) : (null_r && null_f);
        int64_t len = 0;
        uint8_t ok = 0;
        if (!err && !is_null && nr + nf > 0) {
            const int64_t ss = nr > 0 ? 2 * i : 2 * i + 1;
            uint64_t seed_entry = a.pool[(uint64_t)a.block_table[ss * a.max_blocks] * LDBG_PATH_BLOCK];
            int64_t idx = path_idx(seed_entry);
            if (idx >= 0 && (int32_t)graph_cov(a.e.g, idx, a.e.first_trav) > 0) {   // Q5: coverage is a signed int
                ok = 1;
                len = (int64_t)(nr > 0 ? nr - 1 : 0) + (int64_t)(nf > 0 ? nf - 1 : 0) + 1;
            }
        }
        a.walk_len[i] = len;
        a.seed_ok[i] = ok;
        a.contig_len[i] = len > 0 ? (uint32_t)(len + a.k - 1) : 0u;
        n_full += (sr == ST_POOL_FULL) + (sf == ST_POOL_FULL);
        n_err += (sr != ST_OK && !null_r && sr != ST_POOL_FULL) + (sf != ST_OK && !null_f && sf != ST_POOL_FULL);
        n_quirk += (a.quirk[2 * i] != 0) + (a.quirk[2 * i + 1] != 0);
        n_iters += (unsigned long long)a.iters[2 * i] + a.iters[2 * i + 1];
    }
    n_full = wave_sum_u64(n_full); n_err = wave_sum_u64(n_err); n_quirk = wave_sum_u64(n_quirk); n_iters = wave_sum_u64(n_iters);
    if (lane == 0) {
        if (n_full) atomic_add_u64(a.flags + 0, n_full);
        if (n_err) atomic_add_u64(a.flags + 1, n_err);
        if (n_quirk) atomic_add_u64(a.flags + 2, n_quirk);
        if (n_iters) atomic_add_u64(a.flags + 3, n_iters);
    }
}
// exclusive prefix sums of n 32-bit counts as n + 1 64-bit offsets (strand_n -> strand_off, contig_len -> contig_off): partial sums of
// OFF_SCAN_OWNERS stretches, their prefix by one wavefront, the offsets
#define OFF_SCAN_OWNERS 4096
LDBG_KERNEL void k_off_sums(int64_t n, int64_t chunk, const uint32_t* cnt, unsigned long long* sums) {
    for (int64_t t = global_tid(); t < OFF_SCAN_OWNERS; t += global_nthreads()) {
        const int64_t lo = std::min<int64_t>(t * chunk, n), hi = std::min<int64_t>(lo + chunk, n);
        unsigned long long s = 0;
        for (int64_t i = lo; i < hi; i++) s += cnt[i];
        sums[t] = s;
    }
}
LDBG_KERNEL void k_off_top(unsigned long long* sums, int64_t* total) {
    if (global_tid() / wave_size() != 0) return;
    const int WS = wave_size(), lane = wave_lane(), per = OFF_SCAN_OWNERS / WS;
    unsigned long long mine = 0;
    for (int j = 0; j < per; j++) mine += sums[lane * per + j];
    unsigned long long base = 0, all = 0;
    for (int l = 0; l < WS; l++) { const unsigned long long t = wave_bcast_u64(mine, l); if (l < lane) base += t; all += t; }
    for (int j = 0; j < per; j++) { const unsigned long long v = sums[lane * per + j]; sums[lane * per + j] = base; base += v; }
    if (lane == 0) *total = (int64_t)all;
}
LDBG_KERNEL void k_off_apply(int64_t n, int64_t chunk, const uint32_t* cnt, const unsigned long long* sums, const int64_t* total, int64_t* off) {
    for (int64_t t = global_tid(); t < OFF_SCAN_OWNERS; t += global_nthreads()) {
        const int64_t lo = std::min<int64_t>(t * chunk, n), hi = std::min<int64_t>(lo + chunk, n);
        unsigned long long run = sums[t];
        for (int64_t i = lo; i < hi; i++) { off[i] = (int64_t)run; run += cnt[i]; }
        if (t == 0) off[n] = *total;
    }
}
// d_tmp: OFF_SCAN_OWNERS words; d_total: one int64 on the device
static void launch_offsets(const uint32_t* d_cnt, int64_t n, int64_t* d_off, unsigned long long* d_tmp, int64_t* d_total, rt::stream_t s) {
    const int64_t chunk = (n + OFF_SCAN_OWNERS - 1) / OFF_SCAN_OWNERS;
    LDBG_LAUNCH(k_off_sums, OFF_SCAN_OWNERS / 256, 256, s, n, chunk, d_cnt, d_tmp);
    LDBG_LAUNCH(k_off_top, 1, 64, s, d_tmp, d_total);
    LDBG_LAUNCH(k_off_apply, OFF_SCAN_OWNERS / 256, 256, s, n, chunk, d_cnt, (const unsigned long long*)d_tmp, (const int64_t*)d_total, d_off);
}


LDBG_KERNEL void k_compact_paths(const uint64_t* pool, const uint32_t* block_table, int max_blocks, const int64_t* strand_off,
                                 int64_t n_strands, uint64_t* dense) {
    const int64_t wave = global_tid() >> 6, lane = global_tid() & 63, nwaves = (global_nthreads() + 63) >> 6;
    for (int64_t s = wave; s < n_strands; s += nwaves) {
        const int64_t o = strand_off[s], n = strand_off[s + 1] - o;
        for (int64_t j = lane; j < n; j += 64)
            dense[o + j] = pool[(uint64_t)block_table[s * max_blocks + j / LDBG_PATH_BLOCK] * LDBG_PATH_BLOCK + (j & (LDBG_PATH_BLOCK - 1))];
    }
}

// ---- walk paths: stored entries (vertices and run descriptors, strand.h) -> one 8-byte vertex entry per vertex.  One wavefront per
// strand: 64 stored entries at a time, a prefix sum of the vertices they stand for gives every entry its place; a RUN is then
// written by the whole wavefront from the run index (coalesced reads of uo / ubase, coalesced writes), a REPEAT from the vertices
// of this strand that are already in place.  This is where the k-mers of an unbranched stretch are materialised: 8 bytes written
// and 5 read per vertex.
struct ExpandArgs {
    const uint64_t* pool; const uint32_t* block_table; int max_blocks;
    const uint32_t* strand_c;          // stored entries per strand
    const int64_t* strand_off;         // [n_strands + 1] vertices before each strand
    int64_t n_strands;
    RunIndexView runs;
    uint64_t* dense;
    unsigned* overflow;                // a REPEAT that takes a copyIndex out of its range sets this
};
LDBG_DEV uint64_t expand_stored(const ExpandArgs& a, int64_t s, uint32_t j) {
    return a.pool[(uint64_t)a.block_table[s * a.max_blocks + j / LDBG_PATH_BLOCK] * LDBG_PATH_BLOCK + (j & (LDBG_PATH_BLOCK - 1))];
}
LDBG_KERNEL void k_expand_paths(ExpandArgs a) {
    const int64_t wave = global_tid() / wave_size(), nwaves = (global_nthreads() + wave_size() - 1) / wave_size();
    const uint32_t lane = (uint32_t)wave_lane(), WS = (uint32_t)wave_size();
    for (int64_t s = wave; s < a.n_strands; s += nwaves) {
        const uint32_t nc = a.strand_c[s];
        uint64_t* out = a.dense + a.strand_off[s];
        const bool fwd = (s & 1) != 0;
        uint32_t base_v = 0;                                   // vertices before the current group of stored entries
        for (uint32_t j0 = 0; j0 < nc; j0 += WS) {
            const uint32_t j = j0 + lane;
            uint64_t e = 0, prev = 0;
            if (j < nc) {
                e = expand_stored(a, s, j);
                if (j & (LDBG_PATH_BLOCK - 1)) prev = expand_stored(a, s, j - 1);      // a pair never straddles two blocks
            }
            const uint32_t cnt = j < nc ? pd_expanded(prev, e) : 0u;
            const uint32_t incl = wave_incl_scan_u32(cnt);
            const uint32_t at = base_v + incl - cnt;
            const bool head = cnt > 0u && pd_is_head(e);
            if (cnt == 1u && !head) out[at] = e;
            unsigned long long hb = wave_ballot(head);
            while (hb) {
                const int L = __builtin_ctzll(hb);
                hb &= hb - 1;
                const uint64_t he = wave_bcast_u64(e, L);
                const uint32_t hat = wave_bcast_u32(at, L);
                const uint32_t hj = j0 + (uint32_t)L;
                const uint64_t payload = expand_stored(a, s, hj + 1);
                const uint32_t len = wave_bcast_u32(cnt, L);
                if (LDBG_PD_KIND(he) == LDBG_PD_RUN) {
                    const uint32_t acopy = (uint32_t)(he >> 20) & 0xFFFFu;
                    const bool asc = (he >> 36) & 1ull, inv = (he >> 37) & 1ull;
                    const uint32_t first = (uint32_t)payload;
                    const int copy = fwd ? (int)acopy : -(int)acopy;
                    for (uint32_t i = lane; i < len; i += WS) {
                        const uint32_t pos = asc ? first + i : first - i;
                        const uint32_t u = a.runs.uo[pos];
                        const unsigned bb = a.runs.ubase[pos];
                        const unsigned b0 = !inv ? (bb & 3u) : 3u - ((bb >> 2) & 3u), b1 = !inv ? ((bb >> 2) & 3u) : 3u - (bb & 3u);
                        out[hat + i] = path_pack((int64_t)(u & 0x7FFFFFFFu), ((u >> 31) != 0u) != inv, fwd ? b1 : b0, copy);
                    }
                } else {                                       // REPEAT: always the last entry of its strand
                    wave_fence();
                    const uint32_t first = (uint32_t)payload, period = (uint32_t)(payload >> 32);
                    for (uint32_t i = lane; i < len; i += WS) {
                        const uint32_t o = i % period;
                        const uint64_t s1 = LDBG_GLOBAL(const uint64_t, out)[first + o], s2 = LDBG_GLOBAL(const uint64_t, out)[first + period + o];
                        const int c = path_copy(s2) + (int)(i / period + 1u) * (path_copy(s2) - path_copy(s1));
                        if (c > 32767 || c < -32767) *a.overflow = 1u;     // reported as the k-mer-by-k-mer walk reports it (ST_COPY_OVERFLOW)
                        out[hat + i] = (s2 & ~(0xFFFFFFull << 36)) | (((uint64_t)(uint32_t)c & 0xFFFFFFull) << 36);
                    }
                }
            }
            base_v += wave_bcast_u32(incl, (int)WS - 1);
        }
    }
}

// TraversalUtils.toContig (TraversalUtils.java:367-381) over walk = reverse strand (far end first), seed, forward strand
struct ContigArgs {
    GraphView g;
    int64_t n;
    const uint64_t* seeds;
    const uint64_t* dense;
    const int64_t* strand_off;
    const int64_t* walk_len;
    const int64_t* contig_off;
    const uint8_t* quirk;
    const uint64_t* term;
    char* out;
};
template <int W>
LDBG_DEV Kmer<W> walk_vertex_kmer(const ContigArgs& a, int64_t i, int64_t v, int64_t ro, int64_t nr, int64_t fo) {
    const int64_t nrev = nr > 0 ? nr - 1 : 0;
    uint64_t e;
    const uint64_t* term;
    if (v < nrev) { e = a.dense[ro + (nr - 1 - v)]; term = a.term + (2 * i) * W; }
    else if (v == nrev) { e = nr > 0 ? a.dense[ro] : a.dense[fo]; term = a.seeds + i * W; }
    else { e = a.dense[fo + (v - nrev)]; term = a.term + (2 * i + 1) * W; }
    Kmer<W> km;
    const int64_t ri = path_idx(e);
    if (ri >= 0) { km = graph_key<W>(a.g, ri); if (path_flip(e)) km = kmer_revcomp<W>(km, a.g.k); }
    else { for (int w = 0; w < W; w++) km.w[w] = term[w]; }
    return km;
}
template <int W>
LDBG_KERNEL void k_contigs(ContigArgs a) {
    const int k = a.g.k;
    const int64_t wave = global_tid() >> 6, lane = global_tid() & 63, nwaves = (global_nthreads() + 63) >> 6;
    for (int64_t i = wave; i < a.n; i += nwaves) {
        if (a.walk_len[i] == 0) continue;
        const int64_t L = a.contig_off[i + 1] - a.contig_off[i];
        const int64_t ro = a.strand_off[2 * i], nr = a.strand_off[2 * i + 1] - ro;
        const int64_t fo = a.strand_off[2 * i + 1];
        const int64_t nrev = nr > 0 ? nr - 1 : 0;
        char* o = a.out + a.contig_off[i];
        if (a.quirk[2 * i] | a.quirk[2 * i + 1]) {
            // a Q6 vertex breaks the k-1 overlaps: first k-mer, then the last base of every further vertex's k-mer
            for (int64_t p = lane; p < L; p += 64) {
                const int64_t v = p < k ? 0 : p - k + 1;
                const Kmer<W> km = walk_vertex_kmer<W>(a, i, v, ro, nr, fo);
                o[p] = "ACGT"[kmer_base<W>(km, k, p < k ? (int)p : k - 1)];
            }
            continue;
        }
        Kmer<W> sk;
#pragma unroll
        for (int w = 0; w < W; w++) sk.w[w] = a.seeds[i * W + w];
        for (int64_t p = lane; p < L; p += 64) {
            unsigned b;
            if (p < nrev) b = path_base(a.dense[ro + (nr - 1 - p)]);
            else if (p < nrev + k) b = kmer_base<W>(sk, k, (int)(p - nrev));
            else b = path_base(a.dense[fo + (p - nrev - k + 1)]);
            o[p] = "ACGT"[b];
        }
    }
}

// Contigs straight from the STORED paths (vertex entries and descriptors, strand.h): a contig needs one base per vertex, and for the
// vertices of a run that base is in ubase — 1 byte read and 1 written per k-mer instead of expanding 8-byte vertex entries first.
// One wavefront per seed; the dense vertex entries are produced only when somebody asks for vertex lists (Engine::ensure_dense).
struct ContigRleArgs {
    GraphView g;
    RunIndexView runs;
    int64_t n;
    const uint64_t* seeds;
    const uint64_t* pool; const uint32_t* block_table; int max_blocks;
    const uint32_t* strand_c; const uint32_t* strand_n;
    const int64_t* walk_len; const int64_t* contig_off;
    char* out;
};
LDBG_DEV uint64_t rle_stored(const ContigRleArgs& a, int64_t s, uint32_t j) {
    return a.pool[(uint64_t)a.block_table[s * a.max_blocks + j / LDBG_PATH_BLOCK] * LDBG_PATH_BLOCK + (j & (LDBG_PATH_BLOCK - 1))];
}
template <int W>
LDBG_KERNEL void k_contigs_rle(ContigRleArgs a) {
    const int k = a.g.k;
    const int64_t wave = global_tid() / wave_size(), nwaves = (global_nthreads() + wave_size() - 1) / wave_size();
    const uint32_t lane = (uint32_t)wave_lane(), WS = (uint32_t)wave_size();
    for (int64_t i = wave; i < a.n; i += nwaves) {
        if (a.walk_len[i] == 0) continue;
        char* o = a.out + a.contig_off[i];
        const int64_t nr = a.strand_n[2 * i], nrev = nr > 0 ? nr - 1 : 0;
        Kmer<W> sk;
#pragma unroll
        for (int w = 0; w < W; w++) sk.w[w] = a.seeds[i * W + w];
        for (uint32_t p = lane; p < (uint32_t)k; p += WS) o[nrev + p] = "ACGT"[kmer_base<W>(sk, k, (int)p)];
        for (int dir = 0; dir < 2; dir++) {
            const int64_t s = 2 * i + dir;
            const bool fwd = dir == 1;
            const uint32_t nc = a.strand_c[s];
            // place of vertex v >= 1 of this strand in the contig: the reverse strand runs backwards from the seed
            auto place = [&](uint32_t v) -> int64_t { return fwd ? nrev + k - 1 + (int64_t)v : nrev - (int64_t)v; };
            uint32_t base_v = 0;
            uint64_t carry = 0;                                    // the stored entry before this group's first
            for (uint32_t j0 = 0; j0 < nc; j0 += WS) {
                // The kernel is a chain of dependent trips to memory per group of 64 stored entries (few bytes in flight per wavefront:
                // Little's law held it at a fifth of the HBM rate), so every trip that can go, goes: the block's address is read once per
                // group (64 entries never straddle a block), the entry before each lane's comes from its neighbour lane, and a descriptor's
                // payload word is the entry of the lane after its head — three dependent loads per run round became one.
                const uint32_t j = j0 + lane;
                const uint64_t* blk = a.pool + (uint64_t)a.block_table[s * a.max_blocks + j0 / LDBG_PATH_BLOCK] * LDBG_PATH_BLOCK;
                const uint64_t e = j < nc ? LDBG_GLOBAL(const uint64_t, blk)[j & (LDBG_PATH_BLOCK - 1)] : 0ull;
                const uint64_t up = wave_shfl_u64(e, (int)lane - 1);
                const uint64_t prev = (j & (LDBG_PATH_BLOCK - 1)) == 0u ? 0ull : (lane == 0u ? carry : up);
                carry = wave_shfl_u64(e, (int)WS - 1);
                const uint32_t cnt = j < nc ? pd_expanded(prev, e) : 0u;
                const uint32_t incl = wave_incl_scan_u32(cnt);
                const uint32_t at = base_v + incl - cnt;
                const bool head = cnt > 0u && pd_is_head(e);
                if (cnt == 1u && !head && at >= 1u) o[place(at)] = "ACGT"[path_base(e)];
                // RUN heads: a run is a copy of `len` bytes of ubase with a per-byte map (which 2-bit field, complemented or not), forwards or
                // backwards, eight bytes per lane and access, four accesses in flight.  A run is a few hundred bytes (580 on average at C3):
                // given the whole wavefront, three quarters of the lanes have nothing to copy and ONE run is in flight per wavefront — the
                // kernel ran at a fifth of the HBM rate, bound by the latency of run after run.  So the wavefront copies FOUR runs at a time,
                // a group of 16 lanes each (same instruction stream, per-group operands by lane permute).
                const bool is_run = head && LDBG_PD_KIND(e) == LDBG_PD_RUN;
                unsigned long long hb = wave_ballot(is_run);
                const uint32_t GS = WS == 64u ? 16u : WS, g = lane / GS, sub = lane % GS;
                const uint8_t* ub = LDBG_GLOBAL(const uint8_t, a.runs.ubase);
                while (hb) {
                    uint32_t myL = 0;
                    bool gv = false;
                    for (uint32_t q = 0; q < WS / GS; q++) {
                        if (!hb) break;
                        const int L = __builtin_ctzll(hb);
                        hb &= hb - 1;
                        if (g == q) { myL = (uint32_t)L; gv = true; }
                    }
                    const uint64_t he = wave_shfl_u64(e, (int)myL);
                    const uint32_t hat = wave_shfl_u32(at, (int)myL), len_of = wave_shfl_u32(cnt, (int)myL);
                    const uint32_t len = gv ? len_of : 0u;
                    const uint64_t next_e = wave_shfl_u64(e, (int)myL + 1);         // the payload word follows its head
                    const uint64_t payload = !gv ? 0ull : (myL + 1u < WS ? next_e : rle_stored(a, s, j0 + myL + 1u));
                    const bool asc = (he >> 36) & 1ull, inv = (he >> 37) & 1ull;
                    const uint32_t first = (uint32_t)payload;
                    const unsigned shift = (fwd != inv) ? 2u : 0u, comp = inv ? 3u : 0u;
                    auto ascii = [&](unsigned bb) -> unsigned { return (0x54474341u >> (8u * (((bb >> shift) & 3u) ^ comp))) & 0xFFu; };
                    // four bytes of ubase -> four letters: the 2-bit fields of all four at once, then ONE byte permute out of the word "ACGT"
                    // (v_perm_b32: a selector byte 0..3 picks that byte of the second source) — the letters were a third of the kernel's
                    // instructions when they were looked up byte by byte
                    auto ascii4 = [&](uint32_t bb4) -> uint32_t {
                        const uint32_t f = ((bb4 >> shift) & 0x03030303u) ^ (comp * 0x01010101u);
#ifndef LDBG_HOSTSIM
                        return __builtin_amdgcn_perm(0u, 0x54474341u, f);
#else
                        uint32_t v = 0;
                        for (int bI = 0; bI < 4; bI++) v |= ((0x54474341u >> (8u * ((f >> (8 * bI)) & 3u))) & 0xFFu) << (8 * bI);
                        return v;
#endif
                    };
                    // low ends of the two byte ranges; same = both ascend with t or both descend
                    const uint8_t* in_lo = asc ? ub + first : ub + first - (len ? len - 1u : 0u);
                    char* out_lo = LDBG_GLOBAL(char, o) + (fwd ? place(hat) : place(hat + (len ? len - 1u : 0u)));
                    const bool same = asc == fwd;
                    const uint32_t nd = len / 8u;
                    uint32_t max_nd = wave_bcast_u32(nd, 0);
                    for (uint32_t q = 1; q < WS / GS; q++) { const uint32_t o2 = wave_bcast_u32(nd, (int)(q * GS)); max_nd = o2 > max_nd ? o2 : max_nd; }
                    for (uint32_t base = 0; base < max_nd; base += 4u * GS) {      // (the trip count is the same for every lane)
                        uint64_t w[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t ci = base + (uint32_t)q * GS + sub;
                            w[q] = 0;
                            if (ci < nd) __builtin_memcpy(&w[q], in_lo + (same ? 8u * ci : len - 8u - 8u * ci), 8);
                        }
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t ci = base + (uint32_t)q * GS + sub;
                            uint64_t v = (uint64_t)ascii4((uint32_t)w[q]) | ((uint64_t)ascii4((uint32_t)(w[q] >> 32)) << 32);
                            if (!same) v = __builtin_bswap64(v);
                            if (ci < nd) __builtin_memcpy(out_lo + 8u * ci, &v, 8);
                        }
                    }
                    for (uint32_t t = 8u * nd + sub; t < len; t += GS) {           // the last one to seven bytes of every run
                        const unsigned bb = in_lo[same ? t : len - 1u - t];
                        out_lo[t] = (char)ascii(bb);
                    }
                }
                // REPEAT (always the last entry of its strand): the bases of the last recorded revolution, again and again
                hb = wave_ballot(head && !is_run);
                while (hb) {
                    const int L = __builtin_ctzll(hb);
                    hb &= hb - 1;
                    const uint32_t hat = wave_bcast_u32(at, L), len = wave_bcast_u32(cnt, L);
                    const uint64_t payload = (uint32_t)L + 1u < WS ? wave_bcast_u64(e, L + 1) : rle_stored(a, s, j0 + (uint32_t)L + 1);
                    wave_fence();
                    // `len` further bases: the last recorded revolution (vertices first + period .. first + 2 period - 1 of this strand, already
                    // spelled) again and again.  A walk that circles a repeat until maxLength has ~75,000 of them: copied a byte per lane and
                    // trip (a load, then a store the next load had to wait for) one such strand took longer than the rest of the launch
                    // together.  Eight bases per lane and access: eight independent byte loads, one store.
                    const uint32_t first = (uint32_t)payload, period = (uint32_t)(payload >> 32);
                    const uint32_t nch = (len + 7u) / 8u;
                    for (uint32_t c = lane; c < nch; c += WS) {
                        const uint32_t t0 = 8u * c, m = len - t0 < 8u ? len - t0 : 8u;
                        uint32_t r = t0 % period;
                        uint64_t v = 0;
#pragma unroll
                        for (uint32_t b = 0; b < 8u; b++) {
                            if (b < m) {
                                const uint64_t ch = (uint8_t)LDBG_GLOBAL(const char, o)[place(first + period + r)];
                                v |= ch << (8u * (fwd ? b : m - 1u - b));
                            }
                            r++;
                            if (r == period) r = 0;
                        }
                        char* lo = LDBG_GLOBAL(char, o) + (fwd ? place(hat + t0) : place(hat + t0 + m - 1u));
                        if (m == 8u) __builtin_memcpy(lo, &v, 8);
                        else for (uint32_t b = 0; b < m; b++) lo[b] = (char)(v >> (8u * b));
                    }
                }
                base_v += wave_bcast_u32(incl, (int)WS - 1);
            }
        }
    }
}

// vertices of one walk in walk order
template <int W>
LDBG_KERNEL void k_walk_vertices(GraphView g, const uint64_t* dense, int64_t ro, int64_t nr, int64_t fo, int64_t nf,
                                 const uint64_t* term_r, const uint64_t* term_f, int64_t len, uint64_t* words, int64_t* rec,
                                 int32_t* copy, int32_t* index) {
    const int64_t nrev = nr > 0 ? nr - 1 : 0;
    for (int64_t p = global_tid(); p < len; p += global_nthreads()) {
        uint64_t e;
        int idx_label;
        const uint64_t* term;
        if (p < nrev) { e = dense[ro + (nr - 1 - p)]; idx_label = -1; term = term_r; }
        else if (p == nrev) { e = nr > 0 ? dense[ro] : dense[fo]; idx_label = 0; term = nullptr; }
        else { e = dense[fo + (p - nrev)]; idx_label = 1; term = term_f; }
        int64_t ri = path_idx(e);
        Kmer<W> km;
        if (ri >= 0) {
            km = graph_key<W>(g, ri);
            if (path_flip(e)) km = kmer_revcomp<W>(km, g.k);
        } else {
            for (int w = 0; w < W; w++) km.w[w] = term ? term[w] : 0;
        }
        for (int w = 0; w < W; w++) words[p * W + w] = km.w[w];
        rec[p] = ri;
        copy[p] = path_copy(e);
        index[p] = idx_label;
    }
}

// ---- which ROI records the vertices of each walk are (Partition.markUsedRois, J/commands/discover/call/Partition.java:238-257)
// roi_of[record] = number of that record's k-mer in the ROI graph (0xFFFFFFFF: not a ROI k-mer).  One wavefront per walk;
// fill == 0: count hits (and note walks whose graph holds a vertex without a record); fill == 1: write them.
struct RoiHitArgs {
    const uint32_t* roi_of;
    int64_t n;
    const uint64_t* dense; const int64_t* strand_off; const int64_t* walk_len;
    unsigned long long* count;     // [n]
    const int64_t* out_off;        // [n + 1] (fill pass)
    uint32_t* out;
    uint8_t* has_null;             // [n]
    int fill;
};
LDBG_KERNEL void k_roi_hits(RoiHitArgs a) {
    const int64_t wave = global_tid() >> 6, lane = global_tid() & 63, nwaves = (global_nthreads() + 63) >> 6;
    for (int64_t i = wave; i < a.n; i += nwaves) {
        const int64_t ro = a.strand_off[2 * i], nr = a.strand_off[2 * i + 1] - ro;
        const int64_t fo = a.strand_off[2 * i + 1], nf = a.strand_off[2 * i + 2] - fo;
        const bool in_walk = a.walk_len[i] > 0;
        bool null_seen = false;
        for (int64_t p = lane; p < nr + nf; p += 64) {
            const bool fwd_part = p >= nr;
            const uint64_t e = a.dense[fwd_part ? fo + (p - nr) : ro + p];
            const int64_t idx = path_idx(e);
            if (idx < 0) { null_seen = true; continue; }
            if (fwd_part && p == nr && nr > 0) continue;       // the seed is entry 0 of both strands
            if (!in_walk) continue;
            const uint32_t r = a.roi_of[idx];
            if (r == 0xFFFFFFFFu) continue;
            const unsigned long long slot = atomic_add_u64(&a.count[i], 1ull);
            if (a.fill) a.out[a.out_off[i] + (int64_t)slot] = r;
        }
        if (!a.fill && null_seen) a.has_null[i] = 1;
    }
}
template <int W>
LDBG_KERNEL void k_roi_of(GraphView g, GraphView rois, uint32_t* roi_of) {
    for (int64_t i = global_tid(); i < rois.N; i += global_nthreads()) {
        GraphView exact = g;
        exact.java_tiny = 0;
        const int64_t idx = graph_find_canonical<W>(exact, graph_key<W>(rois, i));
        if (idx >= 0) roi_of[idx] = (uint32_t)i;
    }
}

// ------------------------------------------------------------------ host
static uint32_t next_pow2(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return (uint32_t)p; }
static int grid_for(int64_t n, int block, int max_blocks) {
    int64_t b = (n + block - 1) / block;
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, max_blocks));
}

Engine::Engine(const ldbg_engine_config& c) : cfg(c) {
    // TraversalEngineFactory.make :54-88
    if (c.n_traversal <= 0) throw StatusError(LDBG_ERR_CORTEXJDK, "Traversal color(s) must be specified.");
    if (!c.graph) throw StatusError(LDBG_ERR_CORTEXJDK, "Must provide graph to traverse.");
    graph = (const Graph*)c.graph;
    rois = (const Graph*)c.rois;
    stream_ = graph->stream;           // (replaced by the engine's own stream at the end of the constructor)
    const int nc = graph->hdr.C;
    auto fail = [&](const char* what, int col) {
        throw StatusError(LDBG_ERR_CORTEXJDK, std::string(what) + " colors must be between 0 and " + std::to_string(nc) + " (provided " + std::to_string(col) + ")");
    };
    for (int i = 0; i < c.n_traversal; i++) if (c.traversal_colors[i] >= nc || c.traversal_colors[i] < 0) fail("Traversal", c.traversal_colors[i]);
    for (int i = 0; i < c.n_joining; i++) if (c.joining_colors[i] < 0 || c.joining_colors[i] >= nc) fail("Joining", c.joining_colors[i]);
    for (int i = 0; i < c.n_recruitment; i++) if (c.recruitment_colors[i] < 0 || c.recruitment_colors[i] >= nc) fail("Recruitment", c.recruitment_colors[i]);
    for (int i = 0; i < c.n_secondary; i++) if (c.secondary_colors[i] < 0 || c.secondary_colors[i] >= nc) fail("Secondary", c.secondary_colors[i]);
    if (c.stopping_rule < 0 || c.stopping_rule >= LDBG_STOP_COUNT) throw StatusError(LDBG_ERR_CORTEXJDK, "Must provide stopping rule for graph traversal");

    rt::set_device(graph->device);
    view.g = graph->view;
    view.trav_mask = view.recruit_mask = view.join_mask = 0;
    for (int i = 0; i < c.n_traversal; i++) view.trav_mask |= 1u << c.traversal_colors[i];
    for (int i = 0; i < c.n_recruitment; i++) view.recruit_mask |= 1u << c.recruitment_colors[i];
    for (int i = 0; i < c.n_joining; i++) view.join_mask |= 1u << c.joining_colors[i];
    view.trav_sel4 = view.recruit_sel4 = 0;
    for (int col = 0; col < 4; col++) {
        if ((view.trav_mask >> col) & 1u) view.trav_sel4 |= 0xFFu << (8 * col);
        if ((view.recruit_mask >> col) & 1u) view.recruit_sel4 |= 0xFFu << (8 * col);
    }
    view.first_trav = c.traversal_colors[0];
    view.stopper = c.stopping_rule;
    view.max_len = c.max_branch_length;
    view.connect_all = c.connect_all_neighbors;
    view.strict_flip = c.strict_java_flip;
    view.lean_rows = (view.g.k & 1) && row_is_packed(view.g) ? 1 : 0;
    // initializeLinkStore/updateLinkStore :548-597: only link sets whose colour-0 sample is a traversal sample
    for (int i = 0; i < c.nlinks; i++) {
        const Links* l = (const Links*)c.links[i];
        if (!l) continue;
        bool mine = false;
        for (int t = 0; t < c.n_traversal; t++)
            if (!l->sample_names.empty() && l->sample_names[0] == graph->hdr.colors[c.traversal_colors[t]].sample_name) mine = true;
        if (mine) my_links.push_back(l);
    }
    merged_.reset(new MergedLinks(my_links, *graph));
    view.links = merged_->view;
    view.link_flag_mask = merged_->flag_mask;
    // ec.getLinks().isEmpty() (not "my links") decides whether dfs uses the cursor (:363, :379)
    view.cursor_on = c.nlinks > 0 ? 1 : 0;
    // Every engine over a resident table has its own HIP stream: engines on one graph, driven by different host threads, run side by side —
    // a walk launch is bound by its longest strands, and the compute units its early finishers leave idle take the next
    // batch of another engine.  (An engine over the image of a sharded table stays on that graph's stream: its rounds are ordered with
    // the collectives by the caller's stream.)
    if (!graph->is_image) {
        rt::set_device(graph->device);
        rt::stream_sync(graph->stream);
        own_stream_ = rt::stream_create();
        if (own_stream_) stream_ = own_stream_;
    }
}
// entry of every call that queues device work: this engine's device, and whatever the graph's own stream still holds (a link set bound a
// moment ago, the neighbour index) is complete before this engine's stream goes on
void Engine::enter() {
    rt::set_device(graph->device);
    if (stream_ != graph->stream) rt::stream_sync(graph->stream);
}

Engine::~Engine() { if (own_stream_) { try { rt::set_device(graph->device); } catch (...) {} }
                    sharded_abort(); clear_batch(); drop_batch_seeds(); drop_spares(); rt::hfree_pinned(h_small_); rt::hfree_pinned(h_log_); rt::hfree_pinned(h_stage_[0]); rt::hfree_pinned(h_stage_[1]); release_scratch(); rt::dfree(d_frames_); rt::dfree(d_roi_bits_); rt::dfree(d_roi_of_);
                    if (own_stream_) { quiesce(); rt::stream_destroy(own_stream_); } }

// ROI hits of the walks of the last batch: offsets[n+1] into hits (ROI record numbers, order within a walk arbitrary),
// has_null[i] = the dfs graph of seed i holds a vertex without a record
void Engine::walk_roi_hits(int64_t* offsets, uint32_t* hits, int64_t capacity, uint8_t* has_null) {
    if (!rois) throw StatusError(LDBG_ERR_ARG, "walk_roi_hits: the engine has no ROI graph");
    enter();
    rt::stream_t s = stream_;
    const int W = graph->hdr.W;
    if (!d_roi_of_) {
        if (rois->hdr.k != graph->hdr.k) throw StatusError(LDBG_ERR_ARG, "the ROI graph must have the k-mer size of the traversed graph");
        const size_t nrec = (size_t)std::max<int64_t>(1, graph->view.N);
        d_roi_of_ = rt::dmalloc(nrec * 4);
        rt::dmemset(d_roi_of_, 0xFF, nrec * 4, s);
        if (rois->view.N > 0) {
            const int grid = grid_for(rois->view.N, 256, 4096);
            switch (W) {
                case 1: LDBG_LAUNCH(k_roi_of<1>, grid, 256, s, graph->view, rois->view, (uint32_t*)d_roi_of_); break;
                case 2: LDBG_LAUNCH(k_roi_of<2>, grid, 256, s, graph->view, rois->view, (uint32_t*)d_roi_of_); break;
                case 3: LDBG_LAUNCH(k_roi_of<3>, grid, 256, s, graph->view, rois->view, (uint32_t*)d_roi_of_); break;
                default: LDBG_LAUNCH(k_roi_of<4>, grid, 256, s, graph->view, rois->view, (uint32_t*)d_roi_of_); break;
            }
        }
    }
    // pass 1: counts
    std::vector<std::vector<unsigned long long>> counts(chunks.size());
    int64_t total = 0;
    offsets[0] = 0;
    for (size_t ci = 0; ci < chunks.size(); ci++) {
        WalkChunk& c = chunks[ci];
        RoiHitArgs a;
        ensure_dense(c);
        a.roi_of = (const uint32_t*)d_roi_of_; a.n = c.n; a.dense = (const uint64_t*)c.d_path;
        const int64_t* d_soff = (const int64_t*)c.d_strand_off;
        const int64_t* d_wl = (const int64_t*)c.d_walk_len;
        unsigned long long* d_cnt = (unsigned long long*)rt::dmalloc((size_t)std::max<int64_t>(1, c.n) * 8);
        uint8_t* d_null = (uint8_t*)rt::dmalloc((size_t)std::max<int64_t>(1, c.n));
        rt::dmemset(d_cnt, 0, (size_t)std::max<int64_t>(1, c.n) * 8, s);
        rt::dmemset(d_null, 0, (size_t)std::max<int64_t>(1, c.n), s);
        a.strand_off = d_soff; a.walk_len = d_wl; a.count = d_cnt; a.out_off = nullptr; a.out = nullptr; a.has_null = d_null; a.fill = 0;
        const int grid = grid_for(c.n * 64, 256, 4096);
        LDBG_LAUNCH(k_roi_hits, grid, 256, s, a);
        counts[ci].resize((size_t)c.n);
        rt::d2h(counts[ci].data(), d_cnt, (size_t)c.n * 8, s);
        rt::d2h(has_null + c.first, d_null, (size_t)c.n, s);
        rt::stream_sync(s);
        std::vector<int64_t> off((size_t)c.n + 1, 0);
        for (int64_t i = 0; i < c.n; i++) { off[i + 1] = off[i] + (int64_t)counts[ci][i]; offsets[c.first + i + 1] = total + off[i + 1]; }
        const int64_t chunk_total = off[c.n];
        if (total + chunk_total <= capacity && chunk_total > 0) {          // pass 2: fill
            int64_t* d_off = (int64_t*)rt::dmalloc((size_t)(c.n + 1) * 8);
            uint32_t* d_out = (uint32_t*)rt::dmalloc((size_t)chunk_total * 4);
            rt::h2d(d_off, off.data(), (size_t)(c.n + 1) * 8, s);
            rt::dmemset(d_cnt, 0, (size_t)c.n * 8, s);
            a.out_off = d_off; a.out = d_out; a.fill = 1;
            LDBG_LAUNCH(k_roi_hits, grid, 256, s, a);
            rt::d2h(hits + total, d_out, (size_t)chunk_total * 4, s);
            rt::stream_sync(s);
            rt::dfree(d_off); rt::dfree(d_out);
        }
        total += chunk_total;
        rt::dfree(d_cnt); rt::dfree(d_null);
    }
    if (total > capacity) throw StatusError(LDBG_ERR_CAPACITY, "hit buffer too small: need " + std::to_string(total));
}

void Engine::ensure_run_index() {
    if (runs_ || getenv("LDBG_NO_RUNS") || !(view.g.k & 1) || graph->is_image) return;
    runs_.reset(new RunIndex(view, graph->device, stream_));
    profile_add("run_index", runs_->build_ms);
    if (getenv("LDBG_HOST_TIMES"))
        fprintf(stderr, "[ldbg] run index: %lld chains hold %lld of %lld records, built in %.1f ms\n", (long long)runs_->n_chains,
                (long long)runs_->n_in_chains, (long long)view.g.N, runs_->build_ms);
}

// the dense 8-byte vertex entries of a chunk's walks, expanded from the stored paths the first time they are needed
void Engine::ensure_dense(WalkChunk& c) {
    if (!c.dense_pending) return;
    enter();
    rt::stream_t s = stream_;
    const int64_t ns = 2 * c.n;
    c.d_path = result_alloc((size_t)std::max<int64_t>(1, c.total_entries) * 8, &c.path_cap);
    unsigned* d_ovf = (unsigned*)rt::dmalloc(4);
    rt::dmemset(d_ovf, 0, 4, s);
    ExpandArgs xa;
    xa.pool = (const uint64_t*)d_pool_; xa.block_table = (const uint32_t*)d_block_table_; xa.max_blocks = c.max_blocks;
    xa.strand_c = (const uint32_t*)c.d_strand_c; xa.strand_off = (const int64_t*)c.d_strand_off; xa.n_strands = ns;
    xa.runs = c.runs; xa.dense = (uint64_t*)c.d_path; xa.overflow = d_ovf;
    LDBG_LAUNCH(k_expand_paths, grid_for(ns * 64, LDBG_STREAM_BLOCK, LDBG_STREAM_GRID), LDBG_STREAM_BLOCK, s, xa);
    unsigned ovf = 0;
    rt::d2h(&ovf, d_ovf, 4, s);
    rt::stream_sync(s);
    rt::dfree(d_ovf);
    c.dense_pending = false;
    if (ovf) throw StatusError(LDBG_ERR_UNSUPPORTED, "a vertex was visited more than 32767 times in one walk");
}
// before the path pool is used for something else: the vertex entries of the walks still held
void Engine::materialize_pending() { for (auto& c : chunks) ensure_dense(c); }

// vertices (and, in a dfs log, markers) the stored entries of every strand stand for
LDBG_KERNEL void k_path_lengths(const uint64_t* pool, const uint32_t* block_table, int max_blocks, const uint32_t* strand_c, int64_t n_strands, uint32_t* len) {
    const int64_t wave = global_tid() / wave_size(), nwaves = (global_nthreads() + wave_size() - 1) / wave_size();
    const uint32_t lane = (uint32_t)wave_lane(), WS = (uint32_t)wave_size();
    for (int64_t s = wave; s < n_strands; s += nwaves) {
        const uint32_t nc = strand_c[s];
        uint32_t total = 0;
        for (uint32_t j0 = 0; j0 < nc; j0 += WS) {
            const uint32_t j = j0 + lane;
            uint32_t cnt = 0;
            if (j < nc) {
                const uint64_t* blk = pool + (uint64_t)block_table[s * max_blocks + j / LDBG_PATH_BLOCK] * LDBG_PATH_BLOCK;
                const uint32_t o = j & (LDBG_PATH_BLOCK - 1);
                cnt = pd_expanded(o ? blk[o - 1] : 0ull, blk[o]);
            }
            total += wave_bcast_u32(wave_incl_scan_u32(cnt), (int)WS - 1);
        }
        if (lane == 0) len[s] = total;
    }
}
void Engine::launch_path_lengths(const uint32_t* d_strand_c, int64_t n_strands, int max_blocks, uint32_t* d_len) {
    LDBG_LAUNCH(k_path_lengths, grid_for(n_strands * 64, LDBG_STREAM_BLOCK, LDBG_STREAM_GRID), LDBG_STREAM_BLOCK, stream_, (const uint64_t*)d_pool_, (const uint32_t*)d_block_table_, max_blocks,
                d_strand_c, n_strands, d_len);
}
void Engine::launch_expand_paths(const uint32_t* d_strand_c, const int64_t* d_strand_off, int64_t n_strands, uint64_t* d_dense, int max_blocks, const RunIndexView& runs,
                                 unsigned* d_overflow) {
    ExpandArgs xa;
    xa.pool = (const uint64_t*)d_pool_; xa.block_table = (const uint32_t*)d_block_table_; xa.max_blocks = max_blocks;
    xa.strand_c = d_strand_c; xa.strand_off = d_strand_off; xa.n_strands = n_strands;
    xa.runs = runs; xa.dense = d_dense; xa.overflow = d_overflow;
    LDBG_LAUNCH(k_expand_paths, grid_for(n_strands * 64, LDBG_STREAM_BLOCK, LDBG_STREAM_GRID), LDBG_STREAM_BLOCK, stream_, xa);
}

void Engine::launch_compact_paths(const int64_t* d_strand_off, int64_t n_strands, uint64_t* d_dense, int max_blocks) {
    LDBG_LAUNCH(k_compact_paths, grid_for(n_strands * 64, 256, 4096), 256, stream_, (const uint64_t*)d_pool_, (const uint32_t*)d_block_table_, max_blocks,
                d_strand_off, n_strands, d_dense);
}

void Engine::release_scratch() {
    drop_spares();
    rt::dfree(d_vpool_); rt::dfree(d_ls_); rt::dfree(d_snap_); rt::dfree(d_pool_); rt::dfree(d_block_table_);
    d_vpool_ = d_ls_ = d_snap_ = d_pool_ = d_block_table_ = nullptr;
    n_slots_ = 0; n_blocks_ = 0; bt_strands_ = 0; vpool_entries_ = 0; vpool_dirty_ = 0;
}

// Zeroing tens of GB of handed-out tables is a tenth of a step's time with the runtime's fill; 16 bytes per lane, grid-stride,
// runs at the write bandwidth of the device.
LDBG_KERNEL void k_zero16(uint64_t* p, uint64_t n_pairs) {
    for (uint64_t i = (uint64_t)global_tid(); i < n_pairs; i += (uint64_t)global_nthreads()) {
#ifndef LDBG_HOSTSIM
        struct alignas(16) U2 { uint64_t a, b; };
        ((U2*)p)[i] = U2{0ull, 0ull};
#else
        p[2 * i] = 0; p[2 * i + 1] = 0;
#endif
    }
}
void Engine::zero_dirty_tables(rt::stream_t s) {
    if (vpool_dirty_ == 0) { vpool_dirty_ = vpool_entries_; return; }
    const uint64_t n = std::min<uint64_t>((vpool_dirty_ + 1) & ~1ull, vpool_entries_ & ~1ull);     // entries, in pairs
    if (n > 0) LDBG_LAUNCH(k_zero16, 256 * 16, 256, s, (uint64_t*)d_vpool_, n / 2);
    if (n < std::min<uint64_t>(vpool_dirty_, vpool_entries_)) rt::dmemset((uint64_t*)d_vpool_ + n, 0, (size_t)(std::min<uint64_t>(vpool_dirty_, vpool_entries_) - n) * 8, s);
    vpool_dirty_ = vpool_entries_;      // until the launch that follows has reported how much it handed out
}

void* Engine::result_alloc(size_t bytes, size_t* cap) {
    int best = -1;
    for (int i = 0; i < (int)spares_.size(); i++)
        if (spares_[i].bytes >= bytes && (best < 0 || spares_[i].bytes < spares_[best].bytes)) best = i;
    if (best >= 0) {
        Spare sp = spares_[best];
        spares_.erase(spares_.begin() + best);
        *cap = sp.bytes;
        return sp.p;
    }
    void* p = nullptr;
    try { p = rt::dmalloc(bytes); }
    catch (const StatusError&) { drop_spares(); p = rt::dmalloc(bytes); }     // the spares may be what is in the way
    *cap = bytes;
    return p;
}
void Engine::result_free(void* p, size_t cap) {
    if (!p) return;
    if (spares_.size() >= 8 || cap == 0) { rt::dfree(p); return; }
    spares_.push_back({p, cap});
}
void Engine::drop_spares() {
    for (auto& sp : spares_) rt::dfree(sp.p);
    spares_.clear();
}

// Blocks go back to the block cache (rt::tfree) only once nothing on the device can still touch them: wait for the streams this engine
// has work on (hipFree used to do that by itself).  Idle streams: microseconds.
void Engine::quiesce() noexcept {
    try {
        rt::stream_sync(stream_);
        if (sharded_run_ && sharded_stream_ && sharded_stream_ != stream_) rt::stream_sync(sharded_stream_);
    } catch (...) {}
}
void Engine::clear_batch() {
    if (!chunks.empty()) quiesce();
    for (auto& c : chunks) {
        result_free(c.d_path, c.path_cap); result_free(c.d_contigs, c.contigs_cap); rt::tfree(c.d_seed_words); rt::tfree(c.d_term);
        rt::tfree(c.d_strand_c); rt::tfree(c.d_strand_off); rt::tfree(c.d_contig_off); rt::tfree(c.d_walk_len);
    }
    chunks.clear();
    batch_n = batch_bytes = batch_traversed = 0;
}

// per-slot link stores, the visited-table pool, the path block pool and the block table; kept across batches
// pool entries a strand needs at worst: tables of init, 4 x init, ... entries, the last one capped at vmax (grown tables are not given back)
uint64_t vt_series(uint64_t init, uint64_t vmax) {
    uint64_t c = std::min(init, vmax), sum = 0;
    while (true) { sum += c; if (c >= vmax) break; c = std::min(c * 4, vmax); }
    return sum;
}
uint32_t vt_initial_entries() {
    uint32_t v = LDBG_VT_INITIAL;
    if (const char* ev = getenv("LDBG_VT_INITIAL")) v = std::max<uint32_t>(64u, next_pow2((uint64_t)atoll(ev)));   // tuning knob
    return v;
}

void Engine::ensure_scratch(int64_t ns, uint32_t ecap, int max_blocks, uint64_t table_floor, bool small) {
    rt::stream_t s = stream_;
    // small: the pools of a walk with the run index — a strand's table holds fringes and junction vertices, its path a few descriptors per
    // stretch — start at a fraction of the worst case (reserving the worst case was 200 GB of hipMalloc: 1.8 s of a 3.7 s first batch)
    // and are enlarged x4 by walk_batch_run when a batch does run out (ST_POOL_FULL: the batch is walked again, nothing is traded)
    const uint64_t small_v = (uint64_t)ns * 1024ull * scratch_scale_, small_b = (uint64_t)ns * 2ull * scratch_scale_;
    const bool reusable = d_vpool_ && ecap_ == ecap && max_blocks_ == max_blocks && bt_strands_ >= ns && table_floor_ >= table_floor;
    if (reusable && (scratch_full_ || (small && scratch_scale_ == scratch_scale_built_))) return;
    release_scratch();
    size_t free_b = 0, total_b = 0;
    rt::mem_info(&free_b, &total_b);
    const int64_t slots = ((ns + 63) / 64) * 64;           // every strand of the batch gets a lane
    // memory split: 40% of what is free for the visited-table pool, 35% for the path pool (both capped by need)
    const uint64_t vcap_max = next_pow2(2ull * (uint64_t)(cfg.max_branch_length + 12));
    uint64_t want_v = std::max<uint64_t>((uint64_t)ns * vt_series(vt_initial_entries(), vcap_max), table_floor) + LDBG_VT_INITIAL;
    uint64_t want_blocks = (uint64_t)ns * (uint64_t)max_blocks;
    scratch_full_ = !small || (small_v >= want_v && small_b >= want_blocks);
    if (small) { want_v = std::min<uint64_t>(want_v, std::max<uint64_t>(small_v, table_floor) + LDBG_VT_INITIAL); want_blocks = std::min<uint64_t>(want_blocks, std::max<uint64_t>(small_b, 2)); }
    scratch_scale_built_ = scratch_scale_;
    vpool_entries_ = std::max<uint64_t>(LDBG_VT_INITIAL * 2, std::min<uint64_t>(want_v, (uint64_t)(free_b * 0.40) / 8));
    table_floor_ = table_floor;
    n_blocks_ = std::max<uint64_t>(2, std::min<uint64_t>(want_blocks, (uint64_t)(free_b * 0.35) / (LDBG_PATH_BLOCK * 8)));
    d_vpool_ = rt::dmalloc((size_t)vpool_entries_ * 8);
    d_ls_ = rt::dmalloc((size_t)slots * ecap * sizeof(LsElem));
    d_snap_ = rt::dmalloc((size_t)slots * LDBG_SNAP_CAP * sizeof(LsSnap));
    d_pool_ = rt::dmalloc((size_t)n_blocks_ * LDBG_PATH_BLOCK * 8);
    d_block_table_ = rt::dmalloc((size_t)ns * max_blocks * 4);
    rt::dmemset(d_vpool_, 0, (size_t)vpool_entries_ * 8, s);
    vpool_dirty_ = 0;
    n_slots_ = slots; ecap_ = ecap; max_blocks_ = max_blocks; bt_strands_ = ns;
}

// LDBG_HOST_TIMES=1: wall-clock laps of the host side of a batch on stderr (diagnostics)
struct HostLaps {
    bool on = getenv("LDBG_HOST_TIMES") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[ldbg] host %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// the seeds of a batch go to the device as they are (n x k ASCII bytes; nothing to copy when the caller's seeds are there already) and
// become packed words there: on the host that conversion was 0.5 ms of a 7 ms step at C3
void Engine::seeds_to_device(const char* seeds, int64_t n, bool seeds_on_device) {
    rt::stream_t s = stream_;
    const int k = graph->hdr.k, W = graph->hdr.W;
    drop_batch_seeds();
    d_batch_words_ = rt::tmalloc((size_t)std::max<int64_t>(1, n) * W * 8);
    d_batch_valid_ = rt::tmalloc((size_t)std::max<int64_t>(1, n));
    if (n <= 0) return;
    if (!seeds_on_device) {           // (the copy of page-locked seeds is asynchronous: the batch's first wait covers it, the block goes back with the batch's seeds)
        d_batch_ascii_ = rt::tmalloc((size_t)n * k);
        rt::h2d(d_batch_ascii_, seeds, (size_t)n * k, s);
    }
    LDBG_LAUNCH(k_seed_words, grid_for(n, 256, 1024), 256, s, (const unsigned char*)(seeds_on_device ? (const void*)seeds : d_batch_ascii_), n, k, W,
                (uint64_t*)d_batch_words_, (uint8_t*)d_batch_valid_);
}
void Engine::drop_batch_seeds() {
    if (d_batch_words_ || d_batch_valid_ || d_batch_ascii_) quiesce();
    rt::tfree(d_batch_words_); rt::tfree(d_batch_valid_); rt::tfree(d_batch_ascii_);
    d_batch_words_ = d_batch_valid_ = d_batch_ascii_ = nullptr;
}

void Engine::walk_batch_run(const char* seeds, int64_t n, int64_t* total_bytes, int64_t* traversed, bool seeds_on_device) {
    if (cfg.stopping_rule != LDBG_STOP_CONTIG || cfg.connect_all_neighbors)
        throw StatusError(LDBG_ERR_UNSUPPORTED, "walk_batch runs ContigStopper without connectAllNeighbors; use dfs_batch for other rules");
    if (cfg.n_secondary > 0) throw StatusError(LDBG_ERR_UNSUPPORTED, "secondary colours are not supported by walk_batch");
    enter();
    HostLaps laps;
    clear_batch();
    laps.lap("clear_batch");
    seeds_to_device(seeds, n, seeds_on_device);
    laps.lap("seed words (device)");
    batch_n = n;
    int64_t trav = 0;
    // the whole batch in one launch; if the path pool runs dry the batch is split and re-run (exactness first)
    std::vector<std::pair<int64_t, int64_t>> todo{{0, n}};
    while (!todo.empty()) {
        auto [first, cnt] = todo.back();
        todo.pop_back();
        if (cnt <= 0) continue;
        WalkChunk c;
        int64_t t = 0;
        if (run_chunk(first, cnt, c, &t)) { trav += t; chunks.push_back(std::move(c)); }
        else if (!scratch_full_ && scratch_scale_ < (1ull << 24)) {
            scratch_scale_ *= 4;                          // the small pools of a run-index walk were too small for this batch: enlarge, walk it again
            pool_growths_++;
            todo.push_back({first, cnt});
        } else {
            if (cnt == 1) throw StatusError(LDBG_ERR_HIP, "path pool too small for a single walk: not enough device memory");
            todo.push_back({first + cnt / 2, cnt - cnt / 2});
            todo.push_back({first, cnt / 2});
        }
    }
    std::sort(chunks.begin(), chunks.end(), [](const WalkChunk& a, const WalkChunk& b) { return a.first < b.first; });
    batch_traversed = trav;
    batch_bytes = 0;
    for (auto& c : chunks) batch_bytes += c.total_bytes;
    if (total_bytes) *total_bytes = batch_bytes;
    if (traversed) *traversed = trav;
}

// One prepared walk launch: the arguments and the device buffers a batch (or one chunk of it) lives in.  walk_prepare sets it up,
// walk_launch runs the kernel (once for a table that is resident; once per bulk-synchronous round over a sharded table's image),
// walk_finish assembles the results.
struct WalkRun {
    WalkArgs a;
    int W = 0, k = 0, block = 64, grid = 1, max_blocks = 0;
    int64_t first = 0, n = 0, ns = 0;
    uint32_t vcap_max = 0;
    bool want_times = false;
    double walk_ms = 0;
    rt::Event ev0, ev1;                    // around the (last) launch of the walk kernel; read after the batch's first wait (walk_finish)
    bool timed = false;
    uint32_t *d_strand_n = nullptr, *d_strand_c = nullptr, *d_retry = nullptr, *d_status = nullptr, *d_iters = nullptr;
    uint8_t *d_quirk = nullptr, *d_seed_valid = nullptr;
    unsigned long long* d_ctr = nullptr;
    void* d_save = nullptr;
    WalkChunk out;
    void free_tmp() {
        rt::tfree(d_strand_n); rt::tfree(d_strand_c); rt::tfree(d_retry); rt::tfree(d_seed_valid); rt::tfree(d_status); rt::tfree(d_iters); rt::tfree(d_ctr); rt::tfree(d_quirk);
        rt::dfree(d_save);
        d_strand_n = d_strand_c = d_retry = d_status = d_iters = nullptr; d_quirk = d_seed_valid = nullptr; d_ctr = nullptr; d_save = nullptr;
    }
};
#define WALKRUN_ALIASES(r) \
    WalkArgs& a = (r).a; WalkChunk& out = (r).out; const int W = (r).W, k = (r).k; (void)k; const int64_t first = (r).first, n = (r).n, ns = (r).ns; (void)first; \
    const int max_blocks = (r).max_blocks; const uint32_t vcap_max = (r).vcap_max; (void)vcap_max; \
    uint32_t*& d_strand_n = (r).d_strand_n; uint32_t*& d_strand_c = (r).d_strand_c; uint32_t*& d_retry = (r).d_retry; uint32_t*& d_status = (r).d_status; \
    uint32_t*& d_iters = (r).d_iters; uint8_t*& d_quirk = (r).d_quirk; uint8_t*& d_seed_valid = (r).d_seed_valid; unsigned long long*& d_ctr = (r).d_ctr; \
    (void)d_retry; (void)d_seed_valid; rt::stream_t s = stream_; auto free_tmp = [&] { (r).free_tmp(); }; (void)free_tmp

static void launch_k_walk(const WalkRun& r, const WalkArgs& a, rt::stream_t s) {
    const int block = r.block, grid = r.grid;
#define LDBG_WALK_CASE(WW) \
    if (a.img_on) LDBG_LAUNCH((k_walk<WW, 64, true>), grid, 64, s, a); \
    else if (block == 16) LDBG_LAUNCH((k_walk<WW, 16, false>), grid, 16, s, a); \
    else if (block == 64) LDBG_LAUNCH((k_walk<WW, 64, false>), grid, 64, s, a); \
    else LDBG_LAUNCH((k_walk<WW, 32, false>), grid, 32, s, a)
    switch (r.W) {
        case 1: LDBG_WALK_CASE(1); break;
        case 2: LDBG_WALK_CASE(2); break;
        case 3: LDBG_WALK_CASE(3); break;
        default: LDBG_WALK_CASE(4); break;
    }
#undef LDBG_WALK_CASE
}

// strands of this rank that are not done: those a lane holds (suspended) + those still in the queue; and the requests of the round
LDBG_KERNEL void k_round_stats(const unsigned long long* ctr, int64_t ns, const unsigned long long* n_req, int64_t* stats) {
    if (global_tid() != 0) return;
    const int64_t handed = (int64_t)ctr[0] < ns ? (int64_t)ctr[0] : ns;
    stats[0] = (int64_t)ctr[4] + (ns - handed);
    stats[1] = (int64_t)*n_req;
    stats[2] = (int64_t)*(const unsigned*)(n_req + 1);      // the image's overflow flag (image.cpp: d_ctr_[2]): a full image ends the rounds on every rank
}

// img: the walk runs on the local image of a sharded table (image.h): strands suspend where a row is missing, seeds come as image slots
void Engine::walk_prepare(int64_t first_, int64_t n_, WalkRun& r, ShardImage* img, const int32_t* d_seed_slot) {
    r.W = graph->hdr.W; r.k = graph->hdr.k; r.first = first_; r.n = n_; r.ns = 2 * n_;
    r.out.first = first_; r.out.n = n_;
    // a strand's visited table never needs more than this (longest possible branch at load <= 1/2)
    r.vcap_max = std::max<uint32_t>(64u, next_pow2(2ull * (uint64_t)(cfg.max_branch_length + 12)));
    r.max_blocks = (int)(((int64_t)cfg.max_branch_length + 2 + LDBG_PATH_BLOCK - 1) / LDBG_PATH_BLOCK);
    WALKRUN_ALIASES(r);
    HostLaps laps;
    const bool lean_pools = !img && (view.g.k & 1) && !getenv("LDBG_NO_RUNS") && !getenv("LDBG_VT_INITIAL") && !getenv("LDBG_FULL_POOLS");
    ensure_scratch(ns, link_store_capacity, max_blocks, 0, lean_pools);
    zero_dirty_tables(s);
    laps.lap("scratch + zero (issued)");

    out.d_seed_words = rt::tmalloc((size_t)n * W * 8);           // the chunk keeps its seeds (k_contigs, walk_vertices); the batch's go with the next batch
    rt::d2d(out.d_seed_words, (const uint64_t*)d_batch_words_ + first * W, (size_t)n * W * 8, s);
    d_seed_valid = (uint8_t*)rt::tmalloc((size_t)n);
    rt::d2d(d_seed_valid, (const uint8_t*)d_batch_valid_ + first, (size_t)n, s);
    out.d_term = rt::tmalloc((size_t)ns * W * 8);
    d_strand_n = (uint32_t*)rt::tmalloc((size_t)ns * 4);
    d_strand_c = (uint32_t*)rt::tmalloc((size_t)ns * 4);
    d_status = (uint32_t*)rt::tmalloc((size_t)ns * 4);
    d_iters = (uint32_t*)rt::tmalloc((size_t)ns * 4);
    d_quirk = (uint8_t*)rt::tmalloc((size_t)ns);
    d_ctr = (unsigned long long*)rt::tmalloc(256);      // [0..7] queue / pool cursors, [8..] the step-kind counters (strand.h: WalkArgs::kinds)
    rt::dmemset(d_ctr, 0, 256, s);
    rt::dmemset(out.d_term, 0, (size_t)ns * W * 8, s);

    laps.lap("small allocations");
    a.e = view;
    if (!img) ensure_run_index();
    if (runs_ && !img) a.e.runs = runs_->view;
    a.retry = nullptr;
    a.img_on = img ? 1 : 0;
    a.seed_slot = d_seed_slot;
    a.save = nullptr; a.unfinished = d_ctr + 4;
    a.kinds = d_ctr + 8;
    if (img) {
        a.img = img->view((uint64_t*)view.links.rec_of);
        r.d_save = rt::dmalloc((size_t)std::max<int64_t>(64, n_slots_) * sizeof(StrandSave));
        rt::dmemset(r.d_save, 0, (size_t)std::max<int64_t>(64, n_slots_) * sizeof(StrandSave), s);
        a.save = (StrandSave*)r.d_save;
    }
    a.seeds = (const uint64_t*)out.d_seed_words;
    a.seed_valid = d_seed_valid;
    a.n_strands = ns;
    a.n_slots = std::min<int64_t>(n_slots_, ((ns + 63) / 64) * 64);
    if (const char* ev = getenv("LDBG_MAX_SLOTS")) a.n_slots = std::max<int64_t>(64, std::min<int64_t>(a.n_slots, (atoll(ev) / 64) * 64));   // tuning knob
    {   // a stride coprime to the number of strands, a few thousand apart
        auto gcd = [](int64_t x, int64_t y) { while (y) { int64_t t = x % y; x = y; y = t; } return x; };
        int64_t st = 7919;
        if (const char* ev = getenv("LDBG_FETCH_STRIDE")) st = std::max<int64_t>(1, atoll(ev));
        while (gcd(st, ns) != 1) st++;
        a.fetch_stride = st % ns ? st % ns : 1;
    }
    a.lean_run = 4;
    a.grow_at = 2;
    if (const char* ev = getenv("LDBG_VT_GROW_AT")) a.grow_at = (int)std::max<long long>(2, std::min<long long>(8, atoll(ev)));   // tuning knob
    if (const char* ev = getenv("LDBG_LEAN_RUN")) a.lean_run = (int)std::max<long long>(1, atoll(ev));   // tuning knob
    a.run_rev = cfg.direction == LDBG_DIR_BOTH || cfg.direction == LDBG_DIR_REVERSE;
    a.run_fwd = cfg.direction == LDBG_DIR_BOTH || cfg.direction == LDBG_DIR_FORWARD;
    a.next_strand = d_ctr;
    a.next_block = d_ctr + 1;
    a.pool = (uint64_t*)d_pool_; a.n_blocks = n_blocks_;
    a.block_table = (uint32_t*)d_block_table_; a.max_blocks = max_blocks;
    a.strand_n = d_strand_n; a.strand_c = d_strand_c; a.status = d_status; a.iters = d_iters; a.quirk = d_quirk;
    a.term = (uint64_t*)out.d_term;
    a.vpool = (uint64_t*)d_vpool_; a.vnext = d_ctr + 2; a.vpool_entries = vpool_entries_; a.vcap_max = vcap_max;
    a.vcap_init = vt_initial_entries();
    if (!getenv("LDBG_VT_INITIAL")) {
        if (runs_ && !img) {
            // with the run index a strand's table holds the fringes of the stretches it crosses and the junction vertices between
            // them: a few entries per thousand k-mers.  Small tables = little to zero between batches (C3: 8.5 ms -> 0.3 ms)
            // (C3, one step: 128 entries 10.05 ms, 256 9.96, 512 9.71, 1024 9.79, 2048 10.08)
            a.vcap_init = std::min<uint32_t>(512u, vcap_max);
        } else {
            // Regrowing a table stalls the owner's whole wavefront (strand.h), and the regrowths of its 64 strands add up: start as large as
            // half of the pool allows when every strand of the batch takes one (C3: 65,536 entries, launch 291 -> 250 ms against 4,096)
            const uint64_t per = vpool_entries_ / 2 / (uint64_t)std::max<int64_t>(1, ns);
            while ((uint64_t)a.vcap_init * 4 <= per && (uint64_t)a.vcap_init * 4 <= vcap_max) a.vcap_init *= 4;
        }
    }
    a.ls = (LsElem*)d_ls_; a.ecap = ecap_;
    a.snap = getenv("LDBG_NO_REPEAT") ? nullptr : (LsSnap*)d_snap_;
    a.yield_iters = img ? 32u : 0u;           // (C3 over the image, 50,000 seeds: 16 -> 0.41, 32 -> 0.72, 64 -> 0.71, none -> 0.045 G k-mers/s)
    if (const char* ev = getenv("LDBG_IMG_YIELD")) a.yield_iters = img ? (uint32_t)std::max(0, atoi(ev)) : 0u;

    a.wg_times = nullptr; a.st_times = nullptr; a.st_gen = nullptr; a.wave_cat = nullptr;
#ifdef LDBG_WALK_DIAG
    const bool want_times = r.want_times = getenv("LDBG_WG_TIMES") != nullptr && !img;
#else
    const bool want_times = r.want_times = false;       // (the timers are compiled into the -DLDBG_WALK_DIAG build only: make diag)
#endif
    // one (partial) wavefront per workgroup; every workgroup must be resident (lanes refill from the strand queue):
    // LDBG_LS_FAST x block x 24 B of LDS each, at most 32 wavefronts per CU
    // (measured at C3, profiles/r01_exp_block.log: 64 lanes 0.51 s, 32 lanes 0.60 s, 16 lanes 0.63 s per launch — smaller
    // wavefronts finish the bulk sooner but the longest strands run slower with more wavefronts per CU)
    int& block = r.block;
    block = 64;
    if (const char* ev = getenv("LDBG_WALK_BLOCK")) block = atoi(ev) == 16 ? 16 : (atoi(ev) == 32 ? 32 : 64);   // tuning knob
    // residency: LDS per workgroup, and 152 VGPRs per lane leave 3 wavefronts per SIMD = 12 per CU
    int wg_per_cu = std::min<int>(12, (int)(160 * 1024 / (LDBG_LS_FAST * (size_t)block * sizeof(LsElem))));
    if (img) wg_per_cu = std::min(wg_per_cu, 4);        // the image variant of the kernel keeps one wavefront per SIMD (its state save / restore costs registers)
    if (const char* ev = getenv("LDBG_WG_PER_CU")) wg_per_cu = std::max(1, std::min(wg_per_cu, atoi(ev)));   // tuning knob
    a.n_slots = std::min<int64_t>(a.n_slots, (int64_t)wg_per_cu * rt::cu_count(graph->device) * block);
    a.n_slots = (a.n_slots / block) * block;
    if (a.n_slots < block) a.n_slots = block;
    const int grid = r.grid = (int)((a.n_slots + block - 1) / block);
    if (want_times) {
        a.wg_times = (unsigned long long*)rt::dmalloc((size_t)grid * 16); rt::dmemset(a.wg_times, 0, (size_t)grid * 16, s);
        a.st_times = (unsigned long long*)rt::dmalloc((size_t)ns * 16); rt::dmemset(a.st_times, 0, (size_t)ns * 16, s);
        a.st_gen = (unsigned long long*)rt::dmalloc((size_t)ns * 16); rt::dmemset(a.st_gen, 0, (size_t)ns * 16, s);
        a.wave_cat = (unsigned long long*)rt::dmalloc((size_t)grid * 128); rt::dmemset(a.wave_cat, 0, (size_t)grid * 128, s);
#ifdef LDBG_LEAN_PROFILE
        a.st_prof = (unsigned long long*)rt::dmalloc((size_t)ns * 32); rt::dmemset(a.st_prof, 0, (size_t)ns * 32, s);
#endif
    }
}

// one launch of the walk kernel: the whole batch for a resident table, one bulk-synchronous round on an image
void Engine::walk_launch(WalkRun& r) {
    rt::stream_t s = stream_;
    r.ev0.record(s);
    launch_k_walk(r, r.a, s);
    r.ev1.record(s);
    r.timed = true;
}

// returns false when the path pool was exhausted (nothing is kept; the caller splits the chunk)
bool Engine::walk_finish(WalkRun& r, int64_t* traversed) {
    WALKRUN_ALIASES(r);
    const bool want_times = r.want_times;
    const int grid = r.grid, block = r.block; (void)block;
    HostLaps laps;

    // Everything the host must know before it can go on is counted on the device and lands in ONE page-locked block (h_small_):
    // [0..31] the kernel's counters (d_ctr), [32] strand_off[2n], [33] contig_off[n].  The per-strand arrays stay in HBM.
    if (!h_small_) h_small_ = (unsigned long long*)rt::hmalloc_pinned(64 * 8);
    unsigned long long* const ctr = h_small_;
    // strands the run steps handed back (ST_RETRY_PLAIN) are walked again k-mer by k-mer, without the run index
    if (runs_ && !a.img_on) {
        d_retry = (uint32_t*)rt::tmalloc((size_t)ns * 4);
        LDBG_LAUNCH(k_retry_list, grid_for(ns, 256, 1024), 256, s, (const uint32_t*)d_status, ns, d_retry, d_ctr + 28);
        rt::d2h(ctr, d_ctr, 256, s);
        rt::stream_sync(s);
        if (r.timed) { r.walk_ms += rt::Event::elapsed_ms(r.ev0, r.ev1); r.timed = false; }
        const int64_t n_again = (int64_t)ctr[28];
        if (n_again > 0) {
            rt::dmemset(d_ctr, 0, 8, s);                       // the strand queue starts over; the pool cursors carry on
            WalkArgs b = a;
            b.e.runs = RunIndexView{nullptr, nullptr, nullptr};
            b.retry = d_retry;
            b.n_strands = n_again;
            retried_strands_ += n_again;
            launch_k_walk(r, b, s);
        }
    }

    // lengths + seed test + what the batch as a whole has to report; offsets of the strands' vertex lists and of the contigs
    int64_t* d_walk_len = (int64_t*)rt::tmalloc((size_t)n * 8);
    uint8_t* d_seed_ok = (uint8_t*)rt::tmalloc((size_t)n);
    uint32_t* d_contig_len = (uint32_t*)rt::tmalloc((size_t)n * 4);
    int64_t* d_strand_off = (int64_t*)rt::tmalloc((size_t)(ns + 1) * 8);
    int64_t* d_contig_off = (int64_t*)rt::tmalloc((size_t)(n + 1) * 8);
    unsigned long long* d_scan = (unsigned long long*)rt::tmalloc((size_t)(2 * OFF_SCAN_OWNERS + 2) * 8);
    auto free_results = [&] {
        rt::tfree(d_walk_len); rt::tfree(d_seed_ok); rt::tfree(d_contig_len); rt::tfree(d_strand_off); rt::tfree(d_contig_off); rt::tfree(d_scan);
        d_walk_len = nullptr; d_seed_ok = nullptr; d_contig_len = nullptr; d_strand_off = nullptr; d_contig_off = nullptr; d_scan = nullptr;
    };
    AsmArgs aa;
    aa.e = view; aa.n = n; aa.op_and = cfg.combination_operator == LDBG_OP_AND; aa.k = k;
    aa.seeds = a.seeds; aa.pool = a.pool; aa.block_table = a.block_table; aa.max_blocks = max_blocks;
    aa.strand_n = d_strand_n; aa.status = d_status; aa.iters = d_iters; aa.quirk = d_quirk;
    aa.walk_len = d_walk_len; aa.seed_ok = d_seed_ok; aa.contig_len = d_contig_len; aa.flags = d_ctr + 24;
    LDBG_LAUNCH(k_walk_lengths, grid_for(n, 256, 2048), 256, s, aa);
    int64_t* d_totals = (int64_t*)(d_scan + 2 * OFF_SCAN_OWNERS);
    launch_offsets(d_strand_n, ns, d_strand_off, d_scan, d_totals, s);
    launch_offsets(d_contig_len, n, d_contig_off, d_scan + OFF_SCAN_OWNERS, d_totals + 1, s);
    rt::d2h(ctr, d_ctr, 256, s);
    rt::d2h(ctr + 32, d_totals, 16, s);
    rt::stream_sync(s);
    if (r.timed) { r.walk_ms += rt::Event::elapsed_ms(r.ev0, r.ev1); r.timed = false; }
    {
        static const char* const kind_names[8] = {"walk_steps_run", "walk_run_vertices", "walk_steps_lean", "walk_steps_general", "walk_link_adds", "walk_choices",
                                                  "walk_wave_iterations", "walk_wave_general"};
        for (int q = 0; q < 8; q++) profile_add(kind_names[q], (double)ctr[8 + q]);
        profile_add("walk_busiest_general", (double)(ctr[16] >> 32));
        profile_add("walk_busiest_iterations", (double)(ctr[16] & 0xFFFFFFFFull));
        profile_add("walk_wavefronts", (double)grid);
    }
    const bool pool_full = ctr[24] != 0, any_error = ctr[25] != 0, any_quirk = ctr[26] != 0;
    std::vector<uint32_t> iters;
    if (want_times || any_error) {                 // (diagnostics, and the error report below: these want the per-strand arrays)
        out.status.resize(ns);
        rt::d2h(out.status.data(), d_status, (size_t)ns * 4, s);
        if (want_times) { iters.resize(ns); rt::d2h(iters.data(), d_iters, (size_t)ns * 4, s); }
        rt::stream_sync(s);
    }
    if (want_times) {
        std::vector<unsigned long long> t((size_t)grid * 2);
        rt::d2h(t.data(), a.wg_times, (size_t)grid * 16, s);
        rt::stream_sync(s);
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < grid; i++) t0 = std::min(t0, t[2 * i]);
        std::vector<double> st(grid), en(grid);
        for (int i = 0; i < grid; i++) { st[i] = (t[2 * i] - t0) / 1e5; en[i] = (t[2 * i + 1] - t0) / 1e5; }   // ms
        std::vector<double> ss = st, ee = en;
        std::sort(ss.begin(), ss.end()); std::sort(ee.begin(), ee.end());
        fprintf(stderr, "[ldbg] k_walk workgroups=%d start ms p0/p50/p90/p100 = %.2f %.2f %.2f %.2f ; end ms p0/p50/p90/p100 = %.2f %.2f %.2f %.2f\n",
                grid, ss[0], ss[grid / 2], ss[grid * 9 / 10], ss[grid - 1], ee[0], ee[grid / 2], ee[grid * 9 / 10], ee[grid - 1]);
        rt::dfree(a.wg_times);
        {
            std::vector<unsigned long long> wc((size_t)grid * 16);
            rt::d2h(wc.data(), a.wave_cat, (size_t)grid * 128, s);
            rt::stream_sync(s);
            rt::dfree(a.wave_cat);
            unsigned long long sum[16] = {0};
            int slowest = 0;
            for (int i = 0; i < grid; i++) { for (int q = 0; q < 16; q++) sum[q] += wc[16 * i + q]; if (en[i] > en[slowest]) slowest = i; }
            auto line = [&](const char* who, const unsigned long long* w, double div) {
                fprintf(stderr, "[ldbg] %s: %.0f loop iterations; table regrowth %.2f ms, run steps %.2f ms, lean runs %.2f ms; %.0f with a general part, %.2f ms (%.1f us each)\n", who,
                        w[0] / div, w[1] / 1e5 / div, w[2] / 1e5 / div, w[3] / 1e5 / div, w[4] / div, w[5] / 1e5 / div, w[4] ? w[5] / 100.0 / w[4] : 0.0);
                const double g = w[4] ? (double)w[4] : 1.0;
                fprintf(stderr, "[ldbg] %s: general part per iteration: prefetch %.2f us, adds %.2f us, choices %.2f us, step %.2f us; lanes %.1f, add owners %.2f, choice owners %.2f\n", who,
                        w[6] / 100.0 / g, w[7] / 100.0 / g, w[8] / 100.0 / g, w[9] / 100.0 / g, w[10] / g, w[11] / g, w[12] / g);
                fprintf(stderr, "[ldbg] %s: add owners in 16-lane groups %.2f per iteration, of which a store of <= 8 elements %.2f; elements per add owner (store + records) %.1f\n", who,
                        w[13] / g, w[14] / g, w[11] ? (double)w[15] / (double)w[11] : 0.0);
            };
            line("average wavefront", sum, (double)grid);
            line("slowest wavefront", &wc[16 * (size_t)slowest], 1.0);
        }
        std::vector<unsigned long long> tt((size_t)ns * 2);
        rt::d2h(tt.data(), a.st_times, (size_t)ns * 16, s);
        rt::stream_sync(s);
        std::vector<int64_t> order(ns);
        for (int64_t i = 0; i < ns; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return tt[2 * x + 1] - tt[2 * x] > tt[2 * y + 1] - tt[2 * y]; });
        std::vector<unsigned long long> gen((size_t)ns * 2);
        rt::d2h(gen.data(), a.st_gen, (size_t)ns * 16, s);
        rt::stream_sync(s);
        rt::dfree(a.st_gen);
#ifdef LDBG_LEAN_PROFILE
        {
            std::vector<unsigned long long> pr((size_t)ns * 4);
            rt::d2h(pr.data(), a.st_prof, (size_t)ns * 32, s);
            rt::stream_sync(s);
            rt::dfree(a.st_prof);
            for (int r = 0; r < 6 && r < ns; r++) {
                int64_t i = order[r];
                double n = (double)std::max<unsigned long long>(1, pr[4 * i + 3]);
                fprintf(stderr, "[ldbg] strand %lld lean steps %.0f: cycles per step issue %.0f, wait %.0f, rest %.0f\n", (long long)i, n, pr[4 * i] / n, pr[4 * i + 1] / n, pr[4 * i + 2] / n);
            }
        }
#endif
        for (int r = 0; r < 3 && r < ns; r++) {
            int64_t i = order[r];
            fprintf(stderr, "[ldbg] strand %lld: %llu of its %u steps went through the general step, %.1f ms there (%.2f us each, the step itself not included)\n",
                    (long long)i, gen[2 * i + 1], iters[i], gen[2 * i] / 1e5, gen[2 * i + 1] ? gen[2 * i] / 100.0 / gen[2 * i + 1] : 0.0);
        }
        for (int r = 0; r < 12 && r < ns; r++) {
            int64_t i = order[r];
            double ms = (tt[2 * i + 1] - tt[2 * i]) / 1e5;
            fprintf(stderr, "[ldbg] slow strand %lld (seed %lld %s): %.1f ms, %u iterations, %.2f us/iter, lane %lld of its wave, status %u\n", (long long)i, (long long)(first + i / 2),
                    (i & 1) ? "fwd" : "rev", ms, iters[i], iters[i] ? ms * 1e3 / iters[i] : 0.0, (long long)(i & 63), out.status[i]);
        }
        {   // the walks that ran to maxLength: how evenly do they progress?
            std::vector<double> us;
            uint32_t max_it = 0;
            for (int64_t i = 0; i < ns; i++) max_it = std::max(max_it, iters[i]);
            for (int64_t i = 0; i < ns; i++) if (iters[i] == max_it && max_it > 0) us.push_back((tt[2 * i + 1] - tt[2 * i]) / 100.0 / max_it);
            std::sort(us.begin(), us.end());
            if (!us.empty())
                fprintf(stderr, "[ldbg] %zu strands of %u iterations: us/iter p0/p10/p50/p90/p100 = %.2f %.2f %.2f %.2f %.2f\n", us.size(), max_it,
                        us[0], us[us.size() / 10], us[us.size() / 2], us[us.size() * 9 / 10], us.back());
        }
        double tot_ms = 0; for (int64_t i = 0; i < ns; i++) tot_ms += (tt[2 * i + 1] - tt[2 * i]) / 1e5;
        fprintf(stderr, "[ldbg] sum of strand durations %.1f s over %lld strands; %llu wavefront loop iterations in %d wavefronts\n", tot_ms / 1e3, (long long)ns,
                ctr[14], grid);
        rt::dfree(a.st_times);
    }
    vpool_dirty_ = ctr[2];
    profile_add("walk", r.walk_ms);
    laps.lap("launch .. results on host");

    if (pool_full) {
        free_tmp(); free_results();
        rt::tfree(out.d_seed_words); rt::tfree(out.d_term);
        out.d_seed_words = out.d_term = nullptr;
        return false;
    }
    if (any_error) {        // errors the reference raises as exceptions abort the call
        for (int64_t i = 0; i < ns; i++) {
            const uint32_t st = out.status[i];
            if (st != ST_NULLPTR && st != ST_LINKSTORE_FULL && st != ST_COPY_OVERFLOW) continue;
            free_tmp(); free_results();
            rt::tfree(out.d_seed_words); rt::tfree(out.d_term);
            out.d_seed_words = out.d_term = nullptr;
            if (st == ST_NULLPTR)
                throw StatusError(LDBG_ERR_NULLPOINTER, "getNextVertices: record missing while recruitment colours are set (seed " + std::to_string(first + i / 2) + ")");
            if (st == ST_LINKSTORE_FULL) throw StatusError(LDBG_ERR_CAPACITY, "LINKSTORE_FULL");
            throw StatusError(LDBG_ERR_UNSUPPORTED, "a vertex was visited more than 32767 times in one walk");
        }
    }
    *traversed += (int64_t)ctr[27];
    out.total_entries = (int64_t)ctr[32];
    out.total_bytes = (int64_t)ctr[33];
    out.host_ready = false;

    // contigs; the dense vertex entries now (a walk through a quirk-Q6 vertex is spelled k-mer by k-mer from them; a batch that was split
    // reuses the path pool) or when somebody asks for vertex lists (ensure_dense)
    const bool lazy = !any_quirk && first == 0 && n == batch_n && !getenv("LDBG_EAGER_PATHS");
    out.d_contigs = result_alloc((size_t)out.total_bytes, &out.contigs_cap);
    out.d_strand_off = d_strand_off; d_strand_off = nullptr;     // the chunk owns them now (clear_batch frees)
    out.d_contig_off = d_contig_off;
    out.d_walk_len = d_walk_len;
    out.d_strand_c = d_strand_c; d_strand_c = nullptr;
    out.max_blocks = max_blocks;
    out.runs = a.e.runs;
    out.dense_pending = true;
    laps.lap("result buffers");
    rt::Event c0, c1;
    c0.record(s);
    if (lazy) {
        ContigRleArgs ra;
        ra.g = graph->view; ra.runs = a.e.runs; ra.n = n; ra.seeds = a.seeds;
        ra.pool = (const uint64_t*)d_pool_; ra.block_table = (const uint32_t*)d_block_table_; ra.max_blocks = max_blocks;
        ra.strand_c = (const uint32_t*)out.d_strand_c; ra.strand_n = d_strand_n; ra.walk_len = d_walk_len; ra.contig_off = d_contig_off;
        ra.out = (char*)out.d_contigs;
        int rb = LDBG_STREAM_BLOCK, rcap = LDBG_STREAM_GRID;
#ifndef LDBG_HOSTSIM
        if (const char* ev = getenv("LDBG_RLE_BLOCK")) rb = atoi(ev) >= 256 ? 256 : (atoi(ev) >= 128 ? 128 : 64);      // tuning knobs
        if (const char* ev = getenv("LDBG_RLE_GRID")) rcap = std::max(1, atoi(ev));
#endif
        const int rg = grid_for(n * 64, rb, rcap);
        switch (W) {
            case 1: LDBG_LAUNCH(k_contigs_rle<1>, rg, rb, s, ra); break;
            case 2: LDBG_LAUNCH(k_contigs_rle<2>, rg, rb, s, ra); break;
            case 3: LDBG_LAUNCH(k_contigs_rle<3>, rg, rb, s, ra); break;
            default: LDBG_LAUNCH(k_contigs_rle<4>, rg, rb, s, ra); break;
        }
    } else {
        ensure_dense(out);
        const int cg = grid_for(n * 64, 256, 4096);
        ContigArgs ca;
        ca.g = graph->view; ca.n = n; ca.seeds = a.seeds; ca.dense = (const uint64_t*)out.d_path; ca.strand_off = (const int64_t*)out.d_strand_off;
        ca.walk_len = d_walk_len; ca.contig_off = d_contig_off; ca.quirk = d_quirk; ca.term = (const uint64_t*)out.d_term;
        ca.out = (char*)out.d_contigs;
        switch (W) {
            case 1: LDBG_LAUNCH(k_contigs<1>, cg, 256, s, ca); break;
            case 2: LDBG_LAUNCH(k_contigs<2>, cg, 256, s, ca); break;
            case 3: LDBG_LAUNCH(k_contigs<3>, cg, 256, s, ca); break;
            default: LDBG_LAUNCH(k_contigs<4>, cg, 256, s, ca); break;
        }
    }
    c1.record(s);
    rt::stream_sync(s);
    profile_add("contig", rt::Event::elapsed_ms(c0, c1));
    laps.lap("paths + contigs");
    free_tmp();
    d_walk_len = nullptr; d_contig_off = nullptr;                // (the chunk's)
    free_results();
    laps.lap("frees");
    return true;
}


// ---- walk_batch_run over the image of a hash-sharded table (image.h): the same preparation and result assembly, the kernel once
// per bulk-synchronous round
void Engine::sharded_walk_begin(ShardImage& img, const char* seeds, int64_t n, const int32_t* d_seed_slot, rt::stream_t round_stream) {
    if (cfg.stopping_rule != LDBG_STOP_CONTIG || cfg.connect_all_neighbors || cfg.n_secondary > 0)
        throw StatusError(LDBG_ERR_UNSUPPORTED, "walks over a sharded table run ContigStopper without connectAllNeighbors / secondary colours");
    if (&img.graph() != graph) throw StatusError(LDBG_ERR_ARG, "the engine was not created on this image's graph");
    enter();
    sharded_abort();
    clear_batch();
    seeds_to_device(seeds, n, false);
    batch_n = n;
    sharded_run_ = new WalkRun;
    sharded_img_ = &img;
    sharded_stream_ = round_stream ? round_stream : stream_;
    try {
        walk_prepare(0, n, *sharded_run_, &img, d_seed_slot);
        rt::stream_sync(stream_);
    } catch (...) { sharded_abort(); throw; }
}
void Engine::sharded_abort() {
    if (sharded_run_) {
        quiesce();
        sharded_run_->free_tmp();
        rt::tfree(sharded_run_->out.d_seed_words); rt::tfree(sharded_run_->out.d_term);
        delete sharded_run_;
        sharded_run_ = nullptr;
    }
    sharded_img_ = nullptr;
}
// d_stats (device, 3 x int64): strands of this rank still in progress after the round (suspended or not yet handed out), requests filed,
// the image's overflow flag (a full image: the caller stops the rounds on every rank, enlarges the image and runs the batch again)
void Engine::sharded_walk_round(int64_t* d_stats) {
    if (!sharded_run_) throw StatusError(LDBG_ERR_ARG, "sharded_walk_round without sharded_walk_begin");
    enter();
    WalkRun& r = *sharded_run_;
    rt::stream_t s = sharded_stream_;
    rt::dmemset(r.d_ctr + 4, 0, 8, s);
    sharded_img_->reset_requests(s);
    launch_k_walk(r, r.a, s);
    LDBG_LAUNCH(k_round_stats, 1, 64, s, (const unsigned long long*)r.d_ctr, r.ns, (const unsigned long long*)r.a.img.n_req, d_stats);
    sharded_rounds_++;
}
void Engine::sharded_walk_finish(int64_t* total_bytes, int64_t* traversed) {
    if (!sharded_run_) throw StatusError(LDBG_ERR_ARG, "sharded_walk_finish without sharded_walk_begin");
    enter();
    WalkRun& r = *sharded_run_;
    rt::stream_sync(sharded_stream_);
    profile_add("walk_rounds", (double)sharded_rounds_);
    sharded_rounds_ = 0;
    int64_t trav = 0;
    bool ok = false;
    try { ok = walk_finish(r, &trav); } catch (...) { sharded_abort(); throw; }
    if (!ok) { sharded_abort(); throw StatusError(LDBG_ERR_CAPACITY, "path pool exhausted during a walk over a sharded table: use smaller batches"); }
    chunks.push_back(std::move(r.out));
    r.out = WalkChunk();
    sharded_abort();
    batch_traversed = trav;
    batch_bytes = chunks.back().total_bytes;
    if (total_bytes) *total_bytes = batch_bytes;
    if (traversed) *traversed = trav;
}

bool Engine::run_chunk(int64_t first, int64_t n, WalkChunk& out, int64_t* traversed) {
    WalkRun r;
    try {
        walk_prepare(first, n, r, nullptr, nullptr);
        walk_launch(r);
        const bool ok = walk_finish(r, traversed);
        if (ok) out = std::move(r.out);
        r.free_tmp();
        return ok;
    } catch (...) { quiesce(); r.free_tmp(); throw; }
}

// Device -> caller's host buffer.  Into page-locked memory (ldbg_host_alloc) the copy runs at the bus rate as it is.  Into pageable memory
// the runtime's own copy manages about 10 GB/s (780 MB of contigs: 76 ms of an 86 ms step), so it goes through two page-locked staging
// buffers instead: chunk i + 1 crosses the bus while a few host threads move chunk i to its place.
#define LDBG_STAGE_BYTES ((size_t)32 << 20)
void Engine::download(char* dst, const void* d_src, size_t bytes) {
    rt::stream_t s = stream_;
    if (bytes == 0) return;
    if (bytes < ((size_t)4 << 20) || rt::host_is_pinned(dst)) { rt::d2h(dst, d_src, bytes, s); rt::stream_sync(s); return; }
    for (int b = 0; b < 2; b++) if (!h_stage_[b]) h_stage_[b] = rt::hmalloc_pinned(LDBG_STAGE_BYTES);
    const size_t nchunks = (bytes + LDBG_STAGE_BYTES - 1) / LDBG_STAGE_BYTES;
    rt::Event landed[2];
    auto scatter = [&](size_t c) {             // staging buffer of chunk c -> its place, on 4 threads
        const size_t off = c * LDBG_STAGE_BYTES, len = std::min(LDBG_STAGE_BYTES, bytes - off);
        const char* src = (const char*)h_stage_[c & 1];
        const int T = 4;
        std::thread th[T - 1];
        auto part = [&](int t) { const size_t lo = len * (size_t)t / T, hi = len * (size_t)(t + 1) / T; memcpy(dst + off + lo, src + lo, hi - lo); };
        for (int t = 1; t < T; t++) th[t - 1] = std::thread(part, t);
        part(0);
        for (int t = 1; t < T; t++) th[t - 1].join();
    };
    for (size_t c = 0; c < nchunks; c++) {
        const size_t off = c * LDBG_STAGE_BYTES, len = std::min(LDBG_STAGE_BYTES, bytes - off);
        rt::d2h(h_stage_[c & 1], (const char*)d_src + off, len, s);
        landed[c & 1].record(s);
        if (c > 0) { landed[(c - 1) & 1].wait(); scatter(c - 1); }       // (the buffer chunk c + 1 will land in is free again after this)
    }
    landed[(nchunks - 1) & 1].wait();
    scatter(nchunks - 1);
}

// offsets and lengths of a chunk on the host (they are computed and kept on the device, walk_finish)
void Engine::ensure_host(WalkChunk& c) {
    if (c.host_ready) return;
    rt::stream_t s = stream_;
    c.strand_off.resize((size_t)(2 * c.n + 1));
    c.contig_off.resize((size_t)(c.n + 1));
    c.walk_len.resize((size_t)c.n);
    rt::d2h(c.strand_off.data(), c.d_strand_off, (size_t)(2 * c.n + 1) * 8, s);
    rt::d2h(c.contig_off.data(), c.d_contig_off, (size_t)(c.n + 1) * 8, s);
    rt::d2h(c.walk_len.data(), c.d_walk_len, (size_t)c.n * 8, s);
    rt::stream_sync(s);
    c.host_ready = true;
}

void Engine::walk_batch_fetch(char* arena, int64_t cap, int64_t* offsets, int64_t* walk_len) {
    enter();
    if (offsets || walk_len) for (auto& c : chunks) ensure_host(c);
    if (offsets) {
        int64_t o = 0;
        offsets[0] = 0;
        for (auto& c : chunks)
            for (int64_t i = 0; i < c.n; i++) { o += c.contig_off[i + 1] - c.contig_off[i]; offsets[c.first + i + 1] = o; }
    }
    if (walk_len) for (auto& c : chunks) for (int64_t i = 0; i < c.n; i++) walk_len[c.first + i] = c.walk_len[i];
    if (cap < batch_bytes) throw StatusError(LDBG_ERR_CAPACITY, "contig arena too small: need " + std::to_string(batch_bytes) + " bytes");
    if (arena) {
        int64_t o = 0;
        for (auto& c : chunks) {
            download(arena + o, c.d_contigs, (size_t)c.total_bytes);
            o += c.total_bytes;
        }
    }
}

void Engine::walk_vertices(int64_t walk, int64_t capacity, int64_t* len, uint64_t* words, int64_t* rec, int32_t* copy, int32_t* index) {
    enter();
    if (walk < 0 || walk >= batch_n) throw StatusError(LDBG_ERR_ARG, "walk index out of range");
    const int W = graph->hdr.W;
    for (auto& c : chunks) {
        if (walk < c.first || walk >= c.first + c.n) continue;
        int64_t i = walk - c.first;
        ensure_host(c);
        int64_t L = c.walk_len[i];
        *len = L;
        if (L == 0) return;
        if (capacity < L) throw StatusError(LDBG_ERR_CAPACITY, "vertex buffers too small: need " + std::to_string(L));
        ensure_dense(c);
        rt::stream_t s = stream_;
        uint64_t* d_words = (uint64_t*)rt::dmalloc((size_t)L * W * 8);
        int64_t* d_rec = (int64_t*)rt::dmalloc((size_t)L * 8);
        int32_t* d_copy = (int32_t*)rt::dmalloc((size_t)L * 4);
        int32_t* d_index = (int32_t*)rt::dmalloc((size_t)L * 4);
        int64_t ro = c.strand_off[2 * i], nr = c.strand_off[2 * i + 1] - ro, fo = c.strand_off[2 * i + 1], nf = c.strand_off[2 * i + 2] - fo;
        const uint64_t* tr = (const uint64_t*)c.d_term + (2 * i) * W;
        const uint64_t* tf = (const uint64_t*)c.d_term + (2 * i + 1) * W;
        const int g = grid_for(L, 256, 1024);
        switch (W) {
            case 1: LDBG_LAUNCH(k_walk_vertices<1>, g, 256, s, graph->view, (const uint64_t*)c.d_path, ro, nr, fo, nf, tr, tf, L, d_words, d_rec, d_copy, d_index); break;
            case 2: LDBG_LAUNCH(k_walk_vertices<2>, g, 256, s, graph->view, (const uint64_t*)c.d_path, ro, nr, fo, nf, tr, tf, L, d_words, d_rec, d_copy, d_index); break;
            case 3: LDBG_LAUNCH(k_walk_vertices<3>, g, 256, s, graph->view, (const uint64_t*)c.d_path, ro, nr, fo, nf, tr, tf, L, d_words, d_rec, d_copy, d_index); break;
            default: LDBG_LAUNCH(k_walk_vertices<4>, g, 256, s, graph->view, (const uint64_t*)c.d_path, ro, nr, fo, nf, tr, tf, L, d_words, d_rec, d_copy, d_index); break;
        }
        if (words) rt::d2h(words, d_words, (size_t)L * W * 8, s);
        if (rec) rt::d2h(rec, d_rec, (size_t)L * 8, s);
        if (copy) rt::d2h(copy, d_copy, (size_t)L * 4, s);
        if (index) rt::d2h(index, d_index, (size_t)L * 4, s);
        rt::stream_sync(s);
        rt::dfree(d_words); rt::dfree(d_rec); rt::dfree(d_copy); rt::dfree(d_index);
        return;
    }
}

}  // namespace ldbg
