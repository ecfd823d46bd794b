// The 19 traversal stopping rules (J/utils/stoppingrules/*.java) as one device function pair over a
// 12-byte per-branch state.  A rule instance lives for one dfs branch (TraversalEngine.java:366): the kernel
// zeroes the state when a branch opens and keeps the parent's copy in its stack frame.
//
// Every predicate is restated from the Java, including which of them dereference the current record or the
// ROI graph (NullPointerException in the reference -> ST_NULLPTR here) and the fact that keepGoing() always
// evaluates both predicates (AbstractTraversalStoppingRule.java:9-15).
#pragma once
#include "engine.h"

namespace ldbg {

struct StopState {
    uint32_t flags;   // bit 0: foundNovels / startedNovel / haveSeenNovelKmers ; bit 1: hasJoined
    int32_t a;        // distanceFromLastNovel / numSeen / novelKmersSeen / sinceLastLow
    int32_t b;        // distanceSinceJoin
};

// TraversalState (J/utils/traversal/TraversalState.java:9-49), the fields the rules read
struct TravState {
    int graph_size, depth, branch_size, adj;
    bool children_traversed, reached_max;
};

// what the rules can see beside the state: the ROI graph, the sinks of this dfs call
struct StopEnv {
    GraphView rois;            // rois.N < 0: no ROI graph configured (getRois() == null)
    const uint32_t* roi_bits;  // bit i: record i of the traversed graph is a record of the ROI graph
    const uint64_t* sink_keys; // per sink: vt_key(record, flip), 0 = sink k-mer has no record
    const uint64_t* sink_words;// per sink: W packed words (for cursors that have no record either)
};

LDBG_HOSTDEV int destination_junction_limit(int graph_size) {
    // 1 + ceil(5 * exp(-1e-4 * size)), DestinationStopper.java:15-17, as an exact integer table
    if (graph_size <= 2231) return 6;
    if (graph_size <= 5108) return 5;
    if (graph_size <= 9162) return 4;
    if (graph_size <= 16094) return 3;
    if (graph_size <= 7451332) return 2;
    return 1;
}

template <int W>
struct StopEval {
    const EngineView& e;
    const StopEnv& env;
    const int64_t sink_lo, sink_hi;   // this seed's sinks
    const Node& cv;
    const Kmer<W>& nullk;     // the cursor k-mer when cv has no record
    uint32_t status = ST_OK;

    LDBG_HOSTDEV StopEval(const EngineView& e_, const StopEnv& env_, int64_t lo, int64_t hi, const Node& cv_, const Kmer<W>& nk)
        : e(e_), env(env_), sink_lo(lo), sink_hi(hi), cv(cv_), nullk(nk) {}

    LDBG_HOSTDEV bool need_rec() { if (cv.idx < 0) { status = ST_NULLPTR; return false; } return true; }
    LDBG_HOSTDEV bool need_rois() { if (env.rois.N < 0) { status = ST_NULLPTR; return false; } return true; }
    // any joining colour with coverage > 0 (signed, Q5); no record access when there are no joining colours
    LDBG_HOSTDEV bool joined() {
        if (e.join_mask == 0) return false;
        if (!need_rec()) return false;
        bool r = false;
        for (int c = 0; c < e.g.C; c++) if ((e.join_mask >> c) & 1u) r |= (int32_t)graph_cov(e.g, cv.idx, c) > 0;
        return r;
    }
    LDBG_HOSTDEV bool child_cov() {
        if (!need_rec()) return false;
        bool r = false;
        for (int c = 0; c < e.g.C; c++) if ((e.trav_mask >> c) & 1u) r |= (int32_t)graph_cov(e.g, cv.idx, c) > 0;
        return r;
    }
    // CortexRecord.getInDegree / getOutDegree of the stored (canonical) record, per traversal colour
    LDBG_HOSTDEV void degrees(bool& no_in, bool& no_out, bool& busy) {
        no_in = no_out = busy = false;
        if (!need_rec()) return;
        for (int c = 0; c < e.g.C; c++) {
            if (!((e.trav_mask >> c) & 1u)) continue;
            const uint32_t eb = graph_edges(e.g, cv.idx, c);
            const int in = popc4(eb >> 4), out = popc4(eb & 0xf);
            no_in |= in == 0; no_out |= out == 0; busy |= in + out > 4;
        }
    }
    // is the record of the current vertex a record of the ROI graph?  One bit per record of a resident table (k_roi_bits); over the
    // image of a sharded table (roi_bits == nullptr: rows come and go) the ROI graph — small, held by every rank — is asked directly
    LDBG_HOSTDEV bool roi_bit() {
        if (env.roi_bits) return (env.roi_bits[cv.idx >> 5] >> (cv.idx & 31)) & 1u;
        GraphView exact = env.rois;
        exact.java_tiny = 0;           // set membership, not findRecord: no Q1 here
        return graph_find_canonical<W>(exact, graph_key<W>(e.g, cv.idx)) >= 0;
    }
    // rois.findRecord(<cursor k-mer>) != null, with the ROI graph's own Q1 behaviour
    LDBG_HOSTDEV bool rois_find_cur() {
        if (!need_rois()) return false;
        if (env.rois.java_tiny) return false;
        if (cv.idx >= 0) return roi_bit();
        bool fc;
        Kmer<W> c = kmer_canonical<W>(nullk, e.g.k, &fc);
        return graph_find_canonical<W>(env.rois, c) >= 0;
    }
    // HashSet of every ROI record's k-mer contains the cursor record's k-mer (NovelPartitionStopper.java:24-37)
    LDBG_HOSTDEV bool roi_set_contains() {
        if (env.rois.N < 0) { status = ST_STOPPER_CONFIG; return false; }
        if (cv.idx < 0) return false;
        return roi_bit();
    }
    LDBG_HOSTDEV bool at_sink() {                                   // sinks.contains(cv.getKmerAsString())
        for (int64_t i = sink_lo; i < sink_hi; i++) {
            const uint64_t key = env.sink_keys[i];
            if (cv.idx >= 0) { if (key == vt_key(cv.idx, cv.flip != 0)) return true; }
            else if (key == 0) {
                bool eq = true;
                for (int w = 0; w < W; w++) eq &= env.sink_words[i * W + w] == nullk.w[w];
                if (eq) return true;
            }
        }
        return false;
    }
    LDBG_HOSTDEV bool at_canonical_sink() {                         // PairedReadClosingStopper.java:17-31
        if (cv.idx < 0) return false;
        for (int64_t i = sink_lo; i < sink_hi; i++)
            if ((env.sink_keys[i] >> 1) == (uint64_t)(cv.idx + 1)) return true;
        return false;
    }
    static LDBG_HOSTDEV bool novel_stop_now(const StopState& S, const TravState& s) {
        return S.a > 2000 || s.depth > 0 || s.reached_max || s.adj == 0 || (s.adj > 1 && s.children_traversed);
    }

    LDBG_HOSTDEV bool has_succeeded(StopState& S, const TravState& s) {
        switch (e.stopper) {
            case LDBG_STOP_CONTIG: return s.adj != 1 || s.reached_max;
            case LDBG_STOP_CYCLE_COLLAPSING_CONTIG: return s.adj == 0;
            case LDBG_STOP_DESTINATION: return at_sink();
            case LDBG_STOP_EXPLORATION: return s.reached_max || s.adj == 0 || s.depth >= 3;
            case LDBG_STOP_NOVEL_PARTITION: {
                S.a++;
                if (roi_set_contains()) { S.flags |= 1u; S.a = 0; }
                return (S.flags & 1u) && novel_stop_now(S, s);
            }
            case LDBG_STOP_NOVEL_KMER_LIMITED_CONTIG: {
                S.a++;
                if (roi_set_contains()) { S.flags |= 1u; S.a = 0; }
                return (S.flags & 1u) && (S.a > 2000 || s.adj != 1 || s.reached_max);
            }
            case LDBG_STOP_NOVEL_CONTINUATION: {
                if (s.depth > 0 && S.a <= 2 * e.g.k && rois_find_cur()) S.flags |= 1u;
                S.a++;
                return (s.children_traversed && s.adj != 1) || s.reached_max;
            }
            case LDBG_STOP_BUBBLE_CLOSING: return false;
            case LDBG_STOP_BUBBLE_OPENING: {
                if (rois_find_cur()) S.a++;
                if (status != ST_OK) return false;
                if (S.flags & 2u) S.b++;
                if (joined()) S.flags |= 2u;
                return S.a > 0 && (S.flags & 2u) && (S.b >= 30 || s.adj != 1);
            }
            case LDBG_STOP_CONTAMINANT: {
                const bool parents = joined();
                return cv.idx >= 0 && (parents || s.adj == 0);
            }
            case LDBG_STOP_DUST: {
                bool ni, no, busy; degrees(ni, no, busy);
                const bool reunion = joined();
                return ni || no || reunion;
            }
            case LDBG_STOP_GAP_CLOSING: return false;
            case LDBG_STOP_NAHR: {
                if (S.flags & 1u) S.a++;
                if (!need_rois() || !need_rec()) return false;
                if (!env.rois.java_tiny && roi_bit()) { S.flags |= 1u; S.a++; }
                return (S.flags & 1u) && (S.a >= 1000 || s.depth >= 5 || s.adj == 0 || s.children_traversed);
            }
            case LDBG_STOP_NOVEL_KMER_AGGREGATION: {
                const bool child = child_cov();
                if (status != ST_OK) return false;
                const bool parents = joined();
                if (child && !parents) S.flags |= 1u;
                return (S.flags & 1u) && parents;
            }
            case LDBG_STOP_ORPHAN: case LDBG_STOP_TIP_END: { bool ni, no, busy; degrees(ni, no, busy); return ni || no; }
            case LDBG_STOP_PAIRED_READ_CLOSING: return at_canonical_sink();
            case LDBG_STOP_TIP_BEGINNING: return joined();
            case LDBG_STOP_VISUALIZATION: return s.adj == 0 || s.depth > 2 || s.branch_size > 500;
        }
        return false;
    }

    LDBG_HOSTDEV bool has_failed(StopState& S, const TravState& s) {
        switch (e.stopper) {
            case LDBG_STOP_CONTIG: case LDBG_STOP_CYCLE_COLLAPSING_CONTIG: case LDBG_STOP_EXPLORATION:
            case LDBG_STOP_NOVEL_KMER_LIMITED_CONTIG: case LDBG_STOP_VISUALIZATION:
                return false;
            case LDBG_STOP_DESTINATION: return s.depth > destination_junction_limit(s.graph_size) || s.reached_max;
            case LDBG_STOP_NOVEL_PARTITION: return !(S.flags & 1u) && novel_stop_now(S, s);
            case LDBG_STOP_NOVEL_CONTINUATION: return (s.depth > 0 && !(S.flags & 1u)) || s.depth > 3;
            case LDBG_STOP_BUBBLE_CLOSING: return s.branch_size > 10000 || s.depth >= 2 || s.adj == 0;
            case LDBG_STOP_BUBBLE_OPENING: return S.a == 0 && (s.depth >= 5 || s.adj == 0);
            case LDBG_STOP_CONTAMINANT: { const bool parents = joined(); return cv.idx >= 0 && parents; }
            case LDBG_STOP_DUST: {
                bool ni, no, busy; degrees(ni, no, busy);
                if (status != ST_OK) return false;
                if (busy) S.a = 0; else S.a++;
                return S.a >= e.g.k;
            }
            case LDBG_STOP_GAP_CLOSING: return s.depth > 5 || s.adj == 0;
            case LDBG_STOP_NAHR: return !(S.flags & 1u) && (s.branch_size >= 1000 || s.depth >= 2 || s.adj == 0);
            case LDBG_STOP_NOVEL_KMER_AGGREGATION: return !(S.flags & 1u) && (s.branch_size >= 100 || s.depth >= 3);
            case LDBG_STOP_ORPHAN: case LDBG_STOP_TIP_END: return joined();
            case LDBG_STOP_PAIRED_READ_CLOSING: return s.depth >= 5 || s.adj == 0 || s.reached_max;
            case LDBG_STOP_TIP_BEGINNING: { bool ni, no, busy; degrees(ni, no, busy); return ni || no; }
        }
        return true;
    }
};

}  // namespace ldbg
