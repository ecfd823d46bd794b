// General depth-first search with a stopping rule: TraversalEngine.dfs(source, sinks...)
// (J/utils/traversal/TraversalEngine.java:64-106) and its private recursive branch method (:356-482).
//
// Device side.  One strand = (seed, direction) per lane, exactly like the contig walks (walk.cpp), but a
// branch that reaches a junction recurses into its children.  The recursion is an explicit stack of 72-byte
// frames in HBM, touched only at junctions; the lane advances by micro-steps (one loop iteration of :373-481,
// one child launch, or one entry of a visited-set undo) so that the 64 lanes of a wavefront keep working on
// their own strands in lock step.
//
//  * visited (HashSet<CortexVertex>, copied into every child, :360): the strand's ONE visited table is shared by
//    the whole recursion; what a branch added is taken out again when the branch returns (siblings must not see
//    each other's vertices), by replaying the branch's own vertices from the log.
//  * seen (cleared by the seek() that opens every branch when links are configured, :363-365): an epoch number in
//    the table entries.
//  * stopping rule instance per branch (:366): 12 bytes of state (stoppers.h), the parent's copy is kept in its frame.
//  * output: an event log per strand — OPEN, the branch's vertices in order, the logs of the children that
//    returned a graph, CLOSE; a branch that returns null truncates the log back to its OPEN.  The log fully
//    determines the DirectedWeightedPseudograph the reference builds; the JGraphT container semantics
//    (vertex/edge insertion order, Graphs.addGraph merges, index relabelling, OR/AND combination, :75-99) are
//    replayed on the host from that log (assemble_* below).
#include <algorithm>
#include <map>
#include <set>
#include <tuple>
#include <chrono>
#include <functional>
#include <unordered_map>
#include <thread>
#include <unordered_set>

#include "lscoop.h"
#include "runstep.h"
#include "stoppers.h"
#include "shard.h"
#include "strand.h"

namespace ldbg {

// log entries that are not vertices carry bit 63 (strand.h: path descriptors); the markers of this log are kind LDBG_PD_MARK
#define DFS_MARK (1ull << 63)
#define DFS_OPEN (DFS_MARK | ((uint64_t)LDBG_PD_MARK << 60) | 1ull)
#define DFS_CLOSE (DFS_MARK | ((uint64_t)LDBG_PD_MARK << 60) | 2ull)
#define DFS_KMER (DFS_MARK | ((uint64_t)LDBG_PD_MARK << 60) | 3ull)   // the 2 W entries that follow are the packed k-mer of the vertex before (it has no record), 32 bits each
#define DFS_HALF(x) (DFS_MARK | ((uint64_t)LDBG_PD_HALF << 60) | (uint64_t)(uint32_t)(x))

enum : uint8_t { PH_ITER = 0, PH_CHILD = 1, PH_UNDO = 2 };

struct DfsFrame {
    Node cv;                              // the branch's last vertex (where it forked)
    uint32_t log_start, size, gV, nlin;
    StopState ss;
    uint8_t child[4];                     // child bases in the reference's iteration order
    uint8_t nchild, next, any, adj;
    uint32_t cut;                         // run-index position of the branch's first vertex (LDBG_RUN_NONE: it has none)
    uint32_t kids_start, pad;             // log position where the children's logs begin
};
static_assert(sizeof(DfsFrame) % 8 == 0, "frame layout");

struct DfsArgs {
    WalkArgs w;
    StopEnv env;
    const int64_t* sink_off;       // [n + 1] (nullptr: no sinks)
    DfsFrame* frames;              // [n_slots][max_depth]
    int max_depth;
    uint32_t iter_limit;           // loop iterations one seed and direction may take
    int n_trav;
    uint8_t trav_order[LDBG_MAX_COLORS];   // traversal colours in LinkedHashSet order
    void* lane_save;               // over a sharded table's image (image.h): [n_slots] DfsSave<W>, the search a lane keeps from round to round
};

template <int W>
struct DfsLane {
    StrandState st;
    Kmer<W> nullk;                 // k-mer of st.cv when it has no record
    StopState ss;
    int64_t sink_lo, sink_hi;
    uint32_t depth, size, nlin, log_start;
    uint32_t undo_pos, undo_left;
    uint32_t cut;                  // run-index position of this branch's first vertex: it is a piece of its own (runstep.h)
    uint8_t phase;
    bool result, last_prev;
};

template <int W>
struct DfsSave {
    DfsLane<W> L;
    uint32_t ls_n, ls_java_cap, ls_nkeys, ls_next_seq, ls_age, ls_n_new;
    uint8_t ls_overflow, active, begun, pad;
    LsElem fast[LDBG_LS_FAST];
};

// ---- Java iteration order of the neighbour vertices (TraversalEngine.getNextVertices/getPrevVertices :147-239):
// HashSet<CortexVertex> over HashMap<CortexByteKmer, ...> over per-colour HashSet<CortexByteKmer> over
// HashSet<Byte>.  All tables have 16 buckets (at most 4 entries), so the order is a lexicographic key:
//   (bucket of CortexVertex.hashCode, bucket of the k-mer's Arrays.hashCode, first colour that has the edge,
//    rank of the base in HashSet<Byte> order A,C,T,G)
LDBG_HOSTDEV uint32_t jbucket16(uint32_t h) { return (h ^ (h >> 16)) & 15u; }
LDBG_HOSTDEV uint32_t bswap64_fold(uint64_t w) {    // Long.hashCode of the byte-swapped word (CortexRecord keeps big-endian longs)
    uint64_t u = __builtin_bswap64(w);
    return (uint32_t)(u ^ (u >> 32));
}
template <int W>
LDBG_HOSTDEV uint32_t record_java_hash(const GraphView& g, int64_t idx) {   // CortexRecord.hashCode :398-408
    uint32_t hl = 1, hi = 1, hb = 1;
    const Kmer<W> key = graph_key<W>(g, idx);
    for (int i = 0; i < W; i++) hl = 31u * hl + bswap64_fold(kmer_word<W>(key, i));
    for (int c = 0; c < g.C; c++) hi = 31u * hi + graph_cov(g, idx, c);
    for (int c = 0; c < g.C; c++) hb = 31u * hb + (uint32_t)(int32_t)(int8_t)graph_edges(g, idx, c);
    return hl - hi + hb;
}
template <int W>
LDBG_DEV int order_children(const DfsArgs& a, VisitedTable& vt, const Node& cv, bool fwd, uint32_t avs_mask, uint8_t* out) {
    const EngineView& e = a.w.e;
    uint32_t keys[4];
    int n = 0;
    // per-colour neighbour masks of cv (as node_fill, colour by colour)
    const uint8_t* ed = graph_row(e.g, cv.idx) + e.g.edges_off;
    const bool fj = cv.fj != 0;
    for (unsigned b = 0; b < 4; b++) {
        if (!((avs_mask >> b) & 1u)) continue;
        uint32_t rank = 0xFFu;
        // colours in the order the reference concatenates them: traversal colours (insertion order) if they
        // give any neighbour, else recruitment colours (ascending)
        const uint32_t tmask = fwd ? cv.next_mask : cv.prev_mask;
        (void)tmask;
        bool from_trav = false;
        for (int t = 0; t < a.n_trav && !from_trav; t++) {
            const uint32_t eb = ed[a.trav_order[t]];
            const uint32_t lo = eb & 0xf, hi = eb >> 4;
            const uint32_t f = !fj ? lo : hi, rn = !fj ? hi : lo;
            const uint32_t r = ((rn & 1u) << 3) | ((rn & 2u) << 1) | ((rn & 4u) >> 1) | ((rn & 8u) >> 3);
            if ((fwd ? f : r) != 0) from_trav = true;
        }
        if (from_trav) {
            for (int t = 0; t < a.n_trav; t++) {
                const uint32_t eb = ed[a.trav_order[t]];
                const uint32_t lo = eb & 0xf, hi = eb >> 4;
                const uint32_t f = !fj ? lo : hi, rn = !fj ? hi : lo;
                const uint32_t r = ((rn & 1u) << 3) | ((rn & 2u) << 1) | ((rn & 4u) >> 1) | ((rn & 8u) >> 3);
                if ((((fwd ? f : r) >> b) & 1u) && rank == 0xFFu) rank = (uint32_t)t;
            }
        } else {
            uint32_t t = 0;
            for (int c = 0; c < e.g.C; c++) {
                if (!((e.recruit_mask >> c) & 1u)) continue;
                const uint32_t eb = ed[c];
                const uint32_t lo = eb & 0xf, hi = eb >> 4;
                const uint32_t f = !fj ? lo : hi, rn = !fj ? hi : lo;
                const uint32_t r = ((rn & 1u) << 3) | ((rn & 2u) << 1) | ((rn & 4u) >> 1) | ((rn & 8u) >> 3);
                if ((((fwd ? f : r) >> b) & 1u) && rank == 0xFFu) rank = t;
                t++;
            }
        }
        const Kmer<W> ck = child_kmer<W>(e, cv, fwd, b);
        const uint32_t sh = kmer_java_hash<W>(ck, e.g.k);
        Node x;
        node_child(e, cv, fwd, b, x);
        uint32_t vh = sh;
        vh = 31u * vh + (x.idx >= 0 ? record_java_hash<W>(e.g, x.idx) : 0u);   // CortexVertex.hashCode :82-91
        vh = 31u * vh; vh = 31u * vh; vh = 31u * vh; vh = 31u * vh;            // locus, sources, copyIndex 0, index 0
        const uint32_t actg = b == 0 ? 0u : (b == 1 ? 1u : (b == 3 ? 2u : 3u));
        keys[n] = (jbucket16(vh) << 24) | (jbucket16(sh) << 16) | ((rank & 0xFFu) << 8) | (actg << 2) | b;
        n++;
    }
    // insertion sort of at most 4 keys
    for (int i = 1; i < n; i++) {
        const uint32_t x = keys[i];
        int j = i - 1;
        while (j >= 0 && keys[j] > x) { keys[j + 1] = keys[j]; j--; }
        keys[j + 1] = x;
    }
    for (int i = 0; i < n; i++) out[i] = (uint8_t)(keys[i] & 3u);
    return n;
}

// ---- log helpers
// a failed append: the strand's own block table is full (the host retries with a longer one) or the shared pool ran dry
LDBG_DEV uint32_t append_failure(const DfsArgs& a, const StrandState& st) {
    return (int)(st.pw.n / LDBG_PATH_BLOCK) >= a.w.max_blocks ? (uint32_t)ST_LOG_FULL : (uint32_t)ST_POOL_FULL;
}
template <int W>
LDBG_DEV bool log_vertex(const DfsArgs& a, DfsLane<W>& L, const Node& v, const Kmer<W>& nk) {
    StrandState& st = L.st;
#ifdef LDBG_HOSTSIM
    if (getenv("LDBG_DFS_TRACE_IDX") && v.idx == atoll(getenv("LDBG_DFS_TRACE_IDX")))
        fprintf(stderr, "KL log_vertex idx %d flip %d copy %d depth %u iters %u gV %u logpos %u ui_valid %d pos %u cut %u\n", v.idx, (int)v.flip, v.copy, L.depth, st.iters, st.gV, st.pw.n, (int)ui_valid(v.ui), ui_pos(v.ui), L.cut);
#endif
    if (!path_append(a.w, st.s, st.pw, pack_vertex(v))) return false;
    if (v.idx < 0) {
        if (!path_append(a.w, st.s, st.pw, DFS_KMER)) return false;
        for (int i = 0; i < W; i++) {
            const uint64_t w = kmer_word<W>(nk, i);
            if (!path_append(a.w, st.s, st.pw, DFS_HALF(w)) || !path_append(a.w, st.s, st.pw, DFS_HALF(w >> 32))) return false;
        }
    }
    return true;
}

// ---- the size of a branch's graph once children have been merged into it (Graphs.addGraph :454).  dfs() hands every child
// currentGraphSize + g.vertexSet().size() (:445), and g has grown by the graphs of the children that returned one before: by the
// vertices they hold that g did not — siblings do not see each other's `visited`, so two of them can come back with the same
// vertices.  Only DestinationStopper looks at the graph size (its junction limit), so the count is taken for that rule alone, when a
// junction that already has a successful child opens another: the DISTINCT vertices among the log entries of the children so far, in
// a scratch hash set placed behind the log (rare, and as long as the logs it reads).
LDBG_DEV void path_write(const WalkArgs& a, int64_t s, uint32_t pos, uint64_t v) {
    a.pool[(uint64_t)a.block_table[s * a.max_blocks + pos / LDBG_PATH_BLOCK] * LDBG_PATH_BLOCK + (pos & (LDBG_PATH_BLOCK - 1))] = v;
}
template <int W>
LDBG_DEV bool merged_vertices(const DfsArgs& a, DfsLane<W>& L, uint32_t start, uint32_t& count) {
    StrandState& st = L.st;
    const uint32_t end = st.pw.n;
    // vertices the entries stand for (a RUN descriptor = a stretch of the run index: its vertices are counted one by one — siblings cut the
    // same stretch in different places, so a stretch in one log can be single vertices, or part of a longer stretch, in another)
    uint64_t total = 0;
    for (uint32_t pos = start; pos < end; pos++) {
        const uint64_t en = path_read(a.w, st.s, pos);
        if (!(en & DFS_MARK)) { total++; continue; }
        if (en == DFS_KMER) { st.status = ST_MERGE_UNSUPPORTED; return false; }
        if (LDBG_PD_KIND(en) == LDBG_PD_RUN) { total += en & 0xFFFFFu; pos++; }
    }
    uint32_t cap = 64;
    while ((uint64_t)cap < 2ull * total) cap <<= 1;
    for (uint32_t i = 0; i < cap; i++) if (!path_append(a.w, st.s, st.pw, 0ull)) { st.status = append_failure(a, st); return false; }
    count = 0;
    auto insert = [&](uint64_t key) {                                    // key: (record + 1) | flip << 33 | copyIndex << 36, never 0
        uint32_t h = (uint32_t)(sig_mix(key) >> 20) & (cap - 1u);
        while (true) {
            const uint64_t cur = path_read(a.w, st.s, end + h);
            if (cur == 0ull) {
                path_write(a.w, st.s, end + h, key); count++;
#ifdef LDBG_HOSTSIM
                if (getenv("LDBG_DFS_TRACE_KEYS")) fprintf(stderr, "K %u %lld %d %d\n", L.depth, (long long)path_idx(key), (int)path_flip(key), path_copy(key));
#endif
                return;
            }
            if (cur == key) return;
            h = (h + 1u) & (cap - 1u);
        }
    };
    const uint64_t ident = ~((3ull << 34) | (1ull << 60));               // a vertex entry without the base it was reached by and the quirk note
    for (uint32_t pos = start; pos < end; pos++) {
        const uint64_t en = path_read(a.w, st.s, pos);
        if (!(en & DFS_MARK)) {
            if (path_idx(en) < 0) { st.status = ST_MERGE_UNSUPPORTED; return false; }
#ifdef LDBG_HOSTSIM
            if (getenv("LDBG_DFS_TRACE_IDX") && path_idx(en) == atoll(getenv("LDBG_DFS_TRACE_IDX"))) fprintf(stderr, "KV single pos %u flip %d copy %d depth %u\n", pos, (int)path_flip(en), path_copy(en), L.depth);
#endif
            insert(en & ident);
            continue;
        }
        if (LDBG_PD_KIND(en) != LDBG_PD_RUN) continue;                    // OPEN, CLOSE, PAD
        const uint64_t payload = path_read(a.w, st.s, pos + 1);
        pos++;
        const uint32_t len = (uint32_t)(en & 0xFFFFFu), acopy = (uint32_t)(en >> 20) & 0xFFFFu, first = (uint32_t)payload;
        const bool asc = (en >> 36) & 1ull, inv = (en >> 37) & 1ull;
        const int copy = st.fwd ? (int)acopy : -(int)acopy;
        for (uint32_t t = 0; t < len; t++) {                              // as k_expand_paths (walk.cpp) materialises them
            const uint32_t u = LDBG_GLOBAL(const uint32_t, a.w.e.runs.uo)[asc ? first + t : first - t];
#ifdef LDBG_HOSTSIM
            if (getenv("LDBG_DFS_TRACE_IDX") && (int64_t)(u & 0x7FFFFFFFu) == atoll(getenv("LDBG_DFS_TRACE_IDX"))) fprintf(stderr, "KV run pos %u t %u of len %u first %u asc %d inv %d copy %d depth %u\n", pos, t, len, first, (int)asc, (int)inv, copy, L.depth);
#endif
            insert(path_pack((int64_t)(u & 0x7FFFFFFFu), ((u >> 31) != 0u) != inv, 0u, copy) & ident);
        }
    }
    path_truncate(a.w, st.s, st.pw, end);
    return true;
}

// dfs(cv, goForward, size, depth, visited, sinks) entered: TraversalEngine.java:356-371
template <int W>
LDBG_DEV void open_branch(const DfsArgs& a, DfsLane<W>& L, LinkStoreDev& ls, const Node& av, const Kmer<W>& avk, uint32_t size, int64_t slot) {
    StrandState& st = L.st;
    const EngineView& e = a.w.e;
    if ((int)L.depth >= a.max_depth) { st.status = ST_DEPTH_OVERFLOW; return; }
#ifdef LDBG_HOSTSIM
    if (getenv("LDBG_DFS_TRACE")) fprintf(stderr, "P open depth %u size %u iters %u\n", L.depth, size, st.iters);
#endif
    L.log_start = st.pw.n;
    L.size = size;
    L.nlin = 1;
    L.ss.flags = 0; L.ss.a = 0; L.ss.b = 0;
    L.last_prev = false;
    st.gV = 0;
    st.cv = av;
    L.nullk = avk;
    L.cut = av.idx >= 0 && ui_valid(av.ui) ? ui_pos(av.ui) : LDBG_RUN_NONE;     // (Node.ui is 0 where the engine has no run index)
    if (L.cut != LDBG_RUN_NONE && L.depth > 0u) {
        // This branch's first vertex cuts the piece it lies in.  If a branch further up THIS chain has crossed that piece — its interior is
        // marked under the piece's key (runstep.h) — the cut would leave vertices the ancestor visited as interior ones behind per-vertex
        // entries that know nothing of that visit (copy indices and the "visited before" test of :424 would come out wrong: a search
        // that circles back INTO a stretch it crossed; found by the seed sweep, tests/test_soak_hostsim.py seed 231).  The two
        // representations cannot be reconciled here: the search is handed back and run again without the index (ST_RETRY_PLAIN).
        const uint32_t pos = L.cut;
        uint32_t S = pos - ui_dstart(av.ui), E = pos + ui_dend(av.ui);
        for (uint32_t d = 0; d < L.depth; d++) piece_cut(S, E, pos, LDBG_GLOBAL(const uint32_t, &a.frames[(size_t)slot * a.max_depth + d].cut)[0]);
        if (E - S + 1u >= LDBG_RUN_MIN) {
            // (in either orientation: the cut is a position, it shortens the piece for the vertices of both strands of the chain)
            for (uint64_t plus = 0; plus < 2ull && st.status == ST_OK; plus++) {
                const uint64_t key = (1ull << 33) | ((uint64_t)S << 1) | plus;      // runstep.h: piece_key
                uint32_t h = vt_hash(key) & st.vt.mask;
                for (uint32_t probes = 0; probes <= st.vt.mask; probes++, h = (h + 1u) & st.vt.mask) {
                    const uint64_t ev = LDBG_GLOBAL(const uint64_t, st.vt.tab)[h];
                    if (ev == 0ull) break;
                    if ((ev & LDBG_VT_KEY_MASK) == key) { if (vt_count_e(ev) > 0) st.status = ST_RETRY_PLAIN; break; }
                }
            }
            if (st.status != ST_OK) return;
        }
    }
    if (!path_append(a.w, st.s, st.pw, DFS_OPEN) || !log_vertex<W>(a, L, av, avk)) { st.status = append_failure(a, st); return; }
    st.quirk |= av.flip && !av.fj;
    if (e.cursor_on) {                                   // seek(cv.getKmerAsString()) :363-365
        if (++st.cu.epoch > LDBG_VT_EPOCH_MAX) {         // epoch numbers used up: forget every `seen` mark
            for (uint32_t i = 0; i <= st.vt.mask; i++) st.vt.tab[i] = vt_with_seen(st.vt.tab[i], 0);
            st.cu.epoch = 1;
            const uint32_t used = st.vt.used;
            node_locate(st.vt, st.cv);                   // its cached entry carries an old mark
            st.vt.used = used;
        }
        cursor_seek(e, st.cu, ls, st.vt, st.cv, st.fwd);
    }
    L.phase = PH_ITER;
}

// the branch returns: a graph (success) or null
template <int W>
LDBG_DEV bool end_branch(const DfsArgs& a, DfsLane<W>& L, bool success) {
    StrandState& st = L.st;
    L.result = success;
#ifdef LDBG_HOSTSIM
    if (getenv("LDBG_DFS_TRACE")) fprintf(stderr, "P end depth %u success %d gV %u size %u iters %u\n", L.depth, (int)success, st.gV, L.size, st.iters);
#endif
    if (L.depth == 0) {
        if (success) { if (!path_append(a.w, st.s, st.pw, DFS_CLOSE)) st.status = append_failure(a, st); }
        else { path_truncate(a.w, st.s, st.pw, 0); st.branch_null = true; }
        return true;
    }
    L.phase = PH_UNDO;
    L.undo_pos = L.log_start + 1;
    L.undo_left = L.last_prev ? L.nlin - 1 : L.nlin;
    return false;
}

// ---- run steps (runstep.h) in a search.  A branch walks through an unbranched stretch exactly as a contig walk does; what differs
// is that the stopping rule is asked at every vertex, that a branch can end inside a stretch with the search going on elsewhere,
// and that what a branch added to `visited` is taken out again when it returns.
//  * every branch's first vertex is a piece of its own (a cut, like the seed of a walk): the branch opens with seek() and a general
//    step there, and its parents' first vertices stay cut for as long as their `visited` entries are in the table;
//  * for the rules below, "how many iterations until the rule fires" has a closed form over a stretch (the adjacent-vertex count
//    is 1 throughout): thresholds on the branch and graph sizes, and the position of a sink in the stretch;
//  * a branch that ends inside a stretch marks the interior as visited like one that crossed it: the marks are undone at once
//    (PH_UNDO sees the RUN descriptor) or the strand ends.
// Rules that look at every vertex (ROI lookups, degrees) run without the index.
LDBG_HOSTDEV bool dfs_rule_has_closed_form(int stopper) {
    return stopper == LDBG_STOP_CONTIG || stopper == LDBG_STOP_CYCLE_COLLAPSING_CONTIG || stopper == LDBG_STOP_DESTINATION || stopper == LDBG_STOP_EXPLORATION ||
           stopper == LDBG_STOP_GAP_CLOSING || stopper == LDBG_STOP_BUBBLE_CLOSING || stopper == LDBG_STOP_VISUALIZATION;
}
template <int W>
LDBG_DEV Piece dfs_piece(const DfsArgs& a, const DfsLane<W>& L, int64_t slot, const Node& v) {
    const uint32_t pos = ui_pos(v.ui);
    uint32_t S = pos - ui_dstart(v.ui), E = pos + ui_dend(v.ui);
    piece_cut(S, E, pos, L.cut);
    if (E - S + 1u >= LDBG_RUN_MIN)                     // (cuts only shorten a piece)
        for (uint32_t d = 0; d < L.depth; d++) piece_cut(S, E, pos, LDBG_GLOBAL(const uint32_t, &a.frames[(size_t)slot * a.max_depth + d].cut)[0]);
    return piece_make(v.ui, v.flip != 0, L.st.fwd, S, E);
}
template <int W>
LDBG_DEV bool dfs_mode_a(const DfsArgs& a, const DfsLane<W>& L, int64_t slot) {
    const StrandState& st = L.st;
    if (!lean_cursor_ok(a.w.e, st) || !ui_valid(st.cv.ui) || !ui_valid(st.cu.nxt.ui)) return false;
    const Piece pc = dfs_piece<W>(a, L, slot, st.cv);
    if (pc.q != 0u || pc.n < LDBG_RUN_MIN) return false;
    const Piece pt = dfs_piece<W>(a, L, slot, st.cu.nxt);
    return pt.S == pc.S && pt.E == pc.E && pt.plus == pc.plus && pt.q == 1u;
}
template <int W>
LDBG_DEV bool dfs_mode_b(const DfsArgs& a, const DfsLane<W>& L, int64_t slot) {
    const EngineView& e = a.w.e;
    const StrandState& st = L.st;
    const Node& cv = st.cv;
    if (st.status != ST_OK || (e.cursor_on && st.cu.has) || !(e.g.k & 1)) return false;
    if (cv.idx < 0 || cv.npe || cv.flip != cv.fj || !ui_valid(cv.ui)) return false;
    const int acopy = cv.copy < 0 ? -cv.copy : cv.copy;
    if (!(acopy >= vt_count_e(cv.vent) && acopy + 1 <= 32767)) return false;
    const Piece pc = dfs_piece<W>(a, L, slot, cv);
    return pc.q == 1u && pc.n >= LDBG_RUN_MIN;
}
// Iterations i = 1 .. steps of the loop at :373-481, iteration i standing on q_{q0+i-1} with one adjacent vertex.  keep = the
// iterations that pass before the rule ends the branch (== steps: it does not, in this stretch); succ: how it ends.
template <int W>
LDBG_DEV void dfs_rule_run(const DfsArgs& a, DfsLane<W>& L, const Piece& pc, uint32_t q0, uint32_t steps, uint32_t& keep, bool& succ) {
    const EngineView& e = a.w.e;
    StrandState& st = L.st;
    keep = 0; succ = false;
    {   // the first iteration as the k-mer-by-k-mer code would take it (these rules keep no state)
        TravState ts{(int)(L.size + st.gV), (int)L.depth, (int)st.gV, 1, false, st.gV > (uint32_t)e.max_len};
        StopEval<W> ev(e, a.env, L.sink_lo, L.sink_hi, st.cv, L.nullk);
        StopState ss = L.ss;
        if (ev.has_succeeded(ss, ts) || ev.has_failed(ss, ts) || ev.status != ST_OK) return;
    }
    // from the second iteration on the branch size is base + i - 1 (connectVertex counts the first pair as two, :494-516)
    const int64_t base = st.gV == 0 ? 1 : (int64_t)st.gV;
    int64_t fire = (int64_t)steps + 1;                  // first iteration at which the rule ends the branch
    bool fire_succ = false;
    auto threshold = [&](int64_t limit, bool is_succ) {  // the rule fires once the branch size exceeds `limit`
        const int64_t i = limit - base + 2 < 2 ? 2 : limit - base + 2;
        if (i < fire) { fire = i; fire_succ = is_succ; }
    };
    switch (e.stopper) {
        case LDBG_STOP_CONTIG: case LDBG_STOP_EXPLORATION: threshold(e.max_len, true); break;
        case LDBG_STOP_VISUALIZATION: threshold(500, true); break;
        case LDBG_STOP_BUBBLE_CLOSING: threshold(10000, false); break;
        case LDBG_STOP_DESTINATION: {
            threshold(e.max_len, false);
            // junction depth > 1 + ceil(5 exp(-size / 10^4)) (stoppers.h: destination_junction_limit): the first graph size at which it holds
            const int d = (int)L.depth;
            const int64_t s_star = d >= 6 ? 2232 : (d == 5 ? 5109 : (d == 4 ? 9163 : (d == 3 ? 16095 : (d == 2 ? 7451333 : -1))));
            if (s_star >= 0) threshold(s_star - 1 - (int64_t)L.size, false);     // size + branch size >= s_star
            for (int64_t t = L.sink_lo; t < L.sink_hi; t++) {                   // a sink in the stretch: the rule succeeds standing on it
                const uint64_t key = a.env.sink_keys[t];
                if (key == 0) continue;
                const uint64_t ui = LDBG_GLOBAL(const uint64_t, e.runs.uinfo)[(key >> 1) - 1];
                if (!ui_valid(ui)) continue;
                const uint32_t pos = ui_pos(ui);
                if (pos < pc.S || pos > pc.E || (((key & 1ull) != 0ull) == ui_orient(ui)) != pc.plus) continue;
                const int64_t i = (int64_t)(pc.asc ? pos - pc.S : pc.E - pos) - (int64_t)q0 + 1;
                if (i >= 2 && i <= fire) { fire = i; fire_succ = true; }        // (<=: has_succeeded is asked first)
            }
            break;
        }
        default: break;                                  // CycleCollapsingContig, GapClosing: nothing changes along a stretch
    }
    keep = fire > (int64_t)steps ? steps : (uint32_t)(fire - 1);
    succ = fire_succ;
}
LDBG_DEV bool dfs_run_emit(const WalkArgs& a, StrandState& st, uint32_t len, uint32_t acopy, const Piece& pc) {
    if (len == 0u) return true;
    const uint32_t first = pc.asc ? pc.S + 2u : pc.E - 2u;        // interior vertices q_2 ..; the piece's start rides along for PH_UNDO
    return path_append_pair(a, st.s, st.pw, pd_run_head(len, acopy, pc.asc, !pc.plus), (uint64_t)first | ((uint64_t)pc.S << 32));
}
// 0: not taken (the iteration is left to dfs_step); 1: taken; 2: taken and the strand has ended
template <int W>
LDBG_DEV int dfs_run_step(const DfsArgs& a, DfsLane<W>& L, LinkStoreDev& ls, int64_t slot, bool mode_a) {
    const EngineView& e = a.w.e;
    StrandState& st = L.st;
    const bool fwd = st.fwd;
    Node& cv = st.cv;
    const Piece pc = dfs_piece<W>(a, L, slot, cv);
    const uint32_t n = pc.n, nB = n - 4u;
    const bool inv = !pc.plus;
    const uint32_t steps = mode_a ? n - 2u : n - 3u;
    uint32_t keep; bool succ;
    dfs_rule_run<W>(a, L, pc, mode_a ? 0u : 1u, steps, keep, succ);
    if (keep == 0u) return 0;
#ifdef LDBG_HOSTSIM
    if (getenv("LDBG_DFS_TRACE")) fprintf(stderr, "P run depth %u mode %c n %u keep %u succ %d gV %u size %u iters %u S %u E %u q %u\n", L.depth, mode_a ? 'A' : 'B', n, keep, (int)succ, st.gV, L.size, st.iters, pc.S, pc.E, pc.q);
#endif
    const bool full = keep >= steps;
    const uint32_t k = full ? steps : keep;
    Node y, z;
    run_vertex(e, st.vt, pc.asc ? pc.E - 1u : pc.S + 1u, inv, fwd, y);
    run_vertex(e, st.vt, pc.asc ? pc.E : pc.S, inv, fwd, z);
    uint64_t eB = 0;
    const uint64_t kB = piece_key(pc);
    const uint32_t hB = vt_hash(kB) & st.vt.mask;
    const uint32_t slotB = vt_probe_from(st.vt, kB, hB, vt_peek(st.vt, hB), &eB);
    const int cntB = vt_count_e(eB), cntY = vt_count_e(y.vent);
    const uint32_t base = st.gV == 0u ? 1u : st.gV;
    const int acv = cv.copy < 0 ? -cv.copy : cv.copy;
    if (mode_a) {
        // iterations i = 1 .. n-2: cursor onto t = q_i with q_{i+1} looked up, av = q_i, visited.add(q_{i-1}), the rule on q_{i-1}
        Node& t = st.cu.nxt;
        const uint32_t ep = st.cu.epoch;
        const int cntT = vt_count_e(t.vent);
        const bool seenB = vt_seen_e(eB, ep), seenY = vt_seen_e(y.vent, ep), seenZ = vt_seen_e(z.vent, ep);
        bool odd = false;
        if (ls.n == 0u) odd = seenB || (seenY && k >= n - 3u);                 // the cursor would run out inside the piece
        odd = odd || (k >= 2u && cntT + 1 > 32767) || (k >= 3u && cntB + 1 > 32767);
        if (odd) { st.status = ST_RETRY_PLAIN; return 2; }
#ifdef LDBG_HOSTSIM
        ls_debug().runs_a++; ls_debug().run_vertices += k;
#endif
        if (ls_num_new(ls) > 0) ls_increment_ages(ls);                         // :274-276, first step; nothing is new afterwards
        st.iters += k;
        Node tv = t;
        tv.copy = fwd ? cntT : -cntT;
        bool ok = path_append(a.w, st.s, st.pw, pack_vertex(tv));
        ok = ok && dfs_run_emit(a.w, st, k - 1u < nB ? k - 1u : nB, (uint32_t)cntB, pc);
        if (full) { y.copy = fwd ? cntY : -cntY; ok = ok && path_append(a.w, st.s, st.pw, pack_vertex(y)); }
        if (!ok) { st.status = append_failure(a, st); return 2; }
        st.gV = base + k;
        L.nlin += k;
        L.last_prev = false;
        node_store(st.vt, cv, vt_with_count(cv.vent, acv + 1));
        node_store(st.vt, t, vt_with_count(t.vent, cntT + 1));
        if (!full) {
            // iteration k + 1 ends the branch standing on q_k: visited.add(q_0 .. q_k), nothing connected
            st.iters += 1u;
            if (k >= 2u) LDBG_GLOBAL(uint64_t, st.vt.tab)[slotB] = vt_with_count(eB, cntB + 1);
            return end_branch<W>(a, L, succ) ? 2 : 1;
        }
        // visited.add(q_0 .. q_{n-3}); seen.add(q_2 .. q_{n-1})
        uint64_t nb = vt_with_count(eB, cntB + 1);
        if (!seenB) nb = vt_with_seen(nb, ep);
        LDBG_GLOBAL(uint64_t, st.vt.tab)[slotB] = nb;
        if (!seenY) node_store(st.vt, y, vt_with_seen(y.vent, ep));
        if (!seenZ) node_store(st.vt, z, vt_with_seen(z.vent, ep));
        st.cu.has = !seenZ || ls.n > 0u;                                      // :262
        if (st.cu.has) st.cu.nxt = z;
        cv = y;
        st.cu.cur = cv;
        return 1;
    }
    // without the cursor: iterations j = 1 .. n-3 on cv = q_j whose only neighbour q_{j+1} has not been visited.  A visited interior or
    // far fringe under an unvisited q_1 takes a branch that started inside the stretch — those are cut out — so it is not expected
    if (cntB > 0 || cntY > 0) { st.status = ST_RETRY_PLAIN; return 2; }
#ifdef LDBG_HOSTSIM
    ls_debug().runs_b++; ls_debug().run_vertices += k;
#endif
    st.iters += k + (full ? 0u : 1u);
    bool ok = dfs_run_emit(a.w, st, k < nB ? k : nB, 0u, pc);
    if (full) { y.copy = 0; ok = ok && path_append(a.w, st.s, st.pw, pack_vertex(y)); }
    if (!ok) { st.status = append_failure(a, st); return 2; }
    st.gV = base + k;
    L.nlin += k;
    L.last_prev = false;
    node_store(st.vt, cv, vt_with_count(cv.vent, acv + 1));
    LDBG_GLOBAL(uint64_t, st.vt.tab)[slotB] = vt_with_count(eB, 1);
    if (!full) return end_branch<W>(a, L, succ) ? 2 : 1;                      // iteration k + 1 ends the branch on q_{k+1}
    cv = y;
    return 1;
}

// one micro-step; returns true when the strand has ended
template <int W>
LDBG_DEV bool dfs_step(const DfsArgs& a, DfsLane<W>& L, LinkStoreDev& ls, int64_t slot, const StepPre& pre, bool lean) {
    StrandState& st = L.st;
    const EngineView& e = a.w.e;
    const bool fwd = st.fwd;
    if (st.status != ST_OK) return true;
    if ((st.vt.used + 8) * 4 > (st.vt.mask + 1) * 3) { st.status = ST_TABLE_FULL; return true; }   // at its maximum size and filling up
    if (st.iters > a.iter_limit) { st.status = ST_TABLE_FULL; return true; }   // a rule that neither succeeds nor fails on a cycle (the reference spins here)

    if (L.phase == PH_UNDO) {
        if (L.undo_left > 0) {
            const uint64_t en = path_read(a.w, st.s, L.undo_pos);
            if (en == DFS_KMER) { L.undo_pos += 1 + 2 * W; return false; }
            if (en & DFS_MARK) {
                if (LDBG_PD_KIND(en) == LDBG_PD_RUN) {       // a stretch crossed in one step (dfs_run_step): its interior is one table entry
                    const uint64_t payload = path_read(a.w, st.s, L.undo_pos + 1);
                    const uint32_t len = (uint32_t)(en & 0xFFFFFu);
                    const uint64_t key = (1ull << 33) | ((payload >> 32) << 1) | (((en >> 37) & 1ull) ? 0ull : 1ull);   // runstep.h: piece_key
                    uint64_t ev = 0;
                    const uint32_t h0 = vt_hash(key) & st.vt.mask;
                    const uint32_t h = vt_probe_from(st.vt, key, h0, vt_peek(st.vt, h0), &ev);
                    LDBG_GLOBAL(uint64_t, st.vt.tab)[h] = vt_with_count(ev, vt_count_e(ev) - 1);
                    L.undo_left -= len < L.undo_left ? len : L.undo_left;
                    L.undo_pos += 2;
                } else L.undo_pos++;                          // (a PAD before a pair that would have straddled two blocks)
                return false;
            }
            L.undo_pos++;
            L.undo_left--;
            const int64_t idx = path_idx(en);
            if (idx >= 0) {
                uint64_t ev;
                const uint32_t h = vt_locate(st.vt, idx, path_flip(en), &ev);
                st.vt.tab[h] = vt_with_count(ev, vt_count_e(ev) - 1);
            }
            return false;
        }
        if (L.result) { if (!path_append(a.w, st.s, st.pw, DFS_CLOSE)) { st.status = append_failure(a, st); return true; } }
        else path_truncate(a.w, st.s, st.pw, L.log_start);
        L.depth--;
        DfsFrame& F = a.frames[(size_t)slot * a.max_depth + L.depth];
        if (L.result) F.any = 1;
        st.cv = F.cv; st.gV = F.gV;
        L.nlin = F.nlin; L.log_start = F.log_start; L.size = F.size; L.ss = F.ss; L.cut = F.cut;
        L.last_prev = false;
        L.phase = PH_CHILD;
        return false;
    }

    if (L.phase == PH_CHILD) {
        DfsFrame& F = a.frames[(size_t)slot * a.max_depth + L.depth];
        if (F.next < F.nchild) {
            const unsigned b = F.child[F.next];
            F.next++;
            Node av;
            node_child_located(e, st.vt, F.cv, fwd, b, av);
            Kmer<W> avk;
#pragma unroll
            for (int i = 0; i < W; i++) avk.w[i] = 0;
            if (av.idx < 0) avk = child_kmer<W>(e, F.cv, fwd, b);
            uint32_t size = F.size + F.gV;
            if (F.any && e.stopper == LDBG_STOP_DESTINATION) {      // g has grown by the graphs of the children before this one (merged_vertices)
                uint32_t merged = 0;
                if (!merged_vertices<W>(a, L, F.kids_start, merged)) return true;
                size = F.size + (F.gV ? F.gV : 1u) + merged;
#ifdef LDBG_HOSTSIM
                if (getenv("LDBG_DFS_TRACE")) fprintf(stderr, "P sibling depth %u F.size %u F.gV %u merged %u kids_start %u log_end %u\n", L.depth, F.size, F.gV, merged, F.kids_start, st.pw.n);
#endif
            }
            L.depth++;
            open_branch<W>(a, L, ls, av, avk, size, slot);
            if (st.cv.npe && st.status == ST_OK) st.status = ST_NULLPTR;
            return st.status != ST_OK;
        }
        bool ok = F.any != 0;
        if (!ok) {      // hasTraversalSucceeded with childrenWereTraversed = true :462-468
            TravState tc{(int)(F.size + F.gV), (int)L.depth, (int)F.gV, (int)F.adj, true, F.gV > (uint32_t)e.max_len};
            StopEval<W> ev(e, a.env, L.sink_lo, L.sink_hi, st.cv, L.nullk);
            ok = ev.has_succeeded(L.ss, tc);
            if (ev.status != ST_OK) { st.status = ev.status; return true; }
        }
        return end_branch<W>(a, L, ok);
    }

    // ---- PH_ITER: one iteration of the do-loop :373-481
    Node& cv = st.cv;
    int adj = 0;
    uint32_t avs_mask = 0;
    Node av = cv;
    bool previously = false;
    if (lean) {
        // the common case of a link-guided branch (lscoop.h: lean_cursor_ok): cursor step, copyIndex and visited.add(cv) without
        // the general machinery; what follows — the stopping rule, connectVertex or the end of the branch — is the same
        av = lean_cursor_advance<W>(e, st, ls);
        st.cu.cur = av;
        adj = 1;
    } else {
    st.iters++;
    const uint32_t m = fwd ? cv.next_mask : cv.prev_mask;
    if (e.cursor_on && st.cu.has) {                     // :379-407
        av = cursor_step<W, true>(e, st.cu, ls, st.vt, fwd, &pre);   // its link-store part was done by the wavefront (lscoop.h)
        if (st.cu.status != ST_OK) { st.status = st.cu.status; return true; }
        if (st.cu.has) { node_sync(cv, st.cu.nxt); node_sync(av, st.cu.nxt); }   // the `seen` mark may sit in a slot they hold
        const int cnt = node_count(av);                 // first copyIndex not in visited
        av.copy = fwd ? cnt : -cnt;
        adj = 1;
    } else {
        for (unsigned b = 0; b < 4; b++) {
            if (!((m >> b) & 1u)) continue;
            Node x;
            node_child_located(e, st.vt, cv, fwd, b, x);
            if (node_count(x) > 0) continue;            // avs.removeAll(visited) :416-422
            adj++;
            av = x;
            avs_mask |= 1u << b;
        }
    }
    const int acopy = cv.copy < 0 ? -cv.copy : cv.copy;
    const uint64_t ecv = cv.idx >= 0 ? cv.vent : 0ull;
    previously = acopy < vt_count_e(ecv);               // :424
    if (!previously && cv.idx >= 0) {
        if (acopy + 1 > 32767) { st.status = ST_COPY_OVERFLOW; return true; }
        node_store(st.vt, cv, vt_with_count(ecv, acopy + 1));   // visited.add(cv) :425
        node_sync(av, cv);
        if (e.cursor_on) { node_sync(st.cu.cur, cv); if (st.cu.has) node_sync(st.cu.nxt, cv); }
    }
    }
    L.last_prev = previously;
    bool succ = false, keep = false;
    if (!previously) {
        TravState ts{(int)(L.size + st.gV), (int)L.depth, (int)st.gV, adj, false, st.gV > (uint32_t)e.max_len};
        StopEval<W> ev(e, a.env, L.sink_lo, L.sink_hi, cv, L.nullk);
        succ = ev.has_succeeded(L.ss, ts);
        bool fail = false;
        if (ev.status == ST_OK) fail = ev.has_failed(L.ss, ts);
        if (ev.status != ST_OK) { st.status = ev.status; return true; }
        keep = !succ && !fail;
    }
    if (keep) {
        if (adj == 1) {                                 // connectVertex + advance :432-440
            Kmer<W> avk = L.nullk;
            if (av.idx < 0) avk = child_kmer<W>(e, cv, fwd, av.base);
            if (!log_vertex<W>(a, L, av, avk)) { st.status = append_failure(a, st); return true; }
            st.gV = st.gV == 0 ? 2 : st.gV + 1;
            L.nlin++;
            st.quirk |= av.flip && !av.fj;
            L.nullk = avk;
            cv = av;
            if (cv.npe) { st.status = ST_NULLPTR; return true; }
            return false;
        }
        // a junction (or a dead end the rule wants to look beyond): children in the reference's order :441-468
        DfsFrame& F = a.frames[(size_t)slot * a.max_depth + L.depth];
        F.cv = cv; F.log_start = L.log_start; F.size = L.size; F.gV = st.gV; F.nlin = L.nlin; F.ss = L.ss; F.cut = L.cut;
        F.next = 0; F.any = 0; F.adj = (uint8_t)adj; F.kids_start = st.pw.n;
        F.nchild = adj > 0 ? (uint8_t)order_children<W>(a, st.vt, cv, fwd, avs_mask, F.child) : 0;
        L.phase = PH_CHILD;
        return false;
    }
    return end_branch<W>(a, L, !previously && succ);    // :470-478
}

template <int W>
LDBG_DEV bool dfs_begin(const DfsArgs& a, DfsLane<W>& L, LinkStoreDev& ls, int64_t s, int64_t slot) {
    StrandState& st = L.st;
    const EngineView& e = a.w.e;
    st.s = s;
    st.fwd = (s & 1) != 0;
    st.status = ST_OK; st.iters = 0; st.gV = 0; st.branch_null = false; st.quirk = false;
    st.pw.cur = nullptr; st.pw.n = 0; st.pw.nblk = 0;
    st.cu.has = false; st.cu.status = ST_OK; st.cu.first = true; st.cu.epoch = 0;
    ls_clear(ls);
    L.depth = 0;
    L.sink_lo = a.sink_off ? a.sink_off[s >> 1] : 0;
    L.sink_hi = a.sink_off ? a.sink_off[(s >> 1) + 1] : 0;
    if (!vt_alloc(a.w, st.vt, a.w.vcap_init < a.w.vcap_max ? a.w.vcap_init : a.w.vcap_max)) { st.status = ST_POOL_FULL; return false; }
    const uint64_t* sw = a.w.seeds + (s >> 1) * W;
    Kmer<W> sk;
#pragma unroll
    for (int i = 0; i < W; i++) sk.w[i] = sw[i];
    Node v;
    if (a.w.seed_valid[s >> 1]) {
        if (a.w.img_on) seed_node<W>(e, sk, a.w.seed_slot[s >> 1], v);       // (the routed findRecord of the source was done before the first round)
        else node_find<W>(e, sk, v);
        node_locate(st.vt, v);
    } else node_null(e, v);
    if (v.npe) { st.status = ST_NULLPTR; return false; }
    open_branch<W>(a, L, ls, v, sk, 0, slot);
    return st.status == ST_OK;
}

template <int W, bool IMG>
LDBG_WAVE_KERNEL void k_dfs(DfsArgs a) {
    const int64_t slot = global_tid();
    if (slot >= a.w.n_slots) return;
#ifndef LDBG_HOSTSIM
    __shared__ LsElem lds_store[LDBG_LS_FAST * 64];
    LsElem* fast = lds_store + (threadIdx.x & 63u);
    const uint32_t fast_stride = 64;
#else
    static LsElem lds_store[LDBG_LS_FAST * 64];          // (one simulated wavefront at a time: rt.h)
    if ((rt::poison() || getenv("LDBG_HOSTSIM_ZERO_LDS")) && wave_lane() == 0) memset((void*)lds_store, rt::poison() ? 0xAB : 0, sizeof lds_store);      // (lane 0 is the first fibre to run)
    if (wave_lane() == 0) lds_shadow_begin(lds_store, sizeof lds_store);
    LsElem* fast = lds_store + wave_lane();
    const uint32_t fast_stride = (uint32_t)wave_size();
#endif
    LinkStoreDev ls;
    ls.fast = fast; ls.fast_cap = LDBG_LS_FAST; ls.fast_stride = fast_stride;
    ls.el = a.w.ls + (size_t)slot * a.w.ecap;
    ls.cap = a.w.ecap + LDBG_LS_FAST;
    ls_clear(ls);
    LsWave lw;
    lw.fast = fast - wave_lane(); lw.stride = fast_stride; lw.fast_cap = LDBG_LS_FAST;
    lw.el = a.w.ls + (size_t)(slot - wave_lane()) * a.w.ecap; lw.ecap = a.w.ecap;
    DfsLane<W> L;
    L.st.vt.tab = nullptr; L.st.vt.mask = 0; L.st.vt.used = 0; L.st.status = ST_OK;
    L.phase = PH_ITER;
    bool active = false, exhausted = false;
    // over an image (image.h): a search that needs a row that has not been sent yet suspends for the rest of this launch; the one a
    // lane was working on when the previous round ended is taken up again
    bool suspended = false, begun = true;
    const bool runs_on = a.w.e.runs.uinfo != nullptr;
    DfsSave<W>* save = (DfsSave<W>*)a.lane_save;
    if constexpr (IMG) {
        const DfsSave<W>& sv = save[slot];
        if (sv.active) {
            L = sv.L;
            ls.n = sv.ls_n; ls.java_cap = sv.ls_java_cap; ls.nkeys = sv.ls_nkeys; ls.next_seq = sv.ls_next_seq; ls.age = sv.ls_age; ls.n_new = sv.ls_n_new;
            ls.overflow = sv.ls_overflow != 0;
            for (uint32_t i = 0; i < LDBG_LS_FAST && i < sv.ls_n; i++) ls_set(ls, i, sv.fast[i]);
            active = true; begun = sv.begun != 0;
        }
    }
    uint32_t wave_iterations = 0;
    while (wave_ballot((active && !suspended) || (!active && !exhausted)) != 0ull) {
        if (IMG && a.w.yield_iters != 0u && wave_iterations >= a.w.yield_iters) break;      // (walk.cpp: a round lasts as long as its slowest wavefront)
        wave_iterations++;
        if (!active && !exhausted) {
            const int64_t fi = (int64_t)atomic_add_u64(a.w.next_strand, 1ull);
            if (fi >= a.w.n_strands) exhausted = true;
            else {
                const int64_t s = a.w.retry ? (int64_t)a.w.retry[fi]          // (the second launch: only the searches the run steps handed back)
                                            : (int64_t)(((unsigned __int128)fi * (unsigned __int128)a.w.fetch_stride) % (unsigned __int128)a.w.n_strands);
                const bool fwd = (s & 1) != 0;
                if ((fwd && !a.w.run_fwd) || (!fwd && !a.w.run_rev)) {
                    a.w.strand_n[s] = 0; a.w.status[s] = ST_BRANCH_NULL; a.w.iters[s] = 0; a.w.quirk[s] = 0;
                } else if (IMG) {
                    L.st.s = s; L.st.fwd = fwd; active = true; begun = false;
                } else {
                    active = dfs_begin<W>(a, L, ls, s, slot);
                    if (!active) strand_finish(a.w, L.st);
                }
            }
        }
        if (IMG && active && !suspended) {
            StrandState& st = L.st;
            if (!begun) {
                const int32_t sl = a.w.seed_valid[st.s >> 1] ? a.w.seed_slot[st.s >> 1] : -1;
                bool ready = true;
                if (sl >= 0) {
                    Kmer<W> sk;
                    const uint64_t* sw = a.w.seeds + (st.s >> 1) * W;
#pragma unroll
                    for (int i = 0; i < W; i++) sk.w[i] = sw[i];
                    Node sn;
                    seed_node<W>(a.w.e, sk, sl, sn);
                    ready = rows_ready(a.w.img, sn, st.fwd);
                }
                if (!ready) suspended = true;
                else {
                    begun = true;
                    active = dfs_begin<W>(a, L, ls, st.s, slot);
                    if (!active) strand_finish(a.w, L.st);
                }
            }
            if (active && begun && !suspended && st.status == ST_OK) {
                bool ready = true;
                if (L.phase == PH_ITER) {
                    // the iteration materialises the neighbours of cv in the direction of travel (:374-377, children order :441) and, with
                    // the cursor, those of the vertex the cursor is about to step onto (:379-407)
                    ready = rows_ready(a.w.img, st.cv, st.fwd);
                    if (a.w.e.cursor_on && st.cu.has) ready = rows_ready(a.w.img, st.cu.nxt, st.fwd) && ready;
                } else if (L.phase == PH_CHILD) {
                    // the next child's branch opens with seek(child) (:363-365): the child's own neighbours are looked at
                    DfsFrame& F = a.frames[(size_t)slot * a.max_depth + L.depth];
                    if (F.next < F.nchild) {
                        Node av;
                        node_child(a.w.e, F.cv, st.fwd, F.child[F.next], av);
                        ready = rows_ready(a.w.img, av, st.fwd);
                    }
                }
                if (!ready) suspended = true;
            }
        }
        const bool running = active && begun && !suspended;
        wave_grow_tables(a.w, L.st, running);
        // a whole unbranched stretch in one step (dfs_run_step)
        bool stepped = false;
        if (!IMG && runs_on && running && L.phase == PH_ITER && L.st.status == ST_OK && L.st.iters <= a.iter_limit &&
            (L.st.vt.used + 8) * 4 <= (L.st.vt.mask + 1) * 3) {
            const bool ma = dfs_mode_a<W>(a, L, slot);
            const bool mb = !ma && dfs_mode_b<W>(a, L, slot);
            if (ma || mb) {
                const int r = dfs_run_step<W>(a, L, ls, slot, ma);
                stepped = r != 0;
                if (r == 2) { strand_finish(a.w, L.st); active = false; }
            }
        }
        const bool stepping = running && !stepped;
        const bool lean = stepping && L.phase == PH_ITER && lean_cursor_ok(a.w.e, L.st);
        const bool cur_mode = stepping && !lean && L.st.status == ST_OK && L.phase == PH_ITER && a.w.e.cursor_on && L.st.cu.has;
        StepPre pre;
        pre.links_done = true; pre.choice_done = false; pre.choice_ok = false; pre.ch = 0; pre.has_child = false;
        if (wave_ballot(stepping && !lean) != 0ull) coop_step_prepare<W>(a.w.e, L.st, ls, lw, cur_mode, pre);
        if (stepping && dfs_step<W>(a, L, ls, slot, pre, lean)) { strand_finish(a.w, L.st); active = false; }
    }
    if constexpr (IMG) {
        DfsSave<W>& sv = save[slot];
        sv.active = active ? 1 : 0;
        if (active) {
            sv.L = L; sv.begun = begun ? 1 : 0;
            sv.ls_n = ls.n; sv.ls_java_cap = ls.java_cap; sv.ls_nkeys = ls.nkeys; sv.ls_next_seq = ls.next_seq; sv.ls_age = ls.age; sv.ls_n_new = ls.n_new;
            sv.ls_overflow = ls.overflow ? 1 : 0;
            for (uint32_t i = 0; i < LDBG_LS_FAST && i < ls.n; i++) sv.fast[i] = ls_get(ls, i);
            atomic_add_u64(a.w.unfinished, 1ull);
        }
    }
}

LDBG_KERNEL void k_dfs_round_stats(const unsigned long long* ctr, int64_t ns, const unsigned long long* n_req, int64_t* stats) {
    if (global_tid() != 0) return;
    const int64_t handed = (int64_t)ctr[0] < ns ? (int64_t)ctr[0] : ns;
    stats[0] = (int64_t)ctr[4] + (ns - handed);
    stats[1] = (int64_t)*n_req;
    stats[2] = (int64_t)*(const unsigned*)(n_req + 1);      // the image's overflow flag (image.cpp: d_ctr_[2])
}

// sink keys over an image: the sink's record is known by its image slot (-1 = none; -2 = the string is not a k-mer)
template <int W>
LDBG_KERNEL void k_sink_nodes_image(EngineView e, const uint64_t* words, const uint8_t* valid, const int32_t* slots, int64_t n, uint64_t* keys) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        uint64_t key = ~0ull;
        if (valid[i]) {
            key = 0;
            Kmer<W> sk;
            for (int w = 0; w < W; w++) sk.w[w] = words[i * W + w];
            Node v;
            seed_node<W>(e, sk, slots[i], v);
            if (v.idx >= 0) {
                if (graph_row(e.g, v.idx)[e.g.flags_off] & LDBG_ROW_PALINDROME) v.flip = 0;
                key = vt_key(v.idx, v.flip != 0);
            }
        }
        keys[i] = key;
    }
}

// per sink k-mer: its (record, orientation) key for the rules' sink tests
template <int W>
LDBG_KERNEL void k_sink_nodes(EngineView e, const uint64_t* words, const uint8_t* valid, int64_t n, uint64_t* keys) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        uint64_t key = ~0ull;             // a sink string that is not a k-mer equals no vertex's k-mer (neither key nor words are compared)
        if (valid[i]) {
            key = 0;
            Kmer<W> sk;
            for (int w = 0; w < W; w++) sk.w[w] = words[i * W + w];
            Node v;
            node_find<W>(e, sk, v);
            if (v.idx >= 0) {
                if (graph_row(e.g, v.idx)[e.g.flags_off] & LDBG_ROW_PALINDROME) v.flip = 0;
                key = vt_key(v.idx, v.flip != 0);
            }
        }
        keys[i] = key;
    }
}

// bit i: record i of graph g is a record of the ROI graph
template <int W>
LDBG_KERNEL void k_roi_bits(GraphView g, GraphView rois, uint32_t* bits) {
    for (int64_t i = global_tid(); i < rois.N; i += global_nthreads()) {
        const Kmer<W> key = graph_key<W>(rois, i);
        const int64_t idx = graph_find_canonical<W>(g, key);
        if (idx >= 0) atomic_or_u32(&bits[idx >> 5], 1u << (idx & 31));
    }
}

// k-mer words + coverages of a list of vertices (key = ((record + 1) << 1) | flip)
template <int W>
LDBG_KERNEL void k_gather_vertices(GraphView g, const uint64_t* keys, int64_t n, uint64_t* words, uint32_t* cov) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        const int64_t idx = (int64_t)(keys[i] >> 1) - 1;
        Kmer<W> km = graph_key<W>(g, idx);
        if (keys[i] & 1ull) km = kmer_revcomp<W>(km, g.k);
        for (int w = 0; w < W; w++) words[i * W + w] = kmer_word<W>(km, w);
        for (int c = 0; c < g.C; c++) cov[i * g.C + c] = graph_cov(g, idx, c);
    }
}

// edges, flags and neighbour index of a list of vertices (addSecondaryColors needs them for every vertex of a result)
LDBG_KERNEL void k_gather_rows(GraphView g, const uint64_t* keys, int64_t n, uint8_t* edges, uint8_t* flags, uint32_t* nbr) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        const int64_t idx = (int64_t)(keys[i] >> 1) - 1;
        const uint8_t* row = graph_row(g, idx);
        for (int c = 0; c < g.C; c++) edges[i * g.C + c] = row[g.edges_off + c];
        flags[i] = row[g.flags_off];
        for (int q = 0; q < 8; q++) nbr[i * 8 + q] = graph_nbr(g, idx, q);
    }
}

// ------------------------------------------------------------------ host: graph assembly from the event logs
namespace {

struct VKey {
    uint64_t id;       // ((record + 1) << 1) | flip, or (1 << 63) | interned k-mer number for a vertex without a record
    int32_t copy, index;
    bool operator==(const VKey& o) const { return id == o.id && copy == o.copy && index == o.index; }
};
struct VKeyHash {
    size_t operator()(const VKey& k) const {
        uint64_t x = k.id * 0x9E3779B97F4A7C15ull ^ ((uint64_t)(uint32_t)k.copy << 32 | (uint32_t)k.index) * 0xC2B2AE3D27D4EB4Full;
        return (size_t)(x ^ (x >> 29));
    }
};
// The part of JGraphT's DirectedWeightedPseudograph the reference relies on: insertion-ordered vertex and edge
// sets, addVertex/addEdge that refuse duplicates (CortexVertex.equals :67-80, CortexEdge.equals :42-57 — an edge
// equals another with the same two endpoints in EITHER direction and the same colour), Graphs.addGraph.
// The hash indices are built lazily: a branch's own vertices are new by construction (they were not in `visited`),
// and so is everything its FIRST returning child brings; only a second returning child (siblings do not see each
// other's visited sets) can repeat vertices or edges.
struct HGraph {
    std::vector<VKey> verts;
    std::vector<DfsEdge> edges;
    bool indexed = false;
    std::unordered_map<VKey, int, VKeyHash> vmap;
    std::unordered_set<uint64_t> dir, und;
    static uint64_t dir_key(int s, int t) { return ((uint64_t)(uint32_t)s << 32) | (uint32_t)t; }
    static uint64_t und_key(int s, int t, int color) {
        return ((uint64_t)(uint32_t)std::min(s, t) << 38) | ((uint64_t)(uint32_t)std::max(s, t) << 12) | (uint64_t)(uint32_t)(color & 0xFFF);
    }
    void ensure_index() {
        if (indexed) return;
        indexed = true;
        vmap.reserve(verts.size() * 2);
        for (size_t i = 0; i < verts.size(); i++) vmap.emplace(verts[i], (int)i);
        for (auto& e : edges) { dir.insert(dir_key(e.src, e.dst)); und.insert(und_key(e.src, e.dst, e.color)); }
    }
    int add_vertex_new(const VKey& v) {                 // caller knows v is not in the graph
        verts.push_back(v);
        if (indexed) vmap.emplace(v, (int)verts.size() - 1);
        return (int)verts.size() - 1;
    }
    int add_vertex(const VKey& v) {
        ensure_index();
        auto it = vmap.find(v);
        if (it != vmap.end()) return it->second;
        return add_vertex_new(v);
    }
    void add_edge_new(int s, int t, int color) {        // caller knows no equal edge exists
        edges.push_back({s, t, color});
        if (indexed) { dir.insert(dir_key(s, t)); und.insert(und_key(s, t, color)); }
    }
    bool contains_edge(int s, int t) { ensure_index(); return dir.count(dir_key(s, t)) != 0; }
    void add_edge(int s, int t, int color) {            // Graph.addEdge: refused when an equal CortexEdge is present
        ensure_index();
        if (!und.insert(und_key(s, t, color)).second) return;
        dir.insert(dir_key(s, t));
        edges.push_back({s, t, color});
    }
};

struct LogParser {
    const uint64_t* log;
    int64_t n, pos;
    int W, color;
    bool fwd;
    std::vector<std::vector<uint64_t>>& null_kmers;    // interned k-mers of vertices without a record
    std::unordered_map<std::string, uint32_t>& null_ids;

    VKey read_vertex() {
        const uint64_t en = log[pos++];
        VKey v;
        v.copy = path_copy(en); v.index = 0;
        const int64_t idx = path_idx(en);
        if (idx >= 0) v.id = ((uint64_t)(idx + 1) << 1) | (path_flip(en) ? 1ull : 0ull);
        else {
            if (pos >= n || log[pos] != DFS_KMER) throw StatusError(LDBG_ERR_HIP, "dfs log: a vertex without a record lacks its k-mer");
            pos++;
            std::vector<uint64_t> kw((size_t)W);
            for (int i = 0; i < W; i++) kw[(size_t)i] = (log[pos + 2 * i] & 0xFFFFFFFFull) | ((log[pos + 2 * i + 1] & 0xFFFFFFFFull) << 32);
            std::string s((const char*)kw.data(), (size_t)W * 8);
            auto it = null_ids.find(s);
            uint32_t id;
            if (it == null_ids.end()) {
                id = (uint32_t)null_kmers.size();
                null_kmers.push_back(kw);
                null_ids.emplace(s, id);
            } else id = it->second;
            pos += 2 * W;
            v.id = DFS_MARK | id;
        }
        return v;
    }
    void add_link(HGraph& g, int ci, int ai) { if (fwd) g.add_edge_new(ci, ai, color); else g.add_edge_new(ai, ci, color); }
    // One branch: the graph dfs(...) returned, the vertex it started from and where that vertex sits in the graph
    // (-1: not in it — a branch that decided at its first vertex returns an empty graph, :373-481).
    void parse_branch(HGraph& g, VKey& v0, int& v0_index) {
        if (pos >= n || log[pos] != DFS_OPEN) throw StatusError(LDBG_ERR_HIP, "dfs log: OPEN expected");
        pos++;
        v0 = read_vertex();
        v0_index = -1;
        VKey cv = v0;
        int cv_index = -1;
        while (pos < n && !(log[pos] & DFS_MARK)) {          // connectVertex(g, cv, {av}) per step :432-440
            const VKey av = read_vertex();
            if (cv_index < 0) { cv_index = g.add_vertex_new(cv); v0_index = cv_index; }
            const int ai = g.add_vertex_new(av);
            add_link(g, cv_index, ai);
            cv = av; cv_index = ai;
        }
        int merged = 0;
        std::vector<int> map;
        while (pos < n && log[pos] == DFS_OPEN) {
            HGraph br;
            VKey c0;
            int c0_index;
            parse_branch(br, c0, c0_index);
            // connectVertex(branch, cv, {av}) :447-453 — cv was in `visited`, so the branch cannot contain it
            const int br_cv = br.add_vertex_new(cv);
            if (c0_index < 0) c0_index = br.add_vertex_new(c0);
            add_link(br, br_cv, c0_index);
            // Graphs.addGraph(g, branch) :454 — vertices in the branch's order, then its edges
            map.resize(br.verts.size());
            if (merged == 0 && !g.indexed) {
                for (size_t j = 0; j < br.verts.size(); j++) {
                    if ((int)j == br_cv) { if (cv_index < 0) cv_index = g.add_vertex_new(cv); map[j] = cv_index; }
                    else map[j] = g.add_vertex_new(br.verts[j]);
                }
                for (auto& ed : br.edges) g.add_edge_new(map[ed.src], map[ed.dst], ed.color);
            } else {
                for (size_t j = 0; j < br.verts.size(); j++) map[j] = g.add_vertex(br.verts[j]);
                if (cv_index < 0) cv_index = map[(size_t)br_cv];
                for (auto& ed : br.edges) g.add_edge(map[ed.src], map[ed.dst], ed.color);
            }
            if (v0_index < 0 && cv == v0) v0_index = cv_index;
            merged++;
        }
        if (pos >= n || log[pos] != DFS_CLOSE) throw StatusError(LDBG_ERR_HIP, "dfs log: CLOSE expected");
        pos++;
    }
};

// java.util.TimSort on fewer than 32 elements: countRunAndMakeAscending + binarySort.  toWalk's comparators
// never return 0 (TraversalUtils.java:424-428), so the result depends on the algorithm, restated here.
template <class T, class Cmp>
void java_small_sort(std::vector<T>& v, Cmp cmp) {
    const int n = (int)v.size();
    if (n < 2) return;
    int hi = 1;
    if (cmp(v[hi++], v[0]) < 0) {
        while (hi < n && cmp(v[hi], v[hi - 1]) < 0) hi++;
        std::reverse(v.begin(), v.begin() + hi);
    } else {
        while (hi < n && cmp(v[hi], v[hi - 1]) >= 0) hi++;
    }
    for (int start = hi; start < n; start++) {
        T pivot = v[start];
        int l = 0, r = start;
        while (l < r) { int mid = (l + r) >> 1; if (cmp(pivot, v[mid]) < 0) r = mid; else l = mid + 1; }
        for (int i = start; i > l; i--) v[i] = v[i - 1];
        v[l] = pivot;
    }
}

}  // namespace

// ---- getNextVertices / getPrevVertices for a batch of k-mers (TraversalEngine.java:147-239): the neighbour vertices of every query in
// the iteration order of the HashSet<CortexVertex> the reference returns (order_children), at most four each.  One query per thread.
template <int W>
LDBG_KERNEL void k_neighbours(DfsArgs a, const uint64_t* words, const uint8_t* valid, int64_t n, int fwd, uint8_t* cnt, uint64_t* out_words, int64_t* out_rec,
                              uint32_t* status) {
    const EngineView& e = a.w.e;
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        Node v;
        Kmer<W> sk;
#pragma unroll
        for (int w = 0; w < W; w++) sk.w[w] = words[i * W + w];
        if (valid[i]) node_find<W>(e, sk, v); else node_null(e, v);
        cnt[i] = 0; status[i] = ST_OK;
        if (v.npe) { status[i] = ST_NULLPTR; continue; }      // no record, and recruitment colours are set: prevKmers.get(c) on a null map (Q14)
        if (v.idx < 0) continue;
        const uint32_t mask = fwd ? v.next_mask : v.prev_mask;
        if (!mask) continue;
        uint8_t ord[4];
        VisitedTable none;
        none.tab = nullptr; none.mask = 0; none.used = 0;
        const int nn = order_children<W>(a, none, v, fwd != 0, mask, ord);
        for (int c = 0; c < nn; c++) {
            const Kmer<W> ck = child_kmer<W>(e, v, fwd != 0, ord[c]);
            Node x;
            node_child(e, v, fwd != 0, ord[c], x);
            for (int w = 0; w < W; w++) out_words[(i * 4 + c) * W + w] = kmer_word<W>(ck, w);
            out_rec[i * 4 + c] = x.idx;
        }
        cnt[i] = (uint8_t)nn;
    }
}

void Engine::neighbours_batch(const char* kmers, int64_t n, bool forward, int64_t* offsets, uint64_t* kmer_words, int64_t* rec, int64_t capacity) {
    enter();
    rt::stream_t s = stream_;
    const int k = graph->hdr.k, W = graph->hdr.W;
    offsets[0] = 0;
    if (n <= 0) return;
    std::vector<uint64_t> words((size_t)n * W);
    std::vector<uint8_t> valid((size_t)n);
    ascii_batch_to_words(kmers, n, k, W, words.data(), valid.data());
    struct Tmp { std::vector<void*> p; ~Tmp() { for (void* x : p) rt::dfree(x); } void* get(size_t nbytes) { void* x = rt::dmalloc(nbytes); p.push_back(x); return x; } } tmp;
    uint64_t* d_words = (uint64_t*)tmp.get((size_t)n * W * 8);
    uint8_t* d_valid = (uint8_t*)tmp.get((size_t)n);
    uint8_t* d_cnt = (uint8_t*)tmp.get((size_t)n);
    uint64_t* d_out = (uint64_t*)tmp.get((size_t)n * 4 * W * 8);
    int64_t* d_rec = (int64_t*)tmp.get((size_t)n * 4 * 8);
    uint32_t* d_status = (uint32_t*)tmp.get((size_t)n * 4);
    rt::h2d(d_words, words.data(), (size_t)n * W * 8, s);
    rt::h2d(d_valid, valid.data(), (size_t)n, s);
    DfsArgs a;
    memset(&a, 0, sizeof(a));
    a.w.e = view;
    a.n_trav = 0;
    for (int i = 0; i < cfg.n_traversal; i++) {        // traversal colours in LinkedHashSet order
        bool dup = false;
        for (int j = 0; j < a.n_trav; j++) dup |= a.trav_order[j] == (uint8_t)cfg.traversal_colors[i];
        if (!dup) a.trav_order[a.n_trav++] = (uint8_t)cfg.traversal_colors[i];
    }
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096));
    switch (W) {
        case 1: LDBG_LAUNCH(k_neighbours<1>, grid, 256, s, a, (const uint64_t*)d_words, (const uint8_t*)d_valid, n, forward ? 1 : 0, d_cnt, d_out, d_rec, d_status); break;
        case 2: LDBG_LAUNCH(k_neighbours<2>, grid, 256, s, a, (const uint64_t*)d_words, (const uint8_t*)d_valid, n, forward ? 1 : 0, d_cnt, d_out, d_rec, d_status); break;
        case 3: LDBG_LAUNCH(k_neighbours<3>, grid, 256, s, a, (const uint64_t*)d_words, (const uint8_t*)d_valid, n, forward ? 1 : 0, d_cnt, d_out, d_rec, d_status); break;
        default: LDBG_LAUNCH(k_neighbours<4>, grid, 256, s, a, (const uint64_t*)d_words, (const uint8_t*)d_valid, n, forward ? 1 : 0, d_cnt, d_out, d_rec, d_status); break;
    }
    std::vector<uint8_t> cnt((size_t)n);
    std::vector<uint32_t> status((size_t)n);
    std::vector<uint64_t> ow((size_t)n * 4 * W);
    std::vector<int64_t> orec((size_t)n * 4);
    rt::d2h(cnt.data(), d_cnt, (size_t)n, s);
    rt::d2h(status.data(), d_status, (size_t)n * 4, s);
    rt::d2h(ow.data(), d_out, (size_t)n * 4 * W * 8, s);
    rt::d2h(orec.data(), d_rec, (size_t)n * 4 * 8, s);
    rt::stream_sync(s);
    for (int64_t i = 0; i < n; i++)
        if (status[(size_t)i] == ST_NULLPTR)
            throw StatusError(LDBG_ERR_NULLPOINTER, std::string(forward ? "getNextVertices" : "getPrevVertices") + ": record missing while recruitment colours are set (k-mer " + std::to_string(i) + ")");
    int64_t total = 0;
    for (int64_t i = 0; i < n; i++) { total += cnt[(size_t)i]; offsets[i + 1] = total; }
    if (total > capacity) throw StatusError(LDBG_ERR_CAPACITY, "neighbour buffers too small: need " + std::to_string(total));
    for (int64_t i = 0; i < n; i++)
        for (int c = 0; c < cnt[(size_t)i]; c++) {
            const int64_t o = offsets[i] + c;
            if (kmer_words) for (int w = 0; w < W; w++) kmer_words[o * W + w] = ow[((size_t)i * 4 + c) * W + w];
            if (rec) rec[o] = orec[(size_t)i * 4 + c];
        }
}

// ---- dfs(Collection<String> sources, Collection<String> sinks) (TraversalEngine.java:37-62): the graphs of the sources that returned one,
// merged in source order — the first as it is, every further one with Graphs.addGraph (its vertices that the merged graph does not hold
// yet, in their order; then its edges, each refused when an equal CortexEdge — same end points either way round, same colour — is there)
DfsBatch* dfs_merge(DfsBatch& b, const int64_t* which, int64_t m) {
    b.materialize();
    const int W = b.W, C = b.C;
    std::unique_ptr<DfsBatch> out(new DfsBatch);
    out->k = b.k; out->W = W; out->C = C; out->graph = b.graph; out->color = b.color; out->materialized = true;
    out->results.resize(1);
    DfsGraphHost& g = out->results[0];
    struct Key { int64_t rec; uint64_t flip_copy_index; std::string null_kmer; bool operator<(const Key& o) const {
        if (rec != o.rec) return rec < o.rec; if (flip_copy_index != o.flip_copy_index) return flip_copy_index < o.flip_copy_index; return null_kmer < o.null_kmer; } };
    std::map<Key, int> vmap;
    std::set<std::tuple<int, int, int>> emap;        // (min end, max end, colour)
    for (int64_t q = 0; q < m; q++) {
        if (which[q] < 0 || which[q] >= (int64_t)b.results.size()) throw StatusError(LDBG_ERR_ARG, "dfs_merge: result index out of range");
        const DfsGraphHost& s = b.results[(size_t)which[q]];
        if (s.is_null) continue;
        if (s.packed) throw StatusError(LDBG_ERR_HIP, "dfs_merge: a result is still packed after materialize()");
        g.is_null = false;
        std::vector<int> local((size_t)s.verts.size());
        for (size_t i = 0; i < s.verts.size(); i++) {
            const DfsVertex& v = s.verts[i];
            Key key{v.rec, 0ull, std::string()};
            key.flip_copy_index = ((uint64_t)v.flip << 63) | ((uint64_t)(uint32_t)v.copy << 24) | (uint64_t)((uint32_t)v.index & 0xFFFFFFu);
            if (v.rec < 0) key.null_kmer.assign((const char*)&s.words[i * (size_t)W], (size_t)W * 8);
            auto it = vmap.find(key);
            if (it == vmap.end()) {
                it = vmap.emplace(key, (int)g.verts.size()).first;
                DfsVertex nv = v;
                nv.slot = -1;
                g.verts.push_back(nv);
                g.words.insert(g.words.end(), s.words.begin() + (ptrdiff_t)(i * (size_t)W), s.words.begin() + (ptrdiff_t)((i + 1) * (size_t)W));
                g.cov.insert(g.cov.end(), s.cov.begin() + (ptrdiff_t)(i * (size_t)C), s.cov.begin() + (ptrdiff_t)((i + 1) * (size_t)C));
            }
            local[i] = it->second;
        }
        for (const DfsEdge& e : s.edges) {
            const int a = local[(size_t)e.src], c = local[(size_t)e.dst];
            if (!emap.insert(std::make_tuple(std::min(a, c), std::max(a, c), e.color)).second) continue;
            g.edges.push_back({a, c, e.color});
        }
    }
    return out.release();
}

// ------------------------------------------------------------------ host: Engine::dfs_batch
static uint32_t next_pow2_u(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return (uint32_t)std::min<uint64_t>(p, 1ull << 31); }
static int grid_of(int64_t n, int block, int max_blocks) {
    int64_t b = (n + block - 1) / block;
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, max_blocks));
}

void Engine::build_roi_bits() {
    if (!rois || d_roi_bits_) return;
    const GraphView& g = graph->view;
    if (rois->hdr.k != graph->hdr.k) throw StatusError(LDBG_ERR_ARG, "the ROI graph must have the k-mer size of the traversed graph");
    if (rois->device != graph->device) throw StatusError(LDBG_ERR_ARG, "the ROI graph must live on the device of the traversed graph");
    rt::stream_t s = stream_;
    const size_t words = (size_t)((g.N + 31) / 32 + 1);
    d_roi_bits_ = rt::dmalloc(words * 4);
    rt::dmemset(d_roi_bits_, 0, words * 4, s);
    GraphView exact = g;
    exact.java_tiny = 0;           // set membership, not findRecord: no Q1 here
    if (rois->view.N > 0) {
        const int grid = grid_of(rois->view.N, 256, 4096);
        switch (g.W) {
            case 1: LDBG_LAUNCH(k_roi_bits<1>, grid, 256, s, exact, rois->view, (uint32_t*)d_roi_bits_); break;
            case 2: LDBG_LAUNCH(k_roi_bits<2>, grid, 256, s, exact, rois->view, (uint32_t*)d_roi_bits_); break;
            case 3: LDBG_LAUNCH(k_roi_bits<3>, grid, 256, s, exact, rois->view, (uint32_t*)d_roi_bits_); break;
            default: LDBG_LAUNCH(k_roi_bits<4>, grid, 256, s, exact, rois->view, (uint32_t*)d_roi_bits_); break;
        }
    }
    rt::stream_sync(s);
}

DfsBatch* Engine::dfs_batch(const char* sources, int64_t n, const char* sinks, const int64_t* sink_offsets, const ShardedRun* sharded) {
    if (cfg.connect_all_neighbors) throw StatusError(LDBG_ERR_UNSUPPORTED, "dfs_batch: connectAllNeighbors is not supported on the device path");
    enter();
    materialize_pending();            // (the path pool is about to be reused: walks of the last batch keep their vertex lists)
    if (!sharded) build_roi_bits();      // (over an image the rules ask the ROI graph itself)
    const int k = graph->hdr.k, W = graph->hdr.W;
    std::vector<uint64_t> words((size_t)n * W);
    seed_valid_.resize((size_t)std::max<int64_t>(1, n));
    ascii_batch_to_words(sources, n, k, W, words.data(), seed_valid_.data());
    const int64_t nsinks = sink_offsets ? sink_offsets[n] : 0;
    std::vector<uint64_t> sink_words((size_t)std::max<int64_t>(1, nsinks) * W);
    sink_valid_.resize((size_t)std::max<int64_t>(1, nsinks));
    ascii_batch_to_words(sinks, nsinks, k, W, sink_words.data(), sink_valid_.data());

    std::unique_ptr<DfsBatch> out(new DfsBatch);
    out->k = k; out->W = W; out->C = graph->hdr.C; out->graph = graph;
    out->results.resize((size_t)n);
    out->traversed = 0;
    std::vector<std::pair<int64_t, int64_t>> todo{{0, n}};
    while (!todo.empty()) {
        auto [first, cnt] = todo.back();
        todo.pop_back();
        if (cnt <= 0) continue;
        if (!dfs_chunk(words, sink_words, sink_offsets, first, cnt, *out, sharded)) {
            if (sharded) throw StatusError(LDBG_ERR_CAPACITY, "dfs over a sharded table: a device pool ran dry: use smaller batches");
            if (cnt == 1) throw StatusError(LDBG_ERR_HIP, "dfs: pools too small for a single seed: not enough device memory");
            todo.push_back({first + cnt / 2, cnt - cnt / 2});
            todo.push_back({first, cnt / 2});
        }
    }
    dfs_traversed_ += out->traversed;
    return out.release();
}

// returns false when a pool ran dry (the caller splits the chunk)
bool Engine::dfs_chunk(const std::vector<uint64_t>& seed_words, const std::vector<uint64_t>& sink_words, const int64_t* sink_offsets,
                       int64_t first, int64_t n, DfsBatch& out, const ShardedRun* sharded) {
    const int W = graph->hdr.W, C = graph->hdr.C;
    rt::stream_t s = stream_;
    const int64_t ns = 2 * n;
    // a strand's visited table holds the vertices of one root-to-leaf chain of branches
    const uint64_t chain = std::min<uint64_t>((uint64_t)(cfg.max_branch_length + 12) * 64ull, (uint64_t)graph->view.N * 2 + 64);
    const uint32_t vcap_max = std::max<uint32_t>(64u, next_pow2_u(2ull * chain));
    // entries one strand's log may hold: the vertices of every branch that returned a graph
    const int max_blocks = (int)std::min<int64_t>(1 << 20, std::max<int64_t>(dfs_log_blocks, (((int64_t)cfg.max_branch_length + 2) * 8 + LDBG_PATH_BLOCK - 1) / LDBG_PATH_BLOCK + 1));
    // the pool follows the walks' table sizes (a chain of branches is rarely longer than a few branches' worth); what it always holds is
    // one seed's two strands at their largest, so that splitting a batch that ran the pool dry ends in chunks that fit
    // (over a sharded table the batch is not split — every rank must stay in step —, so the pool is sized for all of its searches)
    ensure_scratch(ns, link_store_capacity, max_blocks, (sharded ? (uint64_t)ns : 2ull) * vt_series(vt_initial_entries(), vcap_max));
    zero_dirty_tables(s);

    struct Tmp { std::vector<void*> p; ~Tmp() { for (void* x : p) rt::dfree(x); } void* get(size_t nbytes) { void* x = rt::dmalloc(nbytes); p.push_back(x); return x; } } tmp;
    uint64_t* d_seeds = (uint64_t*)tmp.get((size_t)n * W * 8);
    rt::h2d(d_seeds, &seed_words[first * W], (size_t)n * W * 8, s);
    uint8_t* d_seed_valid = (uint8_t*)tmp.get((size_t)n);
    rt::h2d(d_seed_valid, &seed_valid_[first], (size_t)n, s);
    const int64_t sink_lo = sink_offsets ? sink_offsets[first] : 0, sink_hi = sink_offsets ? sink_offsets[first + n] : 0;
    const int64_t nsk = sink_hi - sink_lo;
    uint64_t* d_sink_words = (uint64_t*)tmp.get((size_t)std::max<int64_t>(1, nsk) * W * 8);
    uint64_t* d_sink_keys = (uint64_t*)tmp.get((size_t)std::max<int64_t>(1, nsk) * 8);
    uint8_t* d_sink_valid = (uint8_t*)tmp.get((size_t)std::max<int64_t>(1, nsk));
    int64_t* d_sink_off = nullptr;
    if (sink_offsets) {
        std::vector<int64_t> off((size_t)n + 1);
        for (int64_t i = 0; i <= n; i++) off[i] = sink_offsets[first + i] - sink_lo;
        d_sink_off = (int64_t*)tmp.get((size_t)(n + 1) * 8);
        rt::h2d(d_sink_off, off.data(), (size_t)(n + 1) * 8, s);
        rt::stream_sync(s);    // `off` leaves scope
        if (nsk > 0) {
            rt::h2d(d_sink_words, &sink_words[sink_lo * W], (size_t)nsk * W * 8, s);
            rt::h2d(d_sink_valid, &sink_valid_[sink_lo], (size_t)nsk, s);
            const int g = grid_of(nsk, 256, 1024);
            if (sharded) {
                const int32_t* sl = sharded->d_sink_slot + sink_lo;
                switch (W) {
                    case 1: LDBG_LAUNCH(k_sink_nodes_image<1>, g, 256, s, view, (const uint64_t*)d_sink_words, (const uint8_t*)d_sink_valid, sl, nsk, d_sink_keys); break;
                    case 2: LDBG_LAUNCH(k_sink_nodes_image<2>, g, 256, s, view, (const uint64_t*)d_sink_words, (const uint8_t*)d_sink_valid, sl, nsk, d_sink_keys); break;
                    case 3: LDBG_LAUNCH(k_sink_nodes_image<3>, g, 256, s, view, (const uint64_t*)d_sink_words, (const uint8_t*)d_sink_valid, sl, nsk, d_sink_keys); break;
                    default: LDBG_LAUNCH(k_sink_nodes_image<4>, g, 256, s, view, (const uint64_t*)d_sink_words, (const uint8_t*)d_sink_valid, sl, nsk, d_sink_keys); break;
                }
            } else
            switch (W) {
                case 1: LDBG_LAUNCH(k_sink_nodes<1>, g, 256, s, view, (const uint64_t*)d_sink_words, (const uint8_t*)d_sink_valid, nsk, d_sink_keys); break;
                case 2: LDBG_LAUNCH(k_sink_nodes<2>, g, 256, s, view, (const uint64_t*)d_sink_words, (const uint8_t*)d_sink_valid, nsk, d_sink_keys); break;
                case 3: LDBG_LAUNCH(k_sink_nodes<3>, g, 256, s, view, (const uint64_t*)d_sink_words, (const uint8_t*)d_sink_valid, nsk, d_sink_keys); break;
                default: LDBG_LAUNCH(k_sink_nodes<4>, g, 256, s, view, (const uint64_t*)d_sink_words, (const uint8_t*)d_sink_valid, nsk, d_sink_keys); break;
            }
        }
    }
    uint64_t* d_term = (uint64_t*)tmp.get((size_t)ns * W * 8);
    uint32_t* d_strand_n = (uint32_t*)tmp.get((size_t)ns * 4);
    uint32_t* d_status = (uint32_t*)tmp.get((size_t)ns * 4);
    uint32_t* d_iters = (uint32_t*)tmp.get((size_t)ns * 4);
    uint8_t* d_quirk = (uint8_t*)tmp.get((size_t)ns);
    unsigned long long* d_ctr = (unsigned long long*)tmp.get(64);
    rt::dmemset(d_ctr, 0, 64, s);

    DfsArgs a;
    memset(&a, 0, sizeof(a));
    a.w.e = view;
    a.w.seeds = d_seeds;
    a.w.seed_valid = d_seed_valid;
    a.w.n_strands = ns;
    a.w.n_slots = std::min<int64_t>(n_slots_, ((ns + 63) / 64) * 64);
    // every workgroup resident: LDBG_LS_FAST x 64 x 24 B of LDS each; 194 VGPRs per lane leave 2 wavefronts per SIMD = 8 per CU
    a.w.n_slots = std::min<int64_t>(a.w.n_slots, (int64_t)std::min<size_t>(sharded ? 4 : 8, 160 * 1024 / (LDBG_LS_FAST * 64 * sizeof(LsElem))) * rt::cu_count(graph->device) * 64);
    a.w.n_slots = std::max<int64_t>(64, (a.w.n_slots / 64) * 64);
    {
        auto gcd = [](int64_t x, int64_t y) { while (y) { int64_t t = x % y; x = y; y = t; } return x; };
        int64_t stp = 7919;
        while (gcd(stp, ns) != 1) stp++;
        a.w.fetch_stride = stp % ns ? stp % ns : 1;
    }
    a.w.run_rev = cfg.direction == LDBG_DIR_BOTH || cfg.direction == LDBG_DIR_REVERSE;
    a.w.run_fwd = cfg.direction == LDBG_DIR_BOTH || cfg.direction == LDBG_DIR_FORWARD;
    a.w.next_strand = d_ctr; a.w.next_block = d_ctr + 1; a.w.vnext = d_ctr + 2;
    a.w.pool = (uint64_t*)d_pool_; a.w.n_blocks = n_blocks_;
    a.w.block_table = (uint32_t*)d_block_table_; a.w.max_blocks = max_blocks;
    a.w.strand_n = d_strand_n; a.w.status = d_status; a.w.iters = d_iters; a.w.quirk = d_quirk;
    a.w.term = d_term;
    a.w.vpool = (uint64_t*)d_vpool_; a.w.vpool_entries = vpool_entries_; a.w.vcap_max = vcap_max; a.w.vcap_init = vt_initial_entries();
    a.w.ls = (LsElem*)d_ls_; a.w.ecap = ecap_;
    a.w.strand_c = nullptr; a.w.retry = nullptr; a.w.snap = nullptr;
    a.w.unfinished = d_ctr + 4;
    a.w.yield_iters = sharded ? 128u : 0u;
    if (const char* ev = getenv("LDBG_IMG_YIELD")) a.w.yield_iters = sharded ? (uint32_t)std::max(0, atoi(ev)) : 0u;
    // unbranched stretches in one step (dfs_run_step), for the rules that allow it and over a resident table
    if (!sharded && dfs_rule_has_closed_form(view.stopper)) {
        ensure_run_index();
        if (runs_) a.w.e.runs = runs_->view;
    }
    if (sharded) {
        a.w.img_on = 1;
        a.w.img = sharded->img->view((uint64_t*)view.links.rec_of);
        a.w.seed_slot = sharded->d_seed_slot + first;
    }
    a.env.rois = rois ? rois->view : GraphView{};
    if (!rois) a.env.rois.N = -1;
    a.env.roi_bits = sharded ? nullptr : (const uint32_t*)d_roi_bits_;      // (over an image the rules ask the ROI graph itself: stoppers.h roi_bit)
    a.env.sink_keys = d_sink_keys; a.env.sink_words = d_sink_words;
    a.sink_off = d_sink_off;
    a.max_depth = dfs_max_depth;
    a.iter_limit = 1u << 26;
    if (const char* ev = getenv("LDBG_DFS_ITER_LIMIT")) a.iter_limit = (uint32_t)std::max<long long>(1, atoll(ev));
    a.n_trav = 0;
    for (int i = 0; i < cfg.n_traversal; i++) {          // LinkedHashSet: first occurrence keeps its place
        bool dup = false;
        for (int j = 0; j < a.n_trav; j++) dup |= a.trav_order[j] == (uint8_t)cfg.traversal_colors[i];
        if (!dup) a.trav_order[a.n_trav++] = (uint8_t)cfg.traversal_colors[i];
    }
    const size_t frame_bytes = (size_t)a.w.n_slots * (size_t)a.max_depth * sizeof(DfsFrame);
    if (frame_bytes > d_frames_bytes_) { rt::dfree(d_frames_); d_frames_ = rt::dmalloc(frame_bytes); d_frames_bytes_ = frame_bytes; }
    a.frames = (DfsFrame*)d_frames_;
    if (sharded) {
        const size_t one = W == 1 ? sizeof(DfsSave<1>) : (W == 2 ? sizeof(DfsSave<2>) : (W == 3 ? sizeof(DfsSave<3>) : sizeof(DfsSave<4>)));
        a.lane_save = tmp.get((size_t)a.w.n_slots * one);
        rt::dmemset(a.lane_save, 0, (size_t)a.w.n_slots * one, s);
    }

    rt::Event e0, e1;
    e0.record(s);
    const int grid = (int)(a.w.n_slots / 64);
    auto launch = [&](rt::stream_t ls) {
        if (sharded) {
            switch (W) {
                case 1: LDBG_LAUNCH((k_dfs<1, true>), grid, 64, ls, a); break;
                case 2: LDBG_LAUNCH((k_dfs<2, true>), grid, 64, ls, a); break;
                case 3: LDBG_LAUNCH((k_dfs<3, true>), grid, 64, ls, a); break;
                default: LDBG_LAUNCH((k_dfs<4, true>), grid, 64, ls, a); break;
            }
        } else {
            switch (W) {
                case 1: LDBG_LAUNCH((k_dfs<1, false>), grid, 64, ls, a); break;
                case 2: LDBG_LAUNCH((k_dfs<2, false>), grid, 64, ls, a); break;
                case 3: LDBG_LAUNCH((k_dfs<3, false>), grid, 64, ls, a); break;
                default: LDBG_LAUNCH((k_dfs<4, false>), grid, 64, ls, a); break;
            }
        }
    };
    const RunIndexView log_runs = a.w.e.runs;       // the logs of the first launch hold RUN descriptors over this index, whatever runs afterwards
    if (!sharded) {
        launch(s);
        if (a.w.e.runs.uinfo) {
            // the searches the run steps handed back (ST_RETRY_PLAIN: cases they leave to the k-mer-by-k-mer code) go round again without the
            // index — those searches only: the others keep their logs; the pool cursors carry on (fresh tables and blocks behind the used ones)
            std::vector<uint32_t> st0((size_t)ns);
            rt::d2h(st0.data(), d_status, (size_t)ns * 4, s);
            rt::stream_sync(s);
            const bool force = getenv("LDBG_DFS_FORCE_RETRY") != nullptr;      // (test hook: every search takes the second launch)
            std::vector<uint32_t> again;
            for (int64_t i = 0; i < ns; i++) if (force || st0[(size_t)i] == ST_RETRY_PLAIN) again.push_back((uint32_t)i);
            if (!again.empty()) {
                uint32_t* d_retry = (uint32_t*)tmp.get(again.size() * 4);
                rt::h2d(d_retry, again.data(), again.size() * 4, s);
                rt::dmemset(d_ctr, 0, 8, s);                           // the strand queue starts over
                a.w.e.runs = RunIndexView{nullptr, nullptr, nullptr};
                a.w.retry = d_retry;
                a.w.n_strands = (int64_t)again.size();
                dfs_retried_ += (int64_t)again.size();
                launch(s);
            }
        }
    } else {
        // bulk-synchronous rounds (image.h): every search runs until it needs a row that is not in the image; the caller's callback
        // carries the requests to their owners and the rows back, and says when no rank has a search left
        rt::stream_sync(s);
        rt::stream_t rs = sharded->stream ? sharded->stream : s;
        bool aborted = false;
        while (true) {
            rt::dmemset(d_ctr + 4, 0, 8, rs);
            sharded->img->reset_requests(rs);
            launch(rs);
            LDBG_LAUNCH(k_dfs_round_stats, 1, 64, rs, (const unsigned long long*)d_ctr, ns, (const unsigned long long*)a.w.img.n_req, sharded->d_stats);
            const int rd = sharded->round_done(sharded->user);
            if (rd == 2) aborted = true;      // the caller gives the batch up on every rank (a full image: searches that wait for a row would wait for ever)
            if (rd) break;
        }
        rt::stream_sync(rs);
        if (aborted) {
            unsigned long long c0[4] = {0, 0, 0, 0};
            rt::d2h(c0, d_ctr, 32, s);
            rt::stream_sync(s);
            vpool_dirty_ = c0[2];
            throw StatusError(LDBG_ERR_CAPACITY, "IMAGE_FULL");
        }
    }
    e1.record(s);
    std::vector<uint32_t> strand_n(ns), status(ns), iters(ns);
    rt::d2h(strand_n.data(), d_strand_n, (size_t)ns * 4, s);
    rt::d2h(status.data(), d_status, (size_t)ns * 4, s);
    rt::d2h(iters.data(), d_iters, (size_t)ns * 4, s);
    unsigned long long ctr[4] = {0, 0, 0, 0};
    rt::d2h(ctr, d_ctr, 32, s);
    rt::stream_sync(s);
    vpool_dirty_ = ctr[2];
    profile_add("dfs", rt::Event::elapsed_ms(e0, e1));

    if (getenv("LDBG_DEBUG_STATUS")) { fprintf(stderr, "[ldbg] dfs statuses:"); for (int64_t i = 0; i < ns && i < 64; i++) fprintf(stderr, " %u/%u/%u", status[i], strand_n[i], iters[i]); fprintf(stderr, " ctr %llu %llu %llu\n", ctr[0], ctr[1], ctr[2]); }
    if (getenv("LDBG_DFS_HIST")) {      // diagnostics: how the loop iterations are spread over the searches (the longest one bounds the launch)
        std::vector<uint32_t> it(iters);
        std::sort(it.begin(), it.end());
        unsigned long long tot = 0;
        for (uint32_t v : it) tot += v;
        fprintf(stderr, "[ldbg] dfs iterations: total %llu  median %u  p90 %u  p99 %u  p99.9 %u  max %u  (kernel %.1f ms)\n", tot, it[(size_t)ns / 2], it[(size_t)(ns * 0.9)],
                it[(size_t)(ns * 0.99)], it[(size_t)(ns * 0.999)], it[(size_t)ns - 1], rt::Event::elapsed_ms(e0, e1));
    }
    for (int64_t i = 0; i < ns; i++) if (status[i] == ST_POOL_FULL) return false;
    // errors the reference raises as exceptions abort the call (first seed in input order)
    for (int64_t i = 0; i < ns; i++) {
        const std::string where = " (seed " + std::to_string(first + i / 2) + ")";
        switch (status[i]) {
            case ST_NULLPTR: throw StatusError(LDBG_ERR_NULLPOINTER, "dfs dereferenced a missing record / ROI graph (NullPointerException in the reference)" + where);
            case ST_STOPPER_CONFIG: throw StatusError(LDBG_ERR_CORTEXJDK, "This stopper requires a list of novel kmers be provided." + where);
            case ST_LINKSTORE_FULL: throw StatusError(LDBG_ERR_CAPACITY, "LINKSTORE_FULL");
            case ST_DEPTH_OVERFLOW: throw StatusError(LDBG_ERR_CAPACITY, "DEPTH_OVERFLOW");
            case ST_LOG_FULL: throw StatusError(LDBG_ERR_CAPACITY, "LOG_FULL");
            case ST_TABLE_FULL: throw StatusError(LDBG_ERR_UNSUPPORTED, "dfs outgrew the per-seed visited table or its step limit (a rule that never stops on a cycle spins in the reference too)" + where);
            case ST_COPY_OVERFLOW: throw StatusError(LDBG_ERR_UNSUPPORTED, "a vertex was visited more than 32767 times in one branch chain" + where);
            case ST_MERGE_UNSUPPORTED: throw StatusError(LDBG_ERR_UNSUPPORTED, "DestinationStopper: a junction with several successful children whose graphs hold vertices without records is not supported" + where);
            default: break;
        }
    }
    for (int64_t i = 0; i < ns; i++) out.traversed += iters[i];
    const bool want_times = getenv("LDBG_DFS_TIMES") != nullptr;     // diagnostics: host phases of the call
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    auto t_phase = now();

    // dense logs -> host: the stored logs with their RUN descriptors expanded (walk.cpp: k_expand_paths)
    std::vector<uint32_t> dense_n((size_t)ns);
    {
        uint32_t* d_len = (uint32_t*)tmp.get((size_t)ns * 4);
        launch_path_lengths(d_strand_n, ns, max_blocks, d_len);
        rt::d2h(dense_n.data(), d_len, (size_t)ns * 4, s);
        rt::stream_sync(s);
    }
    std::vector<int64_t> strand_off((size_t)ns + 1, 0);
    for (int64_t i = 0; i < ns; i++) strand_off[i + 1] = strand_off[i] + dense_n[i];
    const int64_t total = strand_off[ns];
    // page-locked and kept from batch to batch: a pageable vector cost 40 ms to zero and downloaded at 5 GB/s
    if (h_log_cap_ < (size_t)std::max<int64_t>(1, total)) {
        rt::hfree_pinned(h_log_); h_log_ = nullptr; h_log_cap_ = 0;
        const size_t want = (size_t)std::max<int64_t>(1, total) + (size_t)total / 8;
        h_log_ = (uint64_t*)rt::hmalloc_pinned(want * 8);
        h_log_cap_ = want;
    }
    uint64_t* const log = h_log_;
    if (total > 0) {
        int64_t* d_off = (int64_t*)tmp.get((size_t)(ns + 1) * 8);
        uint64_t* d_dense = (uint64_t*)tmp.get((size_t)total * 8);
        rt::h2d(d_off, strand_off.data(), (size_t)(ns + 1) * 8, s);
        unsigned* d_ovf = (unsigned*)tmp.get(4);
        launch_expand_paths(d_strand_n, d_off, ns, d_dense, max_blocks, log_runs, d_ovf);
        rt::d2h(log, d_dense, (size_t)total * 8, s);
        rt::stream_sync(s);
    }

    if (want_times) { fprintf(stderr, "[ldbg] dfs host: log compaction + download %.1f ms (%lld entries)\n", ms_since(t_phase), (long long)total); t_phase = now(); }
    // replay the JGraphT container semantics (dfs(source, sinks) :64-106); seeds are independent -> host threads
    const int color = cfg.traversal_colors[0];
    const bool op_and = cfg.combination_operator == LDBG_OP_AND;
    const bool run_r = a.w.run_rev != 0, run_f = a.w.run_fwd != 0;
    const int n_threads = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), (int64_t)16, n / 64 + 1}));
    std::vector<std::vector<uint64_t>> thread_keys((size_t)n_threads);
    std::vector<std::string> thread_err((size_t)n_threads);
    const size_t seg0 = out.packed_store.size();
    out.packed_store.resize(seg0 + (size_t)n_threads);
    out.color = color;
    bool pack_flat = true;                                 // (the secondary colours below work on vertex lists)
    for (int c = 0; c < graph->hdr.C; c++) {
        bool sec = false, trav = false;
        for (int j = 0; j < cfg.n_secondary; j++) sec |= cfg.secondary_colors[j] == c;
        for (int j = 0; j < cfg.n_traversal; j++) trav |= cfg.traversal_colors[j] == c;
        if (sec && !trav) pack_flat = false;
    }
    auto assemble_range = [&](int t, int64_t lo, int64_t hi) {
        try {
            std::vector<uint64_t>& keys = thread_keys[(size_t)t];
            std::vector<uint64_t>& store = out.packed_store[(size_t)(seg0 + t)];
            for (int64_t i = lo; i < hi; i++) {
                DfsGraphHost& r = out.results[(size_t)(first + i)];
                // the common result: every direction that ran is ONE branch of vertices with records (no junction taken, nothing to merge) —
                // the graph is the two paths joined at the seed (:75-99), written straight from the logs
                {
                    bool flat = true, any_dir = false;
                    for (int d = 0; d < 2 && flat; d++) {
                        const int64_t sidx = 2 * i + d;
                        if (status[sidx] != ST_OK) continue;
                        any_dir = true;
                        const uint64_t* lg = log + strand_off[sidx];
                        const int64_t ln = (int64_t)dense_n[sidx];
                        if (ln == 0) continue;
                        flat = ln >= 3 && lg[0] == DFS_OPEN && lg[ln - 1] == DFS_CLOSE;
                        for (int64_t j = 1; flat && j < ln - 1; j++) flat = !(lg[j] & DFS_MARK) && path_idx(lg[j]) >= 0;
                    }
                    const bool null_r0 = !run_r || status[2 * i] != ST_OK, null_f0 = !run_f || status[2 * i + 1] != ST_OK;
                    if (flat && any_dir && pack_flat && !(op_and ? (null_r0 || null_f0) : (null_r0 && null_f0))) {
                        // kept packed: the vertex entries of the two branches (a branch that decided at its first vertex returns an empty graph)
                        r.is_null = false;
                        r.packed = true;
                        r.p_seg = (uint32_t)(seg0 + t);
                        r.p_off = store.size();
                        for (int d = 0; d < 2; d++) {
                            const int64_t sidx = 2 * i + d;
                            const uint32_t cnt = (status[sidx] == ST_OK && dense_n[sidx] > 3) ? dense_n[sidx] - 2 : 0u;
                            (d == 0 ? r.n_rev : r.n_fwd) = cnt;
                            if (cnt) store.insert(store.end(), log + strand_off[sidx] + 1, log + strand_off[sidx] + 1 + cnt);
                        }
                        continue;
                    }
                }
                std::unordered_map<std::string, uint32_t> null_ids;
                HGraph dir_g[2];
                bool have[2] = {false, false};
                int seed_at[2] = {-1, -1};
                for (int d = 0; d < 2; d++) {
                    const int64_t sidx = 2 * i + d;
                    if (status[sidx] != ST_OK) continue;       // ST_BRANCH_NULL: that direction returned null
                    have[d] = true;
                    if (strand_n[sidx] == 0) continue;
                    LogParser lp{log + strand_off[sidx], (int64_t)dense_n[sidx], 0, W, color, d == 1, r.null_kmers, null_ids};
                    VKey v0;
                    lp.parse_branch(dir_g[d], v0, seed_at[d]);
                }
                const bool null_r = !run_r || !have[0], null_f = !run_f || !have[1];
                r.is_null = op_and ? (null_r || null_f) : (null_r && null_f);
                if (r.is_null) continue;
                // every vertex but the seed gets index -1 (reverse) / +1 (forward) :75-83, then Graphs.addGraph of the
                // reverse and the forward graph :85-99 — the two can only share the seed (all other indices differ)
                int seed_m = -1;
                std::vector<int> map;
                for (int d = 0; d < 2; d++) {
                    if (!have[d]) continue;
                    const HGraph& g = dir_g[d];
                    map.resize(g.verts.size());
                    for (size_t v = 0; v < g.verts.size(); v++) {
                        const bool is_seed = (int)v == seed_at[d];
                        if (is_seed && seed_m >= 0) { map[v] = seed_m; continue; }
                        DfsVertex o;
                        const VKey& kv = g.verts[v];
                        o.copy = kv.copy; o.index = is_seed ? 0 : (d == 0 ? -1 : 1);
                        if (kv.id & DFS_MARK) { o.rec = -1; o.flip = 0; o.slot = -(int64_t)(kv.id & 0xFFFFFFFFull) - 1; }
                        else { o.rec = (int64_t)(kv.id >> 1) - 1; o.flip = (uint8_t)(kv.id & 1ull); o.slot = (int64_t)keys.size(); keys.push_back(kv.id); }
                        map[v] = (int)r.verts.size();
                        if (is_seed) seed_m = map[v];
                        r.verts.push_back(o);
                    }
                    for (auto& ed : g.edges) {
                        const int s2 = map[ed.src], t2 = map[ed.dst];
                        bool dup = false;
                        if (s2 == seed_m && t2 == seed_m)      // a self-loop on the seed is the only edge both directions could hold
                            for (auto& x : r.edges) dup |= x.src == s2 && x.dst == t2 && x.color == ed.color;
                        if (!dup) r.edges.push_back({s2, t2, ed.color});
                    }
                }
            }
        } catch (const std::exception& ex) { thread_err[(size_t)t] = ex.what(); }
    };
    {
        std::vector<std::thread> pool;
        const int64_t per = (n + n_threads - 1) / n_threads;
        for (int t = 1; t < n_threads; t++) pool.emplace_back(assemble_range, t, std::min<int64_t>(n, t * per), std::min<int64_t>(n, (t + 1) * per));
        assemble_range(0, 0, std::min<int64_t>(n, per));
        for (auto& th : pool) th.join();
        for (auto& er : thread_err) if (!er.empty()) throw StatusError(LDBG_ERR_HIP, er);
    }
    if (want_times) { fprintf(stderr, "[ldbg] dfs host: graph assembly %.1f ms on %d threads\n", ms_since(t_phase), n_threads); t_phase = now(); }
    // slots were numbered per thread: a result keeps the offset of its thread's keys among the keys of the batch (its vertices are
    // renumbered only where the secondary colours below need one contiguous list).  The k-mers and coverages of the vertices are
    // gathered when a result is first read (DfsBatch::materialize): the graphs themselves — record numbers, orientations,
    // copy indices, edges — are complete here.
    std::vector<int> sec_cols;
    for (int c = 0; c < graph->hdr.C; c++) {
        bool sec = false, trav = false;
        for (int j = 0; j < cfg.n_secondary; j++) sec |= cfg.secondary_colors[j] == c;
        for (int j = 0; j < cfg.n_traversal; j++) trav |= cfg.traversal_colors[j] == c;
        if (sec && !trav) sec_cols.push_back(c);
    }
    const int64_t base0 = out.n_gather;                  // keys of the chunks before this one
    std::vector<uint64_t> chunk_keys;                   // (secondary colours only) this chunk's keys in one list
    {
        const int64_t per = (n + n_threads - 1) / n_threads;
        std::vector<int64_t> tbase((size_t)n_threads + 1, 0);
        for (int t = 0; t < n_threads; t++) tbase[(size_t)t + 1] = tbase[(size_t)t] + (int64_t)thread_keys[(size_t)t].size();
        if (sec_cols.empty()) {
            for (int t = 0; t < n_threads; t++) {
                for (int64_t i = std::min<int64_t>(n, t * per); i < std::min<int64_t>(n, (t + 1) * per); i++) out.results[(size_t)(first + i)].slot_base = base0 + tbase[(size_t)t];
                out.key_segments.push_back(std::move(thread_keys[(size_t)t]));
            }
            out.n_gather += tbase[(size_t)n_threads];
        } else {
            chunk_keys.reserve((size_t)tbase[(size_t)n_threads]);
            for (int t = 0; t < n_threads; t++) {
                chunk_keys.insert(chunk_keys.end(), thread_keys[(size_t)t].begin(), thread_keys[(size_t)t].end());
                for (int64_t i = std::min<int64_t>(n, t * per); i < std::min<int64_t>(n, (t + 1) * per); i++)
                    for (auto& o : out.results[(size_t)(first + i)].verts) if (o.rec >= 0) o.slot += base0 + tbase[(size_t)t];
            }
        }
    }
    // ---- TraversalEngine.addSecondaryColors (:108-145): for every secondary colour that is not a traversal colour, the edges
    // of that colour at every vertex of the combined graph, to neighbours looked up with findRecord (here: the neighbour index)
    if (!sec_cols.empty()) {
        const int64_t ngk = (int64_t)chunk_keys.size();                      // this chunk's keys
        std::vector<uint8_t> redges((size_t)std::max<int64_t>(1, ngk) * C), rflags((size_t)std::max<int64_t>(1, ngk));
        std::vector<uint32_t> rnbr((size_t)std::max<int64_t>(1, ngk) * 8);
        if (ngk > 0) {
            uint64_t* d_keys = (uint64_t*)tmp.get((size_t)ngk * 8);
            uint8_t* d_e = (uint8_t*)tmp.get((size_t)ngk * C);
            uint8_t* d_f = (uint8_t*)tmp.get((size_t)ngk);
            uint32_t* d_n = (uint32_t*)tmp.get((size_t)ngk * 32);
            rt::h2d(d_keys, chunk_keys.data(), (size_t)ngk * 8, s);
            LDBG_LAUNCH(k_gather_rows, grid_of(ngk, 256, 4096), 256, s, graph->view, (const uint64_t*)d_keys, ngk, d_e, d_f, d_n);
            rt::d2h(redges.data(), d_e, (size_t)ngk * C, s);
            rt::d2h(rflags.data(), d_f, (size_t)ngk, s);
            rt::d2h(rnbr.data(), d_n, (size_t)ngk * 32, s);
            rt::stream_sync(s);
        }
        if (graph->hdr.k % 2 == 0) throw StatusError(LDBG_ERR_UNSUPPORTED, "dfs_batch: secondary colours with an even k-mer size are not supported");
        std::vector<uint64_t> extra_keys;                   // vertices the secondary colours add (single thread: only visualisers ask for this)
        for (int64_t i = 0; i < n; i++) {
            DfsGraphHost& r = out.results[(size_t)(first + i)];
            if (r.is_null) continue;
            HGraph M;
            std::vector<int64_t> slot_of;                   // gather slot per vertex of M
            auto key_of = [&](const DfsVertex& o) { VKey kv; kv.id = ((uint64_t)(o.rec + 1) << 1) | o.flip; kv.copy = o.copy; kv.index = o.index; return kv; };
            for (auto& o : r.verts) {
                if (o.rec < 0) throw StatusError(LDBG_ERR_NULLPOINTER, "addSecondaryColors: findRecord of a vertex returned null (seed " + std::to_string(first + i) + ")");
                M.add_vertex_new(key_of(o)); slot_of.push_back(o.slot);
            }
            for (auto& ed : r.edges) M.add_edge_new(ed.src, ed.dst, ed.color);
            const size_t nv0 = r.verts.size();
            for (int c : sec_cols) {
                HGraph g2;
                std::vector<int64_t> g2_slot;
                auto g2_add = [&](const VKey& kv, int64_t slot) {
                    const size_t before = g2.verts.size();
                    const int id = g2.add_vertex(kv);
                    if (g2.verts.size() > before) g2_slot.push_back(slot);
                    return id;
                };
                for (size_t v = 0; v < nv0; v++) {
                    const DfsVertex& o = r.verts[v];
                    const int64_t sl = o.slot - base0;
                    const bool fj = cfg.strict_java_flip && (rflags[(size_t)sl] & LDBG_ROW_HASH_COLLISION) ? false : o.flip != 0;
                    const uint32_t eb = redges[(size_t)sl * C + c], lo = eb & 0xf, hi = eb >> 4;
                    const int vi = g2_add(key_of(o), o.slot);
                    auto neighbour = [&](bool fwd, unsigned b) {
                        const uint32_t ent = rnbr[(size_t)sl * 8 + nbr_slot(fj, fwd, b)];
                        if ((ent & 0x7FFFFFFFu) == 0) throw StatusError(LDBG_ERR_UNSUPPORTED, "addSecondaryColors: a neighbour without a record is not supported yet");
                        VKey kv;
                        const bool nflip = (((ent >> 31) & 1u) != 0) != fj;
                        kv.id = ((uint64_t)(ent & 0x7FFFFFFFu) << 1) | (nflip ? 1ull : 0ull);
                        kv.copy = 0; kv.index = 0;
                        return kv;
                    };
                    // in-edges (:121-129): record orientation -> bits 3-i of the high nibble, base i; flipped -> complemented out-edges
                    for (unsigned q = 0; q < 4; q++) {
                        const bool have = !fj ? ((hi >> (3 - q)) & 1u) != 0 : ((lo >> q) & 1u) != 0;
                        if (!have) continue;
                        const unsigned b = !fj ? q : 3u - q;
                        const VKey pk = neighbour(false, b);
                        const int pi = g2_add(pk, -1);
                        if (!g2.contains_edge(pi, vi)) g2.add_edge(pi, vi, c);
                    }
                    for (unsigned q = 0; q < 4; q++) {       // out-edges (:131-139)
                        const bool have = !fj ? ((lo >> q) & 1u) != 0 : ((hi >> (3 - q)) & 1u) != 0;
                        if (!have) continue;
                        const unsigned b = !fj ? q : 3u - q;
                        const VKey nk = neighbour(true, b);
                        const int ni = g2_add(nk, -1);
                        if (!g2.contains_edge(vi, ni)) g2.add_edge(vi, ni, c);
                    }
                }
                // Graphs.addGraph(m, g2)
                std::vector<int> map(g2.verts.size());
                for (size_t j = 0; j < g2.verts.size(); j++) {
                    const size_t before = M.verts.size();
                    map[j] = M.add_vertex(g2.verts[j]);
                    if (M.verts.size() > before) {
                        int64_t sl = g2_slot[j];
                        if (sl < 0) { sl = base0 + ngk + (int64_t)extra_keys.size(); extra_keys.push_back(g2.verts[j].id); }
                        slot_of.push_back(sl);
                    }
                }
                for (auto& ed : g2.edges) M.add_edge(map[ed.src], map[ed.dst], ed.color);
            }
            r.verts.resize(M.verts.size());
            for (size_t v = nv0; v < M.verts.size(); v++) {
                DfsVertex& o = r.verts[v];
                const VKey& kv = M.verts[v];
                o.rec = (int64_t)(kv.id >> 1) - 1; o.flip = (uint8_t)(kv.id & 1ull); o.copy = kv.copy; o.index = kv.index; o.slot = slot_of[v];
            }
            r.edges = std::move(M.edges);
        }
        chunk_keys.insert(chunk_keys.end(), extra_keys.begin(), extra_keys.end());
        out.n_gather += (int64_t)chunk_keys.size();
        out.key_segments.push_back(std::move(chunk_keys));
    }
    if (want_times) fprintf(stderr, "[ldbg] dfs host: key lists + secondary colours %.1f ms\n", ms_since(t_phase));
    return true;
}

// k-mers and coverages of every vertex of the batch, gathered from the probe rows in one launch
void DfsBatch::read_packed(int64_t i, int64_t* rec, int32_t* copy_index, int32_t* index, int32_t* edge_src, int32_t* edge_dst, int32_t* edge_color) const {
    const DfsGraphHost& r = results[(size_t)i];
    const uint64_t* en = packed_store[r.p_seg].data() + r.p_off;
    int64_t v = 0, x = 0;
    int seed_m = -1;
    for (int d = 0; d < 2; d++) {
        const uint32_t cnt = d == 0 ? r.n_rev : r.n_fwd;
        int prev = -1;
        for (uint32_t j = 0; j < cnt; j++) {
            int at;
            if (j == 0 && seed_m >= 0) at = seed_m;
            else {
                const uint64_t e = en[j];
                if (rec) rec[v] = path_idx(e);
                if (copy_index) copy_index[v] = path_copy(e);
                if (index) index[v] = j == 0 ? 0 : (d == 0 ? -1 : 1);
                at = (int)v++;
                if (j == 0) seed_m = at;
            }
            if (j > 0) {
                if (edge_src) edge_src[x] = d == 1 ? prev : at;
                if (edge_dst) edge_dst[x] = d == 1 ? at : prev;
                if (edge_color) edge_color[x] = color;
                x++;
            }
            prev = at;
        }
        en += cnt;
    }
}

void DfsBatch::materialize() {
    if (materialized) return;
    materialized = true;
    rt::set_device(graph->device);
    rt::stream_t s = graph->stream;
    {   // packed results -> vertex and edge lists, their keys joining the batch's lists (one more segment per thread)
        const int64_t nres = (int64_t)results.size();
        const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), (int64_t)16, nres / 64 + 1}));
        std::vector<std::vector<uint64_t>> tk((size_t)nt);
        const int64_t per = (nres + nt - 1) / nt;
        auto unpack = [&](int t) {
            std::vector<uint64_t>& keys = tk[(size_t)t];
            std::vector<int64_t> rec;
            std::vector<int32_t> cp, ix, es, et, ec;
            for (int64_t i = std::min<int64_t>(nres, t * per); i < std::min<int64_t>(nres, (t + 1) * per); i++) {
                DfsGraphHost& r = results[(size_t)i];
                if (!r.packed) continue;
                const int64_t nv = r.n_vertices(), ne = r.n_edges();
                rec.resize((size_t)nv); cp.resize((size_t)nv); ix.resize((size_t)nv); es.resize((size_t)ne); et.resize((size_t)ne); ec.resize((size_t)ne);
                read_packed(i, rec.data(), cp.data(), ix.data(), es.data(), et.data(), ec.data());
                const uint64_t* en = packed_store[r.p_seg].data() + r.p_off;
                r.verts.resize((size_t)nv);
                r.edges.resize((size_t)ne);
                int64_t v = 0;
                for (int d = 0; d < 2; d++) {
                    const uint32_t cnt = d == 0 ? r.n_rev : r.n_fwd;
                    for (uint32_t j = (d == 1 && r.n_rev) ? 1u : 0u; j < cnt; j++, v++) {
                        DfsVertex& o = r.verts[(size_t)v];
                        o.rec = rec[(size_t)v]; o.copy = cp[(size_t)v]; o.index = ix[(size_t)v]; o.flip = path_flip(en[j]) ? 1 : 0;
                        o.slot = (int64_t)keys.size();
                        keys.push_back(((uint64_t)(o.rec + 1) << 1) | (uint64_t)o.flip);
                    }
                    en += cnt;
                }
                for (int64_t x = 0; x < ne; x++) r.edges[(size_t)x] = {es[(size_t)x], et[(size_t)x], ec[(size_t)x]};
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(unpack, t);
        unpack(0);
        for (auto& th : pool) th.join();
        for (int t = 0; t < nt; t++) {
            for (int64_t i = std::min<int64_t>(nres, t * per); i < std::min<int64_t>(nres, (t + 1) * per); i++)
                if (results[(size_t)i].packed) { results[(size_t)i].slot_base = n_gather; results[(size_t)i].packed = false; }
            n_gather += (int64_t)tk[(size_t)t].size();
            key_segments.push_back(std::move(tk[(size_t)t]));
        }
        std::vector<std::vector<uint64_t>>().swap(packed_store);
    }
    const int64_t ng = n_gather;
    std::vector<uint64_t> gw((size_t)std::max<int64_t>(1, ng) * W);
    std::vector<uint32_t> gc((size_t)std::max<int64_t>(1, ng) * C);
    if (ng > 0) {
        uint64_t* d_keys = (uint64_t*)rt::dmalloc((size_t)ng * 8);
        uint64_t* d_w = (uint64_t*)rt::dmalloc((size_t)ng * W * 8);
        uint32_t* d_c = (uint32_t*)rt::dmalloc((size_t)ng * C * 4);
        int64_t at = 0;
        for (auto& seg : key_segments) { if (!seg.empty()) rt::h2d(d_keys + at, seg.data(), seg.size() * 8, s); at += (int64_t)seg.size(); }
        const int g = grid_of(ng, 256, 4096);
        switch (W) {
            case 1: LDBG_LAUNCH(k_gather_vertices<1>, g, 256, s, graph->view, (const uint64_t*)d_keys, ng, d_w, d_c); break;
            case 2: LDBG_LAUNCH(k_gather_vertices<2>, g, 256, s, graph->view, (const uint64_t*)d_keys, ng, d_w, d_c); break;
            case 3: LDBG_LAUNCH(k_gather_vertices<3>, g, 256, s, graph->view, (const uint64_t*)d_keys, ng, d_w, d_c); break;
            default: LDBG_LAUNCH(k_gather_vertices<4>, g, 256, s, graph->view, (const uint64_t*)d_keys, ng, d_w, d_c); break;
        }
        rt::d2h(gw.data(), d_w, (size_t)ng * W * 8, s);
        rt::d2h(gc.data(), d_c, (size_t)ng * C * 4, s);
        rt::stream_sync(s);
        rt::dfree(d_keys); rt::dfree(d_w); rt::dfree(d_c);
    }
    std::vector<std::vector<uint64_t>>().swap(key_segments);
    const int64_t n = (int64_t)results.size();
    const int n_threads = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), (int64_t)16, n / 64 + 1}));
    auto fill = [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            DfsGraphHost& r = results[(size_t)i];
            r.words.resize(r.verts.size() * (size_t)W);
            r.cov.assign(r.verts.size() * (size_t)C, 0);
            for (size_t v = 0; v < r.verts.size(); v++) {
                const DfsVertex& o = r.verts[v];
                if (o.rec >= 0) {
                    for (int w = 0; w < W; w++) r.words[v * W + w] = gw[(size_t)(o.slot + r.slot_base) * W + w];
                    for (int c = 0; c < C; c++) r.cov[v * C + c] = gc[(size_t)(o.slot + r.slot_base) * C + c];
                } else {
                    const auto& nk = r.null_kmers[(size_t)(-o.slot - 1)];
                    for (int w = 0; w < W; w++) r.words[v * W + w] = nk[w];
                }
            }
        }
    };
    std::vector<std::thread> pool;
    const int64_t per = (n + n_threads - 1) / n_threads;
    for (int t = 1; t < n_threads; t++) pool.emplace_back(fill, std::min<int64_t>(n, t * per), std::min<int64_t>(n, (t + 1) * per));
    fill(0, std::min<int64_t>(n, per));
    for (auto& th : pool) th.join();
}

// TraversalUtils.toWalk(g, seed, colour) + toContig (TraversalUtils.java:367-488) over an assembled result
std::string DfsBatch::walk_contig(int64_t i, const char* seed, int color) {
    materialize();
    const DfsGraphHost& r = results.at((size_t)i);
    if (r.is_null) return std::string();
    if (color < 0 || color >= C) throw StatusError(LDBG_ERR_ARG, "colour out of range");
    std::vector<uint64_t> sw((size_t)W);
    const bool seed_ok = ascii_to_words(seed, k, sw.data(), W);
    const int nv = (int)r.verts.size();
    auto words_eq = [&](int v) { for (int w = 0; w < W; w++) if (r.words[(size_t)v * W + w] != sw[w]) return false; return true; };
    int sv = -1;
    for (int v = 0; v < nv; v++)      // :392-397 first vertex (insertion order) with the smallest copyIndex
        if (seed_ok && r.verts[v].rec >= 0 && words_eq(v) && (int32_t)r.cov[(size_t)v * C + color] > 0 &&
            (sv < 0 || r.verts[v].copy < r.verts[sv].copy)) sv = v;
    if (sv < 0) return std::string();
    std::vector<std::vector<int>> out_e((size_t)nv), in_e((size_t)nv);
    for (int e = 0; e < (int)r.edges.size(); e++) { out_e[r.edges[e].src].push_back(e); in_e[r.edges[e].dst].push_back(e); }
    std::vector<int> walk{sv}, rev_part;
    auto extend = [&](bool fwd) {
        std::unordered_set<int> seen;
        int cv = sv;
        while (cv >= 0 && !seen.count(cv)) {
            std::vector<int> nvs;
            for (int e : (fwd ? out_e[cv] : in_e[cv]))
                if (r.edges[e].color == color) nvs.push_back(fwd ? r.edges[e].dst : r.edges[e].src);
            auto self = std::find(nvs.begin(), nvs.end(), cv);     // Graphs.successorListOf ... remove(cv) :409-411
            if (self != nvs.end()) nvs.erase(self);
            int nxt = -1;
            if (nvs.size() == 1) nxt = nvs[0];
            else if (nvs.size() > 1) {
                bool same = true;
                for (size_t j = 1; j < nvs.size(); j++) {       // canonical k-mers compared pair by pair :418-422
                    if (r.verts[nvs[0]].rec < 0 || r.verts[nvs[j]].rec < 0) throw StatusError(LDBG_ERR_NULLPOINTER, "toWalk: getCanonicalKmer() on a null record");
                    if (r.verts[nvs[j]].rec != r.verts[nvs[0]].rec) { same = false; break; }
                }
                if (same) {
                    if (fwd) java_small_sort(nvs, [&](int x, int y) { return r.verts[x].copy < r.verts[y].copy ? -1 : 1; });
                    else java_small_sort(nvs, [&](int x, int y) { return r.verts[x].copy > r.verts[y].copy ? -1 : 1; });
                    nxt = nvs[0];
                }
            }
            if (nxt >= 0) { if (fwd) walk.push_back(nxt); else rev_part.push_back(nxt); seen.insert(cv); }
            cv = nxt;
        }
    };
    extend(true);
    extend(false);
    std::reverse(rev_part.begin(), rev_part.end());
    rev_part.insert(rev_part.end(), walk.begin(), walk.end());
    std::string contig, km((size_t)k, 'N');
    for (size_t j = 0; j < rev_part.size(); j++) {       // toContig :367-381: first k-mer, then last characters
        words_to_ascii(&r.words[(size_t)rev_part[j] * W], k, W, &km[0]);
        if (j == 0) contig = km; else contig.push_back(km[(size_t)k - 1]);
    }
    return contig;
}

}  // namespace ldbg
