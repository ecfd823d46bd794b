// Walks over a hash-partitioned table: the device side of the bulk-synchronous walker.
//
// Every rank holds one shard of the sorted table (graph.cpp::k_owner decides the owner of a k-mer).  A walk lives on
// the rank that was given its seed: its visited set, path and stopping rule never move.  What moves is one ROW per
// traversed k-mer: the walk asks the owner of the vertex it steps onto for that record's edges, flags and GLOBAL
// neighbour index (for each of the 8 possible neighbours: owner, record number in the owner's shard, orientation —
// a memoised, routed findRecord built once at load), and the next step needs nothing else.  One step of all walks in
// flight = one all-to-all of requests + one all-to-all of rows (corticall_amd/distributed.py drives the exchanges;
// the kernels here produce requests, serve rows and advance the walks).
//
// Scope (round 1): TraversalEngine.walk with ContigStopper and no link annotations (TraversalEngine.java:64-110,
// 356-482 with ec.getLinks().isEmpty()), odd k (no palindromic k-mers).  A walk that meets a quirk-Q6 vertex is
// reported as unsupported rather than walked differently.
#include <algorithm>

#include "engine_host.h"
#include "shard.h"

namespace ldbg {

// global record id: bits 0..39 record number in its shard + 1 (0 = no record), bits 40..47 owner, bit 63 orientation flag
LDBG_HOSTDEV uint64_t gid_make(int owner, int64_t lidx, bool flip) {
    return lidx < 0 ? 0ull : (((uint64_t)(lidx + 1)) | ((uint64_t)(uint32_t)owner << 40) | (flip ? (1ull << 63) : 0ull));
}
LDBG_HOSTDEV uint64_t gid_key(uint64_t gid) { return gid & 0xFFFFFFFFFFFFull; }      // owner + record: never 0 for a record

// ---- neighbour queries of local records: canonical neighbour k-mers for every edge any colour carries
template <int W>
LDBG_KERNEL void k_nbr_queries(GraphView g, int64_t first, int64_t n, uint64_t* words, uint8_t* flips) {
    for (int64_t t = global_tid(); t < n * 8; t += global_nthreads()) {
        const int64_t i = first + t / 8;
        const unsigned j = (unsigned)(t % 8), b = j & 3u;
        const uint8_t* row = graph_row(g, i);
        uint32_t lo = 0, hi = 0;
        for (int col = 0; col < g.C; col++) { uint32_t e = row[g.edges_off + col]; lo |= e & 0xf; hi |= e >> 4; }
        const bool have = j < 4 ? ((lo >> b) & 1u) != 0 : ((hi >> (3 - b)) & 1u) != 0;
        Kmer<W> x;
        bool f = false;
        if (have) {
            const Kmer<W> c = graph_key<W>(g, i);
            x = kmer_canonical<W>(j < 4 ? kmer_next<W>(c, g.k, b) : kmer_prev<W>(c, g.k, b), g.k, &f);
        }
        for (int w = 0; w < W; w++) words[t * W + w] = have ? kmer_word<W>(x, w) : ~0ull;
        flips[t] = f ? 1 : 0;
    }
}
LDBG_KERNEL void k_set_nbrg(uint64_t* nbrg, int64_t first, int64_t n, const int32_t* owner, const int64_t* lidx, const uint8_t* flips) {
    for (int64_t t = global_tid(); t < n * 8; t += global_nthreads())
        nbrg[first * 8 + t] = gid_make(owner[t], lidx[t], flips[t] != 0);
}
// row served to a walk: 8 global neighbour ids | flags byte | C edge bytes | padding to 8
LDBG_KERNEL void k_serve_rows(GraphView g, const uint64_t* nbrg, const int64_t* lidx, int64_t n, int rowb, uint8_t* rows) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        uint8_t* out = rows + (size_t)i * rowb;
        uint64_t* o64 = (uint64_t*)out;
        const int64_t r = lidx[i];
        for (int q = 0; q < 8; q++) o64[q] = nbrg[r * 8 + q];
        const uint8_t* row = graph_row(g, r);
        out[64] = row[g.flags_off];
        for (int c = 0; c < g.C; c++) out[65 + c] = row[g.edges_off + c];
    }
}

// ---- walker state
struct BspStrand {
    uint64_t cv;          // global id of the current vertex (0 = no record)
    uint32_t n;           // vertices in the branch graph so far
    uint32_t iters, status;
    uint8_t flip, fj, active, waiting;   // waiting: cv's row has been asked for
    uint8_t base, ended_null, pad0, pad1;
    uint32_t used;        // claimed slots of the strand's visited table
};
struct BspView {
    EngineView e;
    int64_t n_strands;
    BspStrand* st;
    uint64_t* vtab; uint32_t vcap;      // [n_strands][vcap] open addressing: gid key | copies << 48
    uint8_t* bases; uint32_t max_path;  // [n_strands][max_path] appended base per vertex (index 0 = the seed: unused)
    int rowb;
    int run_rev, run_fwd;
};
enum : uint32_t { BSP_OK = 0, BSP_NULLPTR = 1, BSP_BRANCH_NULL = 3, BSP_TABLE_FULL = 8, BSP_QUIRK = 11 };

LDBG_DEV uint32_t bsp_hash(uint64_t key) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 32); }
// slot of (gid, flip) in the strand's table; claims it when absent.  Entry: bits 0..48 key (gid key << 1 | flip), bits 49..63 count
LDBG_DEV uint64_t* bsp_slot(const BspView& v, int64_t s, uint64_t gid, bool flip, uint32_t* used) {
    const uint64_t key = (gid_key(gid) << 1) | (flip ? 1ull : 0ull);
    uint64_t* tab = v.vtab + (size_t)s * v.vcap;
    uint32_t h = bsp_hash(key) & (v.vcap - 1);
    while (true) {
        const uint64_t e = tab[h];
        if (e == 0) { tab[h] = key; (*used)++; return &tab[h]; }
        if ((e & 0x1FFFFFFFFFFFFull) == key) return &tab[h];
        h = (h + 1) & (v.vcap - 1);
    }
}

// seeds -> strands: strand 2i walks backwards from seed i, 2i+1 forwards; both first ask for the seed's row
LDBG_KERNEL void k_bsp_start(BspView v, const int32_t* owner, const int64_t* lidx, const uint8_t* flip,
                             int32_t* req_owner, int64_t* req_lidx) {
    for (int64_t s = global_tid(); s < v.n_strands; s += global_nthreads()) {
        const int64_t i = s >> 1;
        const bool fwd = (s & 1) != 0;
        BspStrand x;
        x.cv = gid_make(owner[i], lidx[i], false);
        x.flip = flip[i]; x.fj = flip[i]; x.n = 0; x.iters = 0; x.status = BSP_OK; x.active = 1; x.waiting = 0;
        x.base = 0; x.ended_null = 0; x.pad0 = x.pad1 = 0; x.used = 0;
        req_owner[s] = -1; req_lidx[s] = -1;
        if ((fwd && !v.run_fwd) || (!fwd && !v.run_rev)) { x.active = 0; x.status = BSP_BRANCH_NULL; }
        else if (x.cv == 0) {
            // a seed without a record has no neighbours: the branch decides at once (ContigStopper: adjacent != 1 -> the
            // empty graph is returned); with recruitment colours the reference dereferences the missing record (Q14)
            x.active = 0;
            x.iters = 1;
            if (v.e.recruit_mask != 0) x.status = BSP_NULLPTR;
        } else { x.waiting = 1; req_owner[s] = owner[i]; req_lidx[s] = lidx[i]; }
        v.st[s] = x;
    }
}

// one iteration of the branch loop (TraversalEngine.java:373-481, no links) for every strand whose row arrived
LDBG_KERNEL void k_bsp_step(BspView v, const uint8_t* have_row, const uint8_t* rows, int32_t* req_owner, int64_t* req_lidx) {
    for (int64_t s = global_tid(); s < v.n_strands; s += global_nthreads()) {
        req_owner[s] = -1; req_lidx[s] = -1;
        BspStrand x = v.st[s];
        if (!x.active || !x.waiting || !have_row[s]) continue;
        x.waiting = 0;
        const EngineView& e = v.e;
        const bool fwd = (s & 1) != 0;
        const uint8_t* row = rows + (size_t)s * v.rowb;
        const uint64_t* nb = (const uint64_t*)row;
        const uint8_t fl = row[64];
        // node_fill on the fetched row (engine.h): neighbour masks of cv from the traversal / recruitment colours
        bool fj = x.flip != 0;
        if (e.strict_flip && (fl & LDBG_ROW_HASH_COLLISION)) fj = false;
        if (x.flip && !fj) { x.status = BSP_QUIRK; x.active = 0; v.st[s] = x; continue; }
        uint32_t tf = 0, tr = 0, rf = 0, rr = 0;
        for (int col = 0; col < e.g.C; col++) {
            const uint32_t eb = row[65 + col];
            const uint32_t lo = eb & 0xf, hi = eb >> 4;
            const uint32_t f = !fj ? lo : hi, rn = !fj ? hi : lo;
            const uint32_t r = ((rn & 1u) << 3) | ((rn & 2u) << 1) | ((rn & 4u) >> 1) | ((rn & 8u) >> 3);
            if ((e.trav_mask >> col) & 1u) { tf |= f; tr |= r; }
            if ((e.recruit_mask >> col) & 1u) { rf |= f; rr |= r; }
        }
        const uint32_t m = fwd ? (tf ? tf : rf) : (tr ? tr : rr);
        x.iters++;
        // unvisited neighbours (:416-422)
        int adj = 0;
        uint64_t av = 0;
        bool av_flip = false;
        unsigned av_base = 0;
        for (unsigned b = 0; b < 4; b++) {
            if (!((m >> b) & 1u)) continue;
            const unsigned j = fwd ? (!fj ? b : 4u + (3u - b)) : (!fj ? 4u + b : (3u - b));
            const uint64_t g = nb[j];
            const bool cflip = ((g >> 63) != 0) != fj;
            if (gid_key(g) != 0) {
                const uint64_t en = *bsp_slot(v, s, g, cflip, &x.used);
                if ((en >> 49) > 0) continue;
            }
            adj++;
            av = g; av_flip = cflip; av_base = b;
        }
        // visited.add(cv) (:424-425)
        uint64_t* cs = bsp_slot(v, s, x.cv, x.flip != 0, &x.used);
        const bool previously = (*cs >> 49) > 0;
        if (!previously) *cs += 1ull << 49;
        const bool reached = x.n > (uint32_t)e.max_len;
        if (previously) { x.status = BSP_BRANCH_NULL; x.active = 0; x.n = 0; v.st[s] = x; continue; }
        if (adj != 1 || reached) { x.active = 0; v.st[s] = x; continue; }        // ContigStopper: the branch returns its graph
        if (x.n == 0) x.n = 1;                                                  // connectVertex adds cv, then av
        if (x.n >= v.max_path) { x.status = BSP_TABLE_FULL; x.active = 0; v.st[s] = x; continue; }
        v.bases[(size_t)s * v.max_path + x.n] = (uint8_t)av_base;
        x.n++;
        if (gid_key(av) == 0) {
            // a neighbour without a record: it has no neighbours of its own, the next iteration returns the graph
            x.iters++;
            x.active = 0; x.ended_null = 1;
            if (e.recruit_mask != 0) x.status = BSP_NULLPTR;
            v.st[s] = x;
            continue;
        }
        x.cv = av & ~(1ull << 63); x.flip = av_flip ? 1 : 0; x.fj = x.flip;
        x.waiting = 1;
        req_owner[s] = (int32_t)((av >> 40) & 0xFFu);
        req_lidx[s] = (int64_t)(av & 0xFFFFFFFFFFull) - 1;
        if ((x.used + 8) * 2 > v.vcap) { x.status = BSP_TABLE_FULL; x.active = 0; req_owner[s] = -1; req_lidx[s] = -1; }
        v.st[s] = x;
    }
}

LDBG_KERNEL void k_bsp_export(BspView v, uint32_t* strand_n, uint32_t* status, uint32_t* iters) {
    for (int64_t s = global_tid(); s < v.n_strands; s += global_nthreads()) {
        const BspStrand x = v.st[s];
        strand_n[s] = x.status == BSP_OK ? x.n : 0u;
        status[s] = x.status;
        iters[s] = x.iters;
    }
}

// ------------------------------------------------------------------ host
static int grid_of(int64_t n, int block = 256, int max_blocks = 4096) {
    int64_t b = (n + block - 1) / block;
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, max_blocks));
}

void shard_nbr_queries(const Graph& g, int64_t first, int64_t n, uint64_t* d_words, uint8_t* d_flips) {
    if (n <= 0) return;
    if (first < 0 || first + n > g.view.N) throw StatusError(LDBG_ERR_ARG, "shard_nbr_queries: record range out of bounds");
    rt::stream_t s = g.stream;
    const int grid = grid_of(n * 8);
    switch (g.view.W) {
        case 1: LDBG_LAUNCH(k_nbr_queries<1>, grid, 256, s, g.view, first, n, d_words, d_flips); break;
        case 2: LDBG_LAUNCH(k_nbr_queries<2>, grid, 256, s, g.view, first, n, d_words, d_flips); break;
        case 3: LDBG_LAUNCH(k_nbr_queries<3>, grid, 256, s, g.view, first, n, d_words, d_flips); break;
        default: LDBG_LAUNCH(k_nbr_queries<4>, grid, 256, s, g.view, first, n, d_words, d_flips); break;
    }
    rt::stream_sync(s);
}

void shard_set_nbr(Graph& g, int64_t first, int64_t n, const int32_t* d_owner, const int64_t* d_lidx, const uint8_t* d_flips) {
    if (n <= 0) return;
    if (first < 0 || first + n > g.view.N) throw StatusError(LDBG_ERR_ARG, "shard_set_nbr: record range out of bounds");
    rt::stream_t s = g.stream;
    if (!g.d_nbrg) {
        g.d_nbrg = rt::dmalloc((size_t)std::max<int64_t>(1, g.view.N) * 64);
        rt::dmemset(g.d_nbrg, 0, (size_t)std::max<int64_t>(1, g.view.N) * 64, s);
    }
    LDBG_LAUNCH(k_set_nbrg, grid_of(n * 8), 256, s, (uint64_t*)g.d_nbrg, first, n, d_owner, d_lidx, d_flips);
    rt::stream_sync(s);
}

int shard_row_bytes(const Graph& g) { return 64 + ((1 + g.hdr.C + 7) / 8) * 8; }

void shard_rows(const Graph& g, const int64_t* d_lidx, int64_t n, uint8_t* d_rows) {
    if (n <= 0) return;
    if (!g.d_nbrg) throw StatusError(LDBG_ERR_ARG, "shard_rows: the global neighbour index of this shard has not been built");
    rt::stream_t s = g.stream;
    LDBG_LAUNCH(k_serve_rows, grid_of(n), 256, s, g.view, (const uint64_t*)g.d_nbrg, d_lidx, n, shard_row_bytes(g), d_rows);
    rt::stream_sync(s);
}

struct BspWalker::Impl {
    BspView v{};
    void* d_st = nullptr; void* d_vtab = nullptr; void* d_bases = nullptr;
    int64_t cap_strands = 0;
};

BspWalker::BspWalker(const Engine& e) : eng_(e), impl_(new Impl) {
    const ldbg_engine_config& c = e.cfg;
    if (c.stopping_rule != LDBG_STOP_CONTIG || c.connect_all_neighbors || c.n_secondary > 0)
        throw StatusError(LDBG_ERR_UNSUPPORTED, "walks over a sharded table: ContigStopper without connectAllNeighbors / secondary colours");
    if (c.nlinks > 0) throw StatusError(LDBG_ERR_UNSUPPORTED, "walks over a sharded table: link annotations are not routed yet");
    if ((e.graph->hdr.k & 1) == 0) throw StatusError(LDBG_ERR_UNSUPPORTED, "walks over a sharded table: even k (palindromic k-mers) is not supported yet");
    impl_->v.e = e.view;
    impl_->v.rowb = shard_row_bytes(*e.graph);
    impl_->v.run_rev = c.direction == LDBG_DIR_BOTH || c.direction == LDBG_DIR_REVERSE;
    impl_->v.run_fwd = c.direction == LDBG_DIR_BOTH || c.direction == LDBG_DIR_FORWARD;
}
BspWalker::~BspWalker() { rt::dfree(impl_->d_st); rt::dfree(impl_->d_vtab); rt::dfree(impl_->d_bases); delete impl_; }
int BspWalker::row_bytes() const { return impl_->v.rowb; }

void BspWalker::start(int64_t n_seeds, const int32_t* d_owner, const int64_t* d_lidx, const uint8_t* d_flip, int32_t* d_req_owner, int64_t* d_req_lidx) {
    rt::set_device(eng_.graph->device);
    rt::stream_t s = eng_.graph->stream;
    BspView& v = impl_->v;
    const int64_t ns = 2 * n_seeds;
    uint64_t vcap = 64;
    while (vcap < 4ull * (uint64_t)(eng_.cfg.max_branch_length + 16)) vcap <<= 1;   // every step may also look at 3 vertices it does not take
    const uint32_t max_path = (uint32_t)eng_.cfg.max_branch_length + 4;
    if (ns > impl_->cap_strands) {
        rt::dfree(impl_->d_st); rt::dfree(impl_->d_vtab); rt::dfree(impl_->d_bases);
        impl_->d_st = rt::dmalloc((size_t)std::max<int64_t>(1, ns) * sizeof(BspStrand));
        impl_->d_vtab = rt::dmalloc((size_t)std::max<int64_t>(1, ns) * vcap * 8);
        impl_->d_bases = rt::dmalloc((size_t)std::max<int64_t>(1, ns) * max_path);
        impl_->cap_strands = ns;
    }
    v.n_strands = ns;
    v.st = (BspStrand*)impl_->d_st; v.vtab = (uint64_t*)impl_->d_vtab; v.vcap = (uint32_t)vcap;
    v.bases = (uint8_t*)impl_->d_bases; v.max_path = max_path;
    rt::dmemset(impl_->d_vtab, 0, (size_t)std::max<int64_t>(1, ns) * vcap * 8, s);
    if (ns > 0) LDBG_LAUNCH(k_bsp_start, grid_of(ns), 256, s, v, d_owner, d_lidx, d_flip, d_req_owner, d_req_lidx);
    rt::stream_sync(s);
}

void BspWalker::step(const uint8_t* d_have_row, const uint8_t* d_rows, int32_t* d_req_owner, int64_t* d_req_lidx) {
    rt::set_device(eng_.graph->device);
    rt::stream_t s = eng_.graph->stream;
    if (impl_->v.n_strands > 0) {
        rt::Event e0, e1;
        e0.record(s);
        LDBG_LAUNCH(k_bsp_step, grid_of(impl_->v.n_strands), 256, s, impl_->v, d_have_row, d_rows, d_req_owner, d_req_lidx);
        e1.record(s);
        profile_add("bsp_step", rt::Event::elapsed_ms(e0, e1));
    }
    rt::stream_sync(s);
}

void BspWalker::results(uint32_t* strand_n, uint32_t* status, uint32_t* iters, uint8_t* bases, int64_t bases_stride) {
    rt::set_device(eng_.graph->device);
    rt::stream_t s = eng_.graph->stream;
    const BspView& v = impl_->v;
    const int64_t ns = v.n_strands;
    if (ns <= 0) return;
    uint32_t* d_n = (uint32_t*)rt::dmalloc((size_t)ns * 4);
    uint32_t* d_s = (uint32_t*)rt::dmalloc((size_t)ns * 4);
    uint32_t* d_i = (uint32_t*)rt::dmalloc((size_t)ns * 4);
    LDBG_LAUNCH(k_bsp_export, grid_of(ns), 256, s, v, d_n, d_s, d_i);
    rt::d2h(strand_n, d_n, (size_t)ns * 4, s);
    rt::d2h(status, d_s, (size_t)ns * 4, s);
    rt::d2h(iters, d_i, (size_t)ns * 4, s);
    rt::stream_sync(s);
    if (bases) {
        for (int64_t i = 0; i < ns; i++) {
            const size_t cnt = std::min<size_t>((size_t)strand_n[i], (size_t)bases_stride);
            if (cnt) rt::d2h(bases + (size_t)i * bases_stride, v.bases + (size_t)i * v.max_path, cnt, s);
        }
        rt::stream_sync(s);
    }
    rt::dfree(d_n); rt::dfree(d_s); rt::dfree(d_i);
}

}  // namespace ldbg
