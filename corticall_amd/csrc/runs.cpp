// Building the run index (runs.h) on the device: mutual links -> list ranking by pointer jumping -> canonical chain of
// every mirror pair -> positions by a chunked scan.  All kernels are grid-stride loops over oriented vertices
// a = 2 * record + flip and never wait for one another, so the TEST-ONLY host simulation runs them as they are.
#include "runs.h"

#include <algorithm>

namespace ldbg {

namespace {

LDBG_HOSTDEV bool run_breaker(const EngineView& e, const Node& n) {
    if (n.idx < 0) return true;
    const uint8_t fl = graph_row(e.g, n.idx)[e.g.flags_off];
    return (fl & (LDBG_ROW_HASH_COLLISION | LDBG_ROW_PALINDROME)) != 0 || (n.lflags & e.link_flag_mask) != 0 || n.npe != 0;
}
LDBG_HOSTDEV void run_node(const EngineView& e, uint32_t a, Node& n) {
    n.idx = (int32_t)(a >> 1); n.flip = (uint8_t)(a & 1u);
    n.copy = 0; n.vslot = 0; n.vent = 0; n.base = 0; n.e1 = 0; n.ent1 = 0;
    node_fill(e, n);
}
// the mutual neighbour of oriented vertex a in direction fwd, or NONE
LDBG_HOSTDEV uint32_t run_mutual(const EngineView& e, const Node& n, uint32_t a, bool fwd) {
    const uint32_t m = fwd ? n.next_mask : n.prev_mask;
    if (popc4(m) != 1) return LDBG_RUN_NONE;
    Node c;
    node_child(e, n, fwd, lowbit4(m), c);
    if (c.idx < 0 || run_breaker(e, c)) return LDBG_RUN_NONE;
    const uint32_t bm = fwd ? c.prev_mask : c.next_mask;
    if (popc4(bm) != 1) return LDBG_RUN_NONE;
    Node back;
    node_child(e, c, !fwd, lowbit4(bm), back);
    if (back.idx != n.idx || back.flip != n.flip) return LDBG_RUN_NONE;
    const uint32_t b = ((uint32_t)c.idx << 1) | (uint32_t)c.flip;
    if ((b >> 1) == (a >> 1)) return LDBG_RUN_NONE;        // a -> a and a -> rc(a): leave such vertices alone
    return b;
}

LDBG_KERNEL void k_run_links(EngineView e, int64_t n2, uint32_t* succ, uint32_t* pred) {
    for (int64_t i = global_tid(); i < n2; i += global_nthreads()) {
        const uint32_t a = (uint32_t)i;
        Node n;
        run_node(e, a, n);
        uint32_t s = LDBG_RUN_NONE, p = LDBG_RUN_NONE;
        if (!run_breaker(e, n)) { s = run_mutual(e, n, a, true); p = run_mutual(e, n, a, false); }
        succ[i] = s; pred[i] = p;
    }
}
// pd[a] = ancestor | distance << 32
LDBG_KERNEL void k_run_rank_init(int64_t n2, const uint32_t* pred, unsigned long long* pd) {
    for (int64_t i = global_tid(); i < n2; i += global_nthreads())
        pd[i] = pred[i] == LDBG_RUN_NONE ? (unsigned long long)i : ((unsigned long long)pred[i] | (1ull << 32));
}
// One round of pointer jumping, in place: (ancestor, distance) is one 8-byte word, so whatever interleaving the other
// threads produce, a pair that is read is an ancestor with its true distance, and the update keeps that true.
LDBG_KERNEL void k_run_rank_jump(int64_t n2, unsigned long long* pd, unsigned* changed) {
    bool any = false;
    for (int64_t i = global_tid(); i < n2; i += global_nthreads()) {
        const unsigned long long me = LDBG_GLOBAL(unsigned long long, pd)[i];
        const uint32_t p = (uint32_t)me;
        if (p == (uint32_t)i) continue;
        const unsigned long long up = LDBG_GLOBAL(unsigned long long, pd)[p];
        if ((uint32_t)up == p) continue;                   // p is a head
        LDBG_GLOBAL(unsigned long long, pd)[i] = (unsigned long long)(uint32_t)up | (((me >> 32) + (up >> 32)) << 32);
        any = true;
    }
    if (any) *changed = 1u;
}
// tails report their chain to its head: tail[h] = t, len[h] = distance + 1.  Members of pure cycles (their "head" still has a
// predecessor) become single vertices.
LDBG_KERNEL void k_run_tails(int64_t n2, const uint32_t* succ, const uint32_t* pred, unsigned long long* pd, uint32_t* tail, uint32_t* len) {
    for (int64_t i = global_tid(); i < n2; i += global_nthreads()) {
        const unsigned long long me = pd[i];
        const uint32_t h = (uint32_t)me;
        if (pred[h] != LDBG_RUN_NONE) continue;            // cycle member: handled by k_run_singles
        if (succ[i] == LDBG_RUN_NONE) { tail[h] = (uint32_t)i; len[h] = (uint32_t)(me >> 32) + 1u; }
    }
}
// vertices that stay alone: members of cycles and of chains that are their own mirror image (head == rc(tail))
LDBG_KERNEL void k_run_singles(int64_t n2, const uint32_t* pred, unsigned long long* pd, uint32_t* tail, uint32_t* len) {
    for (int64_t i = global_tid(); i < n2; i += global_nthreads()) {
        const uint32_t h = (uint32_t)pd[i];
        const bool cyc = pred[h] != LDBG_RUN_NONE;
        const bool self_mirror = !cyc && h == (tail[h] ^ 1u) && len[h] > 1u;
        if (cyc || self_mirror) pd[i] = (unsigned long long)i | (1ull << 63);      // bit 63: forced single
    }
}
LDBG_KERNEL void k_run_fix_singles(int64_t n2, unsigned long long* pd, uint32_t* tail, uint32_t* len) {
    for (int64_t i = global_tid(); i < n2; i += global_nthreads())
        if (pd[i] >> 63) { pd[i] = (unsigned long long)i; tail[i] = (uint32_t)i; len[i] = 1u; }
}
// cnt[a] = length of the chain headed by a if that chain is the one of its mirror pair that is laid out
LDBG_KERNEL void k_run_counts(int64_t n2, const unsigned long long* pd, const uint32_t* tail, const uint32_t* len, uint32_t* cnt) {
    for (int64_t i = global_tid(); i < n2; i += global_nthreads()) {
        const bool head = (uint32_t)pd[i] == (uint32_t)i && (pd[i] >> 32) == 0ull;
        cnt[i] = head && (uint32_t)i < (tail[i] ^ 1u) ? len[i] : 0u;
    }
}
#define RUN_SCAN_OWNERS 16384
LDBG_KERNEL void k_run_scan_sums(int64_t n2, int64_t chunk, const uint32_t* cnt, unsigned long long* sums) {
    for (int64_t t = global_tid(); t < RUN_SCAN_OWNERS; t += global_nthreads()) {
        const int64_t lo = std::min<int64_t>(t * chunk, n2), hi = std::min<int64_t>(lo + chunk, n2);
        unsigned long long s = 0;
        for (int64_t i = lo; i < hi; i++) s += cnt[i];
        sums[t] = s;
    }
}
LDBG_KERNEL void k_run_scan_top(unsigned long long* sums, unsigned long long* stats) {
    if (global_tid() != 0) return;
    unsigned long long run = 0;
    for (int t = 0; t < RUN_SCAN_OWNERS; t++) { const unsigned long long v = sums[t]; sums[t] = run; run += v; }
    stats[0] = run;
}
LDBG_KERNEL void k_run_scan_apply(int64_t n2, int64_t chunk, const uint32_t* cnt, const unsigned long long* sums, uint32_t* off, unsigned long long* stats) {
    for (int64_t t = global_tid(); t < RUN_SCAN_OWNERS; t += global_nthreads()) {
        const int64_t lo = std::min<int64_t>(t * chunk, n2), hi = std::min<int64_t>(lo + chunk, n2);
        unsigned long long run = sums[t], chains = 0, members = 0;
        for (int64_t i = lo; i < hi; i++) { off[i] = (uint32_t)run; run += cnt[i]; if (cnt[i] > 1u) { chains++; members += cnt[i]; } }
        if (chains) { atomic_add_u64(stats + 1, chains); atomic_add_u64(stats + 2, members); }
    }
}
template <int W>
LDBG_KERNEL void k_run_assign(EngineView e, int64_t n2, const unsigned long long* pd, const uint32_t* tail, const uint32_t* len, const uint32_t* off,
                              uint64_t* uinfo, uint32_t* uo, uint8_t* ubase) {
    for (int64_t i = global_tid(); i < n2; i += global_nthreads()) {
        const uint32_t h = (uint32_t)pd[i], d = (uint32_t)(pd[i] >> 32);
        if (!(h < (tail[h] ^ 1u))) continue;               // the mirror chain writes the positions
        const uint32_t L = len[h], pos = off[h] + d;
        const uint32_t piece_lo = (d / LDBG_RUN_PIECE) * LDBG_RUN_PIECE;
        const uint32_t piece_hi = std::min<uint32_t>(L, piece_lo + LDBG_RUN_PIECE) - 1u;
        const uint32_t rec = (uint32_t)i >> 1;
        const bool flip = ((uint32_t)i & 1u) != 0;
        uinfo[rec] = ui_pack(pos, d - piece_lo, piece_hi - d, flip);
        uo[pos] = rec | (flip ? 0x80000000u : 0u);
        const Kmer<W> c = graph_key<W>(e.g, rec);
        const unsigned c0 = kmer_base<W>(c, e.g.k, 0), cl = kmer_base<W>(c, e.g.k, e.g.k - 1);
        ubase[pos] = (uint8_t)(!flip ? (c0 | (cl << 2)) : ((3u - cl) | ((3u - c0) << 2)));
    }
}

// A link-flagged record is a chain of its own (run_breaker), and nothing asks for its position (every use tests ui_valid first and treats an
// invalid entry as "in no stretch", which is what a single vertex is).  Its entry is free, then, to carry what the step onto such a vertex
// needs next: the range of its junction records in the merged link table (links.h: rec_of, first | count << 32; bit 63 clear, 0 = none).
// The entry comes with the row (engine.h: node_from_entry reads both in one trip), so the link-store phase of a step starts its
// junction-record load without the rec_of trip in front of it.
LDBG_KERNEL void k_run_link_info(EngineView e, int64_t N, uint64_t* uinfo) {
    for (int64_t i = global_tid(); i < N; i += global_nthreads()) {
        const uint8_t fl = graph_row(e.g, i)[e.g.flags_off];
        if (!(fl & e.link_flag_mask)) continue;
        const uint64_t m = e.links.rec_of[i];
        uinfo[i] = m == ~0ull ? 0ull : (m & ~(1ull << 63));
    }
}

int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 16)); }

}  // namespace

RunIndex::RunIndex(const EngineView& e, int device, rt::stream_t s) {
    const int64_t N = e.g.N, n2 = 2 * N;
    if (N <= 0 || !e.g.nbr_on || n2 >= (int64_t)LDBG_RUN_NONE) return;           // no index: the walk kernel steps k-mer by k-mer
    rt::set_device(device);
    rt::Event e0, e1;
    e0.record(s);
    // Everything is allocated inside the try block (about 64 bytes per record of temporaries: 44 GB at 690 M records — the big tables are
    // where an allocation fails); a failure THERE leaves the engine without an index (the walk kernel then steps k-mer by k-mer)
    // instead of failing every walk.
    uint32_t *succ = nullptr, *pred = nullptr, *tail = nullptr, *len = nullptr, *cnt = nullptr, *off = nullptr;
    unsigned long long *pd = nullptr, *sums = nullptr, *stats = nullptr;
    auto free_tmp = [&] { rt::dfree(succ); rt::dfree(pred); rt::dfree(pd); rt::dfree(tail); rt::dfree(len); rt::dfree(cnt); rt::dfree(off); rt::dfree(sums); rt::dfree(stats); };
    bool allocating = true;
    try {
        d_uinfo_ = rt::dmalloc((size_t)N * 8);
        d_uo_ = rt::dmalloc((size_t)N * 4);
        d_ubase_ = rt::dmalloc((size_t)N);
        succ = (uint32_t*)rt::dmalloc((size_t)n2 * 4);
        pred = (uint32_t*)rt::dmalloc((size_t)n2 * 4);
        pd = (unsigned long long*)rt::dmalloc((size_t)n2 * 8);
        tail = (uint32_t*)rt::dmalloc((size_t)n2 * 4);
        len = (uint32_t*)rt::dmalloc((size_t)n2 * 4);
        cnt = (uint32_t*)rt::dmalloc((size_t)n2 * 4);
        off = (uint32_t*)rt::dmalloc((size_t)n2 * 4);
        sums = (unsigned long long*)rt::dmalloc((size_t)RUN_SCAN_OWNERS * 8);
        stats = (unsigned long long*)rt::dmalloc(64);
        allocating = false;
        const int grid = grid_for(n2);
        rt::dmemset(stats, 0, 64, s);
        rt::dmemset(tail, 0, (size_t)n2 * 4, s);
        rt::dmemset(len, 0, (size_t)n2 * 4, s);
        LDBG_LAUNCH(k_run_links, grid, 256, s, e, n2, succ, pred);
        LDBG_LAUNCH(k_run_rank_init, grid, 256, s, n2, (const uint32_t*)pred, pd);
        // a chain of L vertices is ranked after ceil(log2 L) rounds; pure cycles never settle and are cut off after enough
        // rounds for the longest possible chain
        int max_rounds = 2;
        while ((1ll << max_rounds) < n2) max_rounds++;
        unsigned* d_changed = (unsigned*)(stats + 4);
        for (int r = 0; r < max_rounds; r++) {
            rt::dmemset(d_changed, 0, 4, s);
            LDBG_LAUNCH(k_run_rank_jump, grid, 256, s, n2, pd, d_changed);
            unsigned changed = 0;
            rt::d2h(&changed, d_changed, 4, s);
            rt::stream_sync(s);
            if (!changed) break;
        }
        LDBG_LAUNCH(k_run_tails, grid, 256, s, n2, (const uint32_t*)succ, (const uint32_t*)pred, pd, tail, len);
        LDBG_LAUNCH(k_run_singles, grid, 256, s, n2, (const uint32_t*)pred, pd, tail, len);
        LDBG_LAUNCH(k_run_fix_singles, grid, 256, s, n2, pd, tail, len);
        LDBG_LAUNCH(k_run_counts, grid, 256, s, n2, (const unsigned long long*)pd, (const uint32_t*)tail, (const uint32_t*)len, cnt);
        const int64_t chunk = (n2 + RUN_SCAN_OWNERS - 1) / RUN_SCAN_OWNERS;
        LDBG_LAUNCH(k_run_scan_sums, RUN_SCAN_OWNERS / 256, 256, s, n2, chunk, (const uint32_t*)cnt, sums);
        LDBG_LAUNCH(k_run_scan_top, 1, 64, s, sums, stats);
        LDBG_LAUNCH(k_run_scan_apply, RUN_SCAN_OWNERS / 256, 256, s, n2, chunk, (const uint32_t*)cnt, (const unsigned long long*)sums, off, stats);
#define RUN_ASSIGN(WW) LDBG_LAUNCH(k_run_assign<WW>, grid, 256, s, e, n2, (const unsigned long long*)pd, (const uint32_t*)tail, (const uint32_t*)len, \
                                   (const uint32_t*)off, (uint64_t*)d_uinfo_, (uint32_t*)d_uo_, (uint8_t*)d_ubase_)
        switch (e.g.W) {
            case 1: RUN_ASSIGN(1); break;
            case 2: RUN_ASSIGN(2); break;
            case 3: RUN_ASSIGN(3); break;
            default: RUN_ASSIGN(4); break;
        }
#undef RUN_ASSIGN
        if (e.link_flag_mask && e.links.rec_of) LDBG_LAUNCH(k_run_link_info, grid_for(N), 256, s, e, N, (uint64_t*)d_uinfo_);
        unsigned long long st[4] = {0, 0, 0, 0};
        rt::d2h(st, stats, 32, s);
        e1.record(s);
        rt::stream_sync(s);
        if ((int64_t)st[0] != N) throw StatusError(LDBG_ERR_HIP, "run index: " + std::to_string(st[0]) + " positions for " + std::to_string(N) + " records");
        n_chains = (int64_t)st[1]; n_in_chains = (int64_t)st[2];
        build_ms = rt::Event::elapsed_ms(e0, e1);
    } catch (...) {
        free_tmp();
        rt::dfree(d_uinfo_); rt::dfree(d_uo_); rt::dfree(d_ubase_);
        d_uinfo_ = d_uo_ = d_ubase_ = nullptr;
        if (allocating) {
            if (getenv("LDBG_HOST_TIMES")) fprintf(stderr, "[ldbg] run index: not enough device memory for the build (%lld records); walking without it\n", (long long)N);
            return;
        }
        throw;
    }
    free_tmp();
    view.uinfo = (const uint64_t*)d_uinfo_; view.uo = (const uint32_t*)d_uo_; view.ubase = (const uint8_t*)d_ubase_;
}

RunIndex::~RunIndex() { rt::dfree(d_uinfo_); rt::dfree(d_uo_); rt::dfree(d_ubase_); }

}  // namespace ldbg
