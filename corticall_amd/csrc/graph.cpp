// Device graph: streaming upload of .ctx records, layout kernels, sortedness check, radix index,
// and the two bulk DeBruijnGraph kernels (records = iteration/getRecord, find = findRecord).
#include "graph.h"
#include "links.h"

#include <fcntl.h>
#include <strings.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <mutex>

namespace ldbg {

// ------------------------------------------------------------------ profile registry
namespace {
std::mutex g_prof_mu;
std::map<std::string, std::pair<double, int64_t>> g_prof;
}  // namespace
void profile_add(const char* family, double ms) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    auto& e = g_prof[family];
    e.first += ms;
    e.second += 1;
}
void profile_reset_all() {
    std::lock_guard<std::mutex> l(g_prof_mu);
    g_prof.clear();
}
bool profile_get(const char* family, double* ms, int64_t* n) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    auto it = g_prof.find(family);
    if (it == g_prof.end()) { *ms = 0; *n = 0; return false; }
    *ms = it->second.first; *n = it->second.second;
    return true;
}

// ------------------------------------------------------------------ kernels
// One thread per record of a raw chunk (file layout: W×u64 LE key | C×u32 LE cov | C×u8 edges,
// record_size = 8W+5C bytes, unaligned).  Writes the SoA arrays and the probe row.
LDBG_KERNEL void k_layout(const uint8_t* raw, int64_t first, int64_t n, int64_t N, int W, int C, int rec_size,
                          uint64_t* keys, uint32_t* cov, uint8_t* edges, uint8_t* probe, int stride, int edges_off,
                          int cov_off) {
    // (the link-flags byte at edges_off + C is zeroed with the padding below)
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        const uint8_t* r = raw + i * rec_size;
        int64_t gi = first + i;
        uint8_t* row = probe + (size_t)gi * (size_t)stride;
        for (int w = 0; w < W; w++) {
            uint64_t v = 0;
            for (int b = 0; b < 8; b++) v |= (uint64_t)r[8 * w + b] << (8 * b);
            keys[(size_t)w * (size_t)N + (size_t)gi] = v;
            ((uint64_t*)row)[w] = v;
        }
        const uint8_t* rc = r + 8 * W;
        for (int c = 0; c < C; c++) {
            uint32_t v = (uint32_t)rc[4 * c] | ((uint32_t)rc[4 * c + 1] << 8) | ((uint32_t)rc[4 * c + 2] << 16) | ((uint32_t)rc[4 * c + 3] << 24);
            cov[(size_t)c * (size_t)N + (size_t)gi] = v;
            ((uint32_t*)(row + cov_off))[c] = v;
        }
        const uint8_t* re = rc + 4 * C;
        for (int c = 0; c < C; c++) {
            edges[(size_t)c * (size_t)N + (size_t)gi] = re[c];
            row[edges_off + c] = re[c];
        }
        for (int b = edges_off + C; b < cov_off; b++) row[b] = 0;
        for (int b = cov_off + 4 * C; b < stride; b++) row[b] = 0;
    }
}

// first index i >= 1 with key[i-1] >= key[i] (records must be strictly ascending), else ~0
LDBG_KERNEL void k_verify_sorted(const uint64_t* keys, int64_t N, int W, unsigned long long* first_bad) {
    for (int64_t i = global_tid() + 1; i < N; i += global_nthreads()) {
        int cmp = 0;
        for (int w = 0; w < W && cmp == 0; w++) {
            uint64_t a = keys[(size_t)w * N + i - 1], b = keys[(size_t)w * N + i];
            cmp = a < b ? -1 : (a > b ? 1 : 0);
        }
        if (cmp >= 0) atomic_min_u64(first_bad, (unsigned long long)i);
    }
}

template <int W>
LDBG_KERNEL void k_prefix_index(GraphView g, uint32_t* pstart) {
    const uint32_t np = 1u << (2 * g.p);
    for (int64_t i = global_tid(); i <= g.N; i += global_nthreads()) {
        // entries (prev_px, px] := i ; thread N fills the tail
        int64_t lo, hi;
        if (i == g.N) {
            lo = g.N == 0 ? 0 : (int64_t)kmer_prefix<W>(graph_key<W>(g, g.N - 1), g.k, g.p) + 1;
            hi = (int64_t)np;
        } else {
            hi = (int64_t)kmer_prefix<W>(graph_key<W>(g, i), g.k, g.p);
            lo = i == 0 ? 0 : (int64_t)kmer_prefix<W>(graph_key<W>(g, i - 1), g.k, g.p) + 1;
        }
        for (int64_t x = lo; x <= hi; x++) pstart[x] = (uint32_t)i;
    }
}

// neighbour index: for every record and every edge bit any colour carries, the record of that neighbour
// (memoised findRecord of canon[1:]+b and b+canon[:-1]); entry = (index + 1) | (neighbour flipped << 31)
template <int W>
LDBG_KERNEL void k_build_nbr(GraphView g, uint8_t* probe) {
    for (int64_t i = global_tid(); i < g.N; i += global_nthreads()) {
        uint8_t* row = probe + (size_t)i * (size_t)g.stride;
        Kmer<W> c = graph_key<W>(g, i);
        uint32_t lo = 0, hi = 0;
        for (int col = 0; col < g.C; col++) { uint32_t e = row[g.edges_off + col]; lo |= e & 0xf; hi |= e >> 4; }
        {   // orientation quirks of this record, decided once here instead of on every traversal step
            Kmer<W> rc = kmer_revcomp<W>(c, g.k);
            uint8_t fl = row[g.flags_off] & LDBG_ROW_LINK_BITS;
            if (kmer_eq<W>(rc, c)) fl |= LDBG_ROW_PALINDROME;
            else {
                uint32_t hs, hr;
                kmer_java_hash_mod32<W>(c, g.k, &hs, &hr);
                if (hs == hr && kmer_java_hash<W>(c, g.k) == kmer_java_hash<W>(rc, g.k)) fl |= LDBG_ROW_HASH_COLLISION;
            }
            row[g.flags_off] = fl;
        }
        uint32_t* nbr = (uint32_t*)(row + g.nbr_off);
        for (unsigned b = 0; b < 4; b++) {
            uint32_t ent = 0;
            if (g.nbr_on && ((lo >> b) & 1u)) {         // out-edge base b  (CortexRecord.java:252-275)
                bool f;
                Kmer<W> x = kmer_canonical<W>(kmer_next<W>(c, g.k, b), g.k, &f);
                int64_t idx = graph_find_canonical<W>(g, x);
                if (idx >= 0) ent = (uint32_t)(idx + 1) | (f ? 0x80000000u : 0u);
            }
            nbr[b] = ent;
            ent = 0;
            if (g.nbr_on && ((hi >> (3 - b)) & 1u)) {   // in-edge base b <-> bit 3-b  (:214-238)
                bool f;
                Kmer<W> x = kmer_canonical<W>(kmer_prev<W>(c, g.k, b), g.k, &f);
                int64_t idx = graph_find_canonical<W>(g, x);
                if (idx >= 0) ent = (uint32_t)(idx + 1) | (f ? 0x80000000u : 0u);
            }
            nbr[4 + b] = ent;
        }
    }
}

// Iterator<CortexRecord> / getRecord in bulk: SoA -> caller arrays (n×W, n×C, n×C)
LDBG_KERNEL void k_records(GraphView g, int64_t first, int64_t n, uint64_t* words, uint32_t* cov, uint8_t* edges) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        int64_t gi = first + i;
        for (int w = 0; w < g.W; w++) words[i * g.W + w] = g.keys[(size_t)w * g.N + gi];
        if (cov) for (int c = 0; c < g.C; c++) cov[i * g.C + c] = g.cov[(size_t)c * g.N + gi];
        if (edges) for (int c = 0; c < g.C; c++) edges[i * g.C + c] = g.edges[(size_t)c * g.N + gi];
    }
}

// findRecord in bulk: one query per lane; canonicalise in registers, radix index, block search
template <int W>
LDBG_KERNEL void k_find(GraphView g, const uint64_t* packed, const uint8_t* vmask, int64_t n, int64_t* idx_out, uint32_t* cov_out, uint8_t* edges_out) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        Kmer<W> q;
#pragma unroll
        for (int w = 0; w < W; w++) q.w[w] = packed[i * W + w];
        int64_t idx = -1;
        // a word with bits above 2k set marks "not a k-mer" (non-ACGT ASCII query, Q4)
        const int top = 2 * g.k - 64 * (W - 1);
        bool valid = (top >= 64 || (q.w[0] >> top) == 0) && (!vmask || vmask[i] != 0);
        if (valid) {
            bool f;
            Kmer<W> c = kmer_canonical<W>(q, g.k, &f);
            idx = graph_find_canonical<W>(g, c);
        }
        idx_out[i] = idx;
        if (cov_out) for (int c = 0; c < g.C; c++) cov_out[i * g.C + c] = idx >= 0 ? graph_cov(g, idx, c) : 0u;
        if (edges_out) for (int c = 0; c < g.C; c++) edges_out[i * g.C + c] = idx >= 0 ? graph_edges(g, idx, c) : (uint8_t)0;
    }
}

// owner of a k-mer in a table hash-partitioned over `world` devices: a 64-bit mix of the k-mer's MINIMIZER — the m-mer (in its own
// canonical orientation, so that a k-mer and its reverse complement agree) with the smallest mixed value, m = shard_minimizer_len(k).
// Consecutive k-mers of a walk share their minimizer more often than not (a run of about (k - m) / 2 of them), so most of a walk's
// successors live on the owner of the row that was just asked for and can be sent along with it (image.cpp: k_serve_chain) — the
// table is still spread by a hash, and the cross-shard lookups per traversed k-mer drop by that run length.
// Queries that are not k-mers (Q4) go to shard 0, where they miss like everywhere else.
LDBG_HOSTDEV uint64_t shard_mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    return x;
}
// (at most 32 bases: the m-mer windows below are one 64-bit word — k >= 97 would otherwise shift by 64 bits and more)
LDBG_HOSTDEV int shard_minimizer_len(int k) { const int m = k <= 8 ? k : (k + 2) / 3; return m > 32 ? 32 : m; }
template <int W>
LDBG_HOSTDEV uint64_t shard_minimizer_hash(const Kmer<W>& q, int k) {
    const int m = shard_minimizer_len(k);
    const uint64_t mask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
    uint64_t fw = 0, rv = 0, best = ~0ull;
    for (int j = 0; j < k; j++) {
        const int bit = 2 * (k - 1 - j);                       // base j of the k-mer, first base most significant (kmer.h)
        const uint64_t code = (kmer_word<W>(q, W - 1 - (bit >> 6)) >> (bit & 63)) & 3ull;
        fw = ((fw << 2) | code) & mask;
        rv = (rv >> 2) | ((3ull - code) << (2 * (m - 1)));
        if (j >= m - 1) {
            const uint64_t h = shard_mix64((fw < rv ? fw : rv) ^ 0x9E3779B97F4A7C15ull);
            if (h < best) best = h;
        }
    }
    return best;
}
template <int W>
LDBG_KERNEL void k_owner(int k, const uint64_t* packed, int64_t n, int world, uint64_t* canon_out, int32_t* owner_out) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        Kmer<W> q;
#pragma unroll
        for (int w = 0; w < W; w++) q.w[w] = packed[i * W + w];
        const int top = 2 * k - 64 * (W - 1);
        const bool valid = top >= 64 || (q.w[0] >> top) == 0;
        int32_t owner = 0;
        if (valid) {
            bool f;
            q = kmer_canonical<W>(q, k, &f);
            owner = (int32_t)(shard_minimizer_hash<W>(q, k) % (uint64_t)world);
        }
        if (canon_out) for (int w = 0; w < W; w++) canon_out[i * W + w] = kmer_word<W>(q, w);
        owner_out[i] = owner;
    }
}
static int grid_for(int64_t n, int block, int max_blocks);
void shard_owner_dev(int k, const uint64_t* d_packed, int64_t n, int world, uint64_t* d_canon, int32_t* d_owner, rt::stream_t s) {
    if (n <= 0) return;
    if (k <= 0 || k > 128 || world <= 0) throw StatusError(LDBG_ERR_ARG, "shard_owner: bad k or world size");
    const int W = (k + 31) / 32;
    const int grid = grid_for(n, 256, 256 * 16);
    switch (W) {
        case 1: LDBG_LAUNCH(k_owner<1>, grid, 256, s, k, d_packed, n, world, d_canon, d_owner); break;
        case 2: LDBG_LAUNCH(k_owner<2>, grid, 256, s, k, d_packed, n, world, d_canon, d_owner); break;
        case 3: LDBG_LAUNCH(k_owner<3>, grid, 256, s, k, d_packed, n, world, d_canon, d_owner); break;
        default: LDBG_LAUNCH(k_owner<4>, grid, 256, s, k, d_packed, n, world, d_canon, d_owner); break;
    }
}

// ------------------------------------------------------------------ host side
static int grid_for(int64_t n, int block = 256, int max_blocks = 256 * 8);
static int grid_for(int64_t n, int block, int max_blocks) {
    int64_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

// Records are numbered in 31 bits throughout the device structures: a neighbour-index entry is (record + 1) | flip << 31, a
// vertex is an int32 record number (engine.h: Node), a visited-table key is 34 bits wide, the run index numbers 2N oriented
// vertices in 32 bits.  A table at or beyond 2^31 - 1 records must be hash-sharded (corticall_amd.distributed) — it is refused
// here, never walked with wrapped indices.  LDBG_MAX_RECORDS (tests) lowers the limit.
int64_t max_records_per_device() {
    int64_t lim = (1LL << 31) - 2;
    if (const char* ev = getenv("LDBG_MAX_RECORDS")) lim = std::min<int64_t>(lim, atoll(ev));
    return lim;
}
void check_record_count(int64_t n, const std::string& path) {
    if (n > max_records_per_device())
        throw StatusError(LDBG_ERR_UNSUPPORTED, "Cortex graph file '" + path + "' holds " + std::to_string(n) + " records; one device table holds at most " +
                          std::to_string(max_records_per_device()) + " (hash-shard the table over several devices)");
}

Graph::Graph(const std::string& p, const void* image, int64_t nbytes, int dev) : device(dev), path(p) {
    if (rt::device_count() <= dev) throw StatusError(LDBG_ERR_HIP, "no HIP device " + std::to_string(dev) + " available (libldbg has no CPU fallback)");
    rt::set_device(dev);
    const uint8_t* base = nullptr;
    size_t size = 0;
    int fd = -1;
    void* map = nullptr;
    if (image) {
        base = (const uint8_t*)image;
        size = (size_t)nbytes;
    } else {
        fd = ::open(p.c_str(), O_RDONLY);
        if (fd < 0) throw StatusError(LDBG_ERR_CORTEXJDK, "Cortex graph file '" + p + "' not found");
        struct stat st;
        fstat(fd, &st);
        size = (size_t)st.st_size;
        if (size > 0) {
            map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map == MAP_FAILED) { ::close(fd); throw StatusError(LDBG_ERR_CORTEXJDK, "Error while parsing Cortex graph file '" + p + "': mmap failed"); }
            base = (const uint8_t*)map;
        }
    }
    try {
        hdr = parse_ctx_header(base, size, (int64_t)size, p);
        if (hdr.W > 4) throw StatusError(LDBG_ERR_UNSUPPORTED, "k > 128 is not supported (k=" + std::to_string(hdr.k) + ")");
        check_record_count(hdr.num_records, p);
        stream = rt::stream_create();
        upload(base + hdr.data_offset);
    } catch (...) {
        // ~Graph does not run for a constructor that throws: give back what upload() had taken (a rejected open — "Records are not
        // sorted", then Sort, then open again — must not leak a table's worth of HBM every time)
        release_device();
        if (map) munmap(map, size);
        if (fd >= 0) ::close(fd);
        throw;
    }
    if (map) munmap(map, size);
    if (fd >= 0) ::close(fd);
}

// header bytes on the host, n_records records in DEVICE memory (sorted, as in a file)
Graph::Graph(const std::string& p, const void* header, int64_t header_bytes, const void* d_records, int64_t n_records, int dev) : device(dev), path(p) {
    if (rt::device_count() <= dev) throw StatusError(LDBG_ERR_HIP, "no HIP device " + std::to_string(dev) + " available (libldbg has no CPU fallback)");
    rt::set_device(dev);
    try {
        CtxHeader probe = parse_ctx_header((const uint8_t*)header, (size_t)header_bytes, header_bytes, p);
        hdr = parse_ctx_header((const uint8_t*)header, (size_t)header_bytes, (int64_t)probe.data_offset + n_records * (int64_t)probe.record_size, p);
        if (hdr.W > 4) throw StatusError(LDBG_ERR_UNSUPPORTED, "k > 128 is not supported (k=" + std::to_string(hdr.k) + ")");
        if (hdr.num_records != n_records) throw StatusError(LDBG_ERR_ARG, "open_device: the header and the record count disagree");
        check_record_count(hdr.num_records, p);
        stream = rt::stream_create();
        upload((const uint8_t*)d_records, true);
    } catch (...) { release_device(); throw; }
}

Graph::Graph(const CtxHeader& h, int64_t cap, int dev, const GraphView& like, bool tiny) : device(dev), path("#image") {
    rt::set_device(dev);
    hdr = h;
    is_image = true;
    stream = rt::stream_create();
    view = like;
    view.N = cap; view.keys = nullptr; view.cov = nullptr; view.edges = nullptr; view.pstart = nullptr; view.p = 0;
    view.java_tiny = tiny ? 1 : 0; view.nbr_on = 1;
    try {
        d_probe_ = rt::dmalloc((size_t)std::max<int64_t>(1, cap) * (size_t)view.stride);
        rt::dmemset(d_probe_, 0, (size_t)std::max<int64_t>(1, cap) * (size_t)view.stride, stream);
        rt::stream_sync(stream);
    } catch (...) { release_device(); throw; }
    view.probe = (const uint8_t*)d_probe_;
}

void Graph::release_device() {
    rt::dfree(d_keys_); rt::dfree(d_cov_); rt::dfree(d_edges_); rt::dfree(d_probe_); rt::dfree(d_pstart_); rt::dfree(d_nbrg);
    d_keys_ = d_cov_ = d_edges_ = d_probe_ = d_pstart_ = nullptr; d_nbrg = nullptr;
    rt::stream_destroy(stream);
    stream = nullptr;
}

void Graph::upload(const uint8_t* recs, bool on_device) {
    const int64_t N = hdr.num_records;
    const int W = hdr.W, C = hdr.C;
    view.k = hdr.k; view.W = W; view.C = C; view.N = N;
    view.edges_off = 8 * W;
    view.flags_off = 8 * W + C;
    view.cov_off = 8 * W + ((C + 1 + 3) / 4) * 4;
    view.nbr_off = view.cov_off + 4 * C;
    view.stride = ((view.nbr_off + 32 + 15) / 16) * 16;
    view.nbr_on = N < (1LL << 31) ? 1 : 0;
    // radix index width: ~1-2 records per block for uniform k-mers, capped so the table stays cache-sized
    int p = 1;
    while (p < hdr.k && p < 13 && (1LL << (2 * (p + 1))) <= std::max<int64_t>(N, 1)) p++;
    view.p = p;
    view.java_tiny = N <= 2 ? 1 : 0;

    d_keys_ = rt::dmalloc((size_t)N * W * 8);
    d_cov_ = rt::dmalloc((size_t)N * C * 4);
    d_edges_ = rt::dmalloc((size_t)N * C);
    d_probe_ = rt::dmalloc((size_t)N * view.stride);
    d_pstart_ = rt::dmalloc(((size_t)1 << (2 * p)) * 4 + 4);
    view.keys = (const uint64_t*)d_keys_;
    view.cov = (const uint32_t*)d_cov_;
    view.edges = (const uint8_t*)d_edges_;
    view.probe = (const uint8_t*)d_probe_;
    view.pstart = (const uint32_t*)d_pstart_;

    if (on_device) {
        // the records are in device memory already (a shard cut on the device, distributed.py): the layout kernel reads them where they are
        const int64_t step = std::max<int64_t>(1, (1LL << 30) / hdr.record_size);
        for (int64_t first = 0; first < N; first += step) {
            const int64_t n = std::min(step, N - first);
            LDBG_LAUNCH(k_layout, grid_for(n), 256, stream, recs + (size_t)first * hdr.record_size, first, n, N, W, C, (int)hdr.record_size,
                        (uint64_t*)d_keys_, (uint32_t*)d_cov_, (uint8_t*)d_edges_, (uint8_t*)d_probe_, view.stride, view.edges_off, view.cov_off);
        }
        rt::stream_sync(stream);
    }
    // stream the records: pinned double buffer -> raw device chunk -> layout kernel
    const int64_t chunk_recs = std::max<int64_t>(1, (64LL << 20) / hdr.record_size);
    const size_t chunk_bytes = (size_t)chunk_recs * hdr.record_size;
    void* pin[2] = {rt::hmalloc_pinned(chunk_bytes), rt::hmalloc_pinned(chunk_bytes)};
    void* raw[2] = {rt::dmalloc(chunk_bytes), rt::dmalloc(chunk_bytes)};
    rt::stream_t s2[2] = {rt::stream_create(), rt::stream_create()};
    try {
        int b = 0;
        for (int64_t first = 0; first < N && !on_device; first += chunk_recs, b ^= 1) {
            int64_t n = std::min(chunk_recs, N - first);
            rt::stream_sync(s2[b]);   // buffer b free again
            memcpy(pin[b], recs + first * hdr.record_size, (size_t)n * hdr.record_size);
            rt::h2d(raw[b], pin[b], (size_t)n * hdr.record_size, s2[b]);
            LDBG_LAUNCH(k_layout, grid_for(n), 256, s2[b], (const uint8_t*)raw[b], first, n, N, W, C, (int)hdr.record_size,
                        (uint64_t*)d_keys_, (uint32_t*)d_cov_, (uint8_t*)d_edges_, (uint8_t*)d_probe_, view.stride,
                        view.edges_off, view.cov_off);
        }
        rt::stream_sync(s2[0]);
        rt::stream_sync(s2[1]);
    } catch (...) {
        for (int i = 0; i < 2; i++) { rt::hfree_pinned(pin[i]); rt::dfree(raw[i]); rt::stream_destroy(s2[i]); }
        throw;
    }
    for (int i = 0; i < 2; i++) { rt::hfree_pinned(pin[i]); rt::dfree(raw[i]); rt::stream_destroy(s2[i]); }

    // strictly ascending? (the reference throws "Records are not sorted" lazily, CortexGraph.java:295-301)
    unsigned long long* d_bad = (unsigned long long*)rt::dmalloc(8);
    unsigned long long bad = ~0ULL;
    rt::h2d(d_bad, &bad, 8, stream);
    LDBG_LAUNCH(k_verify_sorted, grid_for(N), 256, stream, (const uint64_t*)d_keys_, N, W, d_bad);
    rt::d2h(&bad, d_bad, 8, stream);
    rt::stream_sync(stream);
    rt::dfree(d_bad);
    if (bad != ~0ULL) {
        std::vector<uint64_t> a(W), b2(W);
        for (int w = 0; w < W; w++) {
            rt::d2h(&a[w], (const uint64_t*)d_keys_ + (size_t)w * N + bad - 1, 8, stream);
            rt::d2h(&b2[w], (const uint64_t*)d_keys_ + (size_t)w * N + bad, 8, stream);
        }
        rt::stream_sync(stream);
        std::string sa(hdr.k, 'A'), sb(hdr.k, 'A');
        words_to_ascii(a.data(), hdr.k, W, &sa[0]);
        words_to_ascii(b2.data(), hdr.k, W, &sb[0]);
        throw StatusError(LDBG_ERR_CORTEXJDK, "Records are not sorted ('" + sa + "' is found before '" + sb + "' but is lexicographically greater)");
    }
    switch (W) {
        case 1: LDBG_LAUNCH(k_prefix_index<1>, grid_for(N + 1), 256, stream, view, (uint32_t*)d_pstart_); break;
        case 2: LDBG_LAUNCH(k_prefix_index<2>, grid_for(N + 1), 256, stream, view, (uint32_t*)d_pstart_); break;
        case 3: LDBG_LAUNCH(k_prefix_index<3>, grid_for(N + 1), 256, stream, view, (uint32_t*)d_pstart_); break;
        default: LDBG_LAUNCH(k_prefix_index<4>, grid_for(N + 1), 256, stream, view, (uint32_t*)d_pstart_); break;
    }
    {
        switch (W) {
            case 1: LDBG_LAUNCH(k_build_nbr<1>, grid_for(N, 256, 256 * 16), 256, stream, view, (uint8_t*)d_probe_); break;
            case 2: LDBG_LAUNCH(k_build_nbr<2>, grid_for(N, 256, 256 * 16), 256, stream, view, (uint8_t*)d_probe_); break;
            case 3: LDBG_LAUNCH(k_build_nbr<3>, grid_for(N, 256, 256 * 16), 256, stream, view, (uint8_t*)d_probe_); break;
            default: LDBG_LAUNCH(k_build_nbr<4>, grid_for(N, 256, 256 * 16), 256, stream, view, (uint8_t*)d_probe_); break;
        }
    }
    rt::stream_sync(stream);
}

Graph::~Graph() {
    for (Links* l : bound_links) l->graph_closed();
    release_device();
}

void Graph::records_dev(int64_t first, int64_t n, uint64_t* d_words, uint32_t* d_cov, uint8_t* d_edges, rt::stream_t s) const {
    if (n <= 0) return;
    rt::Event e0, e1;
    e0.record(s);
    LDBG_LAUNCH(k_records, grid_for(n), 256, s, view, first, n, d_words, d_cov, d_edges);
    e1.record(s);
    profile_add("records", rt::Event::elapsed_ms(e0, e1));
}

void Graph::find_dev(const uint64_t* d_packed, int64_t n, int64_t* d_idx, uint32_t* d_cov, uint8_t* d_edges, rt::stream_t s, const uint8_t* d_valid) const {
    if (n <= 0) return;
    rt::Event e0, e1;
    e0.record(s);
    int grid = grid_for(n, 256, 256 * 16);
    switch (view.W) {
        case 1: LDBG_LAUNCH(k_find<1>, grid, 256, s, view, d_packed, d_valid, n, d_idx, d_cov, d_edges); break;
        case 2: LDBG_LAUNCH(k_find<2>, grid, 256, s, view, d_packed, d_valid, n, d_idx, d_cov, d_edges); break;
        case 3: LDBG_LAUNCH(k_find<3>, grid, 256, s, view, d_packed, d_valid, n, d_idx, d_cov, d_edges); break;
        default: LDBG_LAUNCH(k_find<4>, grid, 256, s, view, d_packed, d_valid, n, d_idx, d_cov, d_edges); break;
    }
    e1.record(s);
    profile_add("find", rt::Event::elapsed_ms(e0, e1));
}

int Graph::color_for_sample_name(const std::string& name) const {
    int color = -1, copies = 0;
    for (int c = 0; c < hdr.C; c++)
        if (strcasecmp(hdr.colors[c].sample_name.c_str(), name.c_str()) == 0) { color = c; copies++; }
    if (color == -1) {
        // Integer.valueOf(sampleName) fallback, CortexGraph.java:348-353
        char* end = nullptr;
        long v = strtol(name.c_str(), &end, 10);
        if (!name.empty() && end && *end == 0) { color = (int)v; copies = 1; }
    }
    return copies == 1 ? color : -1;
}

}  // namespace ldbg
