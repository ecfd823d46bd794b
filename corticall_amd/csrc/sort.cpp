// Sort (J/commands/utils/Sort.java:20-49): the records of a Cortex graph file in k-mer order — the step that produces the
// sorted table the rest of this library requires.  The reference loads every record as an object and calls Arrays.sort
// with CortexRecord.compareTo = String.compareTo of the k-mers (CortexRecord.java:210-212; a stable merge sort).  Here the
// packed k-mer words (word 0 most significant: the same order) go through a least-significant-digit radix sort on the
// device that yields the permutation; the 8W + 5C byte records themselves are moved once, by the writer.
//
// One pass = 4 bits of one word: every thread owns a contiguous chunk of the current order (that is what keeps the sort
// stable), counts its 16 digits in registers, a single-workgroup scan turns the [digit][thread] counts into offsets, and
// the thread scatters its chunk.  Passes in which all keys share the digit (the unused high bits of word 0) are skipped.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <functional>
#include <memory>
#include <numeric>
#include <vector>

#include "ctx_host.h"
#include "graph.h"
#include "rt.h"

namespace ldbg {

#define SORT_THREADS 16384          // 64 workgroups x 256 lanes

LDBG_KERNEL void k_radix_count(const uint64_t* word, const uint32_t* perm, int64_t n, int shift, int64_t chunk, uint32_t* counts) {
    for (int64_t t = global_tid(); t < SORT_THREADS; t += global_nthreads()) {      // SORT_THREADS chunk owners
        uint32_t c[16];
#pragma unroll
        for (int d = 0; d < 16; d++) c[d] = 0;
        const int64_t lo = t * chunk < n ? t * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
        for (int64_t p = lo; p < hi; p++) {
            const unsigned dg = (unsigned)((word[perm[p]] >> shift) & 15ull);
#pragma unroll
            for (int d = 0; d < 16; d++) c[d] += dg == (unsigned)d ? 1u : 0u;
        }
#pragma unroll
        for (int d = 0; d < 16; d++) counts[(size_t)d * SORT_THREADS + t] = c[d];
    }
}
// exclusive scan of counts in (digit, thread) order, in place; totals[d] = number of keys with digit d
LDBG_KERNEL void k_radix_scan(uint32_t* counts, uint32_t* totals) {
    if (global_tid() != 0) return;          // 262,144 additions: not worth a parallel scan next to the passes over the keys
    uint32_t run = 0;
    for (int d = 0; d < 16; d++) {
        uint32_t tot = 0;
        for (int64_t t = 0; t < SORT_THREADS; t++) {
            const uint32_t v = counts[(size_t)d * SORT_THREADS + t];
            counts[(size_t)d * SORT_THREADS + t] = run;
            run += v; tot += v;
        }
        totals[d] = tot;
    }
}
LDBG_KERNEL void k_radix_scatter(const uint64_t* word, const uint32_t* perm_in, uint32_t* perm_out, int64_t n, int shift, int64_t chunk,
                                 const uint32_t* offsets) {
    for (int64_t t = global_tid(); t < SORT_THREADS; t += global_nthreads()) {
        uint32_t o[16];
#pragma unroll
        for (int d = 0; d < 16; d++) o[d] = offsets[(size_t)d * SORT_THREADS + t];
        const int64_t lo = t * chunk < n ? t * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
        for (int64_t p = lo; p < hi; p++) {
            const uint32_t item = perm_in[p];
            const unsigned dg = (unsigned)((word[item] >> shift) & 15ull);
            uint32_t dst = 0;
#pragma unroll
            for (int d = 0; d < 16; d++) { const bool m = dg == (unsigned)d; dst = m ? o[d] : dst; o[d] += m ? 1u : 0u; }
            perm_out[dst] = item;
        }
    }
}

// stable permutation that sorts n keys of W words (word 0 most significant, `used_bits_word0` significant bits in word 0);
// fill(w, col) writes word w of every key into col[n]
static std::vector<uint32_t> radix_sort_permutation(int64_t n, int W, int used_bits_word0, int device,
                                                    const std::function<void(int, uint64_t*)>& fill) {
    if (n >= (1ll << 32)) throw StatusError(LDBG_ERR_UNSUPPORTED, "Sort: more than 2^32 records");
    std::vector<uint32_t> perm((size_t)n);
    std::iota(perm.begin(), perm.end(), 0u);
    if (n <= 1) return perm;
    rt::set_device(device);
    rt::stream_t s = rt::stream_create();
    uint64_t* d_word = (uint64_t*)rt::dmalloc((size_t)n * 8);
    uint32_t* d_a = (uint32_t*)rt::dmalloc((size_t)n * 4);
    uint32_t* d_b = (uint32_t*)rt::dmalloc((size_t)n * 4);
    uint32_t* d_counts = (uint32_t*)rt::dmalloc((size_t)16 * SORT_THREADS * 4);
    uint32_t* d_tot = (uint32_t*)rt::dmalloc(64);
    rt::h2d(d_a, perm.data(), (size_t)n * 4, s);
    const int64_t chunk = (n + SORT_THREADS - 1) / SORT_THREADS;
    std::vector<uint64_t> col((size_t)n);
    rt::Event e0, e1;
    double dev_ms = 0;
    for (int w = W - 1; w >= 0; w--) {                  // least significant word first
        fill(w, col.data());
        rt::h2d(d_word, col.data(), (size_t)n * 8, s);
        const int bits = w == 0 ? used_bits_word0 : 64;
        e0.record(s);
        for (int shift = 0; shift < bits; shift += 4) {
            LDBG_LAUNCH(k_radix_count, SORT_THREADS / 256, 256, s, (const uint64_t*)d_word, (const uint32_t*)d_a, n, shift, chunk, d_counts);
            LDBG_LAUNCH(k_radix_scan, 1, 64, s, d_counts, d_tot);
            uint32_t tot[16];
            rt::d2h(tot, d_tot, 64, s);
            rt::stream_sync(s);
            bool uniform = false;
            for (int d = 0; d < 16; d++) uniform |= (int64_t)tot[d] == n;
            if (uniform) continue;
            LDBG_LAUNCH(k_radix_scatter, SORT_THREADS / 256, 256, s, (const uint64_t*)d_word, (const uint32_t*)d_a, d_b, n, shift, chunk, (const uint32_t*)d_counts);
            std::swap(d_a, d_b);
        }
        e1.record(s);
        dev_ms += rt::Event::elapsed_ms(e0, e1);
    }
    profile_add("sort", dev_ms);
    rt::d2h(perm.data(), d_a, (size_t)n * 4, s);
    rt::stream_sync(s);
    rt::dfree(d_word); rt::dfree(d_a); rt::dfree(d_b); rt::dfree(d_counts); rt::dfree(d_tot);
    rt::stream_destroy(s);
    return perm;
}

struct MappedCtx {
    const uint8_t* base = nullptr;
    size_t size = 0;
    CtxHeader h;
    explicit MappedCtx(const std::string& path) {
        int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw StatusError(LDBG_ERR_CORTEXJDK, "Cortex graph file '" + path + "' cannot be opened");
        struct stat sb;
        fstat(fd, &sb);
        size = (size_t)sb.st_size;
        base = size ? (const uint8_t*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
        close(fd);
        if (size && base == MAP_FAILED) { base = nullptr; throw StatusError(LDBG_ERR_CORTEXJDK, "cannot map '" + path + "'"); }
        h = parse_ctx_header(base, size, (int64_t)size, path);
    }
    ~MappedCtx() { if (base && size) munmap((void*)base, size); }
    MappedCtx(const MappedCtx&) = delete;
    const uint8_t* record(int64_t i) const { return base + h.data_offset + (size_t)i * (size_t)h.record_size; }
};

static void write_all(FILE* f, const std::vector<uint8_t>& b, bool& ok) { if (ok && !b.empty()) ok = fwrite(b.data(), 1, b.size(), f) == b.size(); }

// Sort: in_path -> out_path; returns the number of records
int64_t sort_ctx_file(const std::string& in_path, const std::string& out_path, int device) {
    if (rt::device_count() <= device) throw StatusError(LDBG_ERR_HIP, "no HIP device " + std::to_string(device) + " available (libldbg has no CPU fallback)");
    MappedCtx in(in_path);
    const CtxHeader& h = in.h;
    const int64_t n = h.num_records;
    const size_t rec = (size_t)h.record_size;
    const std::vector<uint32_t> perm = radix_sort_permutation(n, h.W, 2 * h.k - 64 * (h.W - 1), device, [&](int w, uint64_t* col) {
        for (int64_t i = 0; i < n; i++) memcpy(&col[i], in.record(i) + (size_t)w * 8, 8);
    });
    // CortexGraphWriter: the header re-serialised from its parsed values (not a byte copy), then the records in their new order
    FILE* f = fopen(out_path.c_str(), "wb");
    if (!f) throw StatusError(LDBG_ERR_CORTEXJDK, "cannot write '" + out_path + "'");
    bool ok = true;
    write_all(f, serialize_ctx_header(h), ok);
    std::vector<uint8_t> buf;
    buf.reserve((size_t)(1 << 16) * rec);
    for (int64_t i = 0; i < n && ok; i++) {
        const uint8_t* r = in.record(perm[(size_t)i]);
        buf.insert(buf.end(), r, r + rec);
        if (buf.size() >= (size_t)(1 << 16) * rec || i + 1 == n) { write_all(f, buf, ok); buf.clear(); }
    }
    ok = fclose(f) == 0 && ok;
    if (!ok) throw StatusError(LDBG_ERR_CORTEXJDK, "error while writing '" + out_path + "'");
    return n;
}

// CortexGraphWriter over a selection of a graph's records (what FindTips.java:112-131 and the other filters write): the header
// re-serialised from its parsed values, then the chosen records in the order given
int64_t subset_ctx_file(const std::string& in_path, const int64_t* indices, int64_t n, const std::string& out_path) {
    MappedCtx in(in_path);
    const CtxHeader& h = in.h;
    const size_t rec = (size_t)h.record_size;
    for (int64_t i = 0; i < n; i++)
        if (indices[i] < 0 || indices[i] >= h.num_records) throw StatusError(LDBG_ERR_ARG, "record " + std::to_string(indices[i]) + " is not in '" + in_path + "'");
    FILE* f = fopen(out_path.c_str(), "wb");
    if (!f) throw StatusError(LDBG_ERR_CORTEXJDK, "cannot write '" + out_path + "'");
    bool ok = true;
    write_all(f, serialize_ctx_header(h), ok);
    std::vector<uint8_t> buf;
    buf.reserve((size_t)(1 << 16) * rec);
    for (int64_t i = 0; i < n && ok; i++) {
        const uint8_t* r = in.record(indices[i]);
        buf.insert(buf.end(), r, r + rec);
        if (buf.size() >= (size_t)(1 << 16) * rec || i + 1 == n) { write_all(f, buf, ok); buf.clear(); }
    }
    ok = fclose(f) == 0 && ok;
    if (!ok) throw StatusError(LDBG_ERR_CORTEXJDK, "error while writing '" + out_path + "'");
    return n;
}

// Join (J/commands/utils/Join.java:16-60 over CortexCollection, J/utils/io/graph/cortex/CortexCollection.java:34-58, 218-293):
// the union of the k-mers of several sorted graphs, each graph's colours side by side (a k-mer missing from a graph has
// coverage 0 and no edges there).  The reference merges the files' iterators head by head; here the keys of all files are
// concatenated and go through the same stable device sort (ties stay in file order), then equal neighbours are folded.
// emit(bytes, n): the joined graph, header first.  find_view: the graphs as CortexCollection.findRecord sees them (:160-188) — one
// findRecord per member graph, which never finds anything in a graph of two records or fewer, so such a member's records stay out
static int64_t join_ctx(const std::vector<std::string>& paths, int device, bool find_view, const std::function<void(const uint8_t*, size_t)>& emit) {
    if (rt::device_count() <= device) throw StatusError(LDBG_ERR_HIP, "no HIP device " + std::to_string(device) + " available (libldbg has no CPU fallback)");
    if (paths.empty()) throw StatusError(LDBG_ERR_ARG, "Join: no graphs");
    std::vector<std::unique_ptr<MappedCtx>> in;
    for (auto& p : paths) in.emplace_back(new MappedCtx(p));
    CtxHeader out_h;
    out_h.version = 6; out_h.k = in[0]->h.k; out_h.W = in[0]->h.W; out_h.C = 0;
    std::vector<int64_t> first_rec{0};
    std::vector<int> first_col;
    for (size_t g = 0; g < in.size(); g++) {
        const CtxHeader& h = in[g]->h;
        if (h.k != out_h.k)
            throw StatusError(LDBG_ERR_CORTEXJDK, "Graph kmer sizes are not equal.  Expected k=" + std::to_string(out_h.k) + ", but found k=" +
                                                      std::to_string(h.k) + " in graph " + paths[g]);
        first_col.push_back(out_h.C);
        out_h.C += h.C;
        out_h.colors.insert(out_h.colors.end(), h.colors.begin(), h.colors.end());
        first_rec.push_back(first_rec.back() + ((find_view && h.num_records <= 2) ? 0 : h.num_records));
    }
    const int64_t n = first_rec.back();
    const int W = out_h.W, C = out_h.C;
    auto locate = [&](int64_t i, size_t* g) { size_t x = 0; while (i >= first_rec[x + 1]) x++; *g = x; return i - first_rec[x]; };
    std::vector<uint32_t> perm;
    if (n > 0) perm = radix_sort_permutation(n, W, 2 * out_h.k - 64 * (W - 1), device, [&](int w, uint64_t* col) {
        for (size_t g = 0; g < in.size(); g++)
            for (int64_t i = 0; i < first_rec[g + 1] - first_rec[g]; i++) memcpy(&col[first_rec[g] + i], in[g]->record(i) + (size_t)w * 8, 8);
    });
    const std::vector<uint8_t> hb = serialize_ctx_header(out_h);
    emit(hb.data(), hb.size());
    const size_t rec = (size_t)(8 * W + 5 * C);
    std::vector<uint8_t> buf, cur(rec);
    int64_t n_out = 0;
    for (int64_t i = 0; i < n;) {
        size_t g;
        int64_t li = locate((int64_t)perm[(size_t)i], &g);
        const uint8_t* key = in[g]->record(li);
        std::fill(cur.begin(), cur.end(), 0);
        memcpy(cur.data(), key, (size_t)8 * W);
        int64_t j = i;
        for (; j < n; j++) {                               // every file's record of this k-mer (CortexCollection.next :241-283)
            size_t g2;
            const int64_t l2 = locate((int64_t)perm[(size_t)j], &g2);
            const uint8_t* r = in[g2]->record(l2);
            if (memcmp(r, key, (size_t)8 * W) != 0) break;
            const int Cg = in[g2]->h.C;
            memcpy(cur.data() + 8 * W + 4 * first_col[g2], r + 8 * W, (size_t)4 * Cg);
            memcpy(cur.data() + 8 * W + 4 * C + first_col[g2], r + 8 * W + 4 * Cg, (size_t)Cg);
        }
        buf.insert(buf.end(), cur.begin(), cur.end());
        n_out++;
        i = j;
        if (buf.size() >= ((size_t)1 << 16) * rec) { emit(buf.data(), buf.size()); buf.clear(); }
    }
    if (!buf.empty()) emit(buf.data(), buf.size());
    return n_out;
}

int64_t join_ctx_files(const std::vector<std::string>& paths, const std::string& out_path, int device) {
    FILE* f = nullptr;
    bool ok = true;
    int64_t n_out = 0;
    try {
        n_out = join_ctx(paths, device, false, [&](const uint8_t* b, size_t nb) {
            if (!f) { f = fopen(out_path.c_str(), "wb"); if (!f) throw StatusError(LDBG_ERR_CORTEXJDK, "cannot write '" + out_path + "'"); }
            ok = ok && fwrite(b, 1, nb, f) == nb;
        });
    } catch (...) { if (f) fclose(f); throw; }
    ok = f && fclose(f) == 0 && ok;
    if (!ok) throw StatusError(LDBG_ERR_CORTEXJDK, "error while writing '" + out_path + "'");
    return n_out;
}

// the joined graph as a file image in memory: what a CortexCollection over the graphs presents (ldbg_graph_open_collection)
std::vector<uint8_t> join_ctx_image(const std::vector<std::string>& paths, int device, bool find_view) {
    std::vector<uint8_t> img;
    join_ctx(paths, device, find_view, [&](const uint8_t* b, size_t nb) { img.insert(img.end(), b, b + nb); });
    return img;
}

}  // namespace ldbg
