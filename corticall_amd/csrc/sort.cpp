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

// in_path -> out_path; returns the number of records
int64_t sort_ctx_file(const std::string& in_path, const std::string& out_path, int device) {
    if (rt::device_count() <= device) throw StatusError(LDBG_ERR_HIP, "no HIP device " + std::to_string(device) + " available (libldbg has no CPU fallback)");
    int fd = open(in_path.c_str(), O_RDONLY);
    if (fd < 0) throw StatusError(LDBG_ERR_CORTEXJDK, "Cortex graph file '" + in_path + "' cannot be opened");
    struct stat sb;
    fstat(fd, &sb);
    const size_t size = (size_t)sb.st_size;
    const uint8_t* base = size ? (const uint8_t*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    close(fd);
    if (size && base == MAP_FAILED) throw StatusError(LDBG_ERR_CORTEXJDK, "cannot map '" + in_path + "'");
    struct Unmap { const uint8_t* p; size_t n; ~Unmap() { if (p && n) munmap((void*)p, n); } } unmap{base, size};
    const CtxHeader h = parse_ctx_header(base, size, (int64_t)size, in_path);
    const int64_t n = h.num_records;
    const int W = h.W;
    const size_t rec = (size_t)h.record_size;
    if (n >= (1ll << 32)) throw StatusError(LDBG_ERR_UNSUPPORTED, "Sort: more than 2^32 records");
    std::vector<uint32_t> perm((size_t)n);
    std::iota(perm.begin(), perm.end(), 0u);
    if (n > 1) {
        rt::set_device(device);
        rt::stream_t s = rt::stream_create();
        const uint8_t* recs = base + h.data_offset;
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
            for (int64_t i = 0; i < n; i++) memcpy(&col[(size_t)i], recs + (size_t)i * rec + (size_t)w * 8, 8);
            rt::h2d(d_word, col.data(), (size_t)n * 8, s);
            const int bits = w == 0 ? 2 * h.k - 64 * (W - 1) : 64;     // the used bits of this word
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
    }
    // CortexGraphWriter: the header as it was, then the records in their new order
    FILE* f = fopen(out_path.c_str(), "wb");
    if (!f) throw StatusError(LDBG_ERR_CORTEXJDK, "cannot write '" + out_path + "'");
    bool ok = fwrite(base, 1, (size_t)h.data_offset, f) == (size_t)h.data_offset;
    std::vector<uint8_t> buf;
    buf.reserve((size_t)(1 << 16) * rec);
    for (int64_t i = 0; i < n && ok; i++) {
        const uint8_t* r = base + h.data_offset + (size_t)perm[(size_t)i] * rec;
        buf.insert(buf.end(), r, r + rec);
        if (buf.size() >= (size_t)(1 << 16) * rec || i + 1 == n) { ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size(); buf.clear(); }
    }
    ok = fclose(f) == 0 && ok;
    if (!ok) throw StatusError(LDBG_ERR_CORTEXJDK, "error while writing '" + out_path + "'");
    return n;
}

}  // namespace ldbg
