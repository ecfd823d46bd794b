// Device-resident Cortex graph: layout in HBM and the lookup primitive every kernel shares.
//
// HBM layout (DESIGN.md §Data layout).  For N records, W words per k-mer, C colours:
//   keys  [W][N] u64   structure-of-arrays, word-major (word 0 = most significant)   -- scans
//   cov   [C][N] u32   structure-of-arrays, colour-major                            -- scans / filters
//   edges [C][N] u8    structure-of-arrays, colour-major                            -- scans / filters
//   probe [N][stride]  one aligned row per record: W×u64 key | C×u8 edges | u8 link flags | pad4 | C×u32 cov | pad16
//                      -- random access: one 16/32-byte sector yields key + edges (+ coverage); bit s of the
//                      link-flags byte says "link set s (ldbg_links_open order) has a record for this k-mer",
//                      so a walk only searches a link table where there is something to find
//                      ... | 8×u32 neighbour index: record (+1, bit 31 = its flip) of the 4 successors and the
//                      4 predecessors of the canonical k-mer, filled at load time for every edge any colour
//                      has.  A walk step then needs ONE row read (key, edges, link flags, coverage and the
//                      pointers to all neighbours share a 64-byte line for k<=64, C<=3); arbitrary findRecord
//                      queries still go through the radix index + binary search below.
//   pstart[4^p + 1] u32  radix index on the first p bases: records with that prefix are
//                      [pstart[x], pstart[x+1]) -- replaces the top ~2p levels of the reference's
//                      binary search (CortexGraph.java:282-313) with one cached load; the remaining
//                      levels are a binary search over the probe rows of that block.
#pragma once
#include <string>
#include <vector>

#include "ctx_host.h"
#include "kmer.h"
#include "rt.h"

namespace ldbg {

// bits of the probe rows' flag byte
#define LDBG_ROW_LINK_BITS 0x3Fu        // bit s: link set s has a record for this k-mer (6 sets per graph)
#define LDBG_ROW_HASH_COLLISION 0x40u   // Arrays.hashCode(k-mer) == Arrays.hashCode(revcomp), k-mer != revcomp (quirk Q6)
#define LDBG_ROW_PALINDROME 0x80u       // k-mer == its reverse complement (even k only)

struct GraphView {
    int k, W, C, p;
    int64_t N;
    const uint64_t* keys;
    const uint32_t* cov;
    const uint8_t* edges;
    const uint8_t* probe;
    int stride, edges_off, flags_off, cov_off, nbr_off;
    int nbr_on;      // neighbour index built (N < 2^31)
    const uint32_t* pstart;
    int java_tiny;   // N <= 2: findRecord's loop never runs -> every lookup misses
};

// Binary search of a canonical k-mer; returns record index or -1.  CortexGraph.findRecord
// (J/utils/io/graph/cortex/CortexGraph.java:272-317) minus canonicalisation (done by callers).
template <int W>
LDBG_HOSTDEV int64_t graph_find_canonical(const GraphView& g, const Kmer<W>& q) {
    if (g.java_tiny) return -1;
    uint32_t px = kmer_prefix<W>(q, g.k, g.p);
    uint32_t lo = g.pstart[px], hi = g.pstart[px + 1];
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        const uint64_t* kp = (const uint64_t*)(g.probe + (size_t)mid * (size_t)g.stride);
        Kmer<W> m;
#pragma unroll
        for (int i = 0; i < W; i++) m.w[i] = kp[i];
        int c = kmer_cmp<W>(m, q);
        if (c == 0) return (int64_t)mid;
        if (c < 0) lo = mid + 1; else hi = mid;
    }
    return -1;
}
LDBG_HOSTDEV const uint8_t* graph_row(const GraphView& g, int64_t idx) { return g.probe + (size_t)idx * (size_t)g.stride; }
LDBG_HOSTDEV uint8_t graph_edges(const GraphView& g, int64_t idx, int c) { return graph_row(g, idx)[g.edges_off + c]; }
LDBG_HOSTDEV uint8_t graph_link_flags(const GraphView& g, int64_t idx) { return graph_row(g, idx)[g.flags_off]; }
LDBG_HOSTDEV uint32_t graph_nbr(const GraphView& g, int64_t idx, int j) {
    return ((const uint32_t*)(graph_row(g, idx) + g.nbr_off))[j];
}
LDBG_HOSTDEV uint32_t graph_cov(const GraphView& g, int64_t idx, int c) {
    return ((const uint32_t*)(graph_row(g, idx) + g.cov_off))[c];
}
template <int W>
LDBG_HOSTDEV Kmer<W> graph_key(const GraphView& g, int64_t idx) {
    const uint64_t* kp = (const uint64_t*)graph_row(g, idx);
    Kmer<W> m;
#pragma unroll
    for (int i = 0; i < W; i++) m.w[i] = kp[i];
    return m;
}

class Graph {
public:
    // file_or_image: if image != nullptr the graph is built from memory, else `path` is mmapped
    Graph(const std::string& path, const void* image, int64_t nbytes, int device);
    Graph(const std::string& path, const void* header, int64_t header_bytes, const void* d_records, int64_t n_records, int device);
    // the local image of a hash-sharded table (image.h): `cap` zeroed rows laid out like those of `like`, filled as rows arrive
    Graph(const CtxHeader& h, int64_t cap, int device, const GraphView& like, bool java_tiny);
    bool is_image = false;
    ~Graph();
    CtxHeader hdr;
    int device = 0;
    GraphView view{};
    rt::stream_t stream = nullptr;
    std::string path;

    void records_dev(int64_t first, int64_t n, uint64_t* d_words, uint32_t* d_cov, uint8_t* d_edges, rt::stream_t s) const;
    // d_valid (optional): per query 0 = not a k-mer over ACGT -> miss (Q4).  Without it a query whose words carry bits above 2k is
    // taken as "not a k-mer" (a convention callers of the packed entry points may use when k is not a multiple of 32)
    void find_dev(const uint64_t* d_packed, int64_t n, int64_t* d_idx, uint32_t* d_cov, uint8_t* d_edges, rt::stream_t s, const uint8_t* d_valid = nullptr) const;
    int color_for_sample_name(const std::string& name) const;
    // link sets bound to this graph get a flag bit in the probe rows (at most 8)
    mutable uint32_t link_slots = 0;                 // bits of the flag byte in use
    mutable std::vector<class Links*> bound_links;   // the sets holding those bits; told when the graph goes first (a
                                                     // garbage collector closes handles in any order)
    uint8_t* probe_mutable() const { return (uint8_t*)d_probe_; }
    void* d_nbrg = nullptr;   // shard of a partitioned table: global neighbour index [N][8] u64 (shard.cpp)

private:
    void* d_keys_ = nullptr; void* d_cov_ = nullptr; void* d_edges_ = nullptr; void* d_probe_ = nullptr; void* d_pstart_ = nullptr;
    void upload(const uint8_t* records, bool on_device = false);
    void release_device();
};

int64_t max_records_per_device();
void check_record_count(int64_t n, const std::string& path);

// hash partitioning of the table over devices (graph.cpp)
void shard_owner_dev(int k, const uint64_t* d_packed, int64_t n, int world, uint64_t* d_canon, int32_t* d_owner, rt::stream_t s);

// timing registry for bench.py (ldbg_profile_get)
void profile_add(const char* family, double ms);

}  // namespace ldbg
