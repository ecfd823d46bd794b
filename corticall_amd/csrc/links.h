// Link annotations (.ctp.gz) on the device.
// Reference: CortexLinksIterable (J/utils/io/graph/links/CortexLinksIterable.java:49-226) parses the
// text; CortexLinksMap (CortexLinksMap.java:22-43) keys the records by canonical binary k-mer.
// HBM layout: sorted canonical keys [M][W] u64 + radix index, CSR offsets into junction records,
// junction records stored per k-mer in the iteration order of the reference's
// java.util.HashSet<CortexJunctionsRecord> (CortexLinksRecord.java:13-21), junction bases as
// codes 0..3 in one byte pool.
#pragma once
#include <string>
#include <vector>

#include "graph.h"

namespace ldbg {

struct JuncRec {
    uint32_t str_off;
    uint32_t len;
    int32_t hash_asis;   // java.lang.String.hashCode of the junction string
    int32_t hash_comp;   // ... of its complement (LinkStore.add, LinkStore.java:25)
    uint32_t is_fw;      // bit 0: "the link goes forward when the querying k-mer is the canonical orientation":
                         // (record k-mer string is canonical) == CortexJunctionsRecord.isForward();
                         // bits 16..31: the first 8 junction bases (2 bits each), so that adding a link reads no bases
};

// one searchable link table on the device (the link sets of a traversal, merged by canonical k-mer)
struct LinksView {
    int64_t M;
    const uint8_t* keys;      // rows of W u64 (stride 8W), ascending
    const uint32_t* pstart;
    int p;
    const uint32_t* off;      // [M+1]
    const JuncRec* junc;      // per key: set after set (order of addition), each in HashSet iteration order
    const uint8_t* bases;
    const uint64_t* rec_of;   // [graph records] junction records of a graph record: first | count << 32 ; ~0 = none
};

struct HostJunction {
    bool is_fw; int num_kmers; int num_junctions; std::vector<int32_t> cov; std::string junctions;
};
struct HostLinksRecord {
    std::string kmer;                   // as written in the file
    std::vector<HostJunction> juncs;    // HashSet iteration order
};

// one .ctp.gz file: parsed on the host; on the device it only owns a flag bit in the graph's probe rows
class Links {
public:
    Links(const std::string& path, const Graph& g);
    ~Links();                                        // gives the flag bit back if the graph is still open
    void graph_closed() { graph_ = nullptr; }        // called by ~Graph
    int version = 0, num_colors = 0, k = 0;
    int64_t num_kmers_in_graph = 0, num_kmers_with_links = 0, num_links = 0, link_bytes = 0;
    std::vector<std::string> sample_names;
    std::vector<HostLinksRecord> records;            // sorted by canonical k-mer
    std::vector<std::vector<uint64_t>> record_keys;  // canonical packed words, same order
    std::vector<uint8_t> record_is_canonical;        // the record's k-mer string is the canonical orientation
    int device = 0;
    int slot = -1;                                   // bit of the probe rows' link-flags byte
    const HostLinksRecord* get(const std::string& kmer_ascii) const;   // containsKey / get
    const std::string& source() const { return source_; }             // ConnectivityAnnotations.getSource(): the LNKIDX header's, "" without an index
private:
    const Graph* graph_ = nullptr;
    std::string source_;
    void mark_records(bool clear);
};

// IndexLinks: in_path (.ctp / .ctp.gz) -> out_path (BGZF) + out_path.idx; returns the number of records
int64_t links_index_file(const std::string& in_path, const std::string& out_path, const std::string& source);

// the link sets a traversal may use, merged into one device table
class MergedLinks {
public:
    MergedLinks(const std::vector<const Links*>& sets, const Graph& g);
    ~MergedLinks();
    LinksView view{};
    uint32_t flag_mask = 0;
private:
    void* d_keys_ = nullptr; void* d_pstart_ = nullptr; void* d_off_ = nullptr; void* d_junc_ = nullptr; void* d_bases_ = nullptr;
    void* d_rec_of_ = nullptr;
};

// radix-indexed search over sorted key rows (shared by graph and links)
template <int W>
LDBG_HOSTDEV int64_t links_find(const LinksView& l, int k, const Kmer<W>& q) {
    if (l.M == 0) return -1;
    uint32_t px = kmer_prefix<W>(q, k, l.p);
    uint32_t lo = l.pstart[px], hi = l.pstart[px + 1];
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        const uint64_t* kp = (const uint64_t*)(l.keys + (size_t)mid * (size_t)(8 * W));
        Kmer<W> m;
#pragma unroll
        for (int i = 0; i < W; i++) m.w[i] = kp[i];
        int c = kmer_cmp<W>(m, q);
        if (c == 0) return (int64_t)mid;
        if (c < 0) lo = mid + 1; else hi = mid;
    }
    return -1;
}

}  // namespace ldbg
