// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the Corticall (mcveanlab/Corticall, Java) LdBG hot path:
// CortexGraph record iteration / findRecord, link-guided cursor walk and the
// recursive DFS with stopping rules.  It is written to mirror the *observable
// behaviour* of the Java code, including its container-order quirks, so that
// the HIP product path can be checked against it bit for bit.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// build, link or call anything in this directory.  The product library
// (corticall_amd/csrc -> libldbg.so) never includes or links it.
//
// Parity pin: the reference cannot be built here (no JDK), so this oracle is
// pinned by the reference's own known-answer tests;
// see tests/test_oracle_golden.py.  Behaviour outside those vectors is
// "parity unpinned" against real Java.
//
// Path shorthands in citations:  J/ = public/java/src/uk/ac/ox/well/cortexjdk/
#pragma once
#include <cstdint>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace orc {

// J/utils/exceptions/CortexJDKException.java
struct CortexJDKException : std::runtime_error { using std::runtime_error::runtime_error; };
// java.lang.NullPointerException raised by the reference on some inputs (Q14 etc.)
struct JavaNullPointer : std::runtime_error { using std::runtime_error::runtime_error; };
struct NoSuchElement : std::runtime_error { using std::runtime_error::runtime_error; };

// ---------------------------------------------------------------- Java emulation helpers
int32_t jhash_bytes(const std::string& b);              // java.util.Arrays.hashCode(byte[])
int32_t jhash_string(const std::string& s);             // java.lang.String.hashCode()
int32_t jhash_longs(const std::vector<int64_t>& v);     // Arrays.hashCode(long[])
int32_t jhash_ints(const std::vector<int32_t>& v);      // Arrays.hashCode(int[])
int32_t jhash_u8(const std::vector<uint8_t>& v);        // Arrays.hashCode(byte[])
// java.util.HashMap: table capacity after n insertions into a default-constructed map
int jhashmap_capacity_for(size_t n);
inline uint32_t jhashmap_bucket(int32_t h, int cap) {
    uint32_t u = (uint32_t)h; u ^= (u >> 16); return u & (uint32_t)(cap - 1);
}
// iteration order of a default java.util.HashSet/HashMap that received the
// (distinct) keys with these hashCodes in this insertion order, no removals.
std::vector<size_t> jhash_iteration_order(const std::vector<int32_t>& hashes);

// ---------------------------------------------------------------- K1: SequenceUtils
// J/utils/sequence/SequenceUtils.java:61-86, 127-135, 206-225
char complement(char c);
std::string reverse_complement(const std::string& s);
std::string complement_str(const std::string& s);
std::string canonical(const std::string& s);             // alphanumericallyLowestOrientation

// K2: J/utils/kmer/CanonicalKmer.java:13-37  (isFlipped by hashCode inequality, Q6)
struct CanonicalKmer {
    std::string kmer;
    bool flipped = false;
    explicit CanonicalKmer(const std::string& s);
};

// ---------------------------------------------------------------- G4-G6: CortexRecord
// J/utils/io/graph/cortex/CortexRecord.java
struct Record {
    int k = 0, W = 0;
    std::vector<int64_t> bk;       // Java long[] as read big-endian from the file (Q: G2)
    std::vector<int32_t> cov;      // LE u32 reinterpreted as signed int (Q5)
    std::vector<uint8_t> edges;

    Record() = default;
    // CortexRecord(String sk, coverageList, inEdgesList, outEdgesList) :35-41, 83-106
    Record(const std::string& sk, const std::vector<int32_t>& covs,
           const std::vector<std::set<char>>& in, const std::vector<std::set<char>>& out);

    static int kmer_bits(int k);                                                  // :309-311
    static std::vector<int64_t> encode_binary_kmer(const std::string& kmer);      // :313-334
    static std::string decode_binary_kmer(const std::vector<int64_t>& bk, int k, int W);  // :291-307
    static uint8_t encode_binary_edges(const std::set<char>& in, const std::set<char>& out, bool rc);  // :379-408

    std::string kmer_string() const { return decode_binary_kmer(bk, k, W); }
    std::string in_edges(int c, bool complement) const;    // :214-238, emission order A,C,G,T bits
    std::string out_edges(int c, bool complement) const;   // :252-275
    int in_degree(int c) const { return (int)in_edges(c, false).size(); }
    int out_degree(int c) const { return (int)out_edges(c, false).size(); }
    std::string edges_string(int c) const;                 // :117-140
    std::string to_string() const;                         // :166-178
    int32_t jhash() const;                                 // :196-198
    bool operator==(const Record& o) const { return bk == o.bk && cov == o.cov && edges == o.edges; }
    // packed McCortex words (word 0 most significant, LE u64 value of the file bytes)
    std::vector<uint64_t> packed_words() const;
};

struct ColorInfo {
    std::string sample_name;
    uint32_t mean_read_length = 0;
    uint64_t total_sequence = 0;
    bool tip_clipping = false, low_covg_supernodes_removed = false, low_covg_kmers_removed = false,
         cleaned_against_graph = false;
    uint32_t low_cov_supernodes_threshold = 0, low_cov_kmer_threshold = 0;
    std::string cleaned_against_graph_name;
};

// ---------------------------------------------------------------- G1-G3: CortexGraph
// J/utils/io/graph/cortex/CortexGraph.java
class CortexGraph {
public:
    // use_cache mirrors the LRUMap(1,000,000) at :160 (cost structure + quirk Q1)
    explicit CortexGraph(const std::string& path, bool use_cache = true);
    ~CortexGraph();
    int version = 0, k = 0, W = 0, C = 0;
    std::vector<ColorInfo> colors;
    int64_t record_size = 0, num_records = 0, data_offset = 0;
    std::string path;

    // getRecord(i) :183-187 ; returns false for "null" (i >= N, Q2); throws for i < 0
    bool get_record(int64_t i, Record& out);
    // findRecord(byte[]) :272-317.  Returns record index or -1 (null).
    int64_t find_record(const std::string& kmer);
    bool find_record(const std::string& kmer, Record& out, int64_t* idx = nullptr);
    int color_for_sample_name(const std::string& name) const;   // :337-357
    // tuned lookup (packed compare, no allocation) with identical results; used by the
    // "cpu-tuned" baseline only
    int64_t find_record_tuned(const std::string& kmer) const;

    bool tuned = false;   // route find_record through find_record_tuned (same results, test-scale speed)
    uint64_t cache_hits_by_kmer = 0, cache_hits_by_index = 0;
    const uint8_t* data() const { return base_ + data_offset; }

private:
    void decode_at(int64_t i, Record& out) const;
    const uint8_t* base_ = nullptr;
    size_t size_ = 0;
    int fd_ = -1;
    bool use_cache_;
    // LRU keyed by record index and by k-mer string (CortexGraph.java:224-225)
    struct CacheEntry { bool by_kmer; int64_t idx; std::string kmer; };
    std::list<CacheEntry> lru_;
    std::unordered_map<int64_t, std::list<CacheEntry>::iterator> by_idx_;
    std::unordered_map<std::string, std::list<CacheEntry>::iterator> by_kmer_;
    void cache_put(int64_t idx, const std::string& kmer);
    static constexpr size_t kCacheMax = 1000000;
};

// J/utils/io/graph/cortex/CortexGraphWriter.java:31-139
void write_cortex_graph(const std::string& path, int k, const std::vector<ColorInfo>& colors,
                        const std::vector<Record>& records);

// J/utils/assembler/TempGraphAssembler.java:19-99.  samples in map-iteration order.
void temp_graph_assembler(const std::string& out_path,
                          const std::vector<std::pair<std::string, std::vector<std::string>>>& haplotypes,
                          int k);
// Order in which a java.util.HashMap<String,?> iterates these keys (TraversalUtilsTest uses one)
std::vector<size_t> java_string_hashmap_order(const std::vector<std::string>& keys);

// ---------------------------------------------------------------- L3-L4: links
// J/utils/io/graph/links/CortexJunctionsRecord.java
struct JunctionsRecord {
    bool is_fw = true;
    int num_kmers = -1;
    int num_junctions = 0;
    std::vector<int32_t> coverages;
    std::string junctions;
    int32_t jhash() const;    // :87-95
    bool operator==(const JunctionsRecord& o) const {
        return is_fw == o.is_fw && num_junctions == o.num_junctions && num_kmers == o.num_kmers &&
               coverages == o.coverages && junctions == o.junctions;
    }
};
// J/utils/io/graph/links/CortexLinksRecord.java — cjs is a java.util.HashSet
struct LinksRecord {
    std::string kmer;
    std::vector<JunctionsRecord> cjs_insertion;     // deduplicated, insertion order
    std::vector<JunctionsRecord> junctions() const; // HashSet iteration order
    std::string to_string() const;                  // :45-61
};
// J/utils/io/graph/links/CortexLinksIterable.java + CortexLinksMap.java (un-indexed .ctp.gz path)
class CortexLinks {
public:
    explicit CortexLinks(const std::string& path);
    int version = 0, k = 0, num_colors = 0;
    int64_t num_kmers_in_graph = 0, num_kmers_with_links = 0, num_links = 0, link_bytes = 0;
    std::vector<std::string> sample_names;
    std::string source = "unknown";                               // ConnectivityAnnotations.getSource default
    std::vector<LinksRecord> records;                             // file order
    bool contains(const std::string& canonical_kmer) const { return map_.count(canonical_kmer) != 0; }
    const LinksRecord& get(const std::string& canonical_kmer) const { return records[map_.at(canonical_kmer)]; }
private:
    std::unordered_map<std::string, size_t> map_;
};
// J/utils/assembler/TempLinksAssembler.java:29-105
void temp_links_assembler(CortexGraph& graph, const std::vector<std::string>& reads,
                          const std::string& sample, const std::string& out_path);

// ---------------------------------------------------------------- L1-L2: LinkStore
// J/utils/traversal/LinkStore.java — linkElements is a java.util.HashMap<String, List<...>> (Q10)
class LinkStore {
public:
    void add(const std::string& cur_kmer, const LinksRecord& clr, bool go_forward);   // :17-35
    void increment_ages();                                                           // :37-43
    int num_new_paths() const;                                                       // :45-56
    bool next_junction_choice(char& choice);                                         // :122-144 (false = null)
    bool is_active() const { return !keys_.empty(); }                                // :146-148
    int size() const;
private:
    struct Elem { int age = 0, pos = 0; };
    struct Key { std::string jl; int32_t hash; uint64_t seq; std::vector<Elem> elems; };
    std::vector<Key> keys_;          // live keys (unordered storage)
    int cap_ = 0;                    // HashMap table size (0 = not allocated yet)
    size_t size_ = 0;
    uint64_t seq_ = 0;
    std::vector<size_t> order() const;   // HashMap iteration order over keys_
    bool oldest_link(size_t& key_index) const;   // :92-119
    void increment_positions_and_expire(char choice);   // :58-90
};

// ---------------------------------------------------------------- D3: vertex / edge / graph
// J/utils/traversal/CortexVertex.java:67-91 ; kmerSources is always the empty set on this path
struct Vertex {
    std::string sk;
    int64_t rec = -1;      // record index in the traversal graph, -1 = null CortexRecord
    int copy_index = 0;
    int index = 0;
    bool operator==(const Vertex& o) const {
        return rec == o.rec && copy_index == o.copy_index && index == o.index && sk == o.sk;
    }
};
struct VertexHash { size_t operator()(const Vertex& v) const; };
struct Edge { int src, dst, color; };
// org.jgrapht.graph.DirectedWeightedPseudograph<CortexVertex,CortexEdge> (jgrapht-core 1.0.1):
// insertion-ordered vertex and edge sets; edge identity per CortexEdge.equals = unordered {s,t} + colour
class PGraph {
public:
    std::vector<Vertex> verts;
    std::vector<Edge> edges;
    int add_vertex(const Vertex& v);
    int find_vertex(const Vertex& v) const;
    bool contains_edge(int s, int t) const;                  // directional
    bool add_edge(int s, int t, int color);                  // false if an equal CortexEdge exists
    void add_graph(const PGraph& o);                         // org.jgrapht.Graphs.addGraph
private:
    std::unordered_map<Vertex, int, VertexHash> vmap_;
    std::set<std::pair<int, int>> dir_;                      // (s,t)
    std::set<std::tuple<int, int, int>> und_;                // (min,max,color)
};

// ---------------------------------------------------------------- S1-S3: stopping rules
enum StopperId {
    CONTIG = 0, CYCLE_COLLAPSING_CONTIG, DESTINATION, EXPLORATION, NOVEL_PARTITION,
    NOVEL_KMER_LIMITED_CONTIG, NOVEL_CONTINUATION, BUBBLE_CLOSING, BUBBLE_OPENING, CONTAMINANT,
    DUST, GAP_CLOSING, NAHR, NOVEL_KMER_AGGREGATION, ORPHAN, PAIRED_READ_CLOSING, TIP_BEGINNING,
    TIP_END, VISUALIZATION, NUM_STOPPERS
};

// C1: J/utils/traversal/TraversalEngineConfiguration.java:19-37
struct EngineConfig {
    std::vector<int> traversal_colors;          // LinkedHashSet: insertion order
    std::set<int> joining_colors, recruitment_colors, secondary_colors;   // TreeSet
    bool op_and = false;                        // GraphCombinationOperator (default OR)
    int direction = 0;                          // 0 BOTH, 1 FORWARD, 2 REVERSE
    bool connect_all_neighbors = false;
    int max_length = 75000;
    int stopper = CONTIG;
    CortexGraph* graph = nullptr;
    CortexGraph* rois = nullptr;
    std::vector<CortexLinks*> links;            // Java: HashSet (identity order); we use given order
    bool strict_java_flip = true;               // Q6: isFlipped by hash inequality
};

struct TraversalState {     // J/utils/traversal/TraversalState.java
    const Vertex* cur; bool go_forward; int graph_size, junction_depth, branch_size, adj, radj;
    bool children_traversed, reached_max; const std::vector<std::string>* sinks;
};

class StoppingRule;   // per-branch instance

// ---------------------------------------------------------------- E1-E3, D1-D2, N1-N2: engine
// J/utils/traversal/TraversalEngine.java
class TraversalEngine {
public:
    explicit TraversalEngine(const EngineConfig& cfg);   // validation per TraversalEngineFactory.make :54-88
    const EngineConfig& config() const { return ec_; }

    std::unique_ptr<PGraph> dfs(const std::string& source, const std::vector<std::string>& sinks = {});   // :64-106
    std::vector<Vertex> walk(const std::string& seed);                                                  // :108-110
    void seek(const std::string& sk);                                                                   // :321-335
    bool has_next() const { return has_next_; }
    bool has_previous() const { return has_prev_; }
    Vertex next();                                                                                      // :241-279
    Vertex previous();                                                                                  // :281-319
    std::vector<Vertex> next_vertices(const std::string& sk);                                           // :194-239
    std::vector<Vertex> prev_vertices(const std::string& sk);                                           // :147-192
    const std::string& cursor() const { return cur_; }
    uint64_t dfs_iterations = 0;       // dfs loop iterations: the "k-mers traversed" unit of walk/dfs batches
    uint64_t cursor_steps = 0;         // next()/previous() calls (also made from inside dfs when links are present)
    Record record_of(const Vertex& v);
    int32_t vertex_jhash(const Vertex& v);

private:
    EngineConfig ec_;
    std::string cur_, prev_, next_;
    bool has_cur_ = false, has_prev_ = false, has_next_ = false;
    std::unordered_set<std::string> seen_;
    bool specific_links_null_ = true;
    LinkStore store_;
    bool go_forward_ = true;

    std::vector<Vertex> adjacent_vertices(const std::string& sk, bool forward);
    std::unique_ptr<PGraph> dfs_branch(Vertex cv, bool go_forward, int graph_size, int depth,
                                       const std::unordered_set<Vertex, VertexHash>& visited_old,
                                       const std::vector<std::string>& sinks);                         // :356-482
    void connect_vertex(PGraph& g, const Vertex& cv, const std::vector<Vertex>* pvs,
                        const std::vector<Vertex>* nvs);                                               // :494-516
    bool adjacent_kmer(const std::string& kmer, const std::vector<Vertex>& adj, bool fwd, std::string& out);  // :518-546
    void initialize_link_store(bool fwd);                                                               // :548-568
    void update_link_store(bool fwd);                                                                   // :570-597
    std::vector<CortexLinks*> my_links() const;
    Vertex step(bool fwd);
    void add_secondary_colors(PGraph& g);                                                               // :599-645
};

// W1: J/utils/traversal/TraversalUtils.java:367-488
std::vector<Vertex> to_walk(TraversalEngine& e, const PGraph* g, const std::string& sk, int color);
std::string to_contig(const std::vector<Vertex>& walk);

// S2: the exact integer table behind DestinationStopper's junction limit
int destination_junction_limit(int graph_size);

}  // namespace orc
