// Host side of the engine: configuration validation (TraversalEngineFactory.make), batching,
// device scratch management and result storage.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "engine.h"

namespace ldbg {

struct WalkChunk {
    int64_t first = 0, n = 0;              // seeds [first, first+n)
    void* d_path = nullptr;                // dense u64 path entries of all strands of the chunk
    size_t path_cap = 0, contigs_cap = 0;  // bytes behind d_path / d_contigs (buffers are reused from batch to batch)
    void* d_contigs = nullptr;             // dense ASCII contigs
    void* d_seed_words = nullptr;          // [n][W]
    void* d_term = nullptr;                // [2n][W] k-mer of a trailing null-record vertex
    // offsets and lengths are computed on the device and stay there (a batch whose contigs are consumed in HBM never needs them on the
    // host); Engine::ensure_host brings them over the first time fetch / walk_vertices ask
    void* d_contig_off = nullptr;          // [n+1] int64
    void* d_walk_len = nullptr;            // [n] int64
    int64_t total_bytes = 0;               // contig_off[n]
    int64_t total_entries = 0;             // strand_off[2n]
    bool host_ready = false;
    std::vector<int64_t> strand_off;       // [2n+1] offsets into d_path (strand 2i = reverse, 2i+1 = forward)        } host mirrors,
    std::vector<int64_t> contig_off;       // [n+1]                                                                  } valid once
    std::vector<int64_t> walk_len;         // [n]                                                                    } host_ready
    std::vector<uint32_t> status;          // [2n]: filled only where a strand ended in an error (walk_finish)
    // the walks as the kernel stored them (vertex entries and run / repeat descriptors in the engine's path pool): d_path is expanded
    // from them by Engine::ensure_dense the first time vertex lists are asked for — contigs do not need it
    void* d_strand_c = nullptr;            // [2n] stored entries per strand
    void* d_strand_off = nullptr;          // [2n+1] device copy of strand_off
    int max_blocks = 0;
    RunIndexView runs{};
    bool dense_pending = false;
};

// ---- dfs results (dfs.cpp): the DirectedWeightedPseudograph<CortexVertex, CortexEdge> of every seed
struct DfsVertex { int64_t rec; int64_t slot; int32_t copy, index; uint8_t flip; };
struct DfsEdge { int src, dst, color; };
struct DfsGraphHost {
    bool is_null = true;                   // dfs() returned null
    // a result whose directions are each one branch of vertices with records is kept PACKED until its vertex list is asked for with
    // k-mers (DfsBatch::materialize): the log entries of the reverse branch (seed first), then those of the forward branch, in
    // DfsBatch::packed_store[p_seg] from p_off on.  The graph is the two paths joined at the seed (TraversalEngine.java:75-99).
    bool packed = false;
    uint32_t n_rev = 0, n_fwd = 0, p_seg = 0;
    uint64_t p_off = 0;
    int64_t n_vertices() const { return packed ? (int64_t)n_rev + n_fwd - ((n_rev && n_fwd) ? 1 : 0) : (int64_t)verts.size(); }
    int64_t n_edges() const { return packed ? (int64_t)(n_rev ? n_rev - 1 : 0) + (n_fwd ? n_fwd - 1 : 0) : (int64_t)edges.size(); }
    int64_t slot_base = 0;                 // added to the slots of the vertices with records (their place among DfsBatch::key_segments)
    std::vector<DfsVertex> verts;          // insertion order
    std::vector<DfsEdge> edges;            // insertion order
    std::vector<uint64_t> words;           // [verts][W]
    std::vector<uint32_t> cov;             // [verts][C]
    std::vector<std::vector<uint64_t>> null_kmers;
};
struct DfsBatch {
    int k = 0, W = 0, C = 0;
    int64_t traversed = 0;
    std::vector<DfsGraphHost> results;
    // the k-mer words and coverages of the vertices are gathered from the device on first use (the graph must still be open)
    const Graph* graph = nullptr;
    std::vector<std::vector<uint64_t>> packed_store;   // vertex entries (engine.h: path_pack) of the packed results
    int color = 0;                                     // colour of the edges (the first traversal colour, TraversalEngine.java:502)
    // vertices and edges of packed result i in the reference's insertion order, without unpacking it (outputs may be null)
    void read_packed(int64_t i, int64_t* rec, int32_t* copy_index, int32_t* index, int32_t* edge_src, int32_t* edge_dst, int32_t* edge_color) const;
    std::vector<std::vector<uint64_t>> key_segments;   // keys ((record + 1) << 1 | flip) of the vertices to gather, in segments laid end to end
    int64_t n_gather = 0;
    bool materialized = false;
    void materialize();
    std::string walk_contig(int64_t i, const char* seed, int color);
};

DfsBatch* dfs_merge(DfsBatch& b, const int64_t* which, int64_t m);     // Graphs.addGraph over results of one batch (dfs.cpp)

struct WalkRun;
class ShardImage;
// a dfs batch over the local image of a hash-sharded table (image.h): the library runs the rounds, the caller's callback makes the
// exchange of every round (requests out, rows in) and returns non-zero once no rank has a search in progress
struct ShardedRun {
    ShardImage* img;
    const int32_t* d_seed_slot;    // [n] image slot of every source's record (-1 = none): the rows are in the image already
    const int32_t* d_sink_slot;    // [number of sinks] likewise
    int (*round_done)(void* user);  // 0 = another round, 1 = no rank has a search in progress, 2 = give the batch up ("IMAGE_FULL")
    void* user;
    int64_t* d_stats;              // device, 3 x int64: searches of this rank not done yet, requests filed in the round, the image's overflow flag
    rt::stream_t stream;           // the stream the rounds are queued on (the callback's collectives use it too)
};
uint64_t vt_series(uint64_t init, uint64_t vmax);     // walk.cpp
uint32_t vt_initial_entries();

class Engine {
public:
    explicit Engine(const ldbg_engine_config& cfg);
    ~Engine();
    const Graph* graph = nullptr;
    const Graph* rois = nullptr;
    std::vector<const Links*> my_links;   // link sets whose colour-0 sample is a traversal sample (:553-557)
    ldbg_engine_config cfg{};
    EngineView view{};

    // seeds: n x k ASCII bytes — in host memory, or (seeds_on_device) already in this device's memory
    void walk_batch_run(const char* seeds, int64_t n, int64_t* total_contig_bytes, int64_t* kmers_traversed, bool seeds_on_device = false);
    void walk_batch_fetch(char* arena, int64_t cap, int64_t* offsets, int64_t* walk_len);
    void walk_vertices(int64_t walk, int64_t capacity, int64_t* len, uint64_t* words, int64_t* rec, int32_t* copy, int32_t* index);
    void clear_batch();
    void enter();                                                // entry of a call that queues device work (device, graph-stream work complete)
    rt::stream_t estream() const { return stream_; }            // the stream this engine's work is queued on
    void quiesce() noexcept;                                     // waits for the streams this engine has work on (before blocks go back to rt::tfree)
    void walk_roi_hits(int64_t* offsets, uint32_t* hits, int64_t capacity, uint8_t* has_null);
    // walk_batch_run over the local image of a hash-sharded table, one bulk-synchronous round at a time (image.h): begin with the
    // seeds' image slots (device array, -1 = no record), then rounds until no rank has a strand left, then finish
    void sharded_walk_begin(ShardImage& img, const char* seeds, int64_t n, const int32_t* d_seed_slot, rt::stream_t round_stream);
    void sharded_walk_round(int64_t* d_stats);        // d_stats[0] = strands of this rank not done yet, d_stats[1] = requests filed (device memory)
    void sharded_walk_finish(int64_t* total_contig_bytes, int64_t* kmers_traversed);
    void sharded_abort();
    // dfs(source, sinks...) for n sources; sinks as CSR over ASCII k-mers (sink_offsets may be nullptr)
    DfsBatch* dfs_batch(const char* sources, int64_t n, const char* sinks, const int64_t* sink_offsets, const ShardedRun* sharded = nullptr);
    // getNextVertices / getPrevVertices of n k-mers as CSR (dfs.cpp)
    void neighbours_batch(const char* kmers, int64_t n, bool forward, int64_t* offsets, uint64_t* kmer_words, int64_t* rec, int64_t capacity);
    int dfs_max_depth = 64;
    int dfs_log_blocks = 64;          // path blocks (1024 entries) one strand's dfs log may use
    int64_t dfs_traversed() const { return dfs_traversed_; }
    int64_t dfs_retried_ = 0;         // searches that took the second launch without the run index
    int64_t retried_strands() const { return retried_strands_; }

    int64_t batch_n = 0, batch_bytes = 0, batch_traversed = 0;
    std::vector<WalkChunk> chunks;
    uint32_t link_store_capacity = 64;

private:
    rt::stream_t stream_ = nullptr, own_stream_ = nullptr;       // own_stream_: created by this engine (resident tables); else the graph's
    // per-slot scratch kept across batches: visited tables, link stores, table generations
    void* d_vpool_ = nullptr; void* d_ls_ = nullptr; void* d_snap_ = nullptr;
    void* d_pool_ = nullptr; void* d_block_table_ = nullptr;
    int64_t n_slots_ = 0, bt_strands_ = 0;
    uint64_t n_blocks_ = 0, vpool_entries_ = 0, vpool_dirty_ = 0;
    uint32_t ecap_ = 0;
    int max_blocks_ = 0;
    void* d_frames_ = nullptr; size_t d_frames_bytes_ = 0;
    void* d_roi_bits_ = nullptr;
    void* d_roi_of_ = nullptr;
    int64_t dfs_traversed_ = 0;
    std::unique_ptr<MergedLinks> merged_;
    std::unique_ptr<class RunIndex> runs_;     // records in unitig order for this engine's colour masks (runs.h), built by the first walk batch
    int64_t retried_strands_ = 0;
    std::vector<uint8_t> sink_valid_;
    std::vector<uint8_t> seed_valid_;          // per source of the current dfs batch: is the string a k-mer over ACGT (Q4)
    // seeds of the current walk batch on the device: packed words and the Q4 validity byte (walk.cpp: k_seed_words)
    void* d_batch_words_ = nullptr; void* d_batch_valid_ = nullptr; void* d_batch_ascii_ = nullptr;
    void seeds_to_device(const char* seeds, int64_t n, bool seeds_on_device);
    void drop_batch_seeds();
    unsigned long long* h_small_ = nullptr;    // page-locked landing block for the counters a batch hands to the host (64 words)
    void ensure_host(WalkChunk& c);
    void build_roi_bits();
    bool dfs_chunk(const std::vector<uint64_t>& seed_words, const std::vector<uint64_t>& sink_words, const int64_t* sink_offsets,
                   int64_t first, int64_t n, DfsBatch& out, const ShardedRun* sharded);
    void launch_compact_paths(const int64_t* d_strand_off, int64_t n_strands, uint64_t* d_dense, int max_blocks);
    // stored paths with descriptors (strand.h): entries they expand to per strand, and the expansion (walk.cpp: k_expand_paths)
    void launch_path_lengths(const uint32_t* d_strand_c, int64_t n_strands, int max_blocks, uint32_t* d_len);
    void launch_expand_paths(const uint32_t* d_strand_c, const int64_t* d_strand_off, int64_t n_strands, uint64_t* d_dense, int max_blocks, const RunIndexView& runs,
                             unsigned* d_overflow);
    void ensure_run_index();           // the run index of this engine's colour masks (runs.h), built on first use
    void ensure_scratch(int64_t n_strands, uint32_t ecap, int max_blocks, uint64_t table_floor = 0, bool small = false);
    uint64_t scratch_scale_ = 1, scratch_scale_built_ = 1; bool scratch_full_ = true; int64_t pool_growths_ = 0;   // walk.cpp: pools start small with the run index   // table_floor: entries the table pool holds at least
    uint64_t table_floor_ = 0;
    void* h_stage_[2] = {nullptr, nullptr};              // page-locked staging buffers of downloads into pageable memory (walk.cpp: download)
    void download(char* dst, const void* d_src, size_t bytes);
    uint64_t* h_log_ = nullptr; size_t h_log_cap_ = 0;   // page-locked landing buffer of the dfs logs (dfs.cpp)
    void release_scratch();
    // result buffers of a cleared batch are kept for the next one: allocating and freeing GBs costs milliseconds per batch
    struct Spare { void* p; size_t bytes; };
    std::vector<Spare> spares_;
    void* result_alloc(size_t bytes, size_t* cap);
    void result_free(void* p, size_t cap);
    void drop_spares();
    void zero_dirty_tables(rt::stream_t s);      // the part of the table pool the last launch handed out
    bool run_chunk(int64_t first, int64_t n, WalkChunk& out, int64_t* traversed);
    void ensure_dense(WalkChunk& c);
    void materialize_pending();
    void walk_prepare(int64_t first, int64_t n, WalkRun& r, ShardImage* img, const int32_t* d_seed_slot);
    void walk_launch(WalkRun& r);
    bool walk_finish(WalkRun& r, int64_t* traversed);
    WalkRun* sharded_run_ = nullptr;           // the batch in progress over a sharded table's image (sharded_walk_begin .. _finish)
    ShardImage* sharded_img_ = nullptr;
    rt::stream_t sharded_stream_ = nullptr;
    int64_t sharded_rounds_ = 0;
};

}  // namespace ldbg
