/* ldbg.h — C ABI of libldbg.so, the MI355X-native LdBG traversal engine.
 *
 * This is the drop-in boundary for ONE hot path of mcveanlab/Corticall: CortexGraph record
 * iteration, binary-search random access (findRecord) and the link-guided walk / DFS of
 * TraversalEngine.  The reference has no FFI for this path; the seam is two Java interfaces
 * and one façade, so every entry point below cites the Java member it replaces
 * (J/ = public/java/src/uk/ac/ox/well/cortexjdk/).  A JNI / ctypes binding calls exactly
 * these functions.
 *
 * Conventions
 *   - every function returns an ldbg_status; on failure ldbg_last_error() (thread-local)
 *     holds the message the reference would have put in its exception.
 *   - k-mers cross the boundary either as ASCII (n × k bytes, no terminators) or as
 *     "packed words": W = ceil(k/32) uint64 per k-mer, word 0 most significant, 2 bits per
 *     base (A=0,C=1,G=2,T=3), right aligned — the value McCortex stores in a .ctx record
 *     (docs/ctx_spec.md "Binary kmer specification").
 *   - plain pointers and sizes only; "_dev" variants take device pointers (HBM resident
 *     inputs/outputs) and a hipStream_t passed as void*; all others take host pointers.
 *   - handles are not thread-safe (neither are the reference's objects); distinct handles
 *     may be used from distinct threads.
 *   - there is NO CPU fallback: every call that computes runs HIP kernels on the handle's
 *     device and fails with LDBG_ERR_HIP when no GPU is present.
 */
#ifndef LDBG_H
#define LDBG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    LDBG_OK = 0,
    LDBG_ERR_CORTEXJDK = 1,      /* J/utils/exceptions/CortexJDKException (bad magic/version/unsorted/IO/config) */
    LDBG_ERR_NULLPOINTER = 2,    /* input on which the reference throws NullPointerException */
    LDBG_ERR_NOSUCHELEMENT = 3,  /* TraversalEngine.next()/previous() without a successor (TraversalEngine.java:242,282) */
    LDBG_ERR_UNSUPPORTED = 4,
    LDBG_ERR_HIP = 5,            /* no device / HIP runtime failure */
    LDBG_ERR_ARG = 6,
    LDBG_ERR_CAPACITY = 7        /* caller-provided output buffer too small; required size is reported */
} ldbg_status;

typedef struct ldbg_graph ldbg_graph;     /* DeBruijnGraph backed by a .ctx file, resident in HBM */
typedef struct ldbg_links ldbg_links;     /* ConnectivityAnnotations (.ctp.gz), resident in HBM */
typedef struct ldbg_engine ldbg_engine;   /* TraversalEngine */
typedef struct ldbg_dfs_result ldbg_dfs_result;

const char* ldbg_last_error(void);
const char* ldbg_version(void);
/* number of visible HIP devices (0 on a CPU-only host; no compute entry point works then) */
ldbg_status ldbg_device_count(int* count);

/* ------------------------------------------------------------------ k-mer helpers (host, no GPU needed)
 * CortexRecord.encodeBinaryKmer / decodeBinaryKmer (J/utils/io/graph/cortex/CortexRecord.java:291-334) */
ldbg_status ldbg_kmer_encode(const char* ascii, int k, uint64_t* words_out);
ldbg_status ldbg_kmer_decode(const uint64_t* words, int k, char* ascii_out /* k bytes + NUL */);

/* Sort (J/commands/utils/Sort.java:20-49): writes the records of in_path in k-mer order (CortexRecord.compareTo :210-212,
 * stable like Arrays.sort) under the unchanged header — produces the sorted table ldbg_graph_open requires.  The order is
 * computed by a radix sort on the device. */
ldbg_status ldbg_sort_ctx(const char* in_path, const char* out_path, int device, int64_t* num_records);
/* Join (J/commands/utils/Join.java:16-60; CortexCollection.java:34-58, 218-293): the union of the k-mers of several sorted
 * graphs with every graph's colours side by side, written as one graph.  num_records = k-mers written. */
ldbg_status ldbg_join_ctx(const char* const* in_paths, int n_paths, const char* out_path, int device, int64_t* num_records);
/* CortexGraphWriter over a selection of records (J/utils/io/graph/cortex/CortexGraphWriter.java:40-135, as the filters use it:
 * setHeader(in.getHeader()), addRecord for the chosen records in the order given, close — FindTips.java:112-131).  A file
 * operation on the host. */
ldbg_status ldbg_ctx_write_records(const char* in_path, const int64_t* indices, int64_t n, const char* out_path);

/* ------------------------------------------------------------------ graph: G1-G3
 * new CortexGraph(path)              J/utils/io/graph/cortex/CortexGraph.java:40-48, 66-168
 * The file is parsed on the host, streamed to the device, verified strictly ascending
 * (the reference asserts sortedness lazily at :295-301) and laid out in HBM. */
ldbg_status ldbg_graph_open(const char* path, int device, ldbg_graph** out);
/* same, from an in-memory image of a .ctx file (header + records) */
ldbg_status ldbg_graph_open_memory(const void* image, int64_t nbytes, int device, ldbg_graph** out);
/* same, the header bytes on the host and the records (sorted, the file's record layout) already in DEVICE memory: a shard of a
 * hash-partitioned table cut on the device (corticall_amd/distributed.py) is laid out without a round trip through the host */
ldbg_status ldbg_graph_open_device(const void* header, int64_t header_bytes, const void* d_records, int64_t n_records, int device, ldbg_graph** out);
/* new CortexCollection(graphs...)   J/utils/io/graph/cortex/CortexCollection.java:34-58: several sorted graphs of one k-mer size as ONE
 * graph, every member's colours side by side, merged on the device without writing a file.  find_view = 0: the records its
 * iterator yields (:218-293, the union of the members' k-mers); find_view = 1: the graph its findRecord answers from (:160-188, one
 * findRecord per member — a member of two records or fewer never finds anything).  Engines are created on the find
 * view; the two differ only when such a member is present. */
ldbg_status ldbg_graph_open_collection(const char* const* paths, int n_paths, int find_view, int device, ldbg_graph** out);
ldbg_status ldbg_graph_close(ldbg_graph* g);                                     /* DeBruijnGraph.close() */
/* getKmerSize/getKmerBits/getNumColors/getNumRecords/getVersion   CortexGraph.java:325-335 */
ldbg_status ldbg_graph_info(const ldbg_graph* g, int* k, int* W, int* C, int64_t* N, int* version);
ldbg_status ldbg_graph_device(const ldbg_graph* g, int* device);
/* A shard of a hash-partitioned table (corticall_amd/distributed.py) must answer membership exactly: quirk Q1
 * (findRecord never finds anything in a graph of <= 2 records, CortexGraph.java:274-282) is a property of the whole
 * graph and is applied by the partitioned front end.  is_shard != 0 turns Q1 off for this handle. */
ldbg_status ldbg_graph_set_shard(ldbg_graph* g, int is_shard);
/* getSampleName(color) / getColor(color)                          CortexGraph.java:329,335 */
ldbg_status ldbg_graph_sample_name(const ldbg_graph* g, int color, char* buf, int buflen);
typedef struct {
    uint32_t mean_read_length;
    uint64_t total_sequence;            /* as the reference reads it (big-endian) */
    uint8_t tip_clipping, low_covg_supernodes_removed, low_covg_kmers_removed, cleaned_against_graph;
    uint32_t low_cov_supernodes_threshold, low_cov_kmer_threshold;
} ldbg_color_info;
ldbg_status ldbg_graph_color_info(const ldbg_graph* g, int color, ldbg_color_info* out,
                                  char* cleaned_against_name, int buflen);
/* getColorForSampleName(name)                                     CortexGraph.java:337-357 ; -1 if none/ambiguous */
ldbg_status ldbg_graph_color_for_sample_name(const ldbg_graph* g, const char* name, int* color);

/* Iterator<CortexRecord>.next() / getRecord(i) in bulk             CortexGraph.java:183-258
 * records [first, first+n): kmer_words n×W, cov n×C (the reference's signed int view of the
 * LE u32), edges n×C.  Indices >= N yield LDBG_ERR_ARG (the reference returns null, Q2). */
ldbg_status ldbg_graph_records(const ldbg_graph* g, int64_t first, int64_t n,
                               uint64_t* kmer_words, uint32_t* cov, uint8_t* edges);
ldbg_status ldbg_graph_records_dev(const ldbg_graph* g, int64_t first, int64_t n,
                                   uint64_t* d_kmer_words, uint32_t* d_cov, uint8_t* d_edges, void* stream);

/* findRecord(byte[] | CortexByteKmer | CanonicalKmer | String)    CortexGraph.java:272-321
 * Queries are canonicalised on the device (SequenceUtils.alphanumericallyLowestOrientation).
 * idx_out[i] = record index, or -1 where the reference returns null: absent k-mer, non-ACGT
 * query (ASCII form, Q4), or N <= 2 (Q1, cold cache).  cov_out / edges_out (n×C) may be NULL;
 * rows of misses are zero. */
ldbg_status ldbg_graph_find(const ldbg_graph* g, const uint64_t* packed, int64_t n,
                            int64_t* idx_out, uint32_t* cov_out, uint8_t* edges_out);
ldbg_status ldbg_graph_find_ascii(const ldbg_graph* g, const char* kmers, int64_t n,
                                  int64_t* idx_out, uint32_t* cov_out, uint8_t* edges_out);
ldbg_status ldbg_graph_find_dev(const ldbg_graph* g, const uint64_t* d_packed, int64_t n,
                                int64_t* d_idx_out, uint32_t* d_cov_out, uint8_t* d_edges_out, void* stream);

/* ------------------------------------------------------------------ hash partitioning over devices
 * owner[i] = mix64(minimizer of canonical k-mer i) mod world — the rule by which the sorted table is split into per-device
 * shards (each still sorted) and by which a lookup is routed to the shard that can answer it.  The minimizer is the m-mer, m = (k + 2) / 3
 * (k itself up to k = 8), taken in its own canonical orientation, whose mixed value is smallest: a k-mer and its reverse complement
 * agree, and a run of consecutive k-mers of a walk shares it — their rows live on one shard and travel together (ldbg_image_serve_chain).  The _dev form
 * works on device buffers (d_canon, n x W canonical words, may be NULL) and is what the exchange step of
 * corticall_amd/distributed.py calls between its all-to-alls; the host form is used when the shards are cut. */
ldbg_status ldbg_shard_owner_dev(int k, const uint64_t* d_packed, int64_t n, int world, uint64_t* d_canon, int32_t* d_owner, void* stream);
ldbg_status ldbg_shard_owner(int k, const uint64_t* packed, int64_t n, int world, int device, int32_t* owner);

/* The GLOBAL neighbour index of a shard (corticall_amd/csrc/shard.cpp): for each record and each of its 8 possible neighbours the
 * owner, the record number in the owner's shard and the orientation — one routed findRecord per edge, memoised at load.
 * All buffers are device buffers of the calling rank; calls return when the work is done. */
ldbg_status ldbg_shard_nbr_queries(const ldbg_graph* shard, int64_t first, int64_t n, uint64_t* d_words /* [8n][W] */, uint8_t* d_flips /* [8n] */,
                                   uint8_t* d_have /* [8n] 0 = no colour carries that edge: no query */);
ldbg_status ldbg_shard_set_nbr(ldbg_graph* shard, int64_t first, int64_t n, const int32_t* d_owner, const int64_t* d_local_idx, const uint8_t* d_flips);

/* Traversals over the hash-sharded table: the local IMAGE (corticall_amd/csrc/image.h).
 * A walk lives on the rank that was given its seed (visited set, link store, path, stopping rule never move).  The rank keeps an
 * image — the rows it has been sent so far, laid out like a shard's probe table, with the neighbour index rewritten to image
 * slots — and the unchanged traversal kernels run on it: link-guided steps, junction choices, every quirk, even k.  A strand that
 * is about to read a row that is not there suspends and files a request.  One bulk-synchronous round =
 *   ldbg_engine_sharded_walk_round   every strand runs until it suspends or ends
 *   ldbg_image_bucket                requests -> one block per owner  [world][cap] of global id keys
 *   (all-to-all)                     RCCL over xGMI: torch.distributed in corticall_amd/distributed.py
 *   ldbg_image_serve                 the owners write the rows asked for: global id key | 8 global neighbour ids | probe row
 *   (all-to-all)
 *   ldbg_image_insert                arrivals enter the image; rows fetched for one strand serve all strands of the rank
 * `stream`: a HIP stream (e.g. torch's current stream) all calls of a round are queued on — none of them synchronises with the
 * host; NULL = the library's own stream.  A global id key is  (record number in the owner's shard + 1) | owner << 40. */
typedef struct ldbg_image ldbg_image;
struct ldbg_engine;
ldbg_status ldbg_image_create(const ldbg_graph* shard_with_neighbour_index, int64_t capacity_rows, int64_t global_records, ldbg_image** out);
ldbg_status ldbg_image_destroy(ldbg_image* im);
/* the image as a graph handle: engines over the sharded table are created on it (owned by the image; do not close it) */
ldbg_status ldbg_image_graph(ldbg_image* im, ldbg_graph** image_graph);
ldbg_status ldbg_image_row_bytes(const ldbg_image* im, int* bytes);
ldbg_status ldbg_image_clear(ldbg_image* im);
ldbg_status ldbg_image_request(ldbg_image* im, const uint64_t* d_keys, int64_t n, void* stream);      /* explicit requests: seeds, sinks */
ldbg_status ldbg_image_reset_requests(ldbg_image* im, void* stream);
ldbg_status ldbg_image_bucket(ldbg_image* im, int world, uint32_t cap_per_owner, uint64_t* d_send /* [world][cap_per_owner], 0 = unused */, void* stream);
ldbg_status ldbg_image_serve(const ldbg_image* im, int my_rank, const uint64_t* d_keys, int64_t n, uint8_t* d_rows_out, void* stream);
/* as ldbg_image_serve, `depth` row slots per request: the row asked for, then rows around it that this owner holds too (its unique
 * neighbours outwards in both directions; ownership goes by minimizer, so these are mostly the rows the asker wants next).  A slot
 * whose key is 0 is unused.  d_rows_out: [n][depth][row_bytes]; hand all n * depth slots to ldbg_image_insert. */
ldbg_status ldbg_image_serve_chain(const ldbg_image* im, int my_rank, const uint64_t* d_keys, int64_t n, int depth, uint8_t* d_rows_out, void* stream);
ldbg_status ldbg_image_insert(ldbg_image* im, const struct ldbg_engine* engine_or_null, const uint8_t* d_rows, int64_t n, void* stream);
ldbg_status ldbg_image_lookup(const ldbg_image* im, const uint64_t* d_keys, int64_t n, int32_t* d_slots /* -1 = not in the image */, void* stream);
ldbg_status ldbg_image_counters(const ldbg_image* im, int64_t* n_rows, int64_t* n_requests, int* overflow);   /* synchronises */
/* TraversalEngine.walk over the image (ContigStopper; links bound to the shard graph).  begin: seeds (n x k ASCII, host) with the
 * image slots of their records (device, -1 = none; their rows are already in the image); round: d_stats (device, 3 x int64) =
 * {strands of this rank not done yet, requests filed, the image is full}; finish: results as after ldbg_engine_walk_batch_run.
 * A full image never fills a request again: when the third value is set on any rank, every rank stops its rounds, enlarges its
 * image and runs the batch again (begin drops a batch that was not finished). */
ldbg_status ldbg_engine_sharded_walk_begin(struct ldbg_engine* e, ldbg_image* im, const char* seeds, int64_t n, const int32_t* d_seed_slot, void* stream);
ldbg_status ldbg_engine_sharded_walk_round(struct ldbg_engine* e, int64_t* d_stats);
ldbg_status ldbg_engine_sharded_walk_finish(struct ldbg_engine* e, int64_t* total_contig_bytes, int64_t* kmers_traversed);
/* TraversalEngine.dfs(source, sinks...) over the image, any stopping rule that does not consult a ROI graph.  The library runs the
 * rounds and calls round_done(user) after each: the caller makes the round's exchange there (bucket, all-to-all, serve, all-to-all,
 * insert) and returns 1 once no rank has a search in progress, 2 to give the batch up on every rank (d_stats as above: a full image;
 * the call then fails with LDBG_ERR_CAPACITY "IMAGE_FULL"), 0 otherwise.  d_seed_slot / d_sink_slot: image slots of the
 * sources' and sinks' records (-1 = none), their rows already in the image.  Capacity errors ("LINKSTORE_FULL", "LOG_FULL",
 * "DEPTH_OVERFLOW") enlarge the engine's stores: run the batch again on every rank.  Result: as ldbg_engine_dfs_batch; `rec` of a
 * vertex is its image slot (>= 0: the vertex has a record). */
struct ldbg_dfs_result;
ldbg_status ldbg_engine_sharded_dfs_batch(struct ldbg_engine* e, ldbg_image* im, const char* sources, int64_t n, const char* sinks, const int64_t* sink_offsets,
                                          const int32_t* d_seed_slot, const int32_t* d_sink_slot, int (*round_done)(void* user), void* user,
                                          int64_t* d_stats, void* stream, struct ldbg_dfs_result** out);

/* ------------------------------------------------------------------ links: L3-L4
 * new CortexLinks(path) -> CortexLinksMap        J/utils/io/graph/links/CortexLinks.java:16-25,
 * CortexLinksIterable.java:49-226 (.ctp.gz text, JSON header v2/3/4).  Bound to a graph for k / device. */
ldbg_status ldbg_links_open(const char* path, const ldbg_graph* g, ldbg_links** out);
ldbg_status ldbg_links_close(ldbg_links* l);
/* IndexLinks: J/commands/index/links/IndexLinks.java:62-135.  The records of a link file (.ctp / .ctp.gz, v2-4) re-written as a BGZF file
 * (out_path, conventionally .ctp.bgz) with the big-endian LNKIDX index beside it (out_path + ".idx": per record the BGZF virtual offset
 * and the text length, ordered by k-mer string).  ldbg_links_open on out_path then reads every record through that index, as
 * CortexLinksRandomAccess does (CortexLinksRandomAccess.java:33-118), with that back-end's record semantics. */
ldbg_status ldbg_links_index(const char* in_path, const char* out_path, const char* source, int64_t* num_records);
ldbg_status ldbg_links_source(const ldbg_links* l, char* buf, int buflen);       /* ConnectivityAnnotations.getSource() */
ldbg_status ldbg_links_info(const ldbg_links* l, int* version, int* num_colors, int* k,
                            int64_t* num_kmers_in_graph, int64_t* num_kmers_with_links, int64_t* num_links);
ldbg_status ldbg_links_sample_name(const ldbg_links* l, int color, char* buf, int buflen);
/* ConnectivityAnnotations.containsKey / get     J/utils/io/graph/ConnectivityAnnotations.java:23-25
 * Writes the record as text "KMER n\n(F|R) len cov[,cov] junctions\n..." in the reference's HashSet
 * iteration order; *found = 0 and empty text when the k-mer has no links. */
ldbg_status ldbg_links_get(const ldbg_links* l, const char* kmer_ascii, int* found, char* buf, int64_t buflen);

/* ------------------------------------------------------------------ engine: C1, E1-E3, D1-D2, W1, S1-S3 */
typedef enum {   /* J/utils/stoppingrules/ (one enumerator per rule class) */
    LDBG_STOP_CONTIG = 0, LDBG_STOP_CYCLE_COLLAPSING_CONTIG, LDBG_STOP_DESTINATION, LDBG_STOP_EXPLORATION,
    LDBG_STOP_NOVEL_PARTITION, LDBG_STOP_NOVEL_KMER_LIMITED_CONTIG, LDBG_STOP_NOVEL_CONTINUATION,
    LDBG_STOP_BUBBLE_CLOSING, LDBG_STOP_BUBBLE_OPENING, LDBG_STOP_CONTAMINANT, LDBG_STOP_DUST,
    LDBG_STOP_GAP_CLOSING, LDBG_STOP_NAHR, LDBG_STOP_NOVEL_KMER_AGGREGATION, LDBG_STOP_ORPHAN,
    LDBG_STOP_PAIRED_READ_CLOSING, LDBG_STOP_TIP_BEGINNING, LDBG_STOP_TIP_END, LDBG_STOP_VISUALIZATION,
    LDBG_STOP_COUNT
} ldbg_stopper;
enum { LDBG_DIR_BOTH = 0, LDBG_DIR_FORWARD = 1, LDBG_DIR_REVERSE = 2 };   /* TraversalDirection */
enum { LDBG_OP_OR = 0, LDBG_OP_AND = 1 };                                 /* GraphCombinationOperator */
#define LDBG_MAX_COLORS 32

/* TraversalEngineConfiguration (J/utils/traversal/TraversalEngineConfiguration.java:19-37) as a POD;
 * ldbg_engine_config_default() fills the reference's defaults (BOTH, OR, ContigStopper, 75000). */
typedef struct {
    const ldbg_graph* graph;
    const ldbg_graph* rois;                 /* may be NULL */
    const ldbg_links* const* links;         /* nlinks entries; order = order of addition */
    int nlinks;
    int traversal_colors[LDBG_MAX_COLORS]; int n_traversal;      /* LinkedHashSet: insertion order */
    int joining_colors[LDBG_MAX_COLORS]; int n_joining;          /* TreeSet */
    int recruitment_colors[LDBG_MAX_COLORS]; int n_recruitment;  /* TreeSet */
    int secondary_colors[LDBG_MAX_COLORS]; int n_secondary;      /* TreeSet */
    int direction;                          /* LDBG_DIR_* */
    int combination_operator;               /* LDBG_OP_* */
    int stopping_rule;                      /* ldbg_stopper */
    int max_branch_length;                  /* maxLength, default 75000 */
    int connect_all_neighbors;
    int strict_java_flip;                   /* 1: CanonicalKmer.isFlipped by Arrays.hashCode inequality (Q6) */
} ldbg_engine_config;
void ldbg_engine_config_default(ldbg_engine_config* cfg);

/* TraversalEngineFactory.make()                 J/utils/traversal/TraversalEngineFactory.java:54-88
 * Threads: an engine is used by one host thread at a time (as a TraversalEngine object is in the reference: it holds the cursor's
 * state).  DIFFERENT engines — on one graph or on several — may be used from different host threads at the same time: every engine
 * over a resident table queues its work on a HIP stream of its own, and batches of two engines overlap on the device. */
ldbg_status ldbg_engine_create(const ldbg_engine_config* cfg, ldbg_engine** out);
ldbg_status ldbg_engine_destroy(ldbg_engine* e);

/* walk(seed) + TraversalUtils.toContig, for n seeds at once       TraversalEngine.java:108-110,
 * TraversalUtils.java:367-488.  seeds: n × k ASCII.  The contigs are written back to back into
 * contig_arena (ASCII); offsets[n+1]; walk_len[i] = number of vertices of walk i (0: empty walk,
 * empty contig).  kmers_traversed (may be NULL) receives the number of dfs loop iterations
 * (TraversalEngine.java:373: one per vertex a branch stands on) the batch performed (SURVEY §8d's unit).  If the arena is too small, LDBG_ERR_CAPACITY is
 * returned and offsets[n] holds the required size. */
ldbg_status ldbg_engine_walk_batch(ldbg_engine* e, const char* seeds, int64_t n,
                                   char* contig_arena, int64_t arena_capacity, int64_t* offsets,
                                   int64_t* walk_len, int64_t* kmers_traversed);
/* two-phase device form: run the batch (results stay in HBM), then query sizes / copy out */
ldbg_status ldbg_engine_walk_batch_run(ldbg_engine* e, const char* seeds, int64_t n,
                                       int64_t* total_contig_bytes, int64_t* kmers_traversed);
ldbg_status ldbg_engine_walk_batch_fetch(ldbg_engine* e, char* contig_arena, int64_t arena_capacity,
                                         int64_t* offsets, int64_t* walk_len);
/* the same batch with the seeds ALREADY IN DEVICE MEMORY (d_seeds: n x k ASCII bytes on the engine's device, e.g. the output of a seed
 * selection that ran there, or a torch uint8 CUDA tensor's data_ptr): nothing crosses the bus before the walks start.  The seeds must
 * stay valid until the call returns; work queued on other streams that produces them must have completed (the call does not wait
 * for foreign streams).  Results as after ldbg_engine_walk_batch_run. */
ldbg_status ldbg_engine_walk_batch_run_device(ldbg_engine* e, const void* d_seeds, int64_t n,
                                              int64_t* total_contig_bytes, int64_t* kmers_traversed);
/* Page-locked host memory for result arenas (contig_arena above): a download into it runs at the bus rate (C3: 780 MB of contigs in
 * about 16 ms); into ordinary memory the library stages the copy through its own page-locked buffers (about twice as long).  A JNI
 * host wraps the block in a direct ByteBuffer (NewDirectByteBuffer) and reads the contigs in place. */
ldbg_status ldbg_host_alloc(int64_t bytes, void** out);
ldbg_status ldbg_host_free(void* p);
/* vertices of walk i of the last batch: packed k-mer words (len × W), record index (-1 = null
 * CortexRecord), copyIndex, index (CortexVertex.java:20-35) */
ldbg_status ldbg_engine_walk_vertices(ldbg_engine* e, int64_t walk, int64_t capacity, int64_t* len,
                                      uint64_t* kmer_words, int64_t* rec, int32_t* copy_index, int32_t* index);

/* Which ROI k-mers the walks of the last batch pass through (the engine must have been made with a ROI graph):
 * what Partition.markUsedRois needs (J/commands/discover/call/Partition.java:238-257).  offsets[n+1] into hits (numbers
 * of ROI records; order within a walk is arbitrary; empty walks have none); has_null[i] != 0: the dfs graph of seed i
 * holds a vertex without a record (the reference's bookkeeping throws NullPointerException on those).  If hits is too
 * small, LDBG_ERR_CAPACITY is returned and offsets[n] holds the required number. */
ldbg_status ldbg_engine_walk_roi_hits(ldbg_engine* e, int64_t* offsets, uint32_t* hits, int64_t capacity, uint8_t* has_null);

/* dfs(source, sinks...) for n sources                               TraversalEngine.java:64-106, 356-482
 * sources n × k ASCII; sinks as CSR over ASCII k-mers (sink_offsets[n+1] counts k-mers; may be NULL). */
ldbg_status ldbg_engine_dfs_batch(ldbg_engine* e, const char* sources, int64_t n,
                                  const char* sinks, const int64_t* sink_offsets, ldbg_dfs_result** out);
/* result i: is_null = dfs returned null; vertices in insertion order; edges (src,dst index into vertices, colour).
 * The graphs (record numbers, copy indices, indices, edges) are complete when dfs_batch returns; the k-mer words of the
 * vertices are gathered from the device the first time they are asked for (kmer_words != NULL, or _walk), so the graph
 * handle must still be open then. */
ldbg_status ldbg_dfs_result_sizes(const ldbg_dfs_result* r, int64_t i, int* is_null, int64_t* n_vertices, int64_t* n_edges);
ldbg_status ldbg_dfs_result_get(const ldbg_dfs_result* r, int64_t i,
                                uint64_t* kmer_words, int64_t* rec, int32_t* copy_index, int32_t* index,
                                int32_t* edge_src, int32_t* edge_dst, int32_t* edge_color);
/* TraversalUtils.toWalk(g, seed, colour) + toContig on result i    TraversalUtils.java:367-488 */
ldbg_status ldbg_dfs_result_walk(const ldbg_dfs_result* r, int64_t i, const char* seed, int color,
                                 char* contig, int64_t capacity, int64_t* len);
/* dfs(Collection<String> sources, Collection<String> sinks)            TraversalEngine.java:37-62
 * = ldbg_engine_dfs_batch with every source given all the sinks, then this: the graphs of results which[0..m) of `r` that are not null,
 * merged in that order — the first as it is, each further one with Graphs.addGraph (vertices the merged graph does not hold yet; edges
 * refused when an equal CortexEdge is present).  *out is a result with ONE graph (index 0; null when every source returned null). */
ldbg_status ldbg_dfs_result_merge(struct ldbg_dfs_result* r, const int64_t* which, int64_t m, struct ldbg_dfs_result** out);
ldbg_status ldbg_dfs_result_free(ldbg_dfs_result* r);
ldbg_status ldbg_engine_dfs_kmers_traversed(const ldbg_engine* e, int64_t* n);

/* cursor: seek / next / previous / hasNext / hasPrevious          TraversalEngine.java:241-339
 * (stateful, batch of one; each call runs on the device) */
/* getNextVertices / getPrevVertices of n k-mers (n x k ASCII)            TraversalEngine.java:147-239
 * as CSR: offsets[n+1]; vertex j: packed k-mer words kmer_words[j*W .. ], rec[j] (record index, -1 = null CortexRecord).  The vertices of
 * one query come in the iteration order of the HashSet<CortexVertex> the reference returns.  capacity: vertices the arrays hold (4 n is
 * always enough).  A k-mer without a record has no neighbours — NullPointerException (Q14) when recruitment colours are configured. */
ldbg_status ldbg_engine_neighbours_batch(struct ldbg_engine* e, const char* kmers, int64_t n, int forward, int64_t* offsets,
                                         uint64_t* kmer_words, int64_t* rec, int64_t capacity);
/* assemble(seed)                                                        TraversalEngine.java:112-145
 * seek(seed), next() while hasNext() and fewer than maxBranchLength vertices; the same with previous(); result in contig order:
 * the previous() vertices (last one first), the seed's own vertex, the next() vertices.  *len = vertices; LDBG_ERR_CAPACITY with *len set
 * when the arrays are too small (call with capacity 0 to learn the length). */
ldbg_status ldbg_engine_assemble(struct ldbg_engine* e, const char* seed, int64_t capacity, int64_t* len, uint64_t* kmer_words, int64_t* rec);
ldbg_status ldbg_engine_seek(ldbg_engine* e, const char* kmer);
ldbg_status ldbg_engine_has_next(ldbg_engine* e, int* yes);
ldbg_status ldbg_engine_has_previous(ldbg_engine* e, int* yes);
ldbg_status ldbg_engine_next(ldbg_engine* e, char* kmer_out, int64_t* rec_out);
ldbg_status ldbg_engine_previous(ldbg_engine* e, char* kmer_out, int64_t* rec_out);

/* ------------------------------------------------------------------ measurement hooks (bench.py)
 * average device time (ms, HIP events on the launch stream) and launch count of the named kernel
 * family since the last reset: "find", "records", "walk", "dfs", "contig". */
ldbg_status ldbg_profile_reset(void);
ldbg_status ldbg_profile_get(const char* family, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* LDBG_H */
