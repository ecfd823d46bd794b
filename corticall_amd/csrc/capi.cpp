// extern "C" surface of libldbg.so (include/ldbg.h).  Every entry point converts C++ exceptions
// into status codes + a thread-local message; there is no CPU fallback behind any of them.
#include <string.h>

#include <memory>

#include "cursor.h"
#include "engine_host.h"
#include "shard.h"

using namespace ldbg;

namespace ldbg {
int64_t sort_ctx_file(const std::string& in_path, const std::string& out_path, int device);
int64_t subset_ctx_file(const std::string& in_path, const int64_t* indices, int64_t n, const std::string& out_path);
int64_t join_ctx_files(const std::vector<std::string>& paths, const std::string& out_path, int device);
std::vector<uint8_t> join_ctx_image(const std::vector<std::string>& paths, int device, bool find_view);
void profile_reset_all();
bool profile_get(const char* family, double* ms, int64_t* n);
}  // namespace ldbg

struct ldbg_graph {
    Graph g;
    ldbg_graph(const std::string& p, const void* img, int64_t n, int dev) : g(p, img, n, dev) {}
    ldbg_graph(const std::string& p, const void* hdr, int64_t hdr_bytes, const void* d_recs, int64_t n_recs, int dev) : g(p, hdr, hdr_bytes, d_recs, n_recs, dev) {}
};
struct ldbg_links { Links l; ldbg_links(const std::string& p, const Graph& g) : l(p, g) {} };
struct ldbg_engine {
    Engine e;
    std::unique_ptr<CursorHost> cursor;
    explicit ldbg_engine(const ldbg_engine_config& c) : e(c) {}
};
struct ldbg_dfs_result { std::unique_ptr<DfsBatch> b; };
struct ldbg_image { ShardImage img; ldbg_image(const Graph& shard, int64_t cap, int64_t global) : img(shard, cap, global) {} };

namespace {
thread_local std::string g_err;
template <class F>
ldbg_status guard(F f) {
    try { f(); return LDBG_OK; }
    catch (const StatusError& e) { g_err = e.what(); return (ldbg_status)e.status; }
    catch (const std::bad_alloc&) { g_err = "out of host memory"; return LDBG_ERR_HIP; }
    catch (const std::exception& e) { g_err = e.what(); return LDBG_ERR_ARG; }
}
// the engine config refers to graphs/links through opaque handles; unwrap them for Engine
ldbg_engine_config unwrap(const ldbg_engine_config& c, std::vector<const ldbg_links*>& keep) {
    ldbg_engine_config u = c;
    u.graph = c.graph ? (const ldbg_graph*)&c.graph->g : nullptr;
    u.rois = c.rois ? (const ldbg_graph*)&c.rois->g : nullptr;
    keep.clear();
    for (int i = 0; i < c.nlinks; i++) keep.push_back(c.links[i] ? (const ldbg_links*)&c.links[i]->l : nullptr);
    u.links = keep.data();
    return u;
}
}  // namespace

extern "C" {

const char* ldbg_last_error(void) { return g_err.c_str(); }
#ifndef LDBG_SRCID
#define LDBG_SRCID "unstamped"
#endif
const char* ldbg_version(void) { return "ldbg 0.2 (gfx950, src " LDBG_SRCID ")"; }
ldbg_status ldbg_device_count(int* count) { *count = rt::device_count(); return LDBG_OK; }

ldbg_status ldbg_kmer_encode(const char* ascii, int k, uint64_t* words_out) {
    return guard([&] {
        if (k <= 0 || k > 128) throw StatusError(LDBG_ERR_ARG, "bad k");
        if (!ascii_to_words_ci(ascii, k, words_out, (k + 31) / 32))        // encodeBinaryKmer takes either case
            throw StatusError(LDBG_ERR_ARG, "Nucleotide is not a valid character nucleotide");
    });
}
ldbg_status ldbg_kmer_decode(const uint64_t* words, int k, char* ascii_out) {
    return guard([&] {
        if (k <= 0 || k > 128) throw StatusError(LDBG_ERR_ARG, "bad k");
        words_to_ascii(words, k, (k + 31) / 32, ascii_out);
        ascii_out[k] = 0;
    });
}

ldbg_status ldbg_sort_ctx(const char* in_path, const char* out_path, int device, int64_t* num_records) {
    return guard([&] { const int64_t n = sort_ctx_file(in_path, out_path, device); if (num_records) *num_records = n; });
}

ldbg_status ldbg_ctx_write_records(const char* in_path, const int64_t* indices, int64_t n, const char* out_path) {
    return guard([&] { subset_ctx_file(in_path, indices, n, out_path); });
}

ldbg_status ldbg_join_ctx(const char* const* in_paths, int n_paths, const char* out_path, int device, int64_t* num_records) {
    return guard([&] {
        std::vector<std::string> paths;
        for (int i = 0; i < n_paths; i++) paths.push_back(in_paths[i]);
        const int64_t n = join_ctx_files(paths, out_path, device);
        if (num_records) *num_records = n;
    });
}

// ---- graph
ldbg_status ldbg_graph_open(const char* path, int device, ldbg_graph** out) {
    return guard([&] { *out = nullptr; *out = new ldbg_graph(path, nullptr, 0, device); });
}
ldbg_status ldbg_graph_open_memory(const void* image, int64_t nbytes, int device, ldbg_graph** out) {
    return guard([&] { *out = nullptr; *out = new ldbg_graph("<memory>", image, nbytes, device); });
}
ldbg_status ldbg_graph_open_device(const void* header, int64_t header_bytes, const void* d_records, int64_t n_records, int device, ldbg_graph** out) {
    return guard([&] { *out = nullptr; *out = new ldbg_graph("<device>", header, header_bytes, d_records, n_records, device); });
}
ldbg_status ldbg_graph_open_collection(const char* const* paths, int n_paths, int find_view, int device, ldbg_graph** out) {
    return guard([&] {
        *out = nullptr;
        std::vector<std::string> ps;
        for (int i = 0; i < n_paths; i++) ps.push_back(paths[i]);
        const std::vector<uint8_t> img = join_ctx_image(ps, device, find_view != 0);
        *out = new ldbg_graph("<collection>", img.data(), (int64_t)img.size(), device);
    });
}
ldbg_status ldbg_graph_close(ldbg_graph* g) { return guard([&] { delete g; }); }
ldbg_status ldbg_graph_info(const ldbg_graph* g, int* k, int* W, int* C, int64_t* N, int* version) {
    return guard([&] {
        if (k) *k = g->g.hdr.k;
        if (W) *W = g->g.hdr.W;
        if (C) *C = g->g.hdr.C;
        if (N) *N = g->g.hdr.num_records;
        if (version) *version = g->g.hdr.version;
    });
}
ldbg_status ldbg_graph_device(const ldbg_graph* g, int* device) { *device = g->g.device; return LDBG_OK; }
ldbg_status ldbg_graph_set_shard(ldbg_graph* g, int is_shard) {
    return guard([&] { g->g.view.java_tiny = (!is_shard && g->g.view.N <= 2) ? 1 : 0; });
}
ldbg_status ldbg_graph_sample_name(const ldbg_graph* g, int color, char* buf, int buflen) {
    return guard([&] {
        if (color < 0 || color >= g->g.hdr.C) throw StatusError(LDBG_ERR_ARG, "colour out of range");
        snprintf(buf, buflen, "%s", g->g.hdr.colors[color].sample_name.c_str());
    });
}
ldbg_status ldbg_graph_color_info(const ldbg_graph* g, int color, ldbg_color_info* out, char* name, int buflen) {
    return guard([&] {
        if (color < 0 || color >= g->g.hdr.C) throw StatusError(LDBG_ERR_ARG, "colour out of range");
        const CtxColor& c = g->g.hdr.colors[color];
        out->mean_read_length = c.mean_read_length;
        out->total_sequence = c.total_sequence;
        out->tip_clipping = c.tip_clipping;
        out->low_covg_supernodes_removed = c.low_covg_supernodes_removed;
        out->low_covg_kmers_removed = c.low_covg_kmers_removed;
        out->cleaned_against_graph = c.cleaned_against_graph;
        out->low_cov_supernodes_threshold = c.low_cov_supernodes_threshold;
        out->low_cov_kmer_threshold = c.low_cov_kmer_threshold;
        if (name) snprintf(name, buflen, "%s", c.cleaned_against_graph_name.c_str());
    });
}
ldbg_status ldbg_graph_color_for_sample_name(const ldbg_graph* g, const char* name, int* color) {
    return guard([&] { *color = g->g.color_for_sample_name(name); });
}

ldbg_status ldbg_graph_records_dev(const ldbg_graph* g, int64_t first, int64_t n, uint64_t* d_words, uint32_t* d_cov, uint8_t* d_edges, void* stream) {
    return guard([&] {
        if (first < 0 || n < 0 || first + n > g->g.hdr.num_records) throw StatusError(LDBG_ERR_ARG, "record range outside 0.." + std::to_string(g->g.hdr.num_records));
        rt::set_device(g->g.device);
        g->g.records_dev(first, n, d_words, d_cov, d_edges, stream ? (rt::stream_t)stream : g->g.stream);
    });
}
ldbg_status ldbg_graph_records(const ldbg_graph* g, int64_t first, int64_t n, uint64_t* words, uint32_t* cov, uint8_t* edges) {
    return guard([&] {
        if (first < 0 || n < 0 || first + n > g->g.hdr.num_records) throw StatusError(LDBG_ERR_ARG, "record range outside 0.." + std::to_string(g->g.hdr.num_records));
        if (n == 0) return;
        rt::set_device(g->g.device);
        const int W = g->g.hdr.W, C = g->g.hdr.C;
        rt::stream_t s = g->g.stream;
        uint64_t* dw = (uint64_t*)rt::dmalloc((size_t)n * W * 8);
        uint32_t* dc = cov ? (uint32_t*)rt::dmalloc((size_t)n * C * 4) : nullptr;
        uint8_t* de = edges ? (uint8_t*)rt::dmalloc((size_t)n * C) : nullptr;
        g->g.records_dev(first, n, dw, dc, de, s);
        if (words) rt::d2h(words, dw, (size_t)n * W * 8, s);
        if (cov) rt::d2h(cov, dc, (size_t)n * C * 4, s);
        if (edges) rt::d2h(edges, de, (size_t)n * C, s);
        rt::stream_sync(s);
        rt::dfree(dw); rt::dfree(dc); rt::dfree(de);
    });
}

ldbg_status ldbg_graph_find_dev(const ldbg_graph* g, const uint64_t* d_packed, int64_t n, int64_t* d_idx, uint32_t* d_cov, uint8_t* d_edges, void* stream) {
    return guard([&] {
        rt::set_device(g->g.device);
        g->g.find_dev(d_packed, n, d_idx, d_cov, d_edges, stream ? (rt::stream_t)stream : g->g.stream);
    });
}
ldbg_status ldbg_shard_owner_dev(int k, const uint64_t* d_packed, int64_t n, int world, uint64_t* d_canon, int32_t* d_owner, void* stream) {
    return guard([&] {
        shard_owner_dev(k, d_packed, n, world, d_canon, d_owner, (rt::stream_t)stream);
        rt::stream_sync((rt::stream_t)stream);
    });
}
ldbg_status ldbg_shard_owner(int k, const uint64_t* packed, int64_t n, int world, int device, int32_t* owner) {
    return guard([&] {
        if (n <= 0) return;
        if (rt::device_count() <= device) throw StatusError(LDBG_ERR_HIP, "no HIP device " + std::to_string(device) + " available (libldbg has no CPU fallback)");
        rt::set_device(device);
        const int W = (k + 31) / 32;
        rt::stream_t s = rt::stream_create();
        uint64_t* dq = (uint64_t*)rt::dmalloc((size_t)n * W * 8);
        int32_t* d_o = (int32_t*)rt::dmalloc((size_t)n * 4);
        rt::h2d(dq, packed, (size_t)n * W * 8, s);
        shard_owner_dev(k, dq, n, world, nullptr, d_o, s);
        rt::d2h(owner, d_o, (size_t)n * 4, s);
        rt::stream_sync(s);
        rt::dfree(dq); rt::dfree(d_o);
        rt::stream_destroy(s);
    });
}
ldbg_status ldbg_shard_nbr_queries(const ldbg_graph* g, int64_t first, int64_t n, uint64_t* d_words, uint8_t* d_flips, uint8_t* d_have) {
    return guard([&] { rt::set_device(g->g.device); shard_nbr_queries(g->g, first, n, d_words, d_flips, d_have); });
}
ldbg_status ldbg_shard_set_nbr(ldbg_graph* g, int64_t first, int64_t n, const int32_t* d_owner, const int64_t* d_lidx, const uint8_t* d_flips) {
    return guard([&] { rt::set_device(g->g.device); shard_set_nbr(g->g, first, n, d_owner, d_lidx, d_flips); });
}
// ---- the local image of a hash-sharded table (image.h) and walks over it, one bulk-synchronous round at a time
ldbg_status ldbg_image_create(const ldbg_graph* shard, int64_t cap, int64_t global_records, ldbg_image** out) {
    return guard([&] {
        *out = nullptr;
        if (cap < 1 || cap > ldbg::max_records_per_device()) throw StatusError(LDBG_ERR_ARG, "image capacity out of range");
        *out = new ldbg_image(shard->g, cap, global_records);
    });
}
ldbg_status ldbg_image_destroy(ldbg_image* im) { return guard([&] { delete im; }); }
ldbg_status ldbg_image_graph(ldbg_image* im, ldbg_graph** g) { return guard([&] { *g = reinterpret_cast<ldbg_graph*>(&im->img.graph()); }); }
ldbg_status ldbg_image_row_bytes(const ldbg_image* im, int* bytes) { return guard([&] { *bytes = im->img.row_bytes(); }); }
ldbg_status ldbg_image_clear(ldbg_image* im) { return guard([&] { rt::set_device(im->img.graph().device); im->img.clear(); }); }
static rt::stream_t image_stream(const ldbg_image* im, void* stream) { return stream ? (rt::stream_t)stream : im->img.graph().stream; }
ldbg_status ldbg_image_serve(const ldbg_image* im, int my_rank, const uint64_t* d_keys, int64_t n, uint8_t* d_out, void* stream) {
    return guard([&] { rt::set_device(im->img.graph().device); im->img.serve(my_rank, (const unsigned long long*)d_keys, n, 1, d_out, image_stream(im, stream)); });
}
ldbg_status ldbg_image_serve_chain(const ldbg_image* im, int my_rank, const uint64_t* d_keys, int64_t n, int depth, uint8_t* d_out, void* stream) {
    return guard([&] { rt::set_device(im->img.graph().device); im->img.serve(my_rank, (const unsigned long long*)d_keys, n, depth, d_out, image_stream(im, stream)); });
}
ldbg_status ldbg_image_insert(ldbg_image* im, const ldbg_engine* e, const uint8_t* d_rows, int64_t n, void* stream) {
    return guard([&] { rt::set_device(im->img.graph().device); im->img.insert(e ? &e->e : nullptr, d_rows, n, image_stream(im, stream)); });
}
ldbg_status ldbg_image_lookup(const ldbg_image* im, const uint64_t* d_keys, int64_t n, int32_t* d_slots, void* stream) {
    return guard([&] { rt::set_device(im->img.graph().device); im->img.lookup((const unsigned long long*)d_keys, n, d_slots, image_stream(im, stream)); });
}
ldbg_status ldbg_image_bucket(ldbg_image* im, int world, uint32_t cap_per_owner, uint64_t* d_send, void* stream) {
    return guard([&] { rt::set_device(im->img.graph().device); im->img.bucket(world, cap_per_owner, (unsigned long long*)d_send, image_stream(im, stream)); });
}
ldbg_status ldbg_image_request(ldbg_image* im, const uint64_t* d_keys, int64_t n, void* stream) {
    return guard([&] { rt::set_device(im->img.graph().device); im->img.request((const unsigned long long*)d_keys, n, image_stream(im, stream)); });
}
ldbg_status ldbg_image_reset_requests(ldbg_image* im, void* stream) {
    return guard([&] { rt::set_device(im->img.graph().device); im->img.reset_requests(image_stream(im, stream)); });
}
ldbg_status ldbg_image_counters(const ldbg_image* im, int64_t* n_rows, int64_t* n_req, int* overflow) {
    return guard([&] { rt::set_device(im->img.graph().device); im->img.counters(n_rows, n_req, overflow); });
}
ldbg_status ldbg_engine_sharded_walk_begin(ldbg_engine* e, ldbg_image* im, const char* seeds, int64_t n, const int32_t* d_seed_slot, void* stream) {
    return guard([&] { e->e.sharded_walk_begin(im->img, seeds, n, d_seed_slot, (rt::stream_t)stream); });
}
ldbg_status ldbg_engine_sharded_walk_round(ldbg_engine* e, int64_t* d_stats) { return guard([&] { e->e.sharded_walk_round(d_stats); }); }
ldbg_status ldbg_engine_sharded_walk_finish(ldbg_engine* e, int64_t* total_bytes, int64_t* traversed) {
    return guard([&] {
        try { e->e.sharded_walk_finish(total_bytes, traversed); }
        catch (const StatusError& se) {
            // a full per-walk link store: the next batch of this engine gets a larger one (the caller walks the batch again — exactness
            // is never traded away; ldbg_engine_walk_batch_run does the same retry inside the call)
            if (se.status == LDBG_ERR_CAPACITY && std::string(se.what()) == "LINKSTORE_FULL") e->e.link_store_capacity *= 4;
            throw;
        }
    });
}

ldbg_status ldbg_graph_find(const ldbg_graph* g, const uint64_t* packed, int64_t n, int64_t* idx_out, uint32_t* cov_out, uint8_t* edges_out) {
    return guard([&] {
        if (n <= 0) return;
        rt::set_device(g->g.device);
        const int W = g->g.hdr.W, C = g->g.hdr.C;
        rt::stream_t s = g->g.stream;
        uint64_t* dq = (uint64_t*)rt::dmalloc((size_t)n * W * 8);
        int64_t* di = (int64_t*)rt::dmalloc((size_t)n * 8);
        uint32_t* dc = cov_out ? (uint32_t*)rt::dmalloc((size_t)n * C * 4) : nullptr;
        uint8_t* de = edges_out ? (uint8_t*)rt::dmalloc((size_t)n * C) : nullptr;
        rt::h2d(dq, packed, (size_t)n * W * 8, s);
        g->g.find_dev(dq, n, di, dc, de, s);
        rt::d2h(idx_out, di, (size_t)n * 8, s);
        if (cov_out) rt::d2h(cov_out, dc, (size_t)n * C * 4, s);
        if (edges_out) rt::d2h(edges_out, de, (size_t)n * C, s);
        rt::stream_sync(s);
        rt::dfree(dq); rt::dfree(di); rt::dfree(dc); rt::dfree(de);
    });
}
ldbg_status ldbg_graph_find_ascii(const ldbg_graph* g, const char* kmers, int64_t n, int64_t* idx_out, uint32_t* cov_out, uint8_t* edges_out) {
    return guard([&] {
        rt::set_device(g->g.device);
        const int W = g->g.hdr.W, k = g->g.hdr.k, C = g->g.hdr.C;
        if (n <= 0) return;
        std::vector<uint64_t> packed((size_t)n * W);
        std::vector<uint8_t> valid((size_t)n);
        ascii_batch_to_words(kmers, n, k, W, packed.data(), valid.data());      // Q4: a string with a non-ACGT byte never matches
        rt::stream_t s = g->g.stream;
        struct Tmp { std::vector<void*> p; ~Tmp() { for (void* x : p) rt::dfree(x); } void* get(size_t nbytes) { void* x = rt::dmalloc(nbytes); p.push_back(x); return x; } } tmp;
        uint64_t* dq = (uint64_t*)tmp.get((size_t)n * W * 8);
        uint8_t* dv = (uint8_t*)tmp.get((size_t)n);
        int64_t* di = (int64_t*)tmp.get((size_t)n * 8);
        uint32_t* dc = cov_out ? (uint32_t*)tmp.get((size_t)n * C * 4) : nullptr;
        uint8_t* de = edges_out ? (uint8_t*)tmp.get((size_t)n * C) : nullptr;
        rt::h2d(dq, packed.data(), (size_t)n * W * 8, s);
        rt::h2d(dv, valid.data(), (size_t)n, s);
        g->g.find_dev(dq, n, di, dc, de, s, dv);
        rt::d2h(idx_out, di, (size_t)n * 8, s);
        if (cov_out) rt::d2h(cov_out, dc, (size_t)n * C * 4, s);
        if (edges_out) rt::d2h(edges_out, de, (size_t)n * C, s);
        rt::stream_sync(s);
    });
}

// ---- links
ldbg_status ldbg_links_open(const char* path, const ldbg_graph* g, ldbg_links** out) {
    return guard([&] { *out = nullptr; *out = new ldbg_links(path, g->g); });
}
ldbg_status ldbg_links_close(ldbg_links* l) { return guard([&] { delete l; }); }
ldbg_status ldbg_links_index(const char* in_path, const char* out_path, const char* source, int64_t* num_records) {
    return guard([&] { const int64_t n = links_index_file(in_path, out_path, source ? source : ""); if (num_records) *num_records = n; });
}
ldbg_status ldbg_links_source(const ldbg_links* l, char* buf, int buflen) { return guard([&] { snprintf(buf, buflen, "%s", l->l.source().c_str()); }); }
ldbg_status ldbg_links_info(const ldbg_links* l, int* version, int* num_colors, int* k, int64_t* nkg, int64_t* nkl, int64_t* nl) {
    return guard([&] {
        if (version) *version = l->l.version;
        if (num_colors) *num_colors = l->l.num_colors;
        if (k) *k = l->l.k;
        if (nkg) *nkg = l->l.num_kmers_in_graph;
        if (nkl) *nkl = l->l.num_kmers_with_links;
        if (nl) *nl = l->l.num_links;
    });
}
ldbg_status ldbg_links_sample_name(const ldbg_links* l, int color, char* buf, int buflen) {
    return guard([&] {
        if (color < 0 || color >= (int)l->l.sample_names.size()) throw StatusError(LDBG_ERR_ARG, "colour out of range");
        snprintf(buf, buflen, "%s", l->l.sample_names[color].c_str());
    });
}
ldbg_status ldbg_links_get(const ldbg_links* l, const char* kmer, int* found, char* buf, int64_t buflen) {
    return guard([&] {
        const HostLinksRecord* r = l->l.get(std::string(kmer, l->l.k));
        *found = r ? 1 : 0;
        std::string s;
        if (r) {
            s = r->kmer + " " + std::to_string(r->juncs.size()) + "\n";
            for (auto& j : r->juncs) {
                s += j.is_fw ? "F " : "R ";
                s += std::to_string(j.num_junctions) + " ";
                for (size_t c = 0; c < j.cov.size(); c++) { if (c) s += ","; s += std::to_string(j.cov[c]); }
                s += " " + j.junctions + "\n";
            }
        }
        if ((int64_t)s.size() + 1 > buflen) throw StatusError(LDBG_ERR_CAPACITY, "buffer too small: need " + std::to_string(s.size() + 1));
        memcpy(buf, s.c_str(), s.size() + 1);
    });
}

// ---- engine
void ldbg_engine_config_default(ldbg_engine_config* c) {
    memset(c, 0, sizeof(*c));
    c->direction = LDBG_DIR_BOTH;
    c->combination_operator = LDBG_OP_OR;
    c->stopping_rule = LDBG_STOP_CONTIG;
    c->max_branch_length = 75000;
    c->connect_all_neighbors = 0;
    c->strict_java_flip = 1;
}
ldbg_status ldbg_engine_create(const ldbg_engine_config* cfg, ldbg_engine** out) {
    return guard([&] {
        *out = nullptr;
        if (cfg->n_traversal > LDBG_MAX_COLORS || cfg->n_joining > LDBG_MAX_COLORS || cfg->n_recruitment > LDBG_MAX_COLORS || cfg->n_secondary > LDBG_MAX_COLORS)
            throw StatusError(LDBG_ERR_ARG, "too many colours");
        std::vector<const ldbg_links*> keep;
        ldbg_engine_config u = unwrap(*cfg, keep);
        *out = new ldbg_engine(u);
    });
}
ldbg_status ldbg_engine_destroy(ldbg_engine* e) { return guard([&] { delete e; }); }

static ldbg_status walk_batch_run_any(ldbg_engine* e, const char* seeds, int64_t n, int64_t* total_bytes, int64_t* traversed, bool on_device) {
    return guard([&] {
        // a full per-walk link store is retried with a larger one (exactness is never traded away)
        for (int attempt = 0;; attempt++) {
            try { e->e.walk_batch_run(seeds, n, total_bytes, traversed, on_device); return; }
            catch (const StatusError& se) {
                if (se.status == LDBG_ERR_CAPACITY && std::string(se.what()) == "LINKSTORE_FULL" && attempt < 8) { e->e.link_store_capacity *= 4; continue; }
                throw;
            }
        }
    });
}
ldbg_status ldbg_engine_walk_batch_run(ldbg_engine* e, const char* seeds, int64_t n, int64_t* total_bytes, int64_t* traversed) {
    return walk_batch_run_any(e, seeds, n, total_bytes, traversed, false);
}
ldbg_status ldbg_engine_walk_batch_run_device(ldbg_engine* e, const void* d_seeds, int64_t n, int64_t* total_bytes, int64_t* traversed) {
    if (n > 0 && !d_seeds) return guard([&] { throw StatusError(LDBG_ERR_ARG, "ldbg_engine_walk_batch_run_device: null seeds"); });
    return walk_batch_run_any(e, (const char*)d_seeds, n, total_bytes, traversed, true);
}
ldbg_status ldbg_host_alloc(int64_t bytes, void** out) {
    return guard([&] { *out = nullptr; if (bytes < 0) throw StatusError(LDBG_ERR_ARG, "ldbg_host_alloc: negative size"); *out = rt::hmalloc_pinned((size_t)bytes); });
}
ldbg_status ldbg_host_free(void* p) { return guard([&] { rt::hfree_pinned(p); }); }
ldbg_status ldbg_engine_walk_batch_fetch(ldbg_engine* e, char* arena, int64_t cap, int64_t* offsets, int64_t* walk_len) {
    return guard([&] { e->e.walk_batch_fetch(arena, cap, offsets, walk_len); });
}
ldbg_status ldbg_engine_walk_batch(ldbg_engine* e, const char* seeds, int64_t n, char* arena, int64_t cap, int64_t* offsets, int64_t* walk_len, int64_t* traversed) {
    int64_t total = 0;
    ldbg_status st = ldbg_engine_walk_batch_run(e, seeds, n, &total, traversed);
    if (st != LDBG_OK) return st;
    return ldbg_engine_walk_batch_fetch(e, arena, cap, offsets, walk_len);
}
ldbg_status ldbg_engine_walk_vertices(ldbg_engine* e, int64_t walk, int64_t capacity, int64_t* len, uint64_t* words, int64_t* rec, int32_t* copy, int32_t* index) {
    return guard([&] { e->e.walk_vertices(walk, capacity, len, words, rec, copy, index); });
}

ldbg_status ldbg_engine_walk_roi_hits(ldbg_engine* e, int64_t* offsets, uint32_t* hits, int64_t capacity, uint8_t* has_null) {
    return guard([&] { e->e.walk_roi_hits(offsets, hits, capacity, has_null); });
}

ldbg_status ldbg_engine_dfs_batch(ldbg_engine* e, const char* sources, int64_t n, const char* sinks, const int64_t* sink_offsets, ldbg_dfs_result** out) {
    return guard([&] {
        *out = nullptr;
        // a full per-walk link store or frame stack is retried with a larger one (exactness is never traded away)
        for (int attempt = 0;; attempt++) {
            try {
                std::unique_ptr<DfsBatch> b(e->e.dfs_batch(sources, n, sinks, sink_offsets));
                *out = new ldbg_dfs_result{std::move(b)};
                return;
            } catch (const StatusError& se) {
                const std::string what = se.what();
                if (se.status == LDBG_ERR_CAPACITY && what == "LINKSTORE_FULL" && attempt < 8) { e->e.link_store_capacity *= 4; continue; }
                if (se.status == LDBG_ERR_CAPACITY && what == "LOG_FULL" && e->e.dfs_log_blocks < (1 << 20)) { e->e.dfs_log_blocks *= 8; continue; }
                if (se.status == LDBG_ERR_CAPACITY && what == "DEPTH_OVERFLOW" && e->e.dfs_max_depth < 4096) { e->e.dfs_max_depth *= 4; continue; }
                if (se.status == LDBG_ERR_CAPACITY && what == "DEPTH_OVERFLOW")
                    throw StatusError(LDBG_ERR_UNSUPPORTED, "dfs recursion deeper than 4096 branches (the reference overflows its thread stack here)");
                throw;
            }
        }
    });
}
// dfs_batch over the local image of a hash-sharded table: the library runs the bulk-synchronous rounds and calls `round_done` after
// each (the caller exchanges requests and rows there, and returns non-zero once no rank has a search in progress)
ldbg_status ldbg_engine_sharded_dfs_batch(ldbg_engine* e, ldbg_image* im, const char* sources, int64_t n, const char* sinks, const int64_t* sink_offsets,
                                          const int32_t* d_seed_slot, const int32_t* d_sink_slot, int (*round_done)(void*), void* user,
                                          int64_t* d_stats, void* stream, ldbg_dfs_result** out) {
    return guard([&] {
        *out = nullptr;
        if (&im->img.graph() != e->e.graph) throw StatusError(LDBG_ERR_ARG, "the engine was not created on this image's graph");
        ShardedRun sr{&im->img, d_seed_slot, d_sink_slot, round_done, user, d_stats, (rt::stream_t)stream};
        try {
            std::unique_ptr<DfsBatch> b(e->e.dfs_batch(sources, n, sinks, sink_offsets, &sr));
            b->materialize();                         // the vertices' k-mers and coverages live in the image: read them before it changes
            *out = new ldbg_dfs_result{std::move(b)};
        } catch (const StatusError& se) {
            // capacities grow for the next attempt (the caller runs the batch again on every rank: exactness is never traded away)
            const std::string what = se.what();
            if (se.status == LDBG_ERR_CAPACITY && what == "LINKSTORE_FULL") e->e.link_store_capacity *= 4;
            if (se.status == LDBG_ERR_CAPACITY && what == "LOG_FULL" && e->e.dfs_log_blocks < (1 << 20)) e->e.dfs_log_blocks *= 8;
            if (se.status == LDBG_ERR_CAPACITY && what == "DEPTH_OVERFLOW" && e->e.dfs_max_depth < 4096) e->e.dfs_max_depth *= 4;
            throw;
        }
    });
}
static const DfsGraphHost& dfs_at(const ldbg_dfs_result* r, int64_t i) {
    if (!r || !r->b) throw StatusError(LDBG_ERR_ARG, "no dfs result");
    if (i < 0 || i >= (int64_t)r->b->results.size()) throw StatusError(LDBG_ERR_ARG, "dfs result index out of range");
    return r->b->results[(size_t)i];
}
ldbg_status ldbg_dfs_result_sizes(const ldbg_dfs_result* r, int64_t i, int* is_null, int64_t* n_vertices, int64_t* n_edges) {
    return guard([&] {
        const DfsGraphHost& g = dfs_at(r, i);
        if (is_null) *is_null = g.is_null ? 1 : 0;
        if (n_vertices) *n_vertices = g.n_vertices();
        if (n_edges) *n_edges = g.n_edges();
    });
}
ldbg_status ldbg_dfs_result_get(const ldbg_dfs_result* r, int64_t i, uint64_t* kmer_words, int64_t* rec, int32_t* copy_index, int32_t* index,
                                int32_t* edge_src, int32_t* edge_dst, int32_t* edge_color) {
    return guard([&] {
        dfs_at(r, i);
        if (kmer_words) r->b->materialize();
        const DfsGraphHost& g = dfs_at(r, i);
        if (g.packed) { r->b->read_packed(i, rec, copy_index, index, edge_src, edge_dst, edge_color); return; }    // (no k-mers asked for: straight from the packed entries)
        if (kmer_words && !g.words.empty()) memcpy(kmer_words, g.words.data(), g.words.size() * 8);
        for (size_t v = 0; v < g.verts.size(); v++) {
            if (rec) rec[v] = g.verts[v].rec;
            if (copy_index) copy_index[v] = g.verts[v].copy;
            if (index) index[v] = g.verts[v].index;
        }
        for (size_t x = 0; x < g.edges.size(); x++) {
            if (edge_src) edge_src[x] = g.edges[x].src;
            if (edge_dst) edge_dst[x] = g.edges[x].dst;
            if (edge_color) edge_color[x] = g.edges[x].color;
        }
    });
}
ldbg_status ldbg_dfs_result_walk(const ldbg_dfs_result* r, int64_t i, const char* seed, int color, char* contig, int64_t capacity, int64_t* len) {
    return guard([&] {
        dfs_at(r, i);
        const std::string c = r->b->walk_contig(i, seed, color);
        if (len) *len = (int64_t)c.size();
        if ((int64_t)c.size() + 1 > capacity) throw StatusError(LDBG_ERR_CAPACITY, "contig buffer too small: need " + std::to_string(c.size() + 1));
        memcpy(contig, c.c_str(), c.size() + 1);
    });
}
ldbg_status ldbg_dfs_result_merge(ldbg_dfs_result* r, const int64_t* which, int64_t m, ldbg_dfs_result** out) {
    return guard([&] {
        *out = nullptr;
        std::unique_ptr<DfsBatch> b(dfs_merge(*r->b, which, m));
        *out = new ldbg_dfs_result{std::move(b)};
    });
}
ldbg_status ldbg_dfs_result_free(ldbg_dfs_result* r) { delete r; return LDBG_OK; }
ldbg_status ldbg_engine_neighbours_batch(ldbg_engine* e, const char* kmers, int64_t n, int forward, int64_t* offsets, uint64_t* kmer_words, int64_t* rec, int64_t capacity) {
    return guard([&] { e->e.neighbours_batch(kmers, n, forward != 0, offsets, kmer_words, rec, capacity); });
}
ldbg_status ldbg_engine_dfs_kmers_traversed(const ldbg_engine* e, int64_t* n) { return guard([&] { *n = e->e.dfs_traversed(); }); }

// ---- cursor
static CursorHost& cursor_of(ldbg_engine* e) {
    if (!e->cursor) e->cursor.reset(new CursorHost(e->e));
    return *e->cursor;
}
ldbg_status ldbg_engine_assemble(ldbg_engine* e, const char* seed, int64_t capacity, int64_t* len, uint64_t* kmer_words, int64_t* rec) {
    return guard([&] { cursor_of(e).assemble(seed, capacity, len, kmer_words, rec); });
}
ldbg_status ldbg_engine_seek(ldbg_engine* e, const char* kmer) { return guard([&] { cursor_of(e).seek(kmer); }); }
ldbg_status ldbg_engine_has_next(ldbg_engine* e, int* yes) { return guard([&] { *yes = cursor_of(e).has(true) ? 1 : 0; }); }
ldbg_status ldbg_engine_has_previous(ldbg_engine* e, int* yes) { return guard([&] { *yes = cursor_of(e).has(false) ? 1 : 0; }); }
ldbg_status ldbg_engine_next(ldbg_engine* e, char* kmer_out, int64_t* rec_out) { return guard([&] { cursor_of(e).step(true, kmer_out, rec_out); }); }
ldbg_status ldbg_engine_previous(ldbg_engine* e, char* kmer_out, int64_t* rec_out) { return guard([&] { cursor_of(e).step(false, kmer_out, rec_out); }); }

#ifdef LDBG_HOSTSIM
void ldbg_debug_ls(uint64_t* out) {
    auto& d = ldbg::ls_debug();
    out[0] = d.adds; out[1] = d.newkeys; out[2] = d.choices; out[3] = d.scan; out[4] = d.maxn; out[5] = d.steps; out[6] = d.sum_n;
    out[7] = d.runs_a; out[8] = d.runs_b; out[9] = d.run_vertices; out[10] = d.retries; out[11] = d.repeats;
    d = ldbg::LsDebug();
}
#endif
// ---- measurement
#ifdef LDBG_HOSTSIM
// test hook of the host simulation (rt.h): lanes a simulated wavefront runs in lock step (1, 2, 4, ... 64)
extern "C" void ldbg_hostsim_set_lanes(int n) { int v = 1; while (v * 2 <= n && v < 64) v *= 2; ::ldbg::sim::lanes_setting() = v; }
#endif
ldbg_status ldbg_profile_reset(void) { profile_reset_all(); return LDBG_OK; }
ldbg_status ldbg_profile_get(const char* family, double* total_ms, int64_t* launches) {
    profile_get(family, total_ms, launches);
    return LDBG_OK;
}

}  // extern "C"
