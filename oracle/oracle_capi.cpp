// ORACLE — TEST INFRASTRUCTURE ONLY (see ldbg_oracle.hpp).
// Flat C entry points so that pytest / bench.py's cpu_baseline leg can drive the
// oracle through ctypes (oracle/pyoracle.py).
#include <cstring>
#include <string>

#include "ldbg_oracle.hpp"

using namespace orc;

namespace {
thread_local std::string g_err;
template <class F>
int guard(F f) {
    try { return f(); }
    catch (const CortexJDKException& e) { g_err = std::string("CortexJDKException: ") + e.what(); return -1; }
    catch (const JavaNullPointer& e) { g_err = std::string("NullPointerException: ") + e.what(); return -2; }
    catch (const NoSuchElement& e) { g_err = std::string("NoSuchElementException: ") + e.what(); return -3; }
    catch (const std::exception& e) { g_err = std::string("RuntimeException: ") + e.what(); return -4; }
}
struct Engine {
    TraversalEngine e;
    explicit Engine(const EngineConfig& c) : e(c) {}
};
struct DfsResult { std::unique_ptr<PGraph> g; int k; };
}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// ---- k-mer primitives
int orc_canonical(const char* in, char* out) { std::string s = canonical(in); memcpy(out, s.c_str(), s.size() + 1); return 0; }
int orc_revcomp(const char* in, char* out) { std::string s = reverse_complement(in); memcpy(out, s.c_str(), s.size() + 1); return 0; }
int orc_complement_char(int c) { return (unsigned char)complement((char)c); }
int32_t orc_jhash_bytes(const char* s) { return jhash_bytes(s); }
int32_t orc_jhash_string(const char* s) { return jhash_string(s); }
int orc_is_flipped(const char* s) { return CanonicalKmer(s).flipped ? 1 : 0; }
int orc_encode_kmer(const char* kmer, uint64_t* words) {
    return guard([&] {
        Record r; r.bk = Record::encode_binary_kmer(kmer);
        auto w = r.packed_words();
        for (size_t i = 0; i < w.size(); i++) words[i] = w[i];
        return (int)w.size();
    });
}
int orc_decode_kmer(const uint64_t* words, int k, char* out) {
    return guard([&] {
        int W = Record::kmer_bits(k);
        std::vector<int64_t> bk(W);
        for (int i = 0; i < W; i++) bk[i] = (int64_t)__builtin_bswap64(words[i]);
        std::string s = Record::decode_binary_kmer(bk, k, W);
        memcpy(out, s.c_str(), s.size() + 1);
        return 0;
    });
}
int orc_destination_junction_limit(int graph_size) { return destination_junction_limit(graph_size); }
int orc_java_string_hashmap_order(int n, const char** keys, int* order_out) {
    std::vector<std::string> k(keys, keys + n);
    auto o = java_string_hashmap_order(k);
    for (int i = 0; i < n; i++) order_out[i] = (int)o[i];
    return 0;
}

// ---- graph
void* orc_graph_open(const char* path, int use_cache) {
    CortexGraph* g = nullptr;
    int rc = guard([&] { g = new CortexGraph(path, use_cache != 0); return 0; });
    return rc == 0 ? g : nullptr;
}
void orc_graph_close(void* g) { delete (CortexGraph*)g; }
void orc_graph_set_tuned(void* g, int t) { ((CortexGraph*)g)->tuned = t != 0; }
int orc_graph_info(void* gp, int* k, int* W, int* C, int64_t* N, int64_t* data_offset) {
    auto* g = (CortexGraph*)gp;
    *k = g->k; *W = g->W; *C = g->C; *N = g->num_records; *data_offset = g->data_offset;
    return 0;
}
int orc_graph_sample_name(void* gp, int c, char* buf, int buflen) {
    auto* g = (CortexGraph*)gp;
    if (c < 0 || c >= g->C) return -1;
    snprintf(buf, buflen, "%s", g->colors[c].sample_name.c_str());
    return 0;
}
int orc_graph_color_for_sample_name(void* gp, const char* name) { return ((CortexGraph*)gp)->color_for_sample_name(name); }
int orc_graph_get_record(void* gp, int64_t i, uint64_t* words, int32_t* cov, uint8_t* edges) {
    return guard([&] {
        auto* g = (CortexGraph*)gp;
        Record r;
        if (!g->get_record(i, r)) return 0;
        auto w = r.packed_words();
        for (int j = 0; j < g->W; j++) words[j] = w[j];
        for (int c = 0; c < g->C; c++) { cov[c] = r.cov[c]; edges[c] = r.edges[c]; }
        return 1;
    });
}
int orc_graph_record_string(void* gp, int64_t i, char* buf, int buflen) {
    return guard([&] {
        Record r;
        if (!((CortexGraph*)gp)->get_record(i, r)) return 0;
        snprintf(buf, buflen, "%s", r.to_string().c_str());
        return 1;
    });
}
// returns 0 and *idx (-1 = null)
int orc_graph_find(void* gp, const char* kmer, int64_t* idx, int32_t* cov, uint8_t* edges) {
    return guard([&] {
        auto* g = (CortexGraph*)gp;
        Record r;
        bool ok = g->find_record(kmer, r, idx);
        if (ok && cov) for (int c = 0; c < g->C; c++) { cov[c] = r.cov[c]; edges[c] = r.edges[c]; }
        return 0;
    });
}
int orc_graph_find_batch(void* gp, const char* kmers, int64_t n, int64_t* idx_out, int tuned) {
    return guard([&] {
        auto* g = (CortexGraph*)gp;
        for (int64_t i = 0; i < n; i++) {
            std::string s(kmers + i * g->k, g->k);
            idx_out[i] = tuned ? g->find_record_tuned(s) : g->find_record(s);
        }
        return 0;
    });
}

// ---- fixtures
int orc_build_graph(const char* out_path, int k, int nsamples, const char** names, const int* nhaps, const char** haps_flat) {
    return guard([&] {
        std::vector<std::pair<std::string, std::vector<std::string>>> h;
        int p = 0;
        for (int s = 0; s < nsamples; s++) {
            std::vector<std::string> v;
            for (int i = 0; i < nhaps[s]; i++) v.push_back(haps_flat[p++]);
            h.push_back({names[s], v});
        }
        temp_graph_assembler(out_path, h, k);
        return 0;
    });
}
int orc_build_links(void* graph, const char* out_path, const char* sample, int nreads, const char** reads) {
    return guard([&] {
        std::vector<std::string> r(reads, reads + nreads);
        temp_links_assembler(*(CortexGraph*)graph, r, sample, out_path);
        return 0;
    });
}

// ---- links
void* orc_links_open(const char* path) {
    CortexLinks* l = nullptr;
    int rc = guard([&] { l = new CortexLinks(path); return 0; });
    return rc == 0 ? l : nullptr;
}
void orc_links_close(void* l) { delete (CortexLinks*)l; }
int orc_links_info(void* lp, int* version, int* ncolors, int* k, int64_t* nkg, int64_t* nkl, int64_t* nl) {
    auto* l = (CortexLinks*)lp;
    *version = l->version; *ncolors = l->num_colors; *k = l->k;
    *nkg = l->num_kmers_in_graph; *nkl = l->num_kmers_with_links; *nl = l->num_links;
    return 0;
}
// text dump: one "KMER n\n<junction lines>\n" block per record, file order, junctions in HashSet order
int64_t orc_links_dump(void* lp, char* buf, int64_t buflen) {
    auto* l = (CortexLinks*)lp;
    std::string s;
    for (auto& r : l->records) s += r.to_string() + "\n";
    if ((int64_t)s.size() + 1 > buflen) return -(int64_t)s.size() - 1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int64_t)s.size();
}

// ---- engine
void* orc_engine_create(void* graph, void* rois, void** links, int nlinks, const int* trav, int ntrav,
                        const int* join, int njoin, const int* recruit, int nrecruit, const int* secondary, int nsec,
                        int op_and, int direction, int connect_all, int max_length, int stopper) {
    Engine* e = nullptr;
    int rc = guard([&] {
        EngineConfig c;
        c.graph = (CortexGraph*)graph;
        c.rois = (CortexGraph*)rois;
        for (int i = 0; i < nlinks; i++) c.links.push_back((CortexLinks*)links[i]);
        c.traversal_colors.assign(trav, trav + ntrav);
        c.joining_colors.insert(join, join + njoin);
        c.recruitment_colors.insert(recruit, recruit + nrecruit);
        c.secondary_colors.insert(secondary, secondary + nsec);
        c.op_and = op_and != 0;
        c.direction = direction;
        c.connect_all_neighbors = connect_all != 0;
        c.max_length = max_length;
        c.stopper = stopper;
        e = new Engine(c);
        return 0;
    });
    return rc == 0 ? e : nullptr;
}
void orc_engine_destroy(void* e) { delete (Engine*)e; }
int orc_engine_seek(void* ep, const char* kmer) { return guard([&] { ((Engine*)ep)->e.seek(kmer); return 0; }); }
int orc_engine_has_next(void* ep) { return ((Engine*)ep)->e.has_next() ? 1 : 0; }
int orc_engine_has_previous(void* ep) { return ((Engine*)ep)->e.has_previous() ? 1 : 0; }
static int step(void* ep, bool fwd, char* kmer_out, int64_t* rec_out) {
    return guard([&] {
        Vertex v = fwd ? ((Engine*)ep)->e.next() : ((Engine*)ep)->e.previous();
        memcpy(kmer_out, v.sk.c_str(), v.sk.size() + 1);
        *rec_out = v.rec;
        return 0;
    });
}
int orc_engine_next(void* ep, char* kmer_out, int64_t* rec_out) { return step(ep, true, kmer_out, rec_out); }
int orc_engine_previous(void* ep, char* kmer_out, int64_t* rec_out) { return step(ep, false, kmer_out, rec_out); }
uint64_t orc_engine_kmers_traversed(void* ep) { return ((Engine*)ep)->e.dfs_iterations; }
uint64_t orc_engine_cursor_steps(void* ep) { return ((Engine*)ep)->e.cursor_steps; }

// walk(seed) -> contig string; *nverts = walk length (0 = empty walk, contig "")
int orc_engine_walk(void* ep, const char* seed, char* contig_out, int64_t cap, int64_t* len_out, int64_t* nverts) {
    return guard([&] {
        auto w = ((Engine*)ep)->e.walk(seed);
        std::string c = to_contig(w);
        *len_out = (int64_t)c.size();
        *nverts = (int64_t)w.size();
        if ((int64_t)c.size() + 1 > cap) return 1;
        memcpy(contig_out, c.c_str(), c.size() + 1);
        return 0;
    });
}
// batch of walks; seeds = n × k ASCII; arena receives the contigs back to back, offsets[n+1].
// vertex lists (optional): per walk, (rec, copy_index) of every walk vertex, with flip bit = k-mer != record k-mer
int orc_engine_walk_batch(void* ep, const char* seeds, int64_t n, char* arena, int64_t arena_cap, int64_t* offsets,
                          int64_t* walk_nverts) {
    return guard([&] {
        auto& e = ((Engine*)ep)->e;
        int k = e.config().graph->k;
        int64_t off = 0;
        offsets[0] = 0;
        for (int64_t i = 0; i < n; i++) {
            auto w = e.walk(std::string(seeds + i * k, k));
            std::string c = to_contig(w);
            if (off + (int64_t)c.size() > arena_cap) return 1;
            memcpy(arena + off, c.data(), c.size());
            off += (int64_t)c.size();
            offsets[i + 1] = off;
            if (walk_nverts) walk_nverts[i] = (int64_t)w.size();
        }
        return 0;
    });
}

// dfs(source, sinks...) -> result handle (graph may be null)
void* orc_engine_dfs(void* ep, const char* source, const char** sinks, int nsinks, int* status) {
    DfsResult* r = nullptr;
    *status = guard([&] {
        std::vector<std::string> s(sinks, sinks + nsinks);
        auto& e = ((Engine*)ep)->e;
        auto g = e.dfs(source, s);
        r = new DfsResult{std::move(g), e.config().graph->k};
        return 0;
    });
    return r;
}
// dfs(Collection<String> sources, Collection<String> sinks) (TraversalEngine.java:37-62): the first graph that comes back as it is, every
// further one merged into it with Graphs.addGraph
void* orc_engine_dfs_collection(void* ep, const char** sources, int nsources, const char** sinks, int nsinks, int* status) {
    DfsResult* r = nullptr;
    *status = guard([&] {
        std::vector<std::string> s(sinks, sinks + nsinks);
        auto& e = ((Engine*)ep)->e;
        std::unique_ptr<PGraph> dfs;
        for (int i = 0; i < nsources; i++) {
            auto g = e.dfs(sources[i], s);
            if (!g) continue;
            if (!dfs) dfs = std::move(g);
            else dfs->add_graph(*g);
        }
        r = new DfsResult{std::move(dfs), e.config().graph->k};
        return 0;
    });
    return r;
}
// getNextVertices / getPrevVertices (TraversalEngine.java:147-239) in the HashSet's iteration order; kmers_out: cap x k bytes
int orc_engine_adjacent(void* ep, const char* kmer, int forward, char* kmers_out, int64_t* rec_out, int cap, int* n_out) {
    return guard([&] {
        auto& e = ((Engine*)ep)->e;
        std::vector<Vertex> vs = forward ? e.next_vertices(kmer) : e.prev_vertices(kmer);
        *n_out = (int)vs.size();
        const size_t k = (size_t)e.config().graph->k;
        for (size_t i = 0; i < vs.size() && (int)i < cap; i++) { memcpy(kmers_out + i * k, vs[i].sk.data(), k); rec_out[i] = vs[i].rec; }
        return 0;
    });
}
int orc_result_is_null(void* rp) { return ((DfsResult*)rp)->g ? 0 : 1; }
int64_t orc_result_num_vertices(void* rp) { auto* r = (DfsResult*)rp; return r->g ? (int64_t)r->g->verts.size() : 0; }
int64_t orc_result_num_edges(void* rp) { auto* r = (DfsResult*)rp; return r->g ? (int64_t)r->g->edges.size() : 0; }
int orc_result_vertices(void* rp, char* kmers, int64_t* rec, int32_t* copy_index, int32_t* index) {
    auto* r = (DfsResult*)rp;
    if (!r->g) return 0;
    for (size_t i = 0; i < r->g->verts.size(); i++) {
        auto& v = r->g->verts[i];
        memcpy(kmers + i * r->k, v.sk.data(), r->k);
        rec[i] = v.rec; copy_index[i] = v.copy_index; index[i] = v.index;
    }
    return 0;
}
int orc_result_edges(void* rp, int32_t* src, int32_t* dst, int32_t* color) {
    auto* r = (DfsResult*)rp;
    if (!r->g) return 0;
    for (size_t i = 0; i < r->g->edges.size(); i++) {
        src[i] = r->g->edges[i].src; dst[i] = r->g->edges[i].dst; color[i] = r->g->edges[i].color;
    }
    return 0;
}
// toWalk(result, seed, color) -> contig
int orc_result_walk(void* ep, void* rp, const char* seed, int color, char* contig_out, int64_t cap, int64_t* len_out) {
    return guard([&] {
        auto* r = (DfsResult*)rp;
        auto w = to_walk(((Engine*)ep)->e, r->g.get(), seed, color);
        std::string c = to_contig(w);
        *len_out = (int64_t)c.size();
        if ((int64_t)c.size() + 1 > cap) return 1;
        memcpy(contig_out, c.c_str(), c.size() + 1);
        return 0;
    });
}
void orc_result_free(void* rp) { delete (DfsResult*)rp; }

}  // extern "C"
