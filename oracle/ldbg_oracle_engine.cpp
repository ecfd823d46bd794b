// ORACLE — TEST INFRASTRUCTURE ONLY (see ldbg_oracle.hpp).
// Part 2: LinkStore, stopping rules, TraversalEngine (cursor + recursive DFS), toWalk/toContig.
#include <algorithm>
#include <cmath>
#include <functional>
#include <tuple>

#include "ldbg_oracle.hpp"

namespace orc {

// ------------------------------------------------------------------ LinkStore (J/utils/traversal/LinkStore.java)
void LinkStore::add(const std::string& cur_kmer, const LinksRecord& clr, bool go_forward) {
    bool matches = clr.kmer == cur_kmer;                        // :18
    for (const JunctionsRecord& cjr : clr.junctions()) {         // HashSet order (L3)
        bool lgf = matches == cjr.is_fw;                         // :24
        std::string jl = lgf ? cjr.junctions : complement_str(cjr.junctions);   // complemented, not reversed
        if (lgf == go_forward) {
            Key* key = nullptr;
            for (auto& kk : keys_) if (kk.jl == jl) { key = &kk; break; }
            if (!key) {
                keys_.push_back({jl, jhash_string(jl), seq_++, {}});
                key = &keys_.back();
                if (cap_ == 0) cap_ = 16;
                size_++;
                if (size_ > (size_t)cap_ * 3 / 4) cap_ *= 2;     // HashMap.resize()
            }
            key->elems.push_back(Elem{});
        }
    }
}
std::vector<size_t> LinkStore::order() const {
    std::vector<size_t> idx(keys_.size());
    for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
    int cap = cap_ ? cap_ : 16;
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
        uint32_t ba = jhashmap_bucket(keys_[a].hash, cap), bb = jhashmap_bucket(keys_[b].hash, cap);
        if (ba != bb) return ba < bb;
        return keys_[a].seq < keys_[b].seq;
    });
    return idx;
}
void LinkStore::increment_ages() {
    for (auto& k : keys_) for (auto& e : k.elems) e.age++;
}
int LinkStore::num_new_paths() const {
    int n = 0;
    for (auto& k : keys_) for (auto& e : k.elems) if (e.age == 0) n++;
    return n;
}
int LinkStore::size() const {
    int n = 0;
    for (auto& k : keys_) n += (int)k.elems.size();
    return n;
}
bool LinkStore::oldest_link(size_t& key_index) const {
    int age = INT32_MIN;
    for (auto& k : keys_) for (auto& e : k.elems) if (e.age > age) age = e.age;
    bool have_first = false;
    std::set<char> choices;
    for (size_t ki : order()) {
        const Key& k = keys_[ki];
        for (auto& e : k.elems) {
            if (e.age == age) {
                if (!have_first) { have_first = true; key_index = ki; }
                if (e.pos + 1 <= (int)k.jl.size()) choices.insert(k.jl[e.pos]);
            }
        }
    }
    return choices.size() == 1;
}
void LinkStore::increment_positions_and_expire(char choice) {
    for (auto& k : keys_) {
        std::vector<Elem> keep;
        for (auto& e : k.elems) {
            if (e.pos + 1 >= (int)k.jl.size() || k.jl[e.pos] != choice) continue;   // expire
            Elem n = e; n.pos++; keep.push_back(n);
        }
        k.elems.swap(keep);
    }
    for (size_t i = 0; i < keys_.size();) {
        if (keys_[i].elems.empty()) { keys_.erase(keys_.begin() + i); size_--; } else i++;
    }
}
bool LinkStore::next_junction_choice(char& choice) {
    size_t ki = 0;
    if (!oldest_link(ki)) return false;
    const Key& k = keys_[ki];
    for (auto& e : k.elems) choice = k.jl.at(e.pos);     // last element of that key's list wins (:129-133)
    increment_positions_and_expire(choice);
    return true;
}

// ------------------------------------------------------------------ graph container
size_t VertexHash::operator()(const Vertex& v) const {
    size_t h = std::hash<std::string>()(v.sk);
    h ^= std::hash<int64_t>()(v.rec * 1315423911LL + v.copy_index * 2654435761LL + v.index);
    return h;
}
int PGraph::find_vertex(const Vertex& v) const {
    auto it = vmap_.find(v);
    return it == vmap_.end() ? -1 : it->second;
}
int PGraph::add_vertex(const Vertex& v) {
    auto it = vmap_.find(v);
    if (it != vmap_.end()) return it->second;
    verts.push_back(v);
    vmap_[v] = (int)verts.size() - 1;
    return (int)verts.size() - 1;
}
bool PGraph::contains_edge(int s, int t) const { return dir_.count({s, t}) != 0; }
bool PGraph::add_edge(int s, int t, int color) {
    auto key = std::make_tuple(std::min(s, t), std::max(s, t), color);   // CortexEdge.equals: {s,t} as a set + colour (+weight 1.0)
    if (und_.count(key)) return false;
    und_.insert(key);
    dir_.insert({s, t});
    edges.push_back({s, t, color});
    return true;
}
void PGraph::add_graph(const PGraph& o) {
    for (auto& v : o.verts) add_vertex(v);
    for (auto& e : o.edges) {
        int s = add_vertex(o.verts[e.src]);
        int t = add_vertex(o.verts[e.dst]);
        add_edge(s, t, e.color);
    }
}

// ------------------------------------------------------------------ stopping rules
int destination_junction_limit(int graph_size) {
    // 1 + ceil(5 * exp(-1e-4 * size)), DestinationStopper.java:15-17, as an exact integer table
    if (graph_size <= 2231) return 6;
    if (graph_size <= 5108) return 5;
    if (graph_size <= 9162) return 4;
    if (graph_size <= 16094) return 3;
    if (graph_size <= 7451332) return 2;
    return 1;   // exp() underflows to 0.0 (glibc libm; unreachable in practice: maxLength caps branches at 75,000)
}

class StoppingRule {
public:
    StoppingRule(int id, TraversalEngine& e, const std::unordered_set<std::string>* roi_set)
        : id_(id), e_(e), ec_(e.config()), roi_set_(roi_set) {}
    // AbstractTraversalStoppingRule.keepGoing :9-15 — both predicates always evaluated
    bool keep_going(const TraversalState& s) {
        succeeded_ = has_succeeded(s);
        failed_ = has_failed(s);
        return !succeeded_ && !failed_;
    }
    bool traversal_succeeded() const { return succeeded_; }
    bool has_succeeded(const TraversalState& s);
    bool has_failed(const TraversalState& s);

private:
    int id_;
    TraversalEngine& e_;
    const EngineConfig& ec_;
    const std::unordered_set<std::string>* roi_set_;
    bool succeeded_ = false, failed_ = false;
    // per-instance state of the stateful rules
    bool found_novel_ = false, started_novel_ = false, has_joined_ = false, seen_novel_agg_ = false;
    int distance_ = 0, num_seen_ = 0, novel_seen_ = 0, since_join_ = 0, since_low_ = 0;
    std::set<std::string> canon_sinks_;

    Record rec(const TraversalState& s) {
        if (s.cur->rec < 0) throw JavaNullPointer("stopper dereferenced a null CortexRecord");
        return e_.record_of(*s.cur);
    }
    CortexGraph& rois() {
        if (!ec_.rois) throw JavaNullPointer("stopper dereferenced null rois");
        return *ec_.rois;
    }
    bool joined(const TraversalState& s) {
        bool r = false;
        if (ec_.joining_colors.empty()) return false;
        Record cr = rec(s);
        for (int c : ec_.joining_colors) r |= cr.cov[c] > 0;
        return r;
    }
    void degrees(const TraversalState& s, bool& no_in, bool& no_out) {
        no_in = no_out = false;
        if (ec_.traversal_colors.empty()) return;
        Record cr = rec(s);
        for (int c : ec_.traversal_colors) { no_in |= cr.in_degree(c) == 0; no_out |= cr.out_degree(c) == 0; }
    }
    bool roi_canonical_contains(const TraversalState& s) {
        if (!ec_.rois) throw CortexJDKException("This stopper requires a list of novel kmers be provided.");
        if (s.cur->rec < 0) return false;   // getCanonicalKmer() == null; HashSet.contains(null) == false
        return roi_set_->count(rec(s).kmer_string()) != 0;
    }
    bool novel_stop_now(const TraversalState& s) const {
        return distance_ > 2000 || s.junction_depth > 0 || s.reached_max || s.adj == 0 || (s.adj > 1 && s.children_traversed);
    }
};

bool StoppingRule::has_succeeded(const TraversalState& s) {
    switch (id_) {
        case CONTIG: return s.adj != 1 || s.reached_max;
        case CYCLE_COLLAPSING_CONTIG: return s.adj == 0;
        case DESTINATION:
            return std::find(s.sinks->begin(), s.sinks->end(), s.cur->sk) != s.sinks->end();
        case EXPLORATION: return s.reached_max || s.adj == 0 || s.junction_depth >= 3;
        case NOVEL_PARTITION: {
            distance_++;
            if (roi_canonical_contains(s)) { found_novel_ = true; distance_ = 0; }
            return found_novel_ && novel_stop_now(s);
        }
        case NOVEL_KMER_LIMITED_CONTIG: {
            distance_++;
            if (roi_canonical_contains(s)) { found_novel_ = true; distance_ = 0; }
            bool stop_now = distance_ > 2000 || s.adj != 1 || s.reached_max;
            return found_novel_ && stop_now;
        }
        case NOVEL_CONTINUATION: {
            if (s.junction_depth > 0 && num_seen_ <= 2 * (int)s.cur->sk.size() && rois().find_record(s.cur->sk) >= 0)
                started_novel_ = true;
            num_seen_++;
            return (s.children_traversed && s.adj != 1) || s.reached_max;
        }
        case BUBBLE_CLOSING: return false;
        case BUBBLE_OPENING: {
            if (rois().find_record(s.cur->sk) >= 0) novel_seen_++;
            if (has_joined_) since_join_++;
            has_joined_ |= joined(s);
            return novel_seen_ > 0 && has_joined_ && (since_join_ >= 30 || s.adj != 1);
        }
        case CONTAMINANT: {
            bool parents = joined(s);
            return s.cur->rec >= 0 && (parents || s.adj == 0);
        }
        case DUST: {
            bool ni, no; degrees(s, ni, no);
            bool reunion = joined(s);
            return ni || no || reunion;
        }
        case GAP_CLOSING: return false;
        case NAHR: {
            if (found_novel_) distance_++;
            if (rois().find_record(rec(s).kmer_string()) >= 0) { found_novel_ = true; distance_++; }
            return found_novel_ && (distance_ >= 1000 || s.junction_depth >= 5 || s.adj == 0 || s.children_traversed);
        }
        case NOVEL_KMER_AGGREGATION: {
            bool child = false;
            if (!ec_.traversal_colors.empty()) { Record cr = rec(s); for (int c : ec_.traversal_colors) child |= cr.cov[c] > 0; }
            bool parents = joined(s);
            if (child && !parents) seen_novel_agg_ = true;
            return seen_novel_agg_ && parents;
        }
        case ORPHAN: case TIP_END: { bool ni, no; degrees(s, ni, no); return ni || no; }
        case PAIRED_READ_CLOSING: {
            if (!s.sinks->empty() && canon_sinks_.empty())
                for (auto& sk : *s.sinks) canon_sinks_.insert(canonical(sk));
            if (s.cur->rec < 0) return false;
            return canon_sinks_.count(rec(s).kmer_string()) != 0;
        }
        case TIP_BEGINNING: return joined(s);
        case VISUALIZATION: return s.adj == 0 || s.junction_depth > 2 || s.branch_size > 500;
    }
    throw CortexJDKException("Could not instantiate stoppingRule");
}

bool StoppingRule::has_failed(const TraversalState& s) {
    switch (id_) {
        case CONTIG: case CYCLE_COLLAPSING_CONTIG: case EXPLORATION: case NOVEL_KMER_LIMITED_CONTIG:
        case VISUALIZATION:
            return false;
        case DESTINATION:
            return s.junction_depth > destination_junction_limit(s.graph_size) || s.reached_max;
        case NOVEL_PARTITION: return !found_novel_ && novel_stop_now(s);
        case NOVEL_CONTINUATION: return (s.junction_depth > 0 && !started_novel_) || s.junction_depth > 3;
        case BUBBLE_CLOSING: return s.branch_size > 10000 || s.junction_depth >= 2 || s.adj == 0;
        case BUBBLE_OPENING: return novel_seen_ == 0 && (s.junction_depth >= 5 || s.adj == 0);
        case CONTAMINANT: { bool parents = joined(s); return s.cur->rec >= 0 && parents; }
        case DUST: {
            bool low = false;
            if (!ec_.traversal_colors.empty()) {
                Record cr = rec(s);
                for (int c : ec_.traversal_colors) low |= cr.in_degree(c) + cr.out_degree(c) > 4;
            }
            if (low) since_low_ = 0; else since_low_++;
            return since_low_ >= rec(s).k;
        }
        case GAP_CLOSING: return s.junction_depth > 5 || s.adj == 0;
        case NAHR: return !found_novel_ && (s.branch_size >= 1000 || s.junction_depth >= 2 || s.adj == 0);
        case NOVEL_KMER_AGGREGATION: return !seen_novel_agg_ && (s.branch_size >= 100 || s.junction_depth >= 3);
        case ORPHAN: case TIP_END: return joined(s);
        case PAIRED_READ_CLOSING: return s.junction_depth >= 5 || s.adj == 0 || s.reached_max;
        case TIP_BEGINNING: { bool ni, no; degrees(s, ni, no); return ni || no; }
    }
    return true;
}

// ------------------------------------------------------------------ TraversalEngine
TraversalEngine::TraversalEngine(const EngineConfig& cfg) : ec_(cfg) {
    // TraversalEngineFactory.make :54-88
    if (ec_.traversal_colors.empty()) throw CortexJDKException("Traversal color(s) must be specified.");
    if (!ec_.graph) throw CortexJDKException("Must provide graph to traverse.");
    int nc = ec_.graph->C;
    for (int c : ec_.traversal_colors)
        if (c >= nc) throw CortexJDKException("Traversal colors must be between 0 and " + std::to_string(nc) + " (provided " + std::to_string(c) + ")");
    auto chk = [&](const std::set<int>& s, const char* what) {
        for (int c : s) if (c < 0 || c >= nc)
            throw CortexJDKException(std::string(what) + " colors must be between 0 and " + std::to_string(nc) + " (provided " + std::to_string(c) + ")");
    };
    chk(ec_.joining_colors, "Joining");
    chk(ec_.recruitment_colors, "Recruitment");
    chk(ec_.secondary_colors, "Secondary");
    if (ec_.stopper < 0 || ec_.stopper >= NUM_STOPPERS) throw CortexJDKException("Must provide stopping rule for graph traversal");
    // LinkedHashSet semantics for traversal colours
    std::vector<int> uniq;
    for (int c : ec_.traversal_colors) if (std::find(uniq.begin(), uniq.end(), c) == uniq.end()) uniq.push_back(c);
    ec_.traversal_colors = uniq;
}

Record TraversalEngine::record_of(const Vertex& v) {
    Record r;
    if (v.rec >= 0) ec_.graph->get_record(v.rec, r);
    return r;
}

int32_t TraversalEngine::vertex_jhash(const Vertex& v) {
    // CortexVertex.hashCode :82-91 (locus null, kmerSources = empty HashSet -> 0)
    uint32_t r = (uint32_t)jhash_bytes(v.sk);
    r = 31u * r + (v.rec >= 0 ? (uint32_t)record_of(v).jhash() : 0u);
    r = 31u * r + 0u;
    r = 31u * r + 0u;
    r = 31u * r + (uint32_t)v.copy_index;
    r = 31u * r + (uint32_t)v.index;
    return (int32_t)r;
}

namespace {
// iteration order of a java.util.HashSet<T> filled in `ins` order (duplicates ignored)
template <class T, class HashFn>
std::vector<T> jset(const std::vector<T>& ins, HashFn hf) {
    std::vector<T> uniq;
    for (auto& x : ins) if (std::find(uniq.begin(), uniq.end(), x) == uniq.end()) uniq.push_back(x);
    if (uniq.size() < 2) return uniq;
    std::vector<int32_t> h;
    for (auto& x : uniq) h.push_back(hf(x));
    std::vector<T> out;
    for (size_t i : jhash_iteration_order(h)) out.push_back(uniq[i]);
    return out;
}
}  // namespace

// N1 + N2: TraversalUtils.getAllNextKmers/getAllPrevKmers + TraversalEngine.getNextVertices/getPrevVertices
std::vector<Vertex> TraversalEngine::adjacent_vertices(const std::string& sk, bool forward) {
    CanonicalKmer ck(sk);
    bool flipped = ec_.strict_java_flip ? ck.flipped : (ck.kmer != sk);
    Record cr;
    int64_t idx;
    bool found = ec_.graph->find_record(ck.kmer, cr, &idx);
    std::vector<std::vector<std::string>> per_colour;    // per colour: HashSet<CortexByteKmer> iteration order
    if (found) {
        std::string o = !flipped ? cr.kmer_string() : reverse_complement(cr.kmer_string());
        per_colour.resize(cr.cov.size());
        for (int c = 0; c < (int)cr.cov.size(); c++) {
            std::string e;
            if (forward) e = !flipped ? cr.out_edges(c, false) : cr.in_edges(c, true);
            else e = !flipped ? cr.in_edges(c, false) : cr.out_edges(c, true);
            std::vector<char> ev(e.begin(), e.end());
            ev = jset(ev, [](char b) { return (int32_t)b; });            // new HashSet<>(Collection<Byte>)
            std::vector<std::string> ks;
            for (char b : ev) ks.push_back(forward ? o.substr(1) + b : std::string(1, b) + o.substr(0, o.size() - 1));
            per_colour[c] = jset(ks, [](const std::string& s) { return jhash_bytes(s); });
        }
    }
    std::vector<std::string> combined;
    for (int c : ec_.traversal_colors)
        if (found && c < (int)per_colour.size()) combined.insert(combined.end(), per_colour[c].begin(), per_colour[c].end());
    combined = jset(combined, [](const std::string& s) { return jhash_bytes(s); });
    std::vector<std::string> kmers;
    if (!combined.empty()) {
        kmers = combined;
    } else {
        for (int c : ec_.recruitment_colors) {
            if (!found) throw JavaNullPointer("getNextVertices: record missing with recruitment colours set (Q14)");
            kmers.insert(kmers.end(), per_colour[c].begin(), per_colour[c].end());
        }
        kmers = jset(kmers, [](const std::string& s) { return jhash_bytes(s); });   // HashMap keySet order
    }
    std::vector<Vertex> vs;
    for (auto& km : kmers) vs.push_back(Vertex{km, ec_.graph->find_record(km), 0, 0});
    if (vs.size() > 1) vs = jset(vs, [this](const Vertex& v) { return vertex_jhash(v); });
    return vs;
}
std::vector<Vertex> TraversalEngine::next_vertices(const std::string& sk) { return adjacent_vertices(sk, true); }
std::vector<Vertex> TraversalEngine::prev_vertices(const std::string& sk) { return adjacent_vertices(sk, false); }

void TraversalEngine::seek(const std::string& sk) {
    cur_ = sk; has_cur_ = true;
    auto pv = prev_vertices(cur_);
    has_prev_ = pv.size() == 1; if (has_prev_) prev_ = pv[0].sk;
    auto nv = next_vertices(cur_);
    has_next_ = nv.size() == 1; if (has_next_) next_ = nv[0].sk;
    store_ = LinkStore();
    seen_.clear();
    specific_links_null_ = true;
}

std::vector<CortexLinks*> TraversalEngine::my_links() const {
    std::set<std::string> samples;
    for (int c : ec_.traversal_colors) samples.insert(ec_.graph->colors[c].sample_name);
    std::vector<CortexLinks*> out;
    for (auto* lm : ec_.links)
        if (!lm->sample_names.empty() && samples.count(lm->sample_names[0])) out.push_back(lm);
    return out;
}
void TraversalEngine::initialize_link_store(bool fwd) {
    specific_links_null_ = false;
    for (auto* lm : my_links()) {
        CanonicalKmer ck(cur_);
        if (lm->contains(ck.kmer)) store_.add(cur_, lm->get(ck.kmer), fwd);
    }
}
void TraversalEngine::update_link_store(bool fwd) {
    specific_links_null_ = false;
    for (auto* lm : my_links()) {
        bool has = fwd ? has_next_ : has_prev_;
        if (!has) continue;
        const std::string& t = fwd ? next_ : prev_;
        CanonicalKmer ck(t);
        if (lm->contains(ck.kmer)) store_.add(t, lm->get(ck.kmer), fwd);
    }
}
bool TraversalEngine::adjacent_kmer(const std::string& kmer, const std::vector<Vertex>& adj, bool fwd, std::string& out) {
    char choice;
    if (!store_.next_junction_choice(choice)) return false;
    std::string cand = fwd ? kmer.substr(1) + choice : std::string(1, choice) + kmer.substr(0, kmer.size() - 1);
    for (auto& v : adj) if (v.sk == cand) { out = cand; return true; }
    return false;
}

Vertex TraversalEngine::step(bool fwd) {
    if (fwd ? !has_next_ : !has_prev_)
        throw NoSuchElement(std::string("No single ") + (fwd ? "advance" : "prev") + " kmer from cursor '" + cur_ + "'");
    if (specific_links_null_ || go_forward_ != fwd) {
        go_forward_ = fwd;
        seek(cur_);
        initialize_link_store(fwd);
    }
    update_link_store(fwd);
    if (fwd ? !has_next_ : !has_prev_) throw JavaNullPointer("cursor target became null after re-seek");
    std::string t = fwd ? next_ : prev_;
    Vertex cv{t, ec_.graph->find_record(t), 0, 0};
    if (fwd) { prev_ = cur_; has_prev_ = true; } else { next_ = cur_; has_next_ = true; }
    cur_ = t;
    std::vector<Vertex> adj = fwd ? next_vertices(cur_) : prev_vertices(cur_);
    bool have = false;
    std::string target;
    if (adj.size() == 1 && (!seen_.count(adj[0].sk) || store_.is_active())) {
        target = adj[0].sk; have = true;
        seen_.insert(target);
    } else if (adj.size() > 1) {
        have = adjacent_kmer(cur_, adj, fwd, target);
        store_.increment_ages();
    }
    if (fwd) { has_next_ = have; next_ = target; } else { has_prev_ = have; prev_ = target; }
    if (store_.num_new_paths() > 0) store_.increment_ages();    // Q12
    cursor_steps++;
    return cv;
}
Vertex TraversalEngine::next() { return step(true); }
Vertex TraversalEngine::previous() { return step(false); }

void TraversalEngine::connect_vertex(PGraph& g, const Vertex& cv, const std::vector<Vertex>* pvs, const std::vector<Vertex>* nvs) {
    int color = ec_.traversal_colors[0];
    int ci = g.add_vertex(cv);
    if (pvs) for (auto& pv : *pvs) { int pi = g.add_vertex(pv); if (!g.contains_edge(pi, ci)) g.add_edge(pi, ci, color); }
    if (nvs) for (auto& nv : *nvs) { int ni = g.add_vertex(nv); if (!g.contains_edge(ci, ni)) g.add_edge(ci, ni, color); }
}

std::unique_ptr<PGraph> TraversalEngine::dfs_branch(Vertex cv, bool fwd, int graph_size, int depth,
                                                    const std::unordered_set<Vertex, VertexHash>& visited_old,
                                                    const std::vector<std::string>& sinks) {
    auto g = std::make_unique<PGraph>();
    std::unordered_set<Vertex, VertexHash> visited(visited_old);
    bool have_links = !ec_.links.empty();
    if (have_links) seek(cv.sk);

    static thread_local std::unordered_map<const CortexGraph*, std::unordered_set<std::string>> roi_sets;
    const std::unordered_set<std::string>* roi_set = nullptr;
    if (ec_.rois && (ec_.stopper == NOVEL_PARTITION || ec_.stopper == NOVEL_KMER_LIMITED_CONTIG)) {
        auto& rs = roi_sets[ec_.rois];
        if (rs.empty()) { Record r; for (int64_t i = 0; i < ec_.rois->num_records; i++) { ec_.rois->get_record(i, r); rs.insert(r.kmer_string()); } }
        roi_set = &rs;
    }
    StoppingRule stopper(ec_.stopper, *this, roi_set);

    std::vector<Vertex> avs, rvs;
    do {
        dfs_iterations++;
        if (getenv("ORC_TRACE")) fprintf(stderr, "O %d %s\n", depth, cv.sk.c_str());
        std::vector<Vertex> pvs = prev_vertices(cv.sk);
        std::vector<Vertex> nvs = next_vertices(cv.sk);
        avs = fwd ? nvs : pvs;
        rvs = fwd ? pvs : nvs;

        if (have_links) {
            bool have_qv = false;
            Vertex qv;
            if (fwd && has_next()) { qv = next(); have_qv = true; }
            else if (!fwd && has_previous()) { qv = previous(); have_qv = true; }
            if (have_qv) {
                Vertex lv;
                bool first = true;
                do {
                    int ci = first ? 0 : (fwd ? lv.copy_index + 1 : lv.copy_index - 1);
                    first = false;
                    lv = Vertex{qv.sk, qv.rec, ci, 0};
                } while (visited.count(lv));
                avs.clear();
                avs.push_back(lv);
            }
        }
        if (ec_.connect_all_neighbors) connect_vertex(*g, cv, &pvs, &nvs);

        avs.erase(std::remove_if(avs.begin(), avs.end(), [&](const Vertex& v) { return visited.count(v) != 0; }), avs.end());
        bool previously = visited.count(cv) != 0;
        visited.insert(cv);

        int gv = (int)g->verts.size();
        if (getenv("ORC_TRACE")) fprintf(stderr, "O   adj %d prev %d gv %d size %d rvs %d\n", (int)avs.size(), (int)previously, gv, graph_size + gv, (int)rvs.size());
        TraversalState ts{&cv, fwd, graph_size + gv, depth, gv, (int)avs.size(), (int)rvs.size(), false, gv > ec_.max_length, &sinks};
        if (!previously && stopper.keep_going(ts)) {
            if (avs.size() == 1) {
                if (fwd) connect_vertex(*g, cv, nullptr, &avs); else connect_vertex(*g, cv, &avs, nullptr);
                cv = avs[0];
            } else {
                bool children = false;
                for (auto& av : avs) {
                    auto branch = dfs_branch(av, fwd, graph_size + (int)g->verts.size(), depth + 1, visited, sinks);
                    if (branch) {
                        std::vector<Vertex> single{av};
                        if (fwd) connect_vertex(*branch, cv, nullptr, &single); else connect_vertex(*branch, cv, &single, nullptr);
                        g->add_graph(*branch);
                        children = true;
                    }
                }
                int gv2 = (int)g->verts.size();
                TraversalState tc{&cv, fwd, graph_size + gv2, depth, gv2, (int)avs.size(), (int)rvs.size(), true, gv2 > ec_.max_length, &sinks};
                if (children || stopper.has_succeeded(tc)) { if (getenv("ORC_TRACE")) fprintf(stderr, "O end depth %d success 1 iters %llu\n", depth, (unsigned long long)dfs_iterations); return g; }
            }
        } else if (stopper.traversal_succeeded()) {
            if (getenv("ORC_TRACE")) fprintf(stderr, "O end depth %d success 1 iters %llu\n", depth, (unsigned long long)dfs_iterations);
            return g;
        } else {
            if (getenv("ORC_TRACE")) fprintf(stderr, "O end depth %d success 0 iters %llu\n", depth, (unsigned long long)dfs_iterations);
            return nullptr;
        }
    } while (avs.size() == 1);
    if (getenv("ORC_TRACE")) fprintf(stderr, "O end depth %d success 0 iters %llu\n", depth, (unsigned long long)dfs_iterations);
    return nullptr;
}

void TraversalEngine::add_secondary_colors(PGraph& m) {
    if (ec_.secondary_colors.empty()) return;
    PGraph g = m;    // iterate the pre-secondary vertex set
    for (int c : ec_.secondary_colors) {
        if (std::find(ec_.traversal_colors.begin(), ec_.traversal_colors.end(), c) != ec_.traversal_colors.end()) continue;
        PGraph g2;
        for (auto& v : g.verts) {
            CanonicalKmer ck(v.sk);
            bool flipped = ec_.strict_java_flip ? ck.flipped : (ck.kmer != v.sk);
            Record cr;
            if (!ec_.graph->find_record(ck.kmer, cr)) throw JavaNullPointer("addSecondaryColors on missing record");
            std::string o = !flipped ? cr.kmer_string() : reverse_complement(cr.kmer_string());
            int vi = g2.add_vertex(v);
            std::string ine = !flipped ? cr.in_edges(c, false) : cr.out_edges(c, true);
            for (char b : ine) {
                std::string pk = std::string(1, b) + o.substr(0, o.size() - 1);
                int pi = g2.add_vertex(Vertex{pk, ec_.graph->find_record(pk), 0, 0});
                if (!g2.contains_edge(pi, vi)) g2.add_edge(pi, vi, c);
            }
            std::string oute = !flipped ? cr.out_edges(c, false) : cr.in_edges(c, true);
            for (char b : oute) {
                std::string nk = o.substr(1) + b;
                int ni = g2.add_vertex(Vertex{nk, ec_.graph->find_record(nk), 0, 0});
                if (!g2.contains_edge(vi, ni)) g2.add_edge(vi, ni, c);
            }
        }
        m.add_graph(g2);
    }
}

std::unique_ptr<PGraph> TraversalEngine::dfs(const std::string& source, const std::vector<std::string>& sinks) {
    Vertex cv{source, ec_.graph->find_record(source), 0, 0};
    std::unordered_set<Vertex, VertexHash> empty;
    std::unique_ptr<PGraph> dfsr, dfsf;
    if (ec_.direction == 0 || ec_.direction == 2) dfsr = dfs_branch(cv, false, 0, 0, empty, sinks);
    if (ec_.direction == 0 || ec_.direction == 1) dfsf = dfs_branch(cv, true, 0, 0, empty, sinks);
    auto relabel = [&](PGraph& g, int idx) {
        PGraph out;
        std::vector<Vertex> vs = g.verts;
        for (auto& v : vs) if (!(v == cv)) v.index = idx;
        for (auto& v : vs) out.add_vertex(v);
        for (auto& e : g.edges) out.add_edge(out.add_vertex(vs[e.src]), out.add_vertex(vs[e.dst]), e.color);
        g = out;
    };
    if (dfsr) relabel(*dfsr, -1);
    if (dfsf) relabel(*dfsf, 1);
    std::unique_ptr<PGraph> out;
    if (!ec_.op_and) {
        if (dfsr || dfsf) {
            out = std::make_unique<PGraph>();
            if (dfsr) out->add_graph(*dfsr);
            if (dfsf) out->add_graph(*dfsf);
        }
    } else if (dfsr && dfsf) {
        out = std::make_unique<PGraph>();
        out->add_graph(*dfsr);
        out->add_graph(*dfsf);
    }
    if (out) add_secondary_colors(*out);
    return out;
}

std::vector<Vertex> TraversalEngine::walk(const std::string& seed) {
    auto g = dfs(seed);
    return to_walk(*this, g.get(), seed, ec_.traversal_colors[0]);
}

// ------------------------------------------------------------------ toWalk / toContig
namespace {
// java.util.TimSort for n < 32: countRunAndMakeAscending + binarySort (comparators here are not total orders)
template <class T, class Cmp>
void java_small_sort(std::vector<T>& a, Cmp c) {
    int n = (int)a.size();
    if (n < 2) return;
    int run_hi = 1;
    if (c(a[run_hi++], a[0]) < 0) {
        while (run_hi < n && c(a[run_hi], a[run_hi - 1]) < 0) run_hi++;
        std::reverse(a.begin(), a.begin() + run_hi);
    } else {
        while (run_hi < n && c(a[run_hi], a[run_hi - 1]) >= 0) run_hi++;
    }
    for (int start = run_hi; start < n; start++) {
        T pivot = a[start];
        int left = 0, right = start;
        while (left < right) {
            int mid = (left + right) >> 1;
            if (c(pivot, a[mid]) < 0) right = mid; else left = mid + 1;
        }
        for (int i = start; i > left; i--) a[i] = a[i - 1];
        a[left] = pivot;
    }
}
}  // namespace

std::vector<Vertex> to_walk(TraversalEngine& e, const PGraph* g, const std::string& sk, int color) {
    std::vector<Vertex> w;
    if (!g) return w;
    int seed = -1;
    for (int i = 0; i < (int)g->verts.size(); i++) {
        const Vertex& v = g->verts[i];
        if (v.sk == sk && v.rec >= 0 && e.record_of(v).cov[color] > 0 && (seed < 0 || v.copy_index < g->verts[seed].copy_index)) seed = i;
    }
    if (seed < 0) return w;
    std::vector<std::vector<int>> out_e(g->verts.size()), in_e(g->verts.size());
    for (int i = 0; i < (int)g->edges.size(); i++) { out_e[g->edges[i].src].push_back(i); in_e[g->edges[i].dst].push_back(i); }
    auto canon_of = [&](int vi) -> std::string {
        if (g->verts[vi].rec < 0) throw JavaNullPointer("toWalk: getCanonicalKmer() on null record");
        return e.record_of(g->verts[vi]).kmer_string();
    };
    std::vector<Vertex> rev_part;    // w.add(0, pv) in the reference; collected here and reversed once
    auto extend = [&](bool fwd) {
        std::set<int> seen;
        int cv = seed;
        while (cv >= 0 && !seen.count(cv)) {
            std::vector<int> nvs;
            for (int ei : (fwd ? out_e[cv] : in_e[cv]))
                if (g->edges[ei].color == color) nvs.push_back(fwd ? g->edges[ei].dst : g->edges[ei].src);
            auto self = std::find(nvs.begin(), nvs.end(), cv);
            if (self != nvs.end()) nvs.erase(self);
            int nv = -1;
            if (nvs.size() == 1) nv = nvs[0];
            else if (nvs.size() > 1) {
                bool same = true;
                for (size_t i = 1; i < nvs.size(); i++) if (canon_of(nvs[0]) != canon_of(nvs[i])) { same = false; break; }
                if (same) {
                    if (fwd) java_small_sort(nvs, [&](int a, int b) { return g->verts[a].copy_index < g->verts[b].copy_index ? -1 : 1; });
                    else java_small_sort(nvs, [&](int a, int b) { return g->verts[a].copy_index > g->verts[b].copy_index ? -1 : 1; });
                    nv = nvs[0];
                }
            }
            if (nv >= 0) {
                if (fwd) w.push_back(g->verts[nv]); else rev_part.push_back(g->verts[nv]);
                seen.insert(cv);
            }
            cv = nv;
        }
    };
    w.push_back(g->verts[seed]);
    extend(true);
    extend(false);
    if (!rev_part.empty()) {
        std::reverse(rev_part.begin(), rev_part.end());
        rev_part.insert(rev_part.end(), w.begin(), w.end());
        w.swap(rev_part);
    }
    return w;
}

std::string to_contig(const std::vector<Vertex>& walk) {
    std::string s;
    for (auto& v : walk) {
        if (s.empty()) s = v.sk; else s.push_back(v.sk.back());
    }
    return s;
}

}  // namespace orc
