#include "links.h"

#include <zlib.h>

#include <algorithm>
#include <map>
#include <numeric>

namespace ldbg {

namespace {

// ---- tiny JSON reader (objects / arrays / strings / numbers / literals); enough for the .ctp header
struct JVal {
    enum T { NUL, NUM, STR, BOOL, OBJ, ARR } t = NUL;
    double num = 0; bool b = false; std::string str;
    std::vector<std::pair<std::string, JVal>> obj;
    std::vector<JVal> arr;
    const JVal* get(const std::string& k) const {
        for (auto& kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};
struct JParser {
    const std::string& s; size_t i = 0;
    void ws() { while (i < s.size() && isspace((unsigned char)s[i])) i++; }
    [[noreturn]] void fail(const char* m) { throw StatusError(LDBG_ERR_CORTEXJDK, std::string("Cannot parse CortexLinks JSON header: ") + m); }
    JVal parse() {
        ws();
        if (i >= s.size()) fail("unexpected end");
        JVal v;
        char c = s[i];
        if (c == '{') {
            v.t = JVal::OBJ; i++; ws();
            if (s[i] == '}') { i++; return v; }
            while (true) {
                ws(); JVal k = parse(); if (k.t != JVal::STR) fail("object key");
                ws(); if (s[i] != ':') fail("':' expected"); i++;
                v.obj.push_back({k.str, parse()});
                ws(); if (s[i] == ',') { i++; continue; }
                if (s[i] == '}') { i++; break; }
                fail("',' or '}' expected");
            }
        } else if (c == '[') {
            v.t = JVal::ARR; i++; ws();
            if (s[i] == ']') { i++; return v; }
            while (true) {
                v.arr.push_back(parse());
                ws(); if (s[i] == ',') { i++; continue; }
                if (s[i] == ']') { i++; break; }
                fail("',' or ']' expected");
            }
        } else if (c == '"') {
            v.t = JVal::STR; i++;
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '\\' && i + 1 < s.size()) { i++; char e = s[i]; v.str.push_back(e == 'n' ? '\n' : e == 't' ? '\t' : e); }
                else v.str.push_back(s[i]);
                i++;
            }
            i++;
        } else if (c == 't' || c == 'f') {
            v.t = JVal::BOOL; v.b = c == 't'; i += v.b ? 4 : 5;
        } else if (c == 'n') {
            i += 4;
        } else {
            size_t e = i;
            while (e < s.size() && (isdigit((unsigned char)s[e]) || strchr("+-.eE", s[e]))) e++;
            if (e == i) fail("value expected");
            v.t = JVal::NUM; v.num = strtod(s.substr(i, e - i).c_str(), nullptr); i = e;
        }
        return v;
    }
};
int64_t jnum(const JVal* o, const char* key) {
    const JVal* v = o ? o->get(key) : nullptr;
    if (!v || v->t != JVal::NUM) throw StatusError(LDBG_ERR_CORTEXJDK, std::string("CortexLinks header field missing: ") + key);
    return (int64_t)v->num;
}

std::string gunzip_file(const std::string& path) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) throw StatusError(LDBG_ERR_CORTEXJDK, "Unable to load Cortex links file '" + path + "'");
    std::string out;
    std::vector<char> buf(1 << 20);
    int n;
    while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) out.append(buf.data(), n);
    gzclose(f);
    return out;
}

// java.lang.String.hashCode / java.util.Arrays.hashCode(int[])
int32_t jstring_hash(const std::string& s) { uint32_t h = 0; for (unsigned char c : s) h = 31u * h + c; return (int32_t)h; }
int32_t jints_hash(const std::vector<int32_t>& v) { uint32_t h = 1; for (int32_t e : v) h = 31u * h + (uint32_t)e; return (int32_t)h; }
// CortexJunctionsRecord.hashCode (CortexJunctionsRecord.java:87-95)
int32_t junction_hash(const HostJunction& j) {
    uint32_t r = j.is_fw ? 1u : 0u;
    r = 31u * r + (uint32_t)j.num_kmers;
    r = 31u * r + (uint32_t)j.num_junctions;
    r = 31u * r + (uint32_t)jints_hash(j.cov);
    r = 31u * r + (uint32_t)jstring_hash(j.junctions);
    return (int32_t)r;
}
bool junction_eq(const HostJunction& a, const HostJunction& b) {
    return a.is_fw == b.is_fw && a.num_junctions == b.num_junctions && a.num_kmers == b.num_kmers && a.cov == b.cov && a.junctions == b.junctions;
}
// iteration order of a default-constructed java.util.HashSet after inserting distinct elements
// with these hashCodes: bucket index first, insertion order inside a bucket
void hashset_order(std::vector<HostJunction>& js) {
    size_t n = js.size();
    int cap = 16;
    while (n > (size_t)cap * 3 / 4) cap *= 2;
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    auto bucket = [&](size_t i) { uint32_t h = (uint32_t)junction_hash(js[i]); h ^= h >> 16; return h & (uint32_t)(cap - 1); };
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return bucket(a) < bucket(b); });
    std::vector<HostJunction> out;
    for (size_t i : idx) out.push_back(js[i]);
    js.swap(out);
}
std::string complement_ascii(const std::string& s) {
    std::string c(s);
    for (auto& ch : c) {
        switch (ch) {
            case 'A': ch = 'T'; break; case 'C': ch = 'G'; break; case 'G': ch = 'C'; break; case 'T': ch = 'A'; break;
            case 'a': ch = 't'; break; case 'c': ch = 'g'; break; case 'g': ch = 'c'; break; case 't': ch = 'a'; break;
            default: break;
        }
    }
    return c;
}
std::vector<std::string> split_fields(const std::string& s, bool commas) {
    std::vector<std::string> out;
    std::string cur;
    for (char ch : s) {
        if (isspace((unsigned char)ch) || (commas && ch == ',')) { if (!cur.empty()) { out.push_back(cur); cur.clear(); } }
        else cur.push_back(ch);
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

}  // namespace

// sets bit `slot` of the link-flags byte of every graph record that has a link record in this set
template <int W>
LDBG_KERNEL void k_set_link_flags(GraphView g, uint8_t* probe, const uint64_t* keys, int64_t M, int slot, int clear = 0) {
    for (int64_t i = global_tid(); i < M; i += global_nthreads()) {
        Kmer<W> q;
#pragma unroll
        for (int w = 0; w < W; w++) q.w[w] = keys[i * W + w];
        int64_t idx = graph_find_canonical<W>(g, q);
        if (idx >= 0) {
            uint8_t* f = probe + (size_t)idx * (size_t)g.stride + g.flags_off;
            *f = clear ? (uint8_t)(*f & ~(1u << slot)) : (uint8_t)(*f | (1u << slot));
        }
    }
}

// rec_of[record] = the junction records of that graph record as (first | count << 32), ~0 = none: a walk standing on a record finds its links
// with one load instead of a search of the link table
template <int W>
LDBG_KERNEL void k_link_rec_of(GraphView g, const uint64_t* keys, int64_t M, const uint32_t* off, uint64_t* rec_of) {
    for (int64_t i = global_tid(); i < M; i += global_nthreads()) {
        Kmer<W> q;
#pragma unroll
        for (int w = 0; w < W; w++) q.w[w] = keys[i * W + w];
        GraphView exact = g;
        exact.java_tiny = 0;
        int64_t idx = graph_find_canonical<W>(exact, q);
        if (idx >= 0) rec_of[idx] = (uint64_t)off[i] | ((uint64_t)(off[i + 1] - off[i]) << 32);
    }
}

Links::Links(const std::string& path, const Graph& g) : device(g.device) {
    // CortexLinks.initialize (CortexLinks.java:16-25): an ".idx" file next to the links file selects the random-access back-end
    // (BGZF is a series of gzip members, so the whole file is read the same way; the index itself is not needed in HBM)
    const bool indexed = [&] { FILE* f = fopen((path + ".idx").c_str(), "rb"); if (f) fclose(f); return f != nullptr; }();
    std::string text = gunzip_file(path);
    // header = lines from "{" to "}" (CortexLinksIterable.java:58-67)
    size_t pos = 0;
    auto next_line = [&](std::string& line) -> bool {
        if (pos >= text.size()) return false;
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        line.assign(text, pos, e - pos);
        pos = e + 1;
        return true;
    };
    std::string line, header;
    bool in_header = false;
    while (next_line(line)) {
        if (line == "{") in_header = true;
        if (in_header) header += line + "\n";
        if (line == "}") break;
    }
    JParser jp{header};
    JVal h = jp.parse();
    const JVal* fv = h.get("formatVersion") ? h.get("formatVersion") : h.get("format_version");
    if (!fv) throw StatusError(LDBG_ERR_CORTEXJDK, "Cannot parse CortexLinks format version field");
    version = (int)fv->num;
    if (version != 2 && version != 3 && version != 4)
        throw StatusError(LDBG_ERR_CORTEXJDK, "Cannot parse CortexLinks format version '" + std::to_string(version) + "'");
    const JVal* colours = nullptr;
    if (version == 2) {
        num_colors = (int)jnum(&h, "ncols");
        k = (int)jnum(&h, "kmer_size");
        num_kmers_in_graph = jnum(&h, "num_kmers_in_graph");
        num_kmers_with_links = jnum(&h, "num_kmers_with_paths");
        num_links = jnum(&h, "num_paths");
        link_bytes = jnum(&h, "path_bytes");
        colours = h.get("colours");
    } else {
        const JVal* gr = h.get("graph");
        const JVal* pa = h.get("paths");
        num_colors = (int)jnum(gr, "num_colours");
        k = (int)jnum(gr, "kmer_size");
        num_kmers_in_graph = jnum(gr, "num_kmers_in_graph");
        num_kmers_with_links = jnum(pa, "num_kmers_with_paths");
        num_links = jnum(pa, "num_paths");
        link_bytes = jnum(pa, "path_bytes");
        colours = gr ? gr->get("colours") : nullptr;
    }
    if (colours) for (auto& c : colours->arr) { const JVal* s = c.get("sample"); sample_names.push_back(s ? s->str : ""); }
    if (k != g.hdr.k)
        throw StatusError(LDBG_ERR_CORTEXJDK, "links k-mer size " + std::to_string(k) + " does not match the graph's " + std::to_string(g.hdr.k));

    // skip comments and blank lines (:133-144), then numKmersWithLinks records (:172-226)
    bool have = false;
    while (next_line(line)) {
        if (line.empty() || line[0] == '#') continue;
        have = true;
        break;
    }
    const int W = g.hdr.W;
    std::map<std::vector<uint64_t>, HostLinksRecord> by_key;   // canonical packed words -> record (later replaces earlier)
    for (int64_t r = 0; r < num_kmers_with_links && have; r++) {
        auto kl = split_fields(line, false);
        if (kl.size() < 2) throw StatusError(LDBG_ERR_CORTEXJDK, "Unable to parse CortexLinks record");
        HostLinksRecord rec;
        rec.kmer = kl[0];
        int n = atoi(kl[1].c_str());
        for (int i = 0; i < n; i++) {
            if (!next_line(line)) throw StatusError(LDBG_ERR_CORTEXJDK, "Unable to parse CortexLinks record");
            auto f = split_fields(line, true);
            HostJunction j;
            int off = version == 4 ? 2 : 3;
            if ((int)f.size() < off + num_colors + 1) throw StatusError(LDBG_ERR_CORTEXJDK, "Unable to parse CortexLinks record");
            j.is_fw = f[0] == "F";
            if (indexed && version != 4) throw StatusError(LDBG_ERR_CORTEXJDK, "indexed link files are version 4 (IndexLinks.java:62-135)");
            if (indexed) {
                // CortexLinksRandomAccess reads records through CortexLinksRecord(byte[]) (CortexLinksRecord.java:17-43):
                // "orientation x coverages junctions", numKmers := x, numJunctions := junctions.length() — quirk Q11: another
                // hashCode, hence another HashSet order of the junction records than the un-indexed back-end
                j.num_kmers = atoi(f[1].c_str());
                for (int c = 0; c < num_colors; c++) j.cov.push_back(atoi(f[2 + c].c_str()));
                j.junctions = f[2 + num_colors];
                j.num_junctions = (int)j.junctions.size();
            } else {
                j.num_kmers = version == 4 ? -1 : atoi(f[1].c_str());
                j.num_junctions = version == 4 ? atoi(f[1].c_str()) : atoi(f[2].c_str());
                for (int c = 0; c < num_colors; c++) j.cov.push_back(atoi(f[off + c].c_str()));
                j.junctions = f[off + num_colors];
            }
            bool dup = false;
            for (auto& o : rec.juncs) dup |= junction_eq(o, j);
            if (!dup) rec.juncs.push_back(j);
        }
        hashset_order(rec.juncs);
        std::vector<uint64_t> w(W), rc(W);
        if ((int)rec.kmer.size() != k || !ascii_to_words(rec.kmer.c_str(), k, w.data(), W))
            throw StatusError(LDBG_ERR_CORTEXJDK, "Unable to parse CortexLinks record: bad k-mer '" + rec.kmer + "'");
        // canonical key (CortexBinaryKmer(byte[]) canonicalises, CortexBinaryKmer.java:17-19)
        std::string rcs(k, 'A');
        for (int i = 0; i < k; i++) { char ch = rec.kmer[k - 1 - i]; rcs[i] = complement_ascii(std::string(1, ch))[0]; }
        ascii_to_words(rcs.c_str(), k, rc.data(), W);
        by_key[std::min(w, rc)] = rec;
        have = next_line(line);
        while (have && line.empty()) have = next_line(line);
    }

    for (auto& kv : by_key) {
        record_keys.push_back(kv.first);
        records.push_back(kv.second);
        std::vector<uint64_t> w(W);
        ascii_to_words(kv.second.kmer.c_str(), k, w.data(), W);
        record_is_canonical.push_back(w == kv.first ? 1 : 0);
    }
    // claim a flag bit in the graph's probe rows and set it on every record that has links here
    int free_slot = -1;
    for (int b = 0; b < 6 && free_slot < 0; b++) if (!((g.link_slots >> b) & 1u)) free_slot = b;
    if (free_slot < 0) throw StatusError(LDBG_ERR_UNSUPPORTED, "more than 6 link sets bound to one graph at the same time");
    slot = free_slot;
    g.link_slots |= 1u << slot;
    graph_ = &g;
    mark_records(false);
    g.bound_links.push_back(this);
}

// set (or, when the link set is closed, clear) this set's bit on the records it has links for
void Links::mark_records(bool clear) {
    const Graph& g = *graph_;
    const int W = g.hdr.W;
    const int64_t M = (int64_t)records.size();
    if (M > 0) {
        rt::set_device(device);
        std::vector<uint64_t> keys;
        keys.reserve((size_t)M * W);
        for (auto& kk : record_keys) keys.insert(keys.end(), kk.begin(), kk.end());
        void* d_keys = rt::dmalloc(keys.size() * 8);
        rt::h2d(d_keys, keys.data(), keys.size() * 8, g.stream);
        const int grid = (int)std::min<int64_t>((M + 255) / 256, 2048);
        const int cl = clear ? 1 : 0;
        switch (W) {
            case 1: LDBG_LAUNCH(k_set_link_flags<1>, grid, 256, g.stream, g.view, g.probe_mutable(), (const uint64_t*)d_keys, M, slot, cl); break;
            case 2: LDBG_LAUNCH(k_set_link_flags<2>, grid, 256, g.stream, g.view, g.probe_mutable(), (const uint64_t*)d_keys, M, slot, cl); break;
            case 3: LDBG_LAUNCH(k_set_link_flags<3>, grid, 256, g.stream, g.view, g.probe_mutable(), (const uint64_t*)d_keys, M, slot, cl); break;
            default: LDBG_LAUNCH(k_set_link_flags<4>, grid, 256, g.stream, g.view, g.probe_mutable(), (const uint64_t*)d_keys, M, slot, cl); break;
        }
        rt::stream_sync(g.stream);
        rt::dfree(d_keys);
    }
}

Links::~Links() {
    if (!graph_) return;
    auto& bl = graph_->bound_links;
    bl.erase(std::remove(bl.begin(), bl.end(), this), bl.end());
    if (slot >= 0) {
        try { mark_records(true); } catch (...) {}
        graph_->link_slots &= ~(1u << slot);
    }
}

MergedLinks::MergedLinks(const std::vector<const Links*>& sets, const Graph& g) {
    const int W = g.hdr.W, k = g.hdr.k;
    struct Ref { const Links* l; size_t i; };
    std::map<std::vector<uint64_t>, std::vector<Ref>> by_key;
    for (const Links* l : sets) {
        flag_mask |= 1u << l->slot;
        for (size_t i = 0; i < l->records.size(); i++) by_key[l->record_keys[i]].push_back({l, i});
    }
    std::vector<uint64_t> keys;
    std::vector<uint32_t> off{0};
    std::vector<uint8_t> bases;
    std::vector<JuncRec> junc;
    for (auto& kv : by_key) {
        keys.insert(keys.end(), kv.first.begin(), kv.first.end());
        for (auto& r : kv.second) {
            const bool rec_canon = r.l->record_is_canonical[r.i] != 0;
            for (auto& j : r.l->records[r.i].juncs) {
                JuncRec jr;
                jr.str_off = (uint32_t)bases.size();
                jr.len = (uint32_t)j.junctions.size();
                jr.hash_asis = jstring_hash(j.junctions);
                jr.hash_comp = jstring_hash(complement_ascii(j.junctions));
                jr.is_fw = (rec_canon == j.is_fw) ? 1u : 0u;
                uint32_t n_packed = 0;
                for (char ch : j.junctions) {
                    uint8_t code;
                    switch (ch) {
                        case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break;
                        default: throw StatusError(LDBG_ERR_UNSUPPORTED, std::string("junction string with a non-ACGT character '") + ch + "'");
                    }
                    bases.push_back(code);
                    if (n_packed < 8) { jr.is_fw |= (uint32_t)code << (16 + 2 * n_packed); n_packed++; }
                }
                junc.push_back(jr);
            }
        }
        off.push_back((uint32_t)junc.size());
    }
    const int64_t M = (int64_t)by_key.size();
    int p = 1;
    while (p < k && p < 12 && (1LL << (2 * (p + 1))) <= std::max<int64_t>(M, 1)) p++;
    std::vector<uint32_t> pstart(((size_t)1 << (2 * p)) + 1, (uint32_t)M);
    {
        size_t x = 0;
        for (int64_t i = 0; i < M; i++) {
            uint32_t px;
            switch (W) {
                case 1: { Kmer<1> q; q.w[0] = keys[i]; px = kmer_prefix<1>(q, k, p); break; }
                case 2: { Kmer<2> q; q.w[0] = keys[2 * i]; q.w[1] = keys[2 * i + 1]; px = kmer_prefix<2>(q, k, p); break; }
                case 3: { Kmer<3> q; for (int w = 0; w < 3; w++) q.w[w] = keys[3 * i + w]; px = kmer_prefix<3>(q, k, p); break; }
                default: { Kmer<4> q; for (int w = 0; w < 4; w++) q.w[w] = keys[4 * i + w]; px = kmer_prefix<4>(q, k, p); break; }
            }
            while (x <= px) pstart[x++] = (uint32_t)i;
        }
    }
    rt::set_device(g.device);
    rt::stream_t s = g.stream;
    auto up = [&](const void* h, size_t n) { void* d = rt::dmalloc(n); rt::h2d(d, h, n, s); return d; };
    d_keys_ = up(keys.data(), keys.size() * 8);
    d_pstart_ = up(pstart.data(), pstart.size() * 4);
    d_off_ = up(off.data(), off.size() * 4);
    d_junc_ = up(junc.data(), junc.size() * sizeof(JuncRec));
    d_bases_ = up(bases.data(), bases.size());
    const int64_t N = g.view.N;
    d_rec_of_ = rt::dmalloc((size_t)std::max<int64_t>(1, N) * 8);
    rt::dmemset(d_rec_of_, 0xFF, (size_t)std::max<int64_t>(1, N) * 8, s);
    if (M > 0 && !g.is_image) {             // (an image learns the link records of a row when the row arrives: image.cpp)
        const int grid = (int)std::min<int64_t>((M + 255) / 256, 2048);
        switch (W) {
            case 1: LDBG_LAUNCH(k_link_rec_of<1>, grid, 256, s, g.view, (const uint64_t*)d_keys_, M, (const uint32_t*)d_off_, (uint64_t*)d_rec_of_); break;
            case 2: LDBG_LAUNCH(k_link_rec_of<2>, grid, 256, s, g.view, (const uint64_t*)d_keys_, M, (const uint32_t*)d_off_, (uint64_t*)d_rec_of_); break;
            case 3: LDBG_LAUNCH(k_link_rec_of<3>, grid, 256, s, g.view, (const uint64_t*)d_keys_, M, (const uint32_t*)d_off_, (uint64_t*)d_rec_of_); break;
            default: LDBG_LAUNCH(k_link_rec_of<4>, grid, 256, s, g.view, (const uint64_t*)d_keys_, M, (const uint32_t*)d_off_, (uint64_t*)d_rec_of_); break;
        }
    }
    rt::stream_sync(s);
    view.rec_of = (const uint64_t*)d_rec_of_;
    view.M = M;
    view.keys = (const uint8_t*)d_keys_;
    view.pstart = (const uint32_t*)d_pstart_;
    view.p = p;
    view.off = (const uint32_t*)d_off_;
    view.junc = (const JuncRec*)d_junc_;
    view.bases = (const uint8_t*)d_bases_;
}

MergedLinks::~MergedLinks() {
    rt::dfree(d_keys_); rt::dfree(d_pstart_); rt::dfree(d_off_); rt::dfree(d_junc_); rt::dfree(d_bases_); rt::dfree(d_rec_of_);
}

const HostLinksRecord* Links::get(const std::string& kmer_ascii) const {
    if (records.empty() || (int)kmer_ascii.size() != k) return nullptr;
    const int W = (int)record_keys[0].size();
    std::vector<uint64_t> w(W), rc(W);
    if (!ascii_to_words(kmer_ascii.c_str(), k, w.data(), W)) return nullptr;
    std::string rcs(k, 'A');
    for (int i = 0; i < k; i++) rcs[i] = complement_ascii(std::string(1, kmer_ascii[k - 1 - i]))[0];
    ascii_to_words(rcs.c_str(), k, rc.data(), W);
    auto key = std::min(w, rc);
    auto it = std::lower_bound(record_keys.begin(), record_keys.end(), key);
    if (it == record_keys.end() || *it != key) return nullptr;
    return &records[it - record_keys.begin()];
}

}  // namespace ldbg
