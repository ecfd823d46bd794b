// ORACLE — TEST INFRASTRUCTURE ONLY (see ldbg_oracle.hpp).
// Part 1: Java emulation helpers, k-mer primitives, CortexRecord, CortexGraph,
// CortexGraphWriter, link files, TempGraphAssembler, TempLinksAssembler.
#include "ldbg_oracle.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <sstream>

namespace orc {

// ------------------------------------------------------------------ Java hashing
int32_t jhash_bytes(const std::string& b) {
    uint32_t r = 1;
    for (char c : b) r = 31u * r + (uint32_t)(int32_t)(signed char)c;
    return (int32_t)r;
}
int32_t jhash_string(const std::string& s) {
    uint32_t h = 0;
    for (char c : s) h = 31u * h + (uint32_t)(unsigned char)c;
    return (int32_t)h;
}
int32_t jhash_longs(const std::vector<int64_t>& v) {
    uint32_t r = 1;
    for (int64_t e : v) {
        uint64_t u = (uint64_t)e;
        r = 31u * r + (uint32_t)(u ^ (u >> 32));
    }
    return (int32_t)r;
}
int32_t jhash_ints(const std::vector<int32_t>& v) {
    uint32_t r = 1;
    for (int32_t e : v) r = 31u * r + (uint32_t)e;
    return (int32_t)r;
}
int32_t jhash_u8(const std::vector<uint8_t>& v) {
    uint32_t r = 1;
    for (uint8_t e : v) r = 31u * r + (uint32_t)(int32_t)(int8_t)e;
    return (int32_t)r;
}
int jhashmap_capacity_for(size_t n) {
    int cap = 16;
    while (n > (size_t)cap * 3 / 4) cap *= 2;
    return cap;
}
std::vector<size_t> jhash_iteration_order(const std::vector<int32_t>& hashes) {
    int cap = jhashmap_capacity_for(hashes.size());
    std::vector<size_t> idx(hashes.size());
    for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
        return jhashmap_bucket(hashes[a], cap) < jhashmap_bucket(hashes[b], cap);
    });
    return idx;
}
std::vector<size_t> java_string_hashmap_order(const std::vector<std::string>& keys) {
    std::vector<int32_t> h;
    for (auto& k : keys) h.push_back(jhash_string(k));
    return jhash_iteration_order(h);
}

// ------------------------------------------------------------------ K1
char complement(char c) {
    switch (c) {
        case 'A': return 'T'; case 'a': return 't';
        case 'C': return 'G'; case 'c': return 'g';
        case 'G': return 'C'; case 'g': return 'c';
        case 'T': return 'A'; case 't': return 'a';
        case 'N': return 'N'; case 'n': return 'n';
        case '.': return '.';
        default: return c;
    }
}
std::string reverse_complement(const std::string& s) {
    std::string rc(s.size(), 'N');
    for (size_t i = 0; i < s.size(); i++) rc[s.size() - 1 - i] = complement(s[i]);
    return rc;
}
std::string complement_str(const std::string& s) {
    std::string c(s);
    for (auto& ch : c) ch = complement(ch);
    return c;
}
std::string canonical(const std::string& s) {
    for (size_t i = 0; i < s.size(); i++) {
        signed char rc = (signed char)complement(s[s.size() - 1 - i]);
        signed char b = (signed char)s[i];
        if (b < rc) return s;
        if (b > rc) return reverse_complement(s);
    }
    return s;
}
CanonicalKmer::CanonicalKmer(const std::string& s) : kmer(canonical(s)) {
    flipped = jhash_bytes(kmer) != jhash_bytes(s);
}

// ------------------------------------------------------------------ CortexRecord
static inline int64_t bswap64s(int64_t x) { return (int64_t)__builtin_bswap64((uint64_t)x); }

int Record::kmer_bits(int k) { return (k + 31) / 32; }

static int64_t char_to_nuc(char b) {
    switch (b) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: throw std::runtime_error(std::string("Nucleotide '") + b + "' is not a valid character nucleotide");
    }
}

std::vector<int64_t> Record::encode_binary_kmer(const std::string& kmer) {
    int len = (int)kmer.size();
    int nb = kmer_bits(len);
    std::vector<int64_t> bk(nb, 0);
    for (int b = 0; b < nb; b++) {
        uint64_t w = 0;
        for (int i = len - 32 * (b + 1); i < len - 32 * b; i++) {
            if (i >= 0) w |= (uint64_t)char_to_nuc(kmer[i]);
            if (i < len - 32 * b - 1) w <<= 2;
        }
        bk[nb - b - 1] = bswap64s((int64_t)w);
    }
    return bk;
}

std::string Record::decode_binary_kmer(const std::vector<int64_t>& kmer, int k, int W) {
    std::string raw(k, 'A');
    std::vector<uint64_t> b(kmer.size());
    for (size_t i = 0; i < kmer.size(); i++) b[i] = (uint64_t)bswap64s(kmer[i]);
    for (int i = k - 1; i >= 0; i--) {
        raw[i] = "ACGT"[b[W - 1] & 3];
        for (int j = W - 1; j > 0; j--) {
            b[j] >>= 2;
            b[j] |= (b[j - 1] << 62);
        }
        b[0] >>= 2;
    }
    return raw;
}

uint8_t Record::encode_binary_edges(const std::set<char>& in, const std::set<char>& out, bool rc) {
    const char* fwd = "ACGT";
    const char* rev = "TGCA";
    if (rc) std::swap(fwd, rev);
    uint32_t edge = 0;
    for (int i = 0; i < 4; i++) {
        if (in.count(fwd[i])) edge |= 1;
        edge <<= 1;
    }
    for (int i = 0; i < 4; i++) {
        if (out.count(rev[i])) edge |= 1;
        if (i != 3) edge <<= 1;
    }
    return (uint8_t)edge;
}

Record::Record(const std::string& sk, const std::vector<int32_t>& covs,
               const std::vector<std::set<char>>& in, const std::vector<std::set<char>>& out) {
    CanonicalKmer ck(sk);
    k = (int)ck.kmer.size();
    W = kmer_bits(k);
    bk = encode_binary_kmer(canonical(sk));   // new CortexBinaryKmer(sk.getBytes()).getBinaryKmer()
    cov = covs;
    edges.resize(covs.size());
    for (size_t c = 0; c < covs.size(); c++)
        edges[c] = !ck.flipped ? encode_binary_edges(in[c], out[c], false) : encode_binary_edges(out[c], in[c], true);
}

std::string Record::in_edges(int c, bool comp) const {
    const char* str = comp ? "TGCA" : "ACGT";
    int left = ((int)(int8_t)edges[c]) >> 4;
    std::string r;
    for (int i = 0; i < 4; i++)
        if (left & (1 << (3 - i))) r.push_back(str[i]);
    return r;
}
std::string Record::out_edges(int c, bool comp) const {
    const char* str = comp ? "TGCA" : "ACGT";
    int right = edges[c] & 0xf;
    std::string r;
    for (int i = 0; i < 4; i++)
        if (right & (1 << i)) r.push_back(str[i]);
    return r;
}
std::string Record::edges_string(int c) const {
    const char* str = "acgtACGT";
    std::string s(8, '.');
    int left = ((int)(int8_t)edges[c]) >> 4, right = edges[c] & 0xf;
    for (int i = 0; i < 4; i++) {
        if (left & (1 << (3 - i))) s[i] = str[i];
        if (right & (1 << i)) s[i + 4] = str[i + 4];
    }
    return s;
}
std::string Record::to_string() const {
    std::string s = kmer_string();
    for (int32_t c : cov) s += " " + std::to_string(c);
    for (size_t c = 0; c < edges.size(); c++) s += " " + edges_string((int)c);
    return s;
}
int32_t Record::jhash() const {
    return (int32_t)((uint32_t)jhash_longs(bk) - (uint32_t)jhash_ints(cov) + (uint32_t)jhash_u8(edges));
}
std::vector<uint64_t> Record::packed_words() const {
    std::vector<uint64_t> w(bk.size());
    for (size_t i = 0; i < bk.size(); i++) w[i] = (uint64_t)bswap64s(bk[i]);
    return w;
}

// ------------------------------------------------------------------ CortexGraph
namespace {
struct Reader {
    const uint8_t* p; size_t n; size_t off = 0;
    void need(size_t k) { if (off + k > n) throw CortexJDKException("Error while parsing Cortex graph file: unexpected end of file"); }
    uint32_t u32() { need(4); uint32_t v; memcpy(&v, p + off, 4); off += 4; return v; }
    uint64_t u64be() { need(8); uint64_t v; memcpy(&v, p + off, 8); off += 8; return __builtin_bswap64(v); }
    std::string bytes(size_t k) { need(k); std::string s((const char*)p + off, k); off += k; return s; }
    uint8_t u8() { need(1); return p[off++]; }
};
// fixStringsWithEarlyTerminators, CortexGraph.java:50-64
std::string fix_early_terminator(const std::string& s) {
    size_t pos = s.find('\0');
    return pos == std::string::npos ? s : s.substr(0, pos);
}
bool iequals(const std::string& a, const std::string& b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
        if (tolower((unsigned char)a[i]) != tolower((unsigned char)b[i])) return false;
    return true;
}
}  // namespace

CortexGraph::CortexGraph(const std::string& p, bool use_cache) : path(p), use_cache_(use_cache) {
    fd_ = ::open(p.c_str(), O_RDONLY);
    if (fd_ < 0) throw CortexJDKException("Cortex graph file '" + p + "' not found");
    struct stat st;
    fstat(fd_, &st);
    size_ = (size_t)st.st_size;
    if (size_ > 0) {
        void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) throw CortexJDKException("mmap failed for '" + p + "'");
        base_ = (const uint8_t*)m;
    }
    Reader r{base_, size_};
    if (!iequals(r.bytes(6), "CORTEX"))
        throw CortexJDKException("The file '" + p + "' does not appear to be a Cortex graph");
    version = (int)r.u32();
    if (version != 6) throw CortexJDKException("The file '" + p + "' is not a version 6 Cortex graph");
    k = (int)r.u32();
    W = (int)r.u32();
    C = (int)r.u32();
    colors.resize(C);
    for (int c = 0; c < C; c++) colors[c].mean_read_length = r.u32();
    for (int c = 0; c < C; c++) colors[c].total_sequence = r.u64be();   // Q16: big-endian readLong()
    for (int c = 0; c < C; c++) {
        uint32_t len = r.u32();
        colors[c].sample_name = fix_early_terminator(r.bytes(len));
    }
    for (int c = 0; c < C; c++) r.bytes(16);   // error rate skipped :113-117
    for (int c = 0; c < C; c++) {
        colors[c].tip_clipping = r.u8() != 0;
        colors[c].low_covg_supernodes_removed = r.u8() != 0;
        colors[c].low_covg_kmers_removed = r.u8() != 0;
        colors[c].cleaned_against_graph = r.u8() != 0;
        colors[c].low_cov_supernodes_threshold = r.u32();
        colors[c].low_cov_kmer_threshold = r.u32();
        uint32_t len = r.u32();
        colors[c].cleaned_against_graph_name = fix_early_terminator(r.bytes(len));
    }
    if (!iequals(r.bytes(6), "CORTEX"))
        throw CortexJDKException("We didn't see a proper header terminator at the expected place in Cortex graph '" + p + "'");
    data_offset = (int64_t)r.off;
    record_size = 8 * (int64_t)W + 5 * (int64_t)C;
    num_records = ((int64_t)size_ - data_offset) / record_size;
}

CortexGraph::~CortexGraph() {
    if (base_) munmap((void*)base_, size_);
    if (fd_ >= 0) ::close(fd_);
}

void CortexGraph::decode_at(int64_t i, Record& out) const {
    const uint8_t* p = base_ + data_offset + i * record_size;
    out.k = k; out.W = W;
    out.bk.resize(W); out.cov.resize(C); out.edges.resize(C);
    for (int w = 0; w < W; w++) {
        uint64_t v; memcpy(&v, p + 8 * w, 8);
        out.bk[w] = (int64_t)__builtin_bswap64(v);     // ByteBuffer.getLong(): big-endian
    }
    p += 8 * W;
    for (int c = 0; c < C; c++) { uint32_t v; memcpy(&v, p + 4 * c, 4); out.cov[c] = (int32_t)v; }
    p += 4 * C;
    for (int c = 0; c < C; c++) out.edges[c] = p[c];
}

void CortexGraph::cache_put(int64_t idx, const std::string& kmer) {
    if (!use_cache_) return;
    // cache.put(recordsSeen, cr); cache.put(cr.getKmerAsByteKmer(), cr)  :224-225 — one LRUMap, two keys
    auto touch_idx = by_idx_.find(idx);
    if (touch_idx != by_idx_.end()) { lru_.erase(touch_idx->second); by_idx_.erase(touch_idx); }
    lru_.push_front({false, idx, std::string()});
    by_idx_[idx] = lru_.begin();
    auto touch_k = by_kmer_.find(kmer);
    if (touch_k != by_kmer_.end()) { lru_.erase(touch_k->second); by_kmer_.erase(touch_k); }
    lru_.push_front({true, idx, kmer});
    by_kmer_[kmer] = lru_.begin();
    while (lru_.size() > kCacheMax) {
        auto& e = lru_.back();
        if (e.by_kmer) by_kmer_.erase(e.kmer); else by_idx_.erase(e.idx);
        lru_.pop_back();
    }
}

bool CortexGraph::get_record(int64_t i, Record& out) {
    if (i < 0) throw CortexJDKException("Record index is prefix of range (" + std::to_string(i) + " vs 0-" + std::to_string(num_records - 1) + ")");
    if (i >= num_records) return false;    // Q2
    if (use_cache_) {
        auto it = by_idx_.find(i);
        if (it != by_idx_.end()) {
            cache_hits_by_index++;
            lru_.splice(lru_.begin(), lru_, it->second);
            decode_at(i, out);
            return true;
        }
    }
    decode_at(i, out);
    if (use_cache_) cache_put(i, out.kmer_string());
    return true;
}

// CortexByteKmer.compareTo — byte-wise, J/utils/kmer/CortexByteKmer.java:41-49
static int byte_kmer_compare(const std::string& a, const std::string& b) {
    for (size_t i = 0; i < a.size(); i++) {
        if ((signed char)a[i] < (signed char)b[i]) return -1;
        if ((signed char)a[i] > (signed char)b[i]) return 1;
    }
    return 0;
}

int64_t CortexGraph::find_record(const std::string& bk) {
    if (tuned) return find_record_tuned(bk);
    std::string kmer = canonical(bk);
    if (use_cache_) {
        auto it = by_kmer_.find(kmer);
        if (it != by_kmer_.end()) {
            cache_hits_by_kmer++;
            lru_.splice(lru_.begin(), lru_, it->second);
            return it->second->idx;
        }
    }
    int64_t start = 0, stop = num_records - 1, mid = start + (stop - start) / 2;
    Record rs, rm, re;
    while (start != mid && mid != stop) {
        get_record(start, rs);
        get_record(mid, rm);
        get_record(stop, re);
        std::string ks = rs.kmer_string(), km = rm.kmer_string(), ke = re.kmer_string();
        if (byte_kmer_compare(ks, ke) > 0)
            throw CortexJDKException("Records are not sorted ('" + ks + "' is found before '" + ke + "' but is lexicographically greater)");
        if (byte_kmer_compare(ks, km) > 0)
            throw CortexJDKException("Records are not sorted ('" + ks + "' is found before '" + km + "' but is lexicographically greater)");
        if (byte_kmer_compare(kmer, ke) > 0 || byte_kmer_compare(kmer, ks) < 0) return -1;
        else if (ks == kmer) return start;
        else if (km == kmer) return mid;
        else if (ke == kmer) return stop;
        else if (byte_kmer_compare(kmer, ks) > 0 && byte_kmer_compare(kmer, km) < 0) {
            stop = mid;
            mid = start + (stop - start) / 2;
        } else if (byte_kmer_compare(kmer, km) > 0 && byte_kmer_compare(kmer, ke) < 0) {
            start = mid;
            mid = start + ((stop - start) / 2);
        }
    }
    return -1;
}

bool CortexGraph::find_record(const std::string& kmer, Record& out, int64_t* idx) {
    int64_t i = find_record(kmer);
    if (idx) *idx = i;
    if (i < 0) return false;
    decode_at(i, out);
    return true;
}

int64_t CortexGraph::find_record_tuned(const std::string& bk) const {
    if (num_records <= 2) return -1;                      // Q1 (cold cache)
    std::string kmer = canonical(bk);
    uint64_t q[8] = {0};
    for (char ch : kmer) {                                // Q4: non-ACGT queries miss
        if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') return -1;
    }
    {
        int len = (int)kmer.size();
        for (int i = 0; i < len; i++) {
            int bit = 2 * (len - 1 - i);
            int w = W - 1 - bit / 64;
            q[w] |= (uint64_t)char_to_nuc(kmer[i]) << (bit % 64);
        }
    }
    int64_t lo = 0, hi = num_records - 1;
    while (lo <= hi) {
        int64_t mid = lo + (hi - lo) / 2;
        const uint8_t* p = base_ + data_offset + mid * record_size;
        int cmp = 0;
        for (int w = 0; w < W && cmp == 0; w++) {
            uint64_t v; memcpy(&v, p + 8 * w, 8);
            cmp = v < q[w] ? -1 : (v > q[w] ? 1 : 0);
        }
        if (cmp == 0) return mid;
        if (cmp < 0) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

int CortexGraph::color_for_sample_name(const std::string& name) const {
    int color = -1, copies = 0;
    for (int c = 0; c < C; c++)
        if (iequals(colors[c].sample_name, name)) { color = c; copies++; }
    if (color == -1) {
        try {
            size_t pos = 0;
            int v = std::stoi(name, &pos);
            if (pos == name.size()) { color = v; copies = 1; }
        } catch (...) {}
    }
    return copies == 1 ? color : -1;
}

// ------------------------------------------------------------------ CortexGraphWriter
void write_cortex_graph(const std::string& path, int k, const std::vector<ColorInfo>& colors,
                        const std::vector<Record>& records) {
    std::string out;
    auto u32 = [&](uint32_t v) { out.append((const char*)&v, 4); };
    auto u64 = [&](uint64_t v) { out.append((const char*)&v, 8); };
    int C = (int)colors.size(), W = Record::kmer_bits(k);
    out += "CORTEX";
    u32(6); u32((uint32_t)k); u32((uint32_t)W); u32((uint32_t)C);
    for (auto& c : colors) u32(c.mean_read_length);
    for (auto& c : colors) u64(c.total_sequence);
    for (auto& c : colors) { u32((uint32_t)c.sample_name.size()); out += c.sample_name; }
    static const unsigned char err[16] = {0, 0xd8, 0xa3, 0x70, 0x3d, 0x0a, 0xd7, 0xa3, 0xf8, 0x3f, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < C; c++) out.append((const char*)err, 16);
    for (auto& c : colors) {
        out.push_back(c.tip_clipping ? 1 : 0);
        out.push_back(c.low_covg_supernodes_removed ? 1 : 0);
        out.push_back(c.low_covg_kmers_removed ? 1 : 0);
        out.push_back(c.cleaned_against_graph ? 1 : 0);
        u32(c.low_cov_supernodes_threshold);
        u32(c.low_cov_kmer_threshold);
        u32((uint32_t)c.cleaned_against_graph_name.size());
        out += c.cleaned_against_graph_name;
    }
    out += "CORTEX";
    for (auto& r : records) {
        for (int w = 0; w < W; w++) { uint64_t v = __builtin_bswap64((uint64_t)r.bk[w]); u64(v); }   // putLong big-endian
        for (int c = 0; c < C; c++) u32((uint32_t)r.cov[c]);
        for (int c = 0; c < C; c++) out.push_back((char)r.edges[c]);
    }
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw CortexJDKException("Unable to open file '" + path + "'");
    fwrite(out.data(), 1, out.size(), f);
    fclose(f);
}

// ------------------------------------------------------------------ TempGraphAssembler
void temp_graph_assembler(const std::string& out_path,
                          const std::vector<std::pair<std::string, std::vector<std::string>>>& haps, int k) {
    std::map<std::string, Record> crs;    // TreeMap<CanonicalKmer, CortexRecord>: ASCII order
    int nc = (int)haps.size();
    for (int color = 0; color < nc; color++) {
        for (std::string seq : haps[color].second) {
            for (auto& ch : seq) ch = (char)toupper((unsigned char)ch);
            for (int i = 0; i + k <= (int)seq.size(); i++) {
                std::string sk = seq.substr(i, k);
                bool has_prev = i != 0, has_next = i != (int)seq.size() - k;
                char prev = has_prev ? seq[i - 1] : 0, next = has_next ? seq[i + k] : 0;
                CanonicalKmer ck(sk);
                std::vector<int32_t> covs(nc);
                std::vector<std::set<char>> in(nc), out(nc);
                auto old = crs.find(ck.kmer);
                for (int c = 0; c < nc; c++) {
                    int cov = 0;
                    if (old != crs.end()) {
                        cov += old->second.cov[c];
                        for (char e : old->second.in_edges(c, false)) in[c].insert(e);
                        for (char e : old->second.out_edges(c, false)) out[c].insert(e);
                    }
                    if (c == color) {
                        cov++;
                        if (!ck.flipped) {
                            if (has_prev) in[c].insert(prev);
                            if (has_next) out[c].insert(next);
                        } else {
                            if (has_next) in[c].insert(complement(next));
                            if (has_prev) out[c].insert(complement(prev));
                        }
                    }
                    covs[c] = cov;
                }
                crs[ck.kmer] = Record(canonical(sk), covs, in, out);
            }
        }
    }
    std::vector<ColorInfo> colors(nc);
    for (int c = 0; c < nc; c++) colors[c].sample_name = haps[c].first;
    std::vector<Record> recs;
    for (auto& kv : crs) recs.push_back(kv.second);
    write_cortex_graph(out_path, k, colors, recs);
}

// ------------------------------------------------------------------ links
int32_t JunctionsRecord::jhash() const {
    uint32_t r = is_fw ? 1u : 0u;
    r = 31u * r + (uint32_t)num_kmers;
    r = 31u * r + (uint32_t)num_junctions;
    r = 31u * r + (uint32_t)jhash_ints(coverages);
    r = 31u * r + (uint32_t)jhash_string(junctions);
    return (int32_t)r;
}
std::vector<JunctionsRecord> LinksRecord::junctions() const {
    std::vector<int32_t> h;
    for (auto& j : cjs_insertion) h.push_back(j.jhash());
    std::vector<JunctionsRecord> out;
    for (size_t i : jhash_iteration_order(h)) out.push_back(cjs_insertion[i]);
    return out;
}
static std::string junction_to_string(const JunctionsRecord& j) {   // CortexJunctionsRecord.toString :32-47
    std::string s = j.is_fw ? "F " : "R ";
    s += std::to_string(j.num_junctions) + " ";
    for (size_t i = 0; i < j.coverages.size(); i++) { if (i) s += ","; s += std::to_string(j.coverages[i]); }
    s += " " + j.junctions;
    return s;
}
std::string LinksRecord::to_string() const {
    auto js = junctions();
    std::string s = kmer + " " + std::to_string(js.size()) + "\n";
    for (size_t i = 0; i < js.size(); i++) { s += junction_to_string(js[i]); if (i + 1 < js.size()) s += "\n"; }
    return s;
}

static std::string read_gz_all(const std::string& path) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) throw CortexJDKException("Unable to load Cortex links file '" + path + "'");
    std::string out;
    char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof buf)) > 0) out.append(buf, n);
    gzclose(f);
    return out;
}
// minimal key lookup inside the JSON header (org.json is only a container in the reference)
static bool json_find_number(const std::string& js, const std::string& key, int64_t& out) {
    size_t p = js.find("\"" + key + "\"");
    if (p == std::string::npos) return false;
    p = js.find(':', p);
    if (p == std::string::npos) return false;
    p++;
    while (p < js.size() && isspace((unsigned char)js[p])) p++;
    size_t e = p;
    while (e < js.size() && (isdigit((unsigned char)js[e]) || js[e] == '-')) e++;
    if (e == p) return false;
    out = std::stoll(js.substr(p, e - p));
    return true;
}
static std::vector<std::string> json_find_strings(const std::string& js, const std::string& key) {
    std::vector<std::string> out;
    size_t p = 0;
    std::string pat = "\"" + key + "\"";
    while ((p = js.find(pat, p)) != std::string::npos) {
        size_t c = js.find(':', p + pat.size());
        size_t q1 = js.find('"', c);
        size_t q2 = js.find('"', q1 + 1);
        out.push_back(js.substr(q1 + 1, q2 - q1 - 1));
        p = q2;
    }
    return out;
}
static std::vector<std::string> split_ws(const std::string& s, const char* extra = "") {
    std::vector<std::string> out;
    std::string cur;
    for (char ch : s) {
        if (isspace((unsigned char)ch) || strchr(extra, ch)) { if (!cur.empty()) { out.push_back(cur); cur.clear(); } }
        else cur.push_back(ch);
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

CortexLinks::CortexLinks(const std::string& path) {
    // CortexLinks.initialize :16-25: an ".idx" file next to the links file selects the random-access back-end
    const bool indexed = [&] { FILE* f = fopen((path + ".idx").c_str(), "rb"); if (f) fclose(f); return f != nullptr; }();
    std::string text = read_gz_all(path);
    std::vector<std::string> lines;
    { std::string l; std::istringstream is(text); while (std::getline(is, l)) lines.push_back(l); }
    size_t li = 0;
    std::string header;
    bool in_header = false;
    for (; li < lines.size(); li++) {
        if (lines[li] == "{") in_header = true;
        if (in_header) header += lines[li] + "\n";
        if (lines[li] == "}") { li++; break; }
    }
    int64_t v;
    if (!json_find_number(header, "formatVersion", v) && !json_find_number(header, "format_version", v))
        throw CortexJDKException("Cannot parse CortexLinks format version field");
    version = (int)v;
    if (version != 2 && version != 3 && version != 4)
        throw CortexJDKException("Cannot parse CortexLinks format version '" + std::to_string(version) + "'");
    if (version == 2) { json_find_number(header, "ncols", v); num_colors = (int)v; }
    else { json_find_number(header, "num_colours", v); num_colors = (int)v; }
    json_find_number(header, "kmer_size", v); k = (int)v;
    json_find_number(header, "num_kmers_in_graph", num_kmers_in_graph);
    json_find_number(header, "num_kmers_with_paths", num_kmers_with_links);
    json_find_number(header, "num_paths", num_links);
    json_find_number(header, "path_bytes", link_bytes);
    sample_names = json_find_strings(header, "sample");
    if (indexed) {
        // CortexLinksRandomAccess.initialize :33-89 takes the header from the LNKIDX file (big-endian), not from the JSON text
        std::vector<unsigned char> ix;
        { FILE* f = fopen((path + ".idx").c_str(), "rb"); unsigned char b[4096]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) ix.insert(ix.end(), b, b + n); fclose(f); }
        auto be32 = [&](size_t p) { return (int64_t)(((uint32_t)ix[p] << 24) | ((uint32_t)ix[p + 1] << 16) | ((uint32_t)ix[p + 2] << 8) | ix[p + 3]); };
        auto be64 = [&](size_t p) { return (int64_t)(((uint64_t)be32(p) << 32) | (uint64_t)(uint32_t)be32(p + 4)); };
        if (ix.size() < 42) throw CortexJDKException("Error in decoding Cortex links index");
        size_t p = 6;
        num_colors = (int)be32(p); p += 4;
        k = (int)be32(p); p += 4;
        num_kmers_in_graph = be64(p); p += 8;
        num_kmers_with_links = be64(p); p += 8;
        link_bytes = be64(p); p += 8;
        const size_t sl = (size_t)be32(p); p += 4;
        source = std::string((const char*)&ix[p], sl); p += sl;
        sample_names.clear();
        for (int c = 0; c < num_colors; c++) { const size_t n = (size_t)be32(p); p += 4; sample_names.emplace_back((const char*)&ix[p], n); p += n; }
        if (std::string((const char*)&ix[0], 6) != std::string((const char*)&ix[p], 6)) throw CortexJDKException("Error in decoding Cortex links index");
        version = 4;         // the records are CortexLinksRecord.toString() texts whatever the version of the indexed file was
        num_links = 0;
    }
    // skip comments / blank lines  (CortexLinksIterable.java:133-144)
    for (; li < lines.size(); li++) {
        if (lines[li].empty() || lines[li][0] == '#') continue;
        break;
    }
    for (int64_t r = 0; r < num_kmers_with_links && li < lines.size(); r++) {
        auto kl = split_ws(lines[li++]);
        LinksRecord rec;
        rec.kmer = kl[0];
        int n = std::stoi(kl[1]);
        for (int i = 0; i < n; i++) {
            auto f = split_ws(lines[li++], ",");
            JunctionsRecord j;
            j.is_fw = f[0] == "F";
            if (indexed) {
                // CortexLinksRandomAccess -> CortexLinksRecord(byte[]) :17-43: "orientation x coverages junctions" with
                // numKmers := x and numJunctions := junctions.length() (quirk Q11: another hashCode, another set order)
                j.num_kmers = std::stoi(f[1]);
                num_links++;
                for (int c = 0; c < num_colors; c++) j.coverages.push_back(std::stoi(f[2 + c]));
                j.junctions = f[2 + num_colors];
                j.num_junctions = (int)j.junctions.size();
            } else {
            j.num_kmers = version == 4 ? -1 : std::stoi(f[1]);
            j.num_junctions = version == 4 ? std::stoi(f[1]) : std::stoi(f[2]);
            int off = version == 4 ? 2 : 3;
            for (int c = 0; c < num_colors; c++) j.coverages.push_back(std::stoi(f[off + c]));
            j.junctions = f[off + num_colors];
            }
            if (std::find(rec.cjs_insertion.begin(), rec.cjs_insertion.end(), j) == rec.cjs_insertion.end())
                rec.cjs_insertion.push_back(j);
        }
        // recordHash.put(new CortexBinaryKmer(clr.getKmer().getKmerAsBytes()), clr): later records replace earlier
        std::string key = canonical(rec.kmer);
        auto it = map_.find(key);
        records.push_back(rec);
        if (it == map_.end()) map_[key] = records.size() - 1; else it->second = records.size() - 1;
    }
}

// ------------------------------------------------------------------ TempLinksAssembler
void temp_links_assembler(CortexGraph& graph, const std::vector<std::string>& reads,
                          const std::string& sample, const std::string& out_path) {
    int color = graph.color_for_sample_name(sample);
    int k = graph.k;
    // loadGraph :108-149 — directed string graph over both orientations
    std::unordered_map<std::string, std::pair<std::set<std::string>, std::set<std::string>>> g;   // v -> (in, out)
    auto add_vertex = [&](const std::string& v) { g[v]; };
    auto add_edge = [&](const std::string& s, const std::string& t) { g[s].second.insert(t); g[t].first.insert(s); };
    Record cr;
    for (int64_t i = 0; i < graph.num_records; i++) {
        graph.get_record(i, cr);
        if (cr.cov[color] > 0) {
            std::string fwd = cr.kmer_string();
            add_vertex(fwd);
            for (char e : cr.in_edges(color, false)) { std::string s = std::string(1, e) + fwd.substr(0, k - 1); add_vertex(s); add_edge(s, fwd); }
            for (char e : cr.out_edges(color, false)) { std::string s = fwd.substr(1) + e; add_vertex(s); add_edge(fwd, s); }
            std::string rev = reverse_complement(fwd);
            add_vertex(rev);
            for (char e : cr.out_edges(color, true)) { std::string s = std::string(1, e) + rev.substr(0, k - 1); add_vertex(s); add_edge(s, rev); }
            for (char e : cr.in_edges(color, true)) { std::string s = rev.substr(1) + e; add_vertex(s); add_edge(rev, s); }
        }
    }
    auto outdeg = [&](const std::string& v) -> int {
        auto it = g.find(v); if (it == g.end()) throw std::runtime_error("no such vertex in graph: " + v); return (int)it->second.second.size(); };
    auto indeg = [&](const std::string& v) -> int {
        auto it = g.find(v); if (it == g.end()) throw std::runtime_error("no such vertex in graph: " + v); return (int)it->second.first.size(); };

    // linkMap: HashMap<CanonicalKmer, Set<CortexJunctionsRecord>> — insertion order + hashes kept for emulation
    std::vector<std::string> lm_keys;
    std::unordered_map<std::string, std::vector<JunctionsRecord>> lm;
    for (const std::string& hap_fwd : reads) {
        std::string hap_rev = reverse_complement(hap_fwd);
        for (const std::string& hap : {hap_fwd, hap_rev}) {
            // links: HashMap<Pair<String,Integer>, String>; Pair.hashCode = (key==null?0:key.hashCode())*31 ... see below
            std::vector<std::pair<std::string, int>> lkeys;
            std::map<std::pair<std::string, int>, std::string> links;
            for (int j = 1; j <= (int)hap.size() - k; j++) {
                std::string sk0 = hap.substr(j - 1, k), sk1 = hap.substr(j, k);
                char edge = hap[j + k - 1];
                if (outdeg(sk0) > 1 && g.count(sk1)) {
                    for (int i = 1; i <= j; i++) {
                        std::string ski = hap.substr(i, k);
                        if (indeg(ski) > 1) {
                            auto key = std::make_pair(hap.substr(i - 1, k), i);
                            if (!links.count(key)) { links[key] = ""; lkeys.push_back(key); }
                            links[key] += edge;
                        }
                    }
                }
            }
            // iteration order of `links` only affects the insertion order into the per-k-mer HashSet;
            // org.apache.commons.math3.util.Pair.hashCode: result = key.hashCode(); result = 37*result + value.hashCode()
            std::vector<int32_t> lh;
            for (auto& kk : lkeys) {
                uint32_t r = (uint32_t)jhash_string(kk.first);
                uint32_t hv = (uint32_t)kk.second;
                r = (37u * r + hv) ^ (hv >> 16);   // commons-math3 Pair.hashCode: `37 * result + h ^ (h >>> 16)`
                lh.push_back((int32_t)r);
            }
            for (size_t oi : jhash_iteration_order(lh)) {
                auto& p = lkeys[oi];
                CanonicalKmer ck(p.first);
                if (!lm.count(ck.kmer)) { lm[ck.kmer]; lm_keys.push_back(ck.kmer); }
                JunctionsRecord j;
                j.is_fw = !ck.flipped;
                j.num_kmers = (int)links[p].size();
                j.num_junctions = (int)links[p].size();
                j.coverages = {1};
                j.junctions = links[p];
                auto& vec = lm[ck.kmer];
                if (std::find(vec.begin(), vec.end(), j) == vec.end()) vec.push_back(j);
            }
        }
    }
    int num_paths = 0;
    for (auto& kk : lm_keys) num_paths += (int)lm[kk].size();

    std::ostringstream os;
    os << "{\n"
       << "        \"file_format\": \"ctp\",\n"
       << "        \"format_version\": 4,\n"
       << "        \"file_key\": 0,\n"
       << "        \"graph\": {\n"
       << "                \"num_colours\": 1,\n"
       << "                \"kmer_size\": " << k << ",\n"
       << "                \"num_kmers_in_graph\": " << graph.num_records << ",\n"
       << "                \"colours\": [{\n"
       << "                        \"colour\": 0,\n"
       << "                        \"sample\": \"" << sample << "\",\n"
       << "                        \"total_sequence\": 0,\n"
       << "                        \"cleaned_tips\": false,\n"
       << "                        \"cleaned_unitigs\": false\n"
       << "                }]\n"
       << "        },\n"
       << "        \"paths\": {\n"
       << "                \"num_kmers_with_paths\": " << lm_keys.size() << ",\n"
       << "                \"num_paths\": " << num_paths << ",\n"
       << "                \"path_bytes\": " << num_paths << "\n"
       << "        }\n"
       << "}";
    os << "\n\n";
    // for (CanonicalKmer ck : linkMap.keySet()) — HashMap order over Arrays.hashCode(kmer bytes)
    std::vector<int32_t> kh;
    for (auto& kk : lm_keys) kh.push_back(jhash_bytes(kk));
    for (size_t oi : jhash_iteration_order(kh)) {
        const std::string& kk = lm_keys[oi];
        // new CortexLinksRecord(kmer, new ArrayList<>(linkMap.get(ck))): ArrayList in HashSet order, re-hashed into cjs
        LinksRecord tmp; tmp.kmer = kk; tmp.cjs_insertion = lm[kk];
        LinksRecord clr; clr.kmer = kk; clr.cjs_insertion = tmp.junctions();
        os << clr.to_string() << "\n";
    }
    os << "\n";
    std::string text = os.str();
    gzFile f = gzopen(out_path.c_str(), "wb");
    if (!f) throw CortexJDKException("Could not get a temp file for links creation");
    gzwrite(f, text.data(), (unsigned)text.size());
    gzclose(f);
}

}  // namespace orc
