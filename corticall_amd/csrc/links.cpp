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

// ---- BGZF (a series of gzip members of at most 64 KB, each carrying its compressed size in a "BC" extra field; SAM specification
// §4.1) — the container of an indexed link file (.ctp.bgz): a virtual offset = compressed offset of a block << 16 | offset inside
// its uncompressed data.  The reference reads and writes it through htsjdk's BlockCompressedInput/OutputStream.
struct BgzfWriter {
    FILE* f;
    std::string buf;                 // uncompressed bytes of the block being filled
    uint64_t block_addr = 0;         // compressed offset of that block
    static constexpr size_t kBlock = 0xFF00;
    explicit BgzfWriter(const std::string& path) : f(fopen(path.c_str(), "wb")) {
        if (!f) throw StatusError(LDBG_ERR_CORTEXJDK, "cannot write '" + path + "'");
    }
    ~BgzfWriter() { if (f) fclose(f); }
    uint64_t position() { if (buf.size() >= kBlock) flush(); return (block_addr << 16) | (uint64_t)buf.size(); }   // BlockCompressedOutputStream.getPosition
    void write(const std::string& d) {
        size_t o = 0;
        while (o < d.size()) {
            const size_t n = std::min(kBlock - buf.size(), d.size() - o);
            buf.append(d, o, n);
            o += n;
            if (buf.size() >= kBlock) flush();
        }
    }
    void flush() { if (!buf.empty()) { put_block(buf); buf.clear(); } }
    void put_block(const std::string& data) {
        std::vector<uint8_t> out(compressBound((uLong)data.size()) + 64);
        z_stream z{};
        if (deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw StatusError(LDBG_ERR_CORTEXJDK, "deflateInit2 failed");
        z.next_in = (Bytef*)data.data(); z.avail_in = (uInt)data.size();
        z.next_out = out.data() + 18; z.avail_out = (uInt)(out.size() - 26);
        const int rc = deflate(&z, Z_FINISH);
        const size_t clen = z.total_out;
        deflateEnd(&z);
        if (rc != Z_STREAM_END || clen + 26 > 65536) throw StatusError(LDBG_ERR_CORTEXJDK, "BGZF block does not fit");
        static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(out.data(), hdr, 16);
        const uint16_t bsize = (uint16_t)(clen + 25);
        out[16] = (uint8_t)(bsize & 0xff); out[17] = (uint8_t)(bsize >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)data.data(), (uInt)data.size()), isz = (uint32_t)data.size();
        memcpy(out.data() + 18 + clen, &crc, 4);
        memcpy(out.data() + 22 + clen, &isz, 4);
        if (fwrite(out.data(), 1, clen + 26, f) != clen + 26) throw StatusError(LDBG_ERR_CORTEXJDK, "short write");
        block_addr += clen + 26;
    }
    void close() {
        flush();
        put_block(std::string());        // the end-of-file marker: an empty block
        fclose(f); f = nullptr;
    }
};
struct BgzfReader {
    FILE* f;
    uint64_t cur_addr = ~0ull, cur_clen = 0;     // the block held in `data`
    std::string data;
    explicit BgzfReader(const std::string& path) : f(fopen(path.c_str(), "rb")) {
        if (!f) throw StatusError(LDBG_ERR_CORTEXJDK, "Unable to load Cortex links file '" + path + "'");
    }
    ~BgzfReader() { if (f) fclose(f); }
    bool load(uint64_t addr) {
        if (addr == cur_addr) return true;
        uint8_t h[18];
        if (fseeko(f, (off_t)addr, SEEK_SET) != 0 || fread(h, 1, 18, f) != 18) return false;
        if (h[0] != 0x1f || h[1] != 0x8b || !(h[3] & 4) || h[12] != 'B' || h[13] != 'C')
            throw StatusError(LDBG_ERR_CORTEXJDK, "Failed to load links record from disk (not a BGZF block)");
        const size_t bsize = (size_t)(h[16] | (h[17] << 8)) + 1, clen = bsize - 26;
        std::vector<uint8_t> c(clen + 8);
        if (fread(c.data(), 1, clen + 8, f) != clen + 8) throw StatusError(LDBG_ERR_CORTEXJDK, "Failed to load links record from disk (truncated block)");
        uint32_t isz;
        memcpy(&isz, c.data() + clen + 4, 4);
        data.assign(isz, '\0');
        z_stream z{};
        if (inflateInit2(&z, -15) != Z_OK) throw StatusError(LDBG_ERR_CORTEXJDK, "inflateInit2 failed");
        z.next_in = c.data(); z.avail_in = (uInt)clen;
        z.next_out = (Bytef*)data.data(); z.avail_out = isz;
        const int rc = isz ? inflate(&z, Z_FINISH) : Z_STREAM_END;
        inflateEnd(&z);
        if (rc != Z_STREAM_END) throw StatusError(LDBG_ERR_CORTEXJDK, "Failed to load links record from disk (corrupt block)");
        cur_addr = addr; cur_clen = bsize;
        return true;
    }
    // BlockCompressedInputStream.seek(virtual offset) + read(len bytes)
    std::string read(uint64_t voff, size_t len) {
        std::string out;
        uint64_t addr = voff >> 16;
        size_t off = (size_t)(voff & 0xFFFF);
        while (out.size() < len) {
            if (!load(addr)) throw StatusError(LDBG_ERR_CORTEXJDK, "Failed to load links record from disk");
            if (data.empty() && off == 0) throw StatusError(LDBG_ERR_CORTEXJDK, "Failed to load links record from disk (past the end)");
            const size_t n = std::min(len - out.size(), data.size() - std::min(off, data.size()));
            out.append(data, off, n);
            addr += cur_clen; off = 0;
        }
        return out;
    }
};
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
uint64_t be64(const uint8_t* p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }
void put_be32(std::string& o, uint32_t v) { for (int i = 3; i >= 0; i--) o.push_back((char)(v >> (8 * i))); }
void put_be64(std::string& o, uint64_t v) { put_be32(o, (uint32_t)(v >> 32)); put_be32(o, (uint32_t)v); }

// the LNKIDX file (IndexLinks.java:62-135; read by CortexLinksRandomAccess.java:33-89): big-endian throughout
struct LinkIndex {
    int num_colors = 0, k = 0;
    int64_t num_kmers_in_graph = 0, num_kmers_with_links = 0, link_bytes = 0;
    std::string source;
    std::vector<std::string> sample_names;
    struct Entry { uint64_t voff; uint32_t len; };
    std::vector<Entry> entries;          // in index order (the TreeMap order of the records' k-mer strings)
};
LinkIndex read_link_index(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw StatusError(LDBG_ERR_CORTEXJDK, "IOException: cannot open '" + path + "'");
    std::vector<uint8_t> b;
    { uint8_t tmp[1 << 16]; size_t n; while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) b.insert(b.end(), tmp, tmp + n); }
    fclose(f);
    size_t p = 0;
    auto need = [&](size_t n) { if (p + n > b.size()) throw StatusError(LDBG_ERR_CORTEXJDK, "Error in decoding Cortex links index (truncated)"); };
    LinkIndex ix;
    need(6 + 4 + 4 + 24 + 4);
    const std::string magic0((const char*)&b[0], 6);
    p = 6;
    ix.num_colors = (int)be32(&b[p]); p += 4;
    ix.k = (int)be32(&b[p]); p += 4;
    ix.num_kmers_in_graph = (int64_t)be64(&b[p]); p += 8;
    ix.num_kmers_with_links = (int64_t)be64(&b[p]); p += 8;
    ix.link_bytes = (int64_t)be64(&b[p]); p += 8;
    if (ix.num_colors < 0 || ix.num_colors > 4096 || ix.k <= 0 || ix.num_kmers_with_links < 0) throw StatusError(LDBG_ERR_CORTEXJDK, "Error in decoding Cortex links index");
    const uint32_t sl = be32(&b[p]); p += 4;
    need(sl); ix.source.assign((const char*)&b[p], sl); p += sl;
    for (int c = 0; c < ix.num_colors; c++) {
        need(4); const uint32_t n = be32(&b[p]); p += 4;
        need(n); ix.sample_names.emplace_back((const char*)&b[p], n); p += n;
    }
    need(6);
    if (magic0 != std::string((const char*)&b[p], 6)) throw StatusError(LDBG_ERR_CORTEXJDK, "Error in decoding Cortex links index");
    p += 6;
    const int W = (ix.k + 31) / 32;
    for (int64_t i = 0; i < ix.num_kmers_with_links; i++) {
        need((size_t)8 * W + 12);
        p += (size_t)8 * W;              // the binary k-mer: the record's own text names it again
        LinkIndex::Entry e;
        e.voff = be64(&b[p]); p += 8;
        e.len = be32(&b[p]); p += 4;
        ix.entries.push_back(e);
    }
    return ix;
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
    std::string text;
    if (indexed) {
        // CortexLinksRandomAccess (:33-89, 104-118): the header comes from the LNKIDX file, every record is fetched through its BGZF
        // virtual offset (seek + read of `length` bytes) — here all of them at once, in index order, into the text the parser below
        // reads (the JSON header of the .ctp.bgz itself is not consulted by the reference's random-access back-end)
        const LinkIndex ix = read_link_index(path + ".idx");
        BgzfReader br(path);
        std::string sn = "[";
        for (size_t c = 0; c < ix.sample_names.size(); c++) {
            std::string esc;
            for (char ch : ix.sample_names[c]) { if (ch == '"' || ch == '\\') esc.push_back('\\'); esc.push_back(ch); }
            sn += std::string(c ? "," : "") + "{\"colour\":" + std::to_string(c) + ",\"sample\":\"" + esc + "\"}";
        }
        sn += "]";
        size_t n_links = 0;
        std::string body;
        for (const auto& en : ix.entries) {
            const std::string rec = br.read(en.voff, en.len);
            for (char ch : rec) n_links += ch == '\n';
            body += rec + "\n";
        }
        text = "{\n\"format_version\":4,\"graph\":{\"num_colours\":" + std::to_string(ix.num_colors) + ",\"kmer_size\":" + std::to_string(ix.k) +
               ",\"num_kmers_in_graph\":" + std::to_string(ix.num_kmers_in_graph) + ",\"colours\":" + sn + "},\"paths\":{\"num_kmers_with_paths\":" +
               std::to_string(ix.num_kmers_with_links) + ",\"num_paths\":" + std::to_string(n_links) + ",\"path_bytes\":" + std::to_string(ix.link_bytes) + "}\n}\n\n" + body;
        source_ = ix.source;
    } else text = gunzip_file(path);
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
        if ((int)rec.kmer.size() != k || !ascii_to_words_ci(rec.kmer.c_str(), k, w.data(), W))
            throw StatusError(LDBG_ERR_CORTEXJDK, "Unable to parse CortexLinks record: bad k-mer '" + rec.kmer + "'");
        // canonical key (CortexBinaryKmer(byte[]) canonicalises, CortexBinaryKmer.java:17-19)
        std::string rcs(k, 'A');
        for (int i = 0; i < k; i++) { char ch = rec.kmer[k - 1 - i]; rcs[i] = complement_ascii(std::string(1, ch))[0]; }
        ascii_to_words_ci(rcs.c_str(), k, rc.data(), W);
        by_key[std::min(w, rc)] = rec;
        have = next_line(line);
        while (have && line.empty()) have = next_line(line);
    }

    for (auto& kv : by_key) {
        record_keys.push_back(kv.first);
        records.push_back(kv.second);
        std::vector<uint64_t> w(W);
        ascii_to_words_ci(kv.second.kmer.c_str(), k, w.data(), W);
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

// IndexLinks (J/commands/index/links/IndexLinks.java:62-135): the records of a link file re-written as a BGZF file, each record's
// virtual offset and text length in a big-endian LNKIDX file beside it, ordered by the records' k-mer strings
int64_t links_index_file(const std::string& in_path, const std::string& out_path, const std::string& source) {
    const std::string text = gunzip_file(in_path);
    size_t pos = 0;
    auto next_line = [&](std::string& line) -> bool {
        if (pos >= text.size()) return false;
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        line.assign(text, pos, e - pos);
        pos = e + 1;
        return true;
    };
    std::string line, header, comments;
    bool in_header = false;
    while (next_line(line)) {
        if (line == "{") in_header = true;
        if (in_header) header += line + "\n";
        if (line == "}") break;
    }
    JParser jp{header};
    const JVal h = jp.parse();
    const JVal* fv = h.get("formatVersion") ? h.get("formatVersion") : h.get("format_version");
    if (!fv) throw StatusError(LDBG_ERR_CORTEXJDK, "Cannot parse CortexLinks format version field");
    const int version = (int)fv->num;
    if (version != 2 && version != 3 && version != 4) throw StatusError(LDBG_ERR_CORTEXJDK, "Cannot parse CortexLinks format version '" + std::to_string(version) + "'");
    const JVal* gr = version == 2 ? &h : h.get("graph");
    const JVal* pa = version == 2 ? &h : h.get("paths");
    const int num_colors = (int)jnum(gr, version == 2 ? "ncols" : "num_colours"), k = (int)jnum(gr, "kmer_size");
    const int64_t nkg = jnum(gr, "num_kmers_in_graph"), nkl = jnum(pa, "num_kmers_with_paths"), lb = jnum(pa, "path_bytes");
    std::vector<std::string> samples;
    if (const JVal* cols = gr->get("colours")) for (auto& c : cols->arr) { const JVal* sname = c.get("sample"); samples.push_back(sname ? sname->str : ""); }
    samples.resize((size_t)num_colors);
    bool have = false;
    while (next_line(line)) {
        if (line.empty()) continue;
        if (line[0] == '#') { comments += line + "\n"; continue; }
        have = true;
        break;
    }
    BgzfWriter bw(out_path);
    bw.write(header);                       // getJSONHeader() + "\n" + getComments() + "\n"  (:110-113)
    bw.write("\n");
    bw.write(comments);
    bw.write("\n");
    std::map<std::string, std::pair<uint64_t, uint32_t>> table;          // TreeMap<CortexByteKmer, (position, length)>
    for (int64_t r = 0; r < nkl && have; r++) {
        auto kl = split_fields(line, false);
        if (kl.size() < 2) throw StatusError(LDBG_ERR_CORTEXJDK, "Unable to parse CortexLinks record");
        const int n = atoi(kl[1].c_str());
        std::vector<HostJunction> js;
        for (int i = 0; i < n; i++) {
            if (!next_line(line)) throw StatusError(LDBG_ERR_CORTEXJDK, "Unable to parse CortexLinks record");
            auto f = split_fields(line, true);
            const int off = version == 4 ? 2 : 3;
            if ((int)f.size() < off + num_colors + 1) throw StatusError(LDBG_ERR_CORTEXJDK, "Unable to parse CortexLinks record");
            HostJunction j;
            j.is_fw = f[0] == "F";
            j.num_kmers = version == 4 ? -1 : atoi(f[1].c_str());
            j.num_junctions = version == 4 ? atoi(f[1].c_str()) : atoi(f[2].c_str());
            for (int c = 0; c < num_colors; c++) j.cov.push_back(atoi(f[off + c].c_str()));
            j.junctions = f[off + num_colors];
            bool dup = false;
            for (auto& o : js) dup |= junction_eq(o, j);
            if (!dup) js.push_back(j);
        }
        hashset_order(js);                  // CortexLinksRecord.toString iterates its HashSet (:58-74)
        std::string rec = kl[0] + " " + std::to_string(js.size()) + "\n";
        for (size_t i = 0; i < js.size(); i++) {
            rec += std::string(js[i].is_fw ? "F" : "R") + " " + std::to_string(js[i].num_junctions) + " ";
            for (size_t c = 0; c < js[i].cov.size(); c++) rec += (c ? "," : "") + std::to_string(js[i].cov[c]);
            rec += " " + js[i].junctions;
            if (i + 1 < js.size()) rec += "\n";
        }
        table[kl[0]] = {bw.position(), (uint32_t)rec.size()};
        bw.write(rec);
        bw.write("\n");
        have = next_line(line);
        while (have && line.empty()) have = next_line(line);
    }
    bw.close();
    // LNKIDX (:62-95, 122-133)
    std::string ix = "LNKIDX";
    put_be32(ix, (uint32_t)num_colors); put_be32(ix, (uint32_t)k);
    put_be64(ix, (uint64_t)nkg); put_be64(ix, (uint64_t)nkl); put_be64(ix, (uint64_t)lb);
    put_be32(ix, (uint32_t)source.size()); ix += source;
    for (auto& sname : samples) { put_be32(ix, (uint32_t)sname.size()); ix += sname; }
    ix += "LNKIDX";
    const int W = (k + 31) / 32;
    for (auto& kv : table) {
        // new CortexBinaryKmer(byte[]) = the canonical orientation, 2 bits per base, each long byte-reversed (CortexRecord.java:313-334):
        // written big-endian that is the packed words in little-endian byte order, most significant word first
        std::vector<uint64_t> w(W), rc(W);
        std::string rcs(k, 'A');
        for (int i = 0; i < k; i++) rcs[i] = complement_ascii(std::string(1, kv.first[k - 1 - i]))[0];
        if ((int)kv.first.size() != k || !ascii_to_words_ci(kv.first.c_str(), k, w.data(), W) || !ascii_to_words_ci(rcs.c_str(), k, rc.data(), W))
            throw StatusError(LDBG_ERR_CORTEXJDK, "Unable to parse CortexLinks record: bad k-mer '" + kv.first + "'");
        const std::vector<uint64_t>& cw = std::min(w, rc);
        for (int i = 0; i < W; i++) ix.append((const char*)&cw[i], 8);
        put_be64(ix, kv.second.first);
        put_be32(ix, kv.second.second);
    }
    FILE* f = fopen((out_path + ".idx").c_str(), "wb");
    if (!f || fwrite(ix.data(), 1, ix.size(), f) != ix.size()) { if (f) fclose(f); throw StatusError(LDBG_ERR_CORTEXJDK, "cannot write '" + out_path + ".idx'"); }
    fclose(f);
    return (int64_t)table.size();
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
    if (!ascii_to_words_ci(kmer_ascii.c_str(), k, w.data(), W)) return nullptr;
    std::string rcs(k, 'A');
    for (int i = 0; i < k; i++) rcs[i] = complement_ascii(std::string(1, kmer_ascii[k - 1 - i]))[0];
    ascii_to_words_ci(rcs.c_str(), k, rc.data(), W);
    auto key = std::min(w, rc);
    auto it = std::lower_bound(record_keys.begin(), record_keys.end(), key);
    if (it == record_keys.end() || *it != key) return nullptr;
    return &records[it - record_keys.begin()];
}

}  // namespace ldbg
