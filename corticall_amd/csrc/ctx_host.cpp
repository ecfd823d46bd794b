#include "ctx_host.h"

#include <string.h>
#include <strings.h>

#include "../../include/ldbg.h"

namespace ldbg {

namespace {
struct Cursor {
    const uint8_t* p;
    size_t n;
    size_t off;
    const std::string& what;
    void need(size_t k) const {
        if (off + k > n)
            throw StatusError(LDBG_ERR_CORTEXJDK, "Error while parsing Cortex graph file '" + what + "': truncated header");
    }
    uint32_t u32() {
        need(4);
        uint32_t v = (uint32_t)p[off] | ((uint32_t)p[off + 1] << 8) | ((uint32_t)p[off + 2] << 16) | ((uint32_t)p[off + 3] << 24);
        off += 4;
        return v;
    }
    uint64_t u64_be() {
        need(8);
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v = (v << 8) | p[off + i];
        off += 8;
        return v;
    }
    uint8_t u8() { need(1); return p[off++]; }
    std::string str(size_t len) {
        need(len);
        std::string s((const char*)p + off, len);
        off += len;
        size_t z = s.find('\0');   // fixStringsWithEarlyTerminators
        if (z != std::string::npos) s.resize(z);
        return s;
    }
    bool magic() {
        need(6);
        bool ok = strncasecmp((const char*)p + off, "CORTEX", 6) == 0;
        off += 6;
        return ok;
    }
};
}  // namespace

CtxHeader parse_ctx_header(const uint8_t* p, size_t avail, int64_t file_size, const std::string& what) {
    Cursor c{p, avail, 0, what};
    CtxHeader h;
    if (!c.magic()) throw StatusError(LDBG_ERR_CORTEXJDK, "The file '" + what + "' does not appear to be a Cortex graph");
    h.version = (int)c.u32();
    if (h.version != 6) throw StatusError(LDBG_ERR_CORTEXJDK, "The file '" + what + "' is not a version 6 Cortex graph");
    h.k = (int)c.u32();
    h.W = (int)c.u32();
    h.C = (int)c.u32();
    if (h.k <= 0 || h.W != (h.k + 31) / 32 || h.C <= 0 || h.C > LDBG_MAX_COLORS)
        throw StatusError(LDBG_ERR_CORTEXJDK, "Error while parsing Cortex graph file '" + what + "': implausible k/W/colours (" +
                                                  std::to_string(h.k) + "/" + std::to_string(h.W) + "/" + std::to_string(h.C) + ")");
    h.colors.resize(h.C);
    for (auto& col : h.colors) col.mean_read_length = c.u32();
    for (auto& col : h.colors) col.total_sequence = c.u64_be();
    for (auto& col : h.colors) { uint32_t len = c.u32(); col.sample_name = c.str(len); }
    for (int i = 0; i < h.C; i++) { c.need(16); c.off += 16; }   // error rate, not parsed by the reference
    for (auto& col : h.colors) {
        col.tip_clipping = c.u8() != 0;
        col.low_covg_supernodes_removed = c.u8() != 0;
        col.low_covg_kmers_removed = c.u8() != 0;
        col.cleaned_against_graph = c.u8() != 0;
        col.low_cov_supernodes_threshold = c.u32();
        col.low_cov_kmer_threshold = c.u32();
        uint32_t len = c.u32();
        col.cleaned_against_graph_name = c.str(len);
    }
    if (!c.magic())
        throw StatusError(LDBG_ERR_CORTEXJDK, "We didn't see a proper header terminator at the expected place in Cortex graph '" + what + "'");
    h.data_offset = (int64_t)c.off;
    h.record_size = 8LL * h.W + 5LL * h.C;
    h.num_records = (file_size - h.data_offset) / h.record_size;
    return h;
}

std::vector<uint8_t> serialize_ctx_header(const CtxHeader& h) {
    std::vector<uint8_t> o;
    auto put = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; o.insert(o.end(), b, b + n); };
    auto u32 = [&](uint32_t v) { put(&v, 4); };
    put("CORTEX", 6);
    u32((uint32_t)h.version); u32((uint32_t)h.k); u32((uint32_t)h.W); u32((uint32_t)h.C);
    for (auto& c : h.colors) u32(c.mean_read_length);
    for (auto& c : h.colors) { uint64_t v = c.total_sequence; put(&v, 8); }          // putLong, little-endian buffer
    for (auto& c : h.colors) { u32((uint32_t)c.sample_name.size()); put(c.sample_name.data(), c.sample_name.size()); }
    static const uint8_t error_rate[16] = {0, 0xd8, 0xa3, 0x70, 0x3d, 0x0a, 0xd7, 0xa3, 0xf8, 0x3f, 0, 0, 0, 0, 0, 0};   // :72-80
    for (size_t i = 0; i < h.colors.size(); i++) put(error_rate, 16);
    for (auto& c : h.colors) {
        const uint8_t flags[4] = {(uint8_t)(c.tip_clipping ? 1 : 0), (uint8_t)(c.low_covg_supernodes_removed ? 1 : 0),
                                  (uint8_t)(c.low_covg_kmers_removed ? 1 : 0), (uint8_t)(c.cleaned_against_graph ? 1 : 0)};
        put(flags, 4);
        u32(c.low_cov_supernodes_threshold); u32(c.low_cov_kmer_threshold);
        u32((uint32_t)c.cleaned_against_graph_name.size()); put(c.cleaned_against_graph_name.data(), c.cleaned_against_graph_name.size());
    }
    put("CORTEX", 6);
    return o;
}

}  // namespace ldbg
