// Host-side parsing of the Cortex .ctx v6 container (header only; records go to the device raw).
// Format: /root/reference/docs/ctx_spec.md:5-46; behaviour mirrors CortexGraph.loadCortexGraph
// (J/utils/io/graph/cortex/CortexGraph.java:66-168) including the early-NUL trimming of names
// (:50-64) and the big-endian read of total_sequence.
#pragma once
#include <stdint.h>

#include <stdexcept>
#include <string>
#include <vector>

namespace ldbg {

struct StatusError : std::runtime_error {
    int status;
    StatusError(int st, const std::string& m) : std::runtime_error(m), status(st) {}
};

struct CtxColor {
    std::string sample_name;
    uint32_t mean_read_length = 0;
    uint64_t total_sequence = 0;
    uint8_t tip_clipping = 0, low_covg_supernodes_removed = 0, low_covg_kmers_removed = 0, cleaned_against_graph = 0;
    uint32_t low_cov_supernodes_threshold = 0, low_cov_kmer_threshold = 0;
    std::string cleaned_against_graph_name;
};

struct CtxHeader {
    int version = 0, k = 0, W = 0, C = 0;
    std::vector<CtxColor> colors;
    int64_t data_offset = 0, record_size = 0, num_records = 0;
};

// Parses the header out of the first `avail` bytes of a file of `file_size` bytes.
// `what` names the file in error messages.  Throws StatusError(LDBG_ERR_CORTEXJDK, ...).
CtxHeader parse_ctx_header(const uint8_t* p, size_t avail, int64_t file_size, const std::string& what);

// The header as CortexGraphWriter.initialize writes it (J/utils/io/graph/cortex/CortexGraphWriter.java:40-113): the parsed
// values, not the input bytes — total_sequence goes out as the byte-swapped value the reader produced (Q16), the error
// rate as the writer's constant, names as trimmed by the reader.
std::vector<uint8_t> serialize_ctx_header(const CtxHeader& h);

}  // namespace ldbg
