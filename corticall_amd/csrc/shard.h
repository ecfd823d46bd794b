// Walks over a hash-partitioned table: host interface of shard.cpp (device side of the bulk-synchronous walker).
#pragma once
#include "engine_host.h"

namespace ldbg {

// canonical neighbour k-mers of local records [first, first+n): 8 per record (4 successors, 4 predecessors), all-ones
// words where no colour carries the edge; flips[t] = the neighbour's canonical form is its reverse complement
void shard_nbr_queries(const Graph& g, int64_t first, int64_t n, uint64_t* d_words, uint8_t* d_flips);
// the routed findRecord answers for those queries -> the shard's global neighbour index
void shard_set_nbr(Graph& g, int64_t first, int64_t n, const int32_t* d_owner, const int64_t* d_lidx, const uint8_t* d_flips);
int shard_row_bytes(const Graph& g);
// rows of local records for the walks that asked for them: 8 global neighbour ids | flags | C edge bytes
void shard_rows(const Graph& g, const int64_t* d_lidx, int64_t n, uint8_t* d_rows);

class BspWalker {
public:
    explicit BspWalker(const Engine& e);      // e: an engine over this rank's shard (configuration + validation)
    ~BspWalker();
    int row_bytes() const;
    // n seeds -> 2n strands; fills the first requests (owner -1 = none)
    void start(int64_t n_seeds, const int32_t* d_owner, const int64_t* d_lidx, const uint8_t* d_flip, int32_t* d_req_owner, int64_t* d_req_lidx);
    // rows that arrived (have_row[s] != 0, row at d_rows + s * row_bytes) -> one loop iteration each; next requests
    void step(const uint8_t* d_have_row, const uint8_t* d_rows, int32_t* d_req_owner, int64_t* d_req_lidx);
    // host copies: vertices per strand (0 = empty graph / null), status, loop iterations, appended bases [strand][stride]
    void results(uint32_t* strand_n, uint32_t* status, uint32_t* iters, uint8_t* bases, int64_t bases_stride);
private:
    const Engine& eng_;
    struct Impl;
    Impl* impl_;
};

}  // namespace ldbg
