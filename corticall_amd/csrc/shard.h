// Traversals over a hash-partitioned table: host interface of shard.cpp (global neighbour index) and image.cpp (the local image).
#pragma once
#include <memory>

#include "engine_host.h"

namespace ldbg {

// canonical neighbour k-mers of local records [first, first+n): 8 per record (4 successors, 4 predecessors); have[t] = 0 where
// no colour carries the edge (no query); flips[t] = the neighbour's canonical form is its reverse complement
void shard_nbr_queries(const Graph& g, int64_t first, int64_t n, uint64_t* d_words, uint8_t* d_flips, uint8_t* d_have);
// the routed findRecord answers for those queries -> the shard's global neighbour index
void shard_set_nbr(Graph& g, int64_t first, int64_t n, const int32_t* d_owner, const int64_t* d_lidx, const uint8_t* d_flips);

// the local image of the sharded table on one rank (image.h, image.cpp)
struct ImageView;
class ShardImage {
public:
    // shard: this rank's shard with its global neighbour index; cap: rows the image can hold; global_records: records of the whole table (Q1)
    ShardImage(const Graph& shard, int64_t cap, int64_t global_records);
    ~ShardImage();
    const Graph& graph() const { return *graph_; }            // the image as a graph: what engines over the sharded table are created on
    Graph& graph() { return *graph_; }
    const Graph& shard() const { return shard_; }
    int64_t capacity() const { return cap_; }
    int row_bytes() const;                                     // bytes of one served row: global id key | 8 neighbour ids | probe row
    void clear();                                              // forget every row (a new batch starts from an empty image)
    // owner side: rows for `n` requested global id keys -> d_out[n][depth][row_bytes]: the row asked for, then up to depth - 1 rows around it
    // that live on this shard too (key 0 = unused slot)
    void serve(int my_rank, const unsigned long long* d_keys, int64_t n, int depth, uint8_t* d_out, rt::stream_t s) const;
    // requester side: rows that arrived (key 0 = none); e: the engine whose link table names the rows' link records (may be null)
    void insert(const Engine* e, const uint8_t* d_rows, int64_t n, rt::stream_t s);
    void lookup(const unsigned long long* d_keys, int64_t n, int32_t* d_slots, rt::stream_t s) const;
    // this round's requests -> d_send[world][cap_per_owner] global id keys, 0 = unused
    void bucket(int world, uint32_t cap_per_owner, unsigned long long* d_send, rt::stream_t s) const;
    void request(const unsigned long long* d_keys, int64_t n, rt::stream_t s);     // explicit requests (seeds, sinks) join the round's list
    void reset_requests(rt::stream_t s);
    void counters(int64_t* n_rows, int64_t* n_req, int* overflow) const;
    ImageView view(uint64_t* rec_of) const;
private:
    const Graph& shard_;
    std::unique_ptr<Graph> graph_;
    int64_t cap_ = 0;
    uint64_t hcap_ = 0;
    uint32_t req_cap_ = 0;
    void* d_nbrg_ = nullptr; void* d_gkey_ = nullptr; void* d_hkeys_ = nullptr; void* d_hvals_ = nullptr; void* d_ctr_ = nullptr; void* d_req_ = nullptr;
    void* d_rec_of_own_ = nullptr; void* d_bcount_ = nullptr;
    void* d_req_seen_ = nullptr;
    mutable void* d_plan_ = nullptr; mutable size_t plan_bytes_ = 0;      // serve: the records picked for every request's slots
};

}  // namespace ldbg
