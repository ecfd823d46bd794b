// Run index: the records of a graph in UNITIG ORDER, for one traversal configuration.
//
// TraversalEngine.dfs (J/utils/traversal/TraversalEngine.java:373-481) advances one k-mer per loop iteration.  Nearly all
// of those iterations are spent inside unbranched stretches of the graph: vertex v has exactly one successor w in the
// traversal (or recruitment) colours, w has exactly one predecessor (v), neither carries link annotations nor an
// orientation quirk.  What the loop does there is fully determined by a handful of facts about the stretch as a whole
// (has this walk been through it before, is the link store active, how far is maxLength).  The run index lays such
// stretches ("chains") out as consecutive positions, so that the walk kernel (walk.cpp) crosses one in a single step and
// the vertices are materialised afterwards by a bandwidth-bound expansion kernel.
//
//   uinfo[record]  u64   bits 0..31 position in unitig order | 32..46 distance to the start of its piece | 47..61 distance
//                        to the end of its piece | 62 orientation: the chain holds the record's k-mer reverse complemented |
//                        63 valid
//   uo[position]   u32   record | orientation << 31
//   ubase[position] u8   first base | last base << 2 of the k-mer in chain orientation (the base a contig gains there)
//
// A chain is a maximal path of MUTUAL unique links between oriented vertices (record, flip): next(a) = {b} and
// prev(b) = {a} under the engine's colour masks (TraversalEngine.getNextVertices/getPrevVertices :147-239, recruitment
// fallback included), both with records, neither link-flagged nor a quirk-Q6 record nor a palindrome.  Every chain has a
// mirror image (its reverse complement); the one whose head has the smaller oriented id is laid out, the other is the same
// positions read backwards with the flips inverted.  Pure cycles and chains that are their own mirror image are left as
// single vertices.  Chains are cut into pieces of at most LDBG_RUN_PIECE vertices so that the distances fit 15 bits.
#pragma once
#include "engine.h"

namespace ldbg {

#define LDBG_RUN_PIECE 32768u
#define LDBG_RUN_NONE 0xFFFFFFFFu

LDBG_HOSTDEV uint32_t ui_pos(uint64_t u) { return (uint32_t)u; }
LDBG_HOSTDEV uint32_t ui_dstart(uint64_t u) { return (uint32_t)(u >> 32) & 0x7FFFu; }
LDBG_HOSTDEV uint32_t ui_dend(uint64_t u) { return (uint32_t)(u >> 47) & 0x7FFFu; }
LDBG_HOSTDEV bool ui_orient(uint64_t u) { return (u >> 62) & 1ull; }
LDBG_HOSTDEV bool ui_valid(uint64_t u) { return (u >> 63) != 0ull; }
LDBG_HOSTDEV uint64_t ui_pack(uint32_t pos, uint32_t dstart, uint32_t dend, bool orient) {
    return (uint64_t)pos | ((uint64_t)(dstart & 0x7FFFu) << 32) | ((uint64_t)(dend & 0x7FFFu) << 47) | ((uint64_t)(orient ? 1 : 0) << 62) | (1ull << 63);
}

// built once per engine (the masks are the engine's), on the engine's device
class RunIndex {
public:
    RunIndex(const EngineView& e, int device, rt::stream_t s);
    ~RunIndex();
    RunIndexView view{};
    int64_t n_chains = 0;       // chains of two or more vertices laid out
    int64_t n_in_chains = 0;    // records in them
    double build_ms = 0;
    RunIndex(const RunIndex&) = delete;
    RunIndex& operator=(const RunIndex&) = delete;
private:
    void* d_uinfo_ = nullptr; void* d_uo_ = nullptr; void* d_ubase_ = nullptr;
};

}  // namespace ldbg
