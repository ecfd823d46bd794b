// Walks and searches over a HASH-SHARDED table: the local image.
//
// Every rank owns one shard of the sorted table (graph.cpp::k_owner: a 64-bit mix of the canonical k-mer's minimizer, mod the number of
// ranks) with a GLOBAL neighbour index (shard.cpp: for each of a record's 8 possible neighbours the owner, the record number in
// the owner's shard and the orientation — a routed findRecord memoised once at load).  A walk lives on the rank that was given its
// seed; its visited set, link store, path and stopping rule never move.  What moves is ROWS: the rank keeps an IMAGE — a table of
// the rows it has been sent so far, laid out exactly like a shard's probe table, with the neighbour index rewritten to image slots
// (or to REMOTE where the neighbour has not been sent yet) — and the unchanged traversal kernels (walk.cpp, dfs.cpp: link-guided
// steps, junction choices, all stopping rules, quirks, even k) run on the image.  A strand about to read a row that is not there
// SUSPENDS: it files a request (the neighbour's global id) and keeps its state; one bulk-synchronous round = run every strand
// until it suspends or ends -> bucket the requests by owner -> all-to-all -> owners serve the rows -> all-to-all -> insert.
// Rows fetched for one strand serve every other strand of the rank (neighbouring seeds walk the same contigs).
#pragma once
#include "engine.h"

namespace ldbg {

// global record id: bits 0..39 record number in its shard + 1 (0 = no record), bits 40..47 owner, bit 63 orientation flag
LDBG_HOSTDEV uint64_t gid_make(int owner, int64_t lidx, bool flip) {
    return lidx < 0 ? 0ull : (((uint64_t)(lidx + 1)) | ((uint64_t)(uint32_t)owner << 40) | (flip ? (1ull << 63) : 0ull));
}
LDBG_HOSTDEV uint64_t gid_key(uint64_t gid) { return gid & 0xFFFFFFFFFFFFull; }      // owner + record: never 0 for a record
LDBG_HOSTDEV int gid_owner(uint64_t gid) { return (int)((gid >> 40) & 0xFFu); }
LDBG_HOSTDEV int64_t gid_lidx(uint64_t gid) { return (int64_t)(gid & 0xFFFFFFFFFFull) - 1; }

// neighbour-index entry of an image row whose neighbour has not been sent yet (bit 31 still carries the orientation).  A shard or
// a whole table never holds this record number: the loader refuses tables of 2^31 - 2 records and more (graph.cpp)
#define LDBG_NBR_REMOTE 0x7FFFFFFFu
LDBG_HOSTDEV bool nbr_remote(uint32_t ent) { return (ent & 0x7FFFFFFFu) == LDBG_NBR_REMOTE; }

struct ImageView {
    uint8_t* probe;                 // [cap][stride] rows (the engine's GraphView points here too)
    int stride, nbr_off, flags_off;
    uint64_t* nbrg;                 // [cap][8] global ids of the neighbours
    uint64_t* gkey;                 // [cap] the row's own global id key
    uint64_t* rec_of;               // [cap] junction records of the row in the (replicated) merged link table, ~0 = none
    unsigned long long* hkeys;      // open-addressing map global id key -> slot + 1
    uint32_t* hvals;
    uint32_t hmask;
    uint32_t cap;
    unsigned long long* n_rows;     // rows in the image
    unsigned long long* req;        // this round's requests (global id keys; duplicates allowed)
    unsigned long long* n_req;
    uint32_t req_cap;
    unsigned long long* req_seen;   // [LDBG_REQ_SEEN] keys filed this round, direct mapped: the same row is usually wanted by several strands at once
};

LDBG_HOSTDEV uint32_t img_hash(uint64_t key) { uint64_t x = key * 0x9E3779B97F4A7C15ull; return (uint32_t)(x >> 29); }
LDBG_DEV int64_t img_lookup(const ImageView& im, uint64_t key) {
    uint32_t h = img_hash(key) & im.hmask;
    for (uint32_t probes = 0; probes <= im.hmask; probes++) {      // (bounded: a map that an overflowing round has filled has no free slot to stop at)
        const unsigned long long k = LDBG_GLOBAL(const unsigned long long, im.hkeys)[h];
        if (k == 0ull) return -1;
        if (k == key) {
            const uint32_t v = LDBG_GLOBAL(const uint32_t, im.hvals)[h];
            return v ? (int64_t)v - 1 : -1;             // (a slot being filled by another thread: not there yet)
        }
        h = (h + 1) & im.hmask;
    }
    return -1;
}
#define LDBG_REQ_SEEN 65536u
LDBG_DEV void img_request(const ImageView& im, uint64_t key) {
    // (a filter, not a set: a key pushed out by another is filed again, and a request that found no room this round is filed again next round)
    if (atomic_exch_u64(&im.req_seen[img_hash(key) & (LDBG_REQ_SEEN - 1u)], (unsigned long long)key) == (unsigned long long)key) return;
    const unsigned long long at = atomic_add_u64(im.n_req, 1ull);
    if (at < im.req_cap) im.req[at] = key;             // (what does not fit is asked for again next round: the strand stays suspended)
}

// Are the rows that a step from vertex v may read in the image?  Those are v's neighbours in the direction of travel (the loop at
// TraversalEngine.java:373-481 materialises no other vertex).  Entries whose row has arrived since the row of v was written are
// patched on the way; what is still missing is requested.
LDBG_DEV bool rows_ready(const ImageView& im, Node& v, bool fwd) {
    if (v.idx < 0) return true;
    uint32_t* nb = (uint32_t*)(im.probe + (size_t)v.idx * (size_t)im.stride + im.nbr_off);
    const unsigned j0 = (fwd != (v.fj != 0)) ? 0u : 4u;       // engine.h: nbr_slot
    bool ok = true, patched = false;
    for (unsigned q = 0; q < 4; q++) {
        const uint32_t ent = LDBG_GLOBAL(const uint32_t, nb)[j0 + q];
        if (!nbr_remote(ent)) continue;
        const uint64_t key = gid_key(LDBG_GLOBAL(const uint64_t, im.nbrg)[(size_t)v.idx * 8 + j0 + q]);
        const int64_t slot = img_lookup(im, key);
        if (slot >= 0) { LDBG_GLOBAL(uint32_t, nb)[j0 + q] = (uint32_t)(slot + 1) | (ent & 0x80000000u); patched = true; }
        else { ok = false; img_request(im, key); }
    }
    if (v.e1 && (patched || nbr_remote(v.ent1))) {       // (the row may have been patched by another strand since this vertex was read)
        const uint32_t m = fwd ? v.next_mask : v.prev_mask;
        v.ent1 = LDBG_GLOBAL(const uint32_t, nb)[nbr_slot(v.fj != 0, fwd, lowbit4(m))];
    }
    return ok;
}

}  // namespace ldbg
