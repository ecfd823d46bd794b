// Device-side pieces shared by the walk kernel (walk.cpp) and the general dfs kernel (dfs.cpp): the strand
// queue arguments, path-block storage, visited-table allocation and the wave-cooperative table regrowth.
#pragma once
#include "engine_host.h"
#include "image.h"

namespace ldbg {

// Path storage: every strand appends 8-byte vertex entries to 1024-entry blocks drawn from one pool
// (one atomic per 1024 traversed k-mers), so memory follows the actual walk lengths instead of
// maxLength per strand.
#define LDBG_PATH_BLOCK 1024

struct WalkArgs {
    EngineView e;
    const uint64_t* seeds;     // [n][W]
    const uint8_t* seed_valid; // [n] 0 = the seed string is not a k-mer over ACGT (findRecord misses, Q4)
    int64_t n_strands;         // 2n: strand 2i = reverse, 2i+1 = forward
    int64_t n_slots;
    int grow_at;               // a table is regrown when (entries + 8) * grow_at exceeds its size (2 = half full; 0 is read as 2)
    int lean_run;              // lean steps a lane may take in a row before the wavefront looks at the lanes waiting for a general step
    int64_t fetch_stride;      // strands are handed out in the order (i * fetch_stride) mod n_strands (coprime): neighbouring
                               // seeds walk the same contig, and a wavefront full of identical long walks is the worst tail
    int run_rev, run_fwd;
    unsigned long long* next_strand;
    uint64_t* pool;            // path blocks [n_blocks][LDBG_PATH_BLOCK]
    unsigned long long* next_block;
    uint64_t n_blocks;
    uint32_t* block_table;     // [n_strands][max_blocks]
    int max_blocks;
    uint64_t* vpool;           // zeroed visited-table pool (entries)
    unsigned long long* vnext;
    uint64_t vpool_entries;
    uint32_t vcap_max;         // largest table a strand may need
    uint32_t vcap_init;        // size a strand's table starts with
    uint32_t* strand_n;        // vertices in the strand's branch graph (0 = empty graph)
    uint32_t* strand_c;        // walk kernel: entries the strand wrote to its path blocks (runs are written as descriptors)
    const uint32_t* retry;     // walk kernel: the strands of this launch (a second launch that re-walks some strands k-mer by k-mer), or nullptr = all
    uint32_t* status;
    uint32_t* iters;
    uint8_t* quirk;            // [n_strands] the strand contains a Q6 vertex
    uint64_t* term;            // [n_strands][W]
    LsElem* ls;                // [n_slots][ecap]
    uint32_t ecap;
    struct LsSnap* snap;       // walk kernel: [n_slots][LDBG_SNAP_CAP] link-store snapshots of the repeat detection (runstep.h), or nullptr
    // ---- walks over the local image of a hash-sharded table (image.h); img_on == 0: the table is resident
    int img_on;
    ImageView img;
    const int32_t* seed_slot;  // [n] image slot of each seed's record (-1 = the seed has no record); the rows are in the image before the first round
    struct StrandSave* save;   // [n_slots] the strand a lane is working on, kept from one bulk-synchronous round to the next
    unsigned long long* unfinished;   // strands that were still in progress when the round's launch ended
    unsigned long long* kinds;      // walk kernel, or nullptr: [0] run steps [1] vertices they crossed [2] lean steps [3] general steps [4] link-store elements added
                                    // [5] junction choices [6] wavefront loop iterations [7] those with a general part [8] the busiest wavefront's (general << 32 | iterations)
    unsigned long long* wg_times;   // diagnostics (LDBG_WG_TIMES; timers exist in the -DLDBG_WALK_DIAG build only): [n_wg][2] start / end of every workgroup (100 MHz clock)
    unsigned long long* st_times;   // diagnostics: [n_strands][2] begin / finish of every strand
    unsigned long long* st_gen;     // diagnostics: [n_strands][2] ticks spent before general steps (prepare + cooperative phases), their number
    uint32_t yield_iters;           // over an image: loop iterations after which a wavefront ends its launch (0 = never): a bulk-synchronous round lasts as long as its
                                    // slowest wavefront, and a strand deep in rows that are already there would keep every strand that waits for rows waiting
    unsigned long long* wave_cat;   // diagnostics: [n_workgroups][8] per wavefront: loop iterations and 100 MHz ticks by kind (table regrowth, run steps, lean runs, general part)
#ifdef LDBG_LEAN_PROFILE
    unsigned long long* st_prof;
#endif
};
#define LDBG_VT_INITIAL 4096u
#ifndef LDBG_LS_FAST
#define LDBG_LS_FAST 16u          // link-store elements per lane kept in LDS: 24 KB per 64-lane workgroup, so that six workgroups
                                  // fit a CU and every strand of a 50,000-seed batch has a lane (profiles/r01_exp_ls_fast.log:
                                  // 8/12/16 elements 0.356 s per launch, 24 elements 0.41 s, 32 elements 0.44 s)
#endif

// ---- path descriptors (walk kernel).  A vertex entry (engine.h: path_pack) uses bits 0..60; an entry with bit 63 set is the head
// of a descriptor, followed by one payload word (a pair never straddles two path blocks: a PAD entry fills the gap).
//   RUN     head: len (bits 0..19) | |copyIndex| (20..35) | ascending positions (36) | flips inverted (37); payload: first position.
//           = len consecutive vertices of the run index (runs.h), all with the same copyIndex
//   REPEAT  head: count (bits 0..31); payload: first | period << 32.  The 2 * period vertices of this strand from vertex `first` on
//           are two revolutions of a walk that has been shown to repeat itself (runstep.h: periodic_check); `count` further
//           vertices follow: revolution after revolution the same records, each copyIndex moving on by the difference between
//           the two revolutions on record
#define LDBG_PD_TAG (1ull << 63)
#define LDBG_PD_KIND(e) ((unsigned)((e) >> 60) & 7u)
#define LDBG_PD_RUN 0u
#define LDBG_PD_REPEAT 1u
#define LDBG_PD_MARK 2u          // dfs logs (dfs.cpp): OPEN / CLOSE / KMER markers, kept as they are by the expansion
#define LDBG_PD_HALF 3u          // dfs logs: 32 bits of the k-mer of a vertex without a record (bits 0..31), kept as they are
#define LDBG_PD_PAD 7u
LDBG_HOSTDEV uint64_t pd_run_head(uint32_t len, uint32_t acopy, bool asc, bool inv) {
    return LDBG_PD_TAG | ((uint64_t)LDBG_PD_RUN << 60) | (uint64_t)(len & 0xFFFFFu) | ((uint64_t)(acopy & 0xFFFFu) << 20) | ((uint64_t)(asc ? 1 : 0) << 36) | ((uint64_t)(inv ? 1 : 0) << 37);
}
LDBG_HOSTDEV uint64_t pd_repeat_head(uint32_t count) { return LDBG_PD_TAG | ((uint64_t)LDBG_PD_REPEAT << 60) | (uint64_t)count; }
LDBG_HOSTDEV uint64_t pd_pad() { return LDBG_PD_TAG | ((uint64_t)LDBG_PD_PAD << 60); }
// vertices an entry stands for, given the entry before it (0 if there is none)
LDBG_HOSTDEV uint32_t pd_expanded(uint64_t prev, uint64_t e) {
    if ((prev & LDBG_PD_TAG) && LDBG_PD_KIND(prev) <= LDBG_PD_REPEAT) return 0u;     // e is a payload word
    if (!(e & LDBG_PD_TAG)) return 1u;
    const unsigned kd = LDBG_PD_KIND(e);
    return kd == LDBG_PD_RUN ? (uint32_t)(e & 0xFFFFFu) : (kd == LDBG_PD_REPEAT ? (uint32_t)e : ((kd == LDBG_PD_MARK || kd == LDBG_PD_HALF) ? 1u : 0u));
}
// is e (not a payload word) the head of a RUN or REPEAT descriptor?
LDBG_HOSTDEV bool pd_is_head(uint64_t e) { return (e & LDBG_PD_TAG) && LDBG_PD_KIND(e) <= LDBG_PD_REPEAT; }

LDBG_DEV uint64_t pack_vertex(const Node& v) { return path_pack(v.idx, v.flip != 0, v.base, v.copy, v.flip && !v.fj); }

struct PathWriter {
    uint64_t* cur;       // current block
    uint32_t n;          // entries written
    uint32_t nblk;       // blocks this strand owns (a truncated log keeps its blocks and writes them again)
};
LDBG_DEV bool path_append(const WalkArgs& a, int64_t s, PathWriter& pw, uint64_t entry) {
    const uint32_t off = pw.n & (LDBG_PATH_BLOCK - 1);
    if (off == 0) {
        const uint32_t bi = pw.n / LDBG_PATH_BLOCK;
        if (bi < pw.nblk) pw.cur = a.pool + (uint64_t)a.block_table[s * a.max_blocks + bi] * LDBG_PATH_BLOCK;
        else {
            if ((int)bi >= a.max_blocks) return false;
            const uint64_t b = (uint64_t)atomic_add_u64(a.next_block, 1ull);
            if (b >= a.n_blocks) return false;
            a.block_table[s * a.max_blocks + bi] = (uint32_t)b;
            pw.cur = a.pool + b * LDBG_PATH_BLOCK;
            pw.nblk = bi + 1;
        }
    }
    pw.cur[off] = entry;
    pw.n++;
    return true;
}
LDBG_DEV bool path_append_pair(const WalkArgs& a, int64_t s, PathWriter& pw, uint64_t head, uint64_t payload) {
    if ((pw.n & (LDBG_PATH_BLOCK - 1)) == LDBG_PATH_BLOCK - 1 && !path_append(a, s, pw, pd_pad())) return false;
    return path_append(a, s, pw, head) && path_append(a, s, pw, payload);
}
LDBG_DEV uint64_t path_read(const WalkArgs& a, int64_t s, uint32_t pos) {
    return a.pool[(uint64_t)a.block_table[s * a.max_blocks + pos / LDBG_PATH_BLOCK] * LDBG_PATH_BLOCK + (pos & (LDBG_PATH_BLOCK - 1))];
}
// drop everything from entry `n` on
LDBG_DEV void path_truncate(const WalkArgs& a, int64_t s, PathWriter& pw, uint32_t n) {
    pw.n = n;
    if (n & (LDBG_PATH_BLOCK - 1)) pw.cur = a.pool + (uint64_t)a.block_table[s * a.max_blocks + n / LDBG_PATH_BLOCK] * LDBG_PATH_BLOCK;
}

// carve a zeroed table of `cap` entries out of the pool
LDBG_DEV bool vt_alloc(const WalkArgs& a, VisitedTable& vt, uint32_t cap) {
    const uint64_t o = (uint64_t)atomic_add_u64(a.vnext, (unsigned long long)cap);
    if (o + cap > a.vpool_entries) {
#ifdef LDBG_HOSTSIM
        if (getenv("LDBG_DEBUG_STATUS")) fprintf(stderr, "[ldbg] vt_alloc fails: o %llu cap %u pool %llu\n", (unsigned long long)o, cap, (unsigned long long)a.vpool_entries);
#endif
        return false;
    }
    vt.tab = a.vpool + o;
    vt.mask = cap - 1;
    vt.used = 0;
    return true;
}

// one strand = private dfs(cv, goForward, 0, 0, {}, sinks) for ContigStopper (TraversalEngine.java:356-482),
// advanced one loop iteration per call so that the lanes of a wave stay busy with different strands
struct StrandState {
    int64_t s;
    Node cv;
    Cursor cu;
    PathWriter pw;
    VisitedTable vt;
    uint32_t gV, iters, status;
    bool fwd, branch_null, quirk;
};

LDBG_DEV void strand_finish(const WalkArgs& a, StrandState& st) {
    a.strand_n[st.s] = (st.branch_null || st.status != ST_OK) ? 0u : st.pw.n;
    a.status[st.s] = st.status != ST_OK ? st.status : (st.branch_null ? (uint32_t)ST_BRANCH_NULL : (uint32_t)ST_OK);
    a.iters[st.s] = st.iters;
    a.quirk[st.s] = st.quirk ? 1 : 0;
#ifdef LDBG_WALK_DIAG
    if (a.st_times) a.st_times[2 * st.s + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// node_find (engine.h) for a seed whose record is already known: its image slot, -1 = no record
template <int W>
LDBG_DEV void seed_node(const EngineView& e, const Kmer<W>& sk, int32_t slot, Node& n) {
    bool fc;
    const Kmer<W> c = kmer_canonical<W>(sk, e.g.k, &fc);
    n.idx = slot;
    n.copy = 0; n.vslot = 0; n.vent = 0; n.base = 0; n.e1 = 0; n.ent1 = 0;
    n.flip = fc ? 1 : 0;
    node_fill(e, n);
    if (n.idx < 0 && e.strict_flip && fc) n.fj = kmer_java_hash<W>(c, e.g.k) != kmer_java_hash<W>(sk, e.g.k) ? 1 : 0;
}

// returns false when the strand ended at once
template <int W>
LDBG_DEV bool strand_begin(const WalkArgs& a, StrandState& st, LinkStoreDev& ls, int64_t s) {
    const EngineView& e = a.e;
    st.s = s;
#ifdef LDBG_WALK_DIAG
    if (a.st_times) a.st_times[2 * s] = __builtin_amdgcn_s_memrealtime();
#endif
    st.fwd = (s & 1) != 0;
    st.status = ST_OK; st.iters = 0; st.gV = 0; st.branch_null = false; st.quirk = false;
    st.pw.cur = nullptr; st.pw.n = 0; st.pw.nblk = 0;
    st.cu.has = false; st.cu.status = ST_OK; st.cu.first = true; st.cu.epoch = 1;
    ls_clear(ls);
    if (!vt_alloc(a, st.vt, a.vcap_init < a.vcap_max ? a.vcap_init : a.vcap_max)) { st.status = ST_POOL_FULL; return false; }
    const uint64_t* sw = a.seeds + (s >> 1) * W;
    Kmer<W> sk;
#pragma unroll
    for (int i = 0; i < W; i++) sk.w[i] = sw[i];
    if (a.seed_valid[s >> 1]) {
        if (a.img_on) seed_node<W>(e, sk, a.seed_slot[s >> 1], st.cv);     // the routed findRecord of the seed was done before the first round
        else node_find<W>(e, sk, st.cv);
        node_locate(st.vt, st.cv);
    } else node_null(e, st.cv);   // not a k-mer: findRecord misses (Q4)
    if (st.cv.npe) { st.status = ST_NULLPTR; return false; }
    if (e.cursor_on) cursor_seek(e, st.cu, ls, st.vt, st.cv, st.fwd);   // :363-365
    return true;
}

// Regrowing a strand's visited table is a wave-cooperative operation: a lane that rehashed its own table
// alone would stall the other 63 lanes of its wavefront for as long as the table is big.  Every lane whose
// table is half full raises its hand (ballot); for each of them in turn the whole wavefront moves that
// lane's entries into a table 4x the size (CAS inserts), then the owner re-locates the vertices whose
// slots it carries.
LDBG_DEV void wave_grow_tables(const WalkArgs& a, StrandState& st, bool active) {
    const uint32_t cap = st.vt.mask + 1;
    // (room for what ONE iteration can claim: a run step's three entries, a lean run's four, a general step's children)
    const bool need = active && st.status == ST_OK && (st.vt.used + 16) * (uint32_t)(a.grow_at > 2 ? a.grow_at : 2) > cap && cap < a.vcap_max;
    unsigned long long ballot = wave_ballot(need);
    const int lane = wave_lane();
    while (ballot) {
        const int L = __builtin_ctzll(ballot);
        ballot &= ballot - 1;
        uint64_t new_tab = 0;
        uint32_t new_cap = 0;
        if (lane == L) {
            uint64_t c = (uint64_t)cap * 4;
            if (c > a.vcap_max) c = a.vcap_max;
            VisitedTable nt;
            if (vt_alloc(a, nt, (uint32_t)c)) { new_tab = (uint64_t)(uintptr_t)nt.tab; new_cap = (uint32_t)c; }
            else st.status = ST_POOL_FULL;
        }
        const uint64_t* old_tab = (const uint64_t*)(uintptr_t)wave_bcast_u64((uint64_t)(uintptr_t)st.vt.tab, L);
        const uint32_t old_mask = wave_bcast_u32(st.vt.mask, L);
        unsigned long long* nt_tab = (unsigned long long*)(uintptr_t)wave_bcast_u64(new_tab, L);
        const uint32_t nt_mask = wave_bcast_u32(new_cap, L) - 1;
        if (nt_tab) {
            for (uint32_t i = (uint32_t)lane; i <= old_mask; i += (uint32_t)wave_size()) {
                const uint64_t e = LDBG_GLOBAL(const uint64_t, old_tab)[i];
                if (e == 0) continue;
                uint32_t h = vt_hash(e & LDBG_VT_KEY_MASK) & nt_mask;
                while (atomic_cas_u64(&nt_tab[h], 0ull, (unsigned long long)e) != 0ull) h = (h + 1) & nt_mask;
            }
            wave_fence();
            if (lane == L) {
                st.vt.tab = (uint64_t*)nt_tab;
                st.vt.mask = nt_mask;       // `used` is unchanged: every entry moved
                const uint32_t used = st.vt.used;
                node_locate(st.vt, st.cv);
                if (a.e.cursor_on) { node_locate(st.vt, st.cu.cur); if (st.cu.has) node_locate(st.vt, st.cu.nxt); }
                st.vt.used = used;
            }
        }
    }
}

}  // namespace ldbg
