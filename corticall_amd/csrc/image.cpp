// The local image of a hash-sharded table (image.h): serving rows of the own shard, inserting the rows that arrive, bucketing a
// round's requests by owner.  The exchanges themselves are driven from corticall_amd/distributed.py (torch.distributed: RCCL
// all-to-all over xGMI on the device, gloo in the CPU tests).
#include "image.h"

#include <algorithm>

#include "engine_host.h"
#include "shard.h"

namespace ldbg {

namespace {

int grid_of(int64_t n, int block = 256, int max_blocks = 4096) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((n + block - 1) / block, max_blocks));
}

// ---- owner side: a "fat row" = own global id key | 8 global neighbour ids | the probe row
// The row asked for and, in the `depth - 1` slots after it, rows the asker is likely to want next: from the record outwards in both
// directions along edges (of any colour) for as long as the neighbour is unique and lives on this shard too — with ownership by
// minimizer (graph.cpp) that is the usual case.  Where the way forks, the local neighbours of the fork are sent and that direction
// ends.  What is sent beyond the first row is a prefetch: results never depend on it (an image row is the owner's row, whoever asked).
// Two kernels: one thread per request picks the records (plan[n][depth], -1 = unused slot), then one thread per word copies the rows.
LDBG_KERNEL void k_serve_plan(GraphView g, const uint64_t* nbrg, int my_rank, const unsigned long long* keys, int64_t n, int depth, int64_t* plan) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        int64_t* pl = plan + i * depth;
        for (int d = 0; d < depth; d++) pl[d] = -1;
        const uint64_t key = keys[i];
        const int64_t r0 = gid_lidx(key);
        if (!(gid_owner(key) == my_rank && r0 >= 0 && r0 < g.N)) continue;                    // (a request that does not belong here)
        pl[0] = r0;
        int used = 1;
        for (int dir = 0; dir < 2 && used < depth; dir++) {
            const int budget = used + (depth - used) / (2 - dir);                              // half of what is left per direction
            int64_t r = r0;
            bool rc = dir == 1;                                                                // dir 1: backwards = forwards along the reverse complement
            while (used < budget) {
                const uint64_t* nb = nbrg + r * 8 + (rc ? 4 : 0);
                int cnt = 0, last = -1;
                for (int b = 0; b < 4; b++) if (nb[b] != 0ull) { cnt++; last = b; }
                if (cnt == 0) break;
                if (cnt > 1) {                                                                 // a fork: its local arms, then stop
                    for (int b = 0; b < 4 && used < budget; b++)
                        if (nb[b] != 0ull && gid_owner(nb[b]) == my_rank) pl[used++] = gid_lidx(nb[b]);
                    break;
                }
                const uint64_t nx = nb[last];
                if (gid_owner(nx) != my_rank) break;
                r = gid_lidx(nx);
                if (r == r0) break;                                                            // round a cycle
                rc = rc != ((nx >> 63) != 0ull);
                pl[used++] = r;
            }
        }
    }
}
// a "fat row" = own global id key | 8 global neighbour ids | the probe row
LDBG_KERNEL void k_serve_rows(GraphView g, const uint64_t* nbrg, int my_rank, const int64_t* plan, int64_t n_slots, int rowb, uint8_t* out) {
    const int words = 9 + g.stride / 8;
    for (int64_t t = global_tid(); t < n_slots * words; t += global_nthreads()) {
        const int64_t i = t / words;
        const int w = (int)(t % words);
        const int64_t r = plan[i];
        uint64_t v = 0;
        if (r >= 0) {
            if (w == 0) v = gid_key(gid_make(my_rank, r, false));
            else if (w < 9) v = nbrg[r * 8 + (w - 1)];
            else v = ((const uint64_t*)graph_row(g, r))[w - 9];
        } else if (w != 0) continue;                                                           // (an unused slot: only its key, 0, is written)
        ((uint64_t*)(out + (size_t)i * rowb))[w] = v;
    }
}

// ---- requester side: insert the rows that arrived
template <int W>
LDBG_KERNEL void k_img_insert(ImageView im, LinksView links, int k, uint32_t link_flag_mask, const uint8_t* rows, int rowb, int64_t n, unsigned* overflow) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        const uint64_t* in = (const uint64_t*)(rows + (size_t)i * rowb);
        const uint64_t key = in[0];
        if (key == 0ull) continue;
        // A full image takes no more keys: the flag ends the rounds on every rank (k_round_stats) and the host enlarges the image.  (Keys
        // claimed without a slot would fill the hash map, and a probe of a map without a free slot would never end.)
        if (LDBG_GLOBAL(const unsigned long long, im.n_rows)[0] >= (unsigned long long)im.cap) { *overflow = 1u; continue; }
        // claim the key (several strands may have asked for the same row in one round: the first copy is kept)
        uint32_t h = img_hash(key) & im.hmask;
        bool claimed = false;
        for (uint32_t probes = 0; probes <= im.hmask; probes++) {          // (bounded: threads that passed the test above together may have filled the map)
            const unsigned long long prev = atomic_cas_u64(&im.hkeys[h], 0ull, (unsigned long long)key);
            if (prev == 0ull) { claimed = true; break; }
            if (prev == key) break;
            h = (h + 1) & im.hmask;
        }
        if (!claimed) { if (LDBG_GLOBAL(const unsigned long long, im.hkeys)[h] != key) *overflow = 1u; continue; }
        const unsigned long long slot = atomic_add_u64(im.n_rows, 1ull);
        if (slot >= im.cap) { *overflow = 1u; continue; }                // (the key stays claimed without a slot: lookups miss until the host has enlarged the image)
        uint64_t* dst = (uint64_t*)(im.probe + (size_t)slot * im.stride);
        for (int w = 0; w < im.stride / 8; w++) dst[w] = in[9 + w];
        uint32_t* nb = (uint32_t*)((uint8_t*)dst + im.nbr_off);
        for (int j = 0; j < 8; j++) {
            const uint64_t g = in[1 + j];
            im.nbrg[slot * 8 + j] = g;
            uint32_t ent = 0;
            if (gid_key(g) != 0ull) {
                const int64_t s = img_lookup(im, gid_key(g));
                ent = (s >= 0 ? (uint32_t)(s + 1) : LDBG_NBR_REMOTE) | ((g >> 63) ? 0x80000000u : 0u);
            }
            nb[j] = ent;
        }
        im.gkey[slot] = key;
        uint64_t ro = ~0ull;
        if (links.M > 0 && (((const uint8_t*)dst)[im.flags_off] & link_flag_mask)) {
            Kmer<W> c;
            for (int w = 0; w < W; w++) c.w[w] = dst[w];
            const int64_t m = links_find<W>(links, k, c);
            if (m >= 0) ro = (uint64_t)links.off[m] | ((uint64_t)(links.off[m + 1] - links.off[m]) << 32);
        }
        im.rec_of[slot] = ro;
        device_fence();
        im.hvals[h] = (uint32_t)slot + 1u;                               // published last: a reader that sees the slot sees the row
    }
}
// After the rows of a round are in: every arrival links itself to the neighbours that are in the image by now (rows of the same round
// were not published yet when k_img_insert looked) and its neighbours' entries to itself, so that a strand finds a chain of rows that
// came together fully linked and takes its lean steps through it without stopping at every hop to patch an entry (image.h: rows_ready)
LDBG_KERNEL void k_img_link(ImageView im, const uint8_t* rows, int rowb, int64_t n) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) {
        const uint64_t key = ((const uint64_t*)(rows + (size_t)i * rowb))[0];
        if (key == 0ull) continue;
        const int64_t slot = img_lookup(im, key);
        if (slot < 0) continue;
        uint32_t* nb = (uint32_t*)(im.probe + (size_t)slot * (size_t)im.stride + im.nbr_off);
        for (int j = 0; j < 8; j++) {
            const uint64_t g = im.nbrg[(size_t)slot * 8 + j];
            if (gid_key(g) == 0ull) continue;
            const int64_t s = img_lookup(im, gid_key(g));
            if (s < 0) continue;
            nb[j] = (uint32_t)(s + 1) | ((g >> 63) ? 0x80000000u : 0u);
            uint32_t* nbs = (uint32_t*)(im.probe + (size_t)s * (size_t)im.stride + im.nbr_off);
            for (int q = 0; q < 8; q++) {
                const uint64_t back = im.nbrg[(size_t)s * 8 + q];
                if (gid_key(back) == key) nbs[q] = (uint32_t)(slot + 1) | ((back >> 63) ? 0x80000000u : 0u);
            }
        }
    }
}
LDBG_KERNEL void k_img_lookup(ImageView im, const unsigned long long* keys, int64_t n, int32_t* slots) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads())
        slots[i] = keys[i] ? (int32_t)img_lookup(im, gid_key(keys[i])) : -1;
}
// ---- requests of a round -> per-owner send blocks [world][cap] of global id keys (0 = unused).  The lanes of a wavefront that
// ask the same owner share ONE atomic on that owner's counter (ballot + prefix count).
LDBG_KERNEL void k_bucket_requests(const unsigned long long* req, const unsigned long long* n_req, uint32_t req_cap, int world, uint32_t cap,
                                   unsigned long long* counts, unsigned long long* send) {
    const unsigned long long n = std::min<unsigned long long>(*n_req, (unsigned long long)req_cap);
    const int64_t total = (int64_t)((n + (unsigned)wave_size() - 1) / (unsigned)wave_size()) * wave_size();
    for (int64_t i = global_tid(); i < total; i += global_nthreads()) {
        const bool have = (unsigned long long)i < n;
        const unsigned long long key = have ? req[i] : 0ull;
        const int owner = have ? gid_owner(key) : -1;
        unsigned long long todo = wave_ballot(have && owner >= 0 && owner < world);
        while (todo) {
            const int L = __builtin_ctzll(todo);
            const int ow = (int)wave_bcast_u32((uint32_t)owner, L);
            const unsigned long long same = wave_ballot(have && owner == ow);
            unsigned long long base = 0;
            if (wave_lane() == L) base = atomic_add_u64(&counts[ow], (unsigned long long)__builtin_popcountll(same));
            base = wave_bcast_u64(base, L);
            if (have && owner == ow) {
                const unsigned long long at = base + (unsigned long long)wave_count_below(same);
                if (at < cap) send[(size_t)ow * cap + at] = key;
            }
            todo &= ~same;
        }
    }
}

LDBG_KERNEL void k_img_request(ImageView im, const unsigned long long* keys, int64_t n) {
    for (int64_t i = global_tid(); i < n; i += global_nthreads()) if (keys[i]) img_request(im, gid_key(keys[i]));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- host
ShardImage::ShardImage(const Graph& shard, int64_t cap, int64_t global_records) : shard_(shard) {
    rt::set_device(shard.device);
    CtxHeader h = shard.hdr;
    h.num_records = cap;
    graph_.reset(new Graph(h, cap, shard.device, shard.view, global_records <= 2));
    cap_ = cap;
    uint64_t hc = 64;
    while (hc < 2ull * (uint64_t)cap) hc <<= 1;
    hcap_ = hc;
    d_nbrg_ = rt::dmalloc((size_t)cap * 64);
    d_gkey_ = rt::dmalloc((size_t)cap * 8);
    d_hkeys_ = rt::dmalloc((size_t)hc * 8);
    d_hvals_ = rt::dmalloc((size_t)hc * 4);
    d_ctr_ = rt::dmalloc(64);
    req_cap_ = (uint32_t)std::min<int64_t>(std::max<int64_t>(1024, cap), 1 << 22);
    d_req_ = rt::dmalloc((size_t)req_cap_ * 8);
    d_req_seen_ = rt::dmalloc((size_t)LDBG_REQ_SEEN * 8);
    rt::dmemset(d_req_seen_, 0, (size_t)LDBG_REQ_SEEN * 8, graph_->stream);
    d_bcount_ = rt::dmalloc(256 * 8);
    clear();
}
ShardImage::~ShardImage() { rt::dfree(d_nbrg_); rt::dfree(d_gkey_); rt::dfree(d_hkeys_); rt::dfree(d_hvals_); rt::dfree(d_ctr_); rt::dfree(d_req_); rt::dfree(d_bcount_); rt::dfree(d_rec_of_own_); rt::dfree(d_plan_); rt::dfree(d_req_seen_); }

void ShardImage::clear() {
    rt::stream_t s = graph_->stream;
    rt::dmemset(d_hkeys_, 0, (size_t)hcap_ * 8, s);
    rt::dmemset(d_hvals_, 0, (size_t)hcap_ * 4, s);
    rt::dmemset(d_ctr_, 0, 64, s);
    rt::stream_sync(s);
}

ImageView ShardImage::view(uint64_t* rec_of) const {
    ImageView im;
    im.probe = graph_->probe_mutable(); im.stride = graph_->view.stride; im.nbr_off = graph_->view.nbr_off; im.flags_off = graph_->view.flags_off;
    im.nbrg = (uint64_t*)d_nbrg_; im.gkey = (uint64_t*)d_gkey_; im.rec_of = rec_of;
    im.hkeys = (unsigned long long*)d_hkeys_; im.hvals = (uint32_t*)d_hvals_; im.hmask = (uint32_t)(hcap_ - 1); im.cap = (uint32_t)cap_;
    unsigned long long* c = (unsigned long long*)d_ctr_;
    im.n_rows = c; im.n_req = c + 1; im.req = (unsigned long long*)d_req_; im.req_cap = req_cap_; im.req_seen = (unsigned long long*)d_req_seen_;
    return im;
}
int ShardImage::row_bytes() const { return 72 + shard_.view.stride; }

void ShardImage::serve(int my_rank, const unsigned long long* d_keys, int64_t n, int depth, uint8_t* d_out, rt::stream_t s) const {
    if (n <= 0) return;
    if (!shard_.d_nbrg) throw StatusError(LDBG_ERR_ARG, "image: the global neighbour index of this shard has not been built");
    if (depth < 1) throw StatusError(LDBG_ERR_ARG, "image: serve depth < 1");
    const size_t need = (size_t)n * (size_t)depth * 8;
    if (need > plan_bytes_) { rt::dfree(d_plan_); d_plan_ = rt::dmalloc(need); plan_bytes_ = need; }
    LDBG_LAUNCH(k_serve_plan, grid_of(n), 256, s, shard_.view, (const uint64_t*)shard_.d_nbrg, my_rank, d_keys, n, depth, (int64_t*)d_plan_);
    const int words = 9 + shard_.view.stride / 8;
    LDBG_LAUNCH(k_serve_rows, grid_of(n * depth * words), 256, s, shard_.view, (const uint64_t*)shard_.d_nbrg, my_rank, (const int64_t*)d_plan_, n * depth, row_bytes(), d_out);
}

void ShardImage::insert(const Engine* e, const uint8_t* d_rows, int64_t n, rt::stream_t s) {
    if (n <= 0) return;
    LinksView lv{};
    uint32_t mask = 0;
    uint64_t* rec_of = nullptr;
    if (e) { lv = e->view.links; mask = e->view.link_flag_mask; rec_of = (uint64_t*)e->view.links.rec_of; }
    if (!rec_of) {
        if (!d_rec_of_own_) d_rec_of_own_ = rt::dmalloc((size_t)cap_ * 8);
        rec_of = (uint64_t*)d_rec_of_own_;
        lv.M = 0;
    }
    ImageView im = view(rec_of);
    unsigned* ovf = (unsigned*)((unsigned long long*)d_ctr_ + 2);
    const int k = shard_.hdr.k, rb = row_bytes();
    switch (shard_.hdr.W) {
        case 1: LDBG_LAUNCH(k_img_insert<1>, grid_of(n), 256, s, im, lv, k, mask, d_rows, rb, n, ovf); break;
        case 2: LDBG_LAUNCH(k_img_insert<2>, grid_of(n), 256, s, im, lv, k, mask, d_rows, rb, n, ovf); break;
        case 3: LDBG_LAUNCH(k_img_insert<3>, grid_of(n), 256, s, im, lv, k, mask, d_rows, rb, n, ovf); break;
        default: LDBG_LAUNCH(k_img_insert<4>, grid_of(n), 256, s, im, lv, k, mask, d_rows, rb, n, ovf); break;
    }
    LDBG_LAUNCH(k_img_link, grid_of(n), 256, s, im, d_rows, rb, n);
}
void ShardImage::lookup(const unsigned long long* d_keys, int64_t n, int32_t* d_slots, rt::stream_t s) const {
    if (n <= 0) return;
    LDBG_LAUNCH(k_img_lookup, grid_of(n), 256, s, view(nullptr), d_keys, n, d_slots);
}
void ShardImage::bucket(int world, uint32_t cap_per_owner, unsigned long long* d_send, rt::stream_t s) const {
    if (world > 256) throw StatusError(LDBG_ERR_ARG, "image: more than 256 ranks");
    rt::dmemset(d_send, 0, (size_t)world * cap_per_owner * 8, s);
    rt::dmemset(d_bcount_, 0, 256 * 8, s);
    const unsigned long long* c = (const unsigned long long*)d_ctr_;
    LDBG_LAUNCH(k_bucket_requests, 64, 64, s, (const unsigned long long*)d_req_, c + 1, req_cap_, world, cap_per_owner, (unsigned long long*)d_bcount_, d_send);
}
// explicit requests (the seeds of a batch, the sinks of a search): they join the round's request list
void ShardImage::request(const unsigned long long* d_keys, int64_t n, rt::stream_t s) {
    if (n <= 0) return;
    LDBG_LAUNCH(k_img_request, grid_of(n), 256, s, view(nullptr), d_keys, n);
}
void ShardImage::reset_requests(rt::stream_t s) {
    rt::dmemset((unsigned long long*)d_ctr_ + 1, 0, 8, s);
    rt::dmemset(d_req_seen_, 0, (size_t)LDBG_REQ_SEEN * 8, s);
}
void ShardImage::counters(int64_t* n_rows, int64_t* n_req, int* overflow) const {
    unsigned long long c[4];
    rt::d2h(c, d_ctr_, 32, graph_->stream);
    rt::stream_sync(graph_->stream);
    if (n_rows) *n_rows = (int64_t)c[0];
    if (n_req) *n_req = (int64_t)c[1];
    if (overflow) *overflow = (int)(unsigned)c[2];
}

}  // namespace ldbg
