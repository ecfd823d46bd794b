// Stateful cursor API: TraversalEngine.seek / next / previous / hasNext / hasPrevious
// (J/utils/traversal/TraversalEngine.java:241-339).  The state lives in HBM; each call is a
// one-thread kernel that reuses the device primitives of the batched walk (engine.h), so the cursor
// and the walk cannot drift apart.
#include "cursor.h"

namespace ldbg {

template <int W>
struct CursorStateDev {
    Node cur;
    Node nxt, prv;
    uint32_t has_next, has_prev, first, go_forward, status;
    uint32_t ls_n, ls_java_cap, ls_nkeys, ls_next_seq, ls_age, ls_n_new;
    uint32_t vt_used, nxt_words_valid;
    uint64_t cur_words[W];     // k-mer of the cursor vertex (vertices without a record have no row to read it from)
    // outputs of the last step
    uint64_t out_words[W];
    int64_t out_rec;
};

template <int W>
LDBG_DEV void cs_reseek(const EngineView& e, CursorStateDev<W>& st, uint64_t* vtab, uint32_t vcap) {
    // seek(sk) :321-335 — unique neighbours, fresh LinkStore, fresh `seen`
    if (st.vt_used != 0) for (uint32_t i = 0; i < vcap; i++) vtab[i] = 0;    // seen = new HashSet<>()
    VisitedTable vt;
    vt.tab = vtab; vt.mask = vcap - 1; vt.used = 0;
    node_locate(vt, st.cur);
    st.has_next = popc4(st.cur.next_mask) == 1;
    if (st.has_next) { node_child_located(e, vt, st.cur, true, lowbit4(st.cur.next_mask), st.nxt); st.nxt_words_valid = 0; }
    st.has_prev = popc4(st.cur.prev_mask) == 1;
    if (st.has_prev) { node_child_located(e, vt, st.cur, false, lowbit4(st.cur.prev_mask), st.prv); }
    st.ls_n = st.ls_java_cap = st.ls_nkeys = st.ls_next_seq = st.ls_age = st.ls_n_new = 0;
    st.first = 1;
    st.vt_used = vt.used;
    st.status = st.cur.npe ? (uint32_t)ST_NULLPTR : (uint32_t)ST_OK;
}

template <int W>
LDBG_KERNEL void k_cursor_seek(EngineView e, CursorStateDev<W>* stp, const uint64_t* words, int is_kmer, uint64_t* vtab, uint32_t vcap) {
    if (global_tid() != 0) return;
    CursorStateDev<W>& st = *stp;
    Kmer<W> sk;
    for (int i = 0; i < W; i++) sk.w[i] = words[i];
    if (is_kmer) node_find<W>(e, sk, st.cur);
    else node_null(e, st.cur);
    for (int i = 0; i < W; i++) st.cur_words[i] = sk.w[i];
    cs_reseek<W>(e, st, vtab, vcap);
}

// one next() / previous() on the device-resident state
template <int W>
LDBG_DEV void cs_step(const EngineView& e, CursorStateDev<W>& st, bool fwd, uint64_t* vtab, uint32_t vcap, LsElem* els, uint32_t ecap) {
    st.status = ST_OK;
    if (st.first || (st.go_forward != 0) != fwd) {      // :243-248 / :283-288
        st.go_forward = fwd ? 1 : 0;
        cs_reseek<W>(e, st, vtab, vcap);
        if (st.status != ST_OK) return;
    }
    VisitedTable vt;
    vt.tab = vtab; vt.mask = vcap - 1; vt.used = st.vt_used;
    if (vt.used * 2 > vcap) { st.status = ST_POOL_FULL; return; }   // more steps since seek() than the cursor's `seen` table holds
    // the nodes were saved by an earlier launch: bring their cached table entries up to date
    node_locate(vt, st.cur);
    if (st.has_next) node_locate(vt, st.nxt);
    if (st.has_prev) node_locate(vt, st.prv);
    vt.used = st.vt_used;
    LinkStoreDev ls;
    ls.fast = nullptr; ls.fast_cap = 0; ls.fast_stride = 0;
    ls.el = els; ls.cap = ecap; ls.n = st.ls_n; ls.java_cap = st.ls_java_cap; ls.nkeys = st.ls_nkeys; ls.next_seq = st.ls_next_seq;
    ls.age = st.ls_age; ls.n_new = st.ls_n_new;
    ls.overflow = false;
    Cursor cu;
    cu.cur = st.cur; cu.first = st.first != 0; cu.status = ST_OK; cu.epoch = 1;
    cu.has = fwd ? st.has_next != 0 : st.has_prev != 0;
    if (!cu.has) { st.status = ST_NULLPTR; return; }   // target vanished after the re-seek: NPE in the reference
    cu.nxt = fwd ? st.nxt : st.prv;
    Node old = st.cur;
    // k-mer of the vertex stepped onto: from its row, or (no record) from the cursor k-mer and the edge base
    Kmer<W> tk;
    if (cu.nxt.idx >= 0) tk = node_kmer<W>(e, cu.nxt);
    else if (old.idx >= 0) tk = child_kmer<W>(e, old, fwd, cu.nxt.base);
    else { for (int i = 0; i < W; i++) tk.w[i] = st.cur_words[i]; }
    Node t = cursor_step<W>(e, cu, ls, vt, fwd);
    st.first = 0;
    st.cur = cu.cur;
    if (fwd) { st.prv = old; st.has_prev = 1; st.nxt = cu.nxt; st.has_next = cu.has ? 1 : 0; }
    else { st.nxt = old; st.has_next = 1; st.prv = cu.nxt; st.has_prev = cu.has ? 1 : 0; }
    st.ls_n = ls.n; st.ls_java_cap = ls.java_cap; st.ls_nkeys = ls.nkeys; st.ls_next_seq = ls.next_seq;
    st.ls_age = ls.age; st.ls_n_new = ls.n_new;
    st.status = cu.status;
    st.vt_used = vt.used;
    for (int i = 0; i < W; i++) { st.out_words[i] = tk.w[i]; st.cur_words[i] = tk.w[i]; }
    st.out_rec = t.idx;
}
template <int W>
LDBG_KERNEL void k_cursor_step(EngineView e, CursorStateDev<W>* stp, int fwd_i, uint64_t* vtab, uint32_t vcap, LsElem* els, uint32_t ecap) {
    if (global_tid() != 0) return;
    cs_step<W>(e, *stp, fwd_i != 0, vtab, vcap, els, ecap);
}
// assemble(seed, goForward) (TraversalEngine.java:125-145) after a seek(seed): next() / previous() for as long as the cursor has one and
// fewer than max_len vertices have been collected, all in ONE launch; out_n[0] = vertices, out_n[1] = status of the last step
template <int W>
LDBG_KERNEL void k_cursor_assemble(EngineView e, CursorStateDev<W>* stp, int fwd_i, uint64_t* vtab, uint32_t vcap, LsElem* els, uint32_t ecap, int64_t max_len,
                                   uint64_t* out_words, int64_t* out_rec, int64_t* out_n) {
    if (global_tid() != 0) return;
    CursorStateDev<W>& st = *stp;
    const bool fwd = fwd_i != 0;
    int64_t n = 0;
    uint32_t status = st.status;
    while (status == ST_OK && (fwd ? st.has_next : st.has_prev) && n < max_len) {
        cs_step<W>(e, st, fwd, vtab, vcap, els, ecap);
        status = st.status;
        if (status != ST_OK) break;
        for (int i = 0; i < W; i++) out_words[n * W + i] = st.out_words[i];
        out_rec[n] = st.out_rec;
        n++;
    }
    out_n[0] = n; out_n[1] = (int64_t)status;
}

// ------------------------------------------------------------------ host
struct CursorHost::Impl {
    void* d_state = nullptr;
    void* d_vtab = nullptr;
    void* d_ls = nullptr;
    void* d_words = nullptr;
    uint32_t vcap = 1u << 17, ecap = 256;
    size_t state_bytes = 0;
    bool sought = false;
};

template <int W>
static size_t state_size() { return sizeof(CursorStateDev<W>); }

CursorHost::CursorHost(Engine& e) : eng_(e), impl_(new Impl) {
    e.enter();
    const int W = e.graph->hdr.W;
    impl_->state_bytes = W == 1 ? state_size<1>() : W == 2 ? state_size<2>() : W == 3 ? state_size<3>() : state_size<4>();
    impl_->d_state = rt::dmalloc(impl_->state_bytes);
    impl_->d_vtab = rt::dmalloc((size_t)impl_->vcap * 8);
    impl_->d_ls = rt::dmalloc((size_t)impl_->ecap * sizeof(LsElem));
    impl_->d_words = rt::dmalloc((size_t)W * 8);
    rt::dmemset(impl_->d_state, 0, impl_->state_bytes, e.estream());
    rt::dmemset(impl_->d_vtab, 0, (size_t)impl_->vcap * 8, e.estream());
    rt::stream_sync(e.estream());
}
CursorHost::~CursorHost() {
    rt::dfree(impl_->d_state); rt::dfree(impl_->d_vtab); rt::dfree(impl_->d_ls); rt::dfree(impl_->d_words);
    delete impl_;
}

template <int W>
static void read_state(void* d, CursorStateDev<W>& h, rt::stream_t s) {
    rt::d2h(&h, d, sizeof(h), s);
    rt::stream_sync(s);
}

void CursorHost::check_status(uint32_t st) {
    if (st == ST_NULLPTR) throw StatusError(LDBG_ERR_NULLPOINTER, "cursor dereferenced a missing record / vanished target (NullPointerException in the reference)");
    if (st == ST_LINKSTORE_FULL) throw StatusError(LDBG_ERR_CAPACITY, "cursor link store capacity exceeded");
    if (st == ST_POOL_FULL) throw StatusError(LDBG_ERR_CAPACITY, "cursor walked more than 65536 steps since the last seek()");
}

void CursorHost::seek(const char* kmer) {
    eng_.enter();
    const int W = eng_.graph->hdr.W, k = eng_.graph->hdr.k;
    rt::stream_t s = eng_.estream();
    std::vector<uint64_t> w(W);
    const int is_kmer = ascii_to_words(kmer, k, w.data(), W) ? 1 : 0;       // validity beside the words: at k = 32, 64, ... no bit pattern is free (Q4)
    if (!is_kmer) std::fill(w.begin(), w.end(), 0ull);
    rt::h2d(impl_->d_words, w.data(), (size_t)W * 8, s);
    switch (W) {
        case 1: LDBG_LAUNCH(k_cursor_seek<1>, 1, 64, s, eng_.view, (CursorStateDev<1>*)impl_->d_state, (const uint64_t*)impl_->d_words, is_kmer, (uint64_t*)impl_->d_vtab, impl_->vcap); break;
        case 2: LDBG_LAUNCH(k_cursor_seek<2>, 1, 64, s, eng_.view, (CursorStateDev<2>*)impl_->d_state, (const uint64_t*)impl_->d_words, is_kmer, (uint64_t*)impl_->d_vtab, impl_->vcap); break;
        case 3: LDBG_LAUNCH(k_cursor_seek<3>, 1, 64, s, eng_.view, (CursorStateDev<3>*)impl_->d_state, (const uint64_t*)impl_->d_words, is_kmer, (uint64_t*)impl_->d_vtab, impl_->vcap); break;
        default: LDBG_LAUNCH(k_cursor_seek<4>, 1, 64, s, eng_.view, (CursorStateDev<4>*)impl_->d_state, (const uint64_t*)impl_->d_words, is_kmer, (uint64_t*)impl_->d_vtab, impl_->vcap); break;
    }
    rt::stream_sync(s);
    impl_->sought = true;
    bool hn, hp;
    uint32_t st;
    peek(&hn, &hp, &st, nullptr, nullptr);
    check_status(st);
}

void CursorHost::peek(bool* has_next, bool* has_prev, uint32_t* status, uint64_t* out_words, int64_t* out_rec) {
    const int W = eng_.graph->hdr.W;
    rt::stream_t s = eng_.estream();
#define LDBG_PEEK(WW)                                                                       \
    {                                                                                       \
        CursorStateDev<WW> h;                                                               \
        read_state<WW>(impl_->d_state, h, s);                                               \
        *has_next = h.has_next != 0; *has_prev = h.has_prev != 0; *status = h.status;       \
        if (out_words) for (int i = 0; i < WW; i++) out_words[i] = h.out_words[i];          \
        if (out_rec) *out_rec = h.out_rec;                                                  \
    }
    switch (W) { case 1: LDBG_PEEK(1) break; case 2: LDBG_PEEK(2) break; case 3: LDBG_PEEK(3) break; default: LDBG_PEEK(4) break; }
#undef LDBG_PEEK
}

int64_t CursorHost::cur_record() {
    rt::stream_t s = eng_.estream();
    int64_t idx = -1;
#define LDBG_CUR(WW) { CursorStateDev<WW> h; read_state<WW>(impl_->d_state, h, s); idx = h.cur.idx; }
    switch (eng_.graph->hdr.W) { case 1: LDBG_CUR(1) break; case 2: LDBG_CUR(2) break; case 3: LDBG_CUR(3) break; default: LDBG_CUR(4) break; }
#undef LDBG_CUR
    return idx;
}

bool CursorHost::has(bool fwd) {
    eng_.enter();
    if (!impl_->sought) return false;
    bool hn, hp; uint32_t st;
    peek(&hn, &hp, &st, nullptr, nullptr);
    return fwd ? hn : hp;
}

// TraversalEngine.assemble(seed) (:112-123): [previous() ... in contig order] + the seed's vertex + [next() ...]
void CursorHost::assemble(const char* seed, int64_t capacity, int64_t* len, uint64_t* words, int64_t* rec) {
    eng_.enter();
    const int W = eng_.graph->hdr.W;
    rt::stream_t s = eng_.estream();
    const int64_t max_len = std::max<int64_t>(0, eng_.cfg.max_branch_length);
    // the cursor's `seen` table holds every vertex stepped onto since the seek (load <= 1/2)
    uint32_t need = 1u << 17;
    while ((uint64_t)need < 2ull * (uint64_t)(max_len + 16)) need <<= 1;
    if (need > impl_->vcap) {
        rt::dfree(impl_->d_vtab);
        impl_->d_vtab = nullptr;
        impl_->d_vtab = rt::dmalloc((size_t)need * 8);
        impl_->vcap = need;
        rt::dmemset(impl_->d_vtab, 0, (size_t)need * 8, s);
    }
    struct Tmp { std::vector<void*> p; ~Tmp() { for (void* x : p) rt::dfree(x); } void* get(size_t nbytes) { void* x = rt::dmalloc(nbytes); p.push_back(x); return x; } } tmp;
    uint64_t* d_w = (uint64_t*)tmp.get((size_t)std::max<int64_t>(1, max_len) * W * 8);
    int64_t* d_r = (int64_t*)tmp.get((size_t)std::max<int64_t>(1, max_len) * 8);
    int64_t* d_n = (int64_t*)tmp.get(16);
    std::vector<uint64_t> part_w[2];
    std::vector<int64_t> part_r[2];
    std::vector<uint64_t> seed_w(W);
    int64_t seed_rec = -1;
    for (int dir = 1; dir >= 0; dir--) {                 // forward first, as the reference does
        seek(seed);
        if (dir == 1) {
            // the seed's own vertex: bases(seed), record(findRecord(seed))
            const bool ok = ascii_to_words(seed, eng_.graph->hdr.k, seed_w.data(), W);
            if (!ok) std::fill(seed_w.begin(), seed_w.end(), 0ull);
            seed_rec = cur_record();                     // (the seek has just looked the seed up)
        }
#define LDBG_ASM(WW) LDBG_LAUNCH(k_cursor_assemble<WW>, 1, 64, s, eng_.view, (CursorStateDev<WW>*)impl_->d_state, dir, (uint64_t*)impl_->d_vtab, impl_->vcap, \
                                 (LsElem*)impl_->d_ls, impl_->ecap, max_len, d_w, d_r, d_n)
        switch (W) { case 1: LDBG_ASM(1); break; case 2: LDBG_ASM(2); break; case 3: LDBG_ASM(3); break; default: LDBG_ASM(4); break; }
#undef LDBG_ASM
        int64_t hn[2] = {0, 0};
        rt::d2h(hn, d_n, 16, s);
        rt::stream_sync(s);
        check_status((uint32_t)hn[1]);
        part_w[dir].resize((size_t)hn[0] * W);
        part_r[dir].resize((size_t)hn[0]);
        rt::d2h(part_w[dir].data(), d_w, (size_t)hn[0] * W * 8, s);
        rt::d2h(part_r[dir].data(), d_r, (size_t)hn[0] * 8, s);
        rt::stream_sync(s);
    }
    const int64_t nr = (int64_t)part_r[0].size(), nf = (int64_t)part_r[1].size();
    *len = nr + 1 + nf;
    if (capacity < *len) throw StatusError(LDBG_ERR_CAPACITY, "vertex buffers too small: need " + std::to_string(*len));
    for (int64_t i = 0; i < nr; i++) {                   // contig.add(0, cv): the last one stepped onto comes first
        const int64_t from = nr - 1 - i;
        if (words) for (int w = 0; w < W; w++) words[i * W + w] = part_w[0][(size_t)from * W + w];
        if (rec) rec[i] = part_r[0][(size_t)from];
    }
    if (words) for (int w = 0; w < W; w++) words[nr * W + w] = seed_w[(size_t)w];
    if (rec) rec[nr] = seed_rec;
    for (int64_t i = 0; i < nf; i++) {
        if (words) for (int w = 0; w < W; w++) words[(nr + 1 + i) * W + w] = part_w[1][(size_t)i * W + w];
        if (rec) rec[nr + 1 + i] = part_r[1][(size_t)i];
    }
}

void CursorHost::step(bool fwd, char* kmer_out, int64_t* rec_out) {
    eng_.enter();
    const int W = eng_.graph->hdr.W, k = eng_.graph->hdr.k;
    if (!has(fwd))
        throw StatusError(LDBG_ERR_NOSUCHELEMENT, std::string("No single ") + (fwd ? "advance" : "prev") + " kmer from cursor");
    rt::stream_t s = eng_.estream();
    switch (W) {
        case 1: LDBG_LAUNCH(k_cursor_step<1>, 1, 64, s, eng_.view, (CursorStateDev<1>*)impl_->d_state, fwd ? 1 : 0, (uint64_t*)impl_->d_vtab, impl_->vcap, (LsElem*)impl_->d_ls, impl_->ecap); break;
        case 2: LDBG_LAUNCH(k_cursor_step<2>, 1, 64, s, eng_.view, (CursorStateDev<2>*)impl_->d_state, fwd ? 1 : 0, (uint64_t*)impl_->d_vtab, impl_->vcap, (LsElem*)impl_->d_ls, impl_->ecap); break;
        case 3: LDBG_LAUNCH(k_cursor_step<3>, 1, 64, s, eng_.view, (CursorStateDev<3>*)impl_->d_state, fwd ? 1 : 0, (uint64_t*)impl_->d_vtab, impl_->vcap, (LsElem*)impl_->d_ls, impl_->ecap); break;
        default: LDBG_LAUNCH(k_cursor_step<4>, 1, 64, s, eng_.view, (CursorStateDev<4>*)impl_->d_state, fwd ? 1 : 0, (uint64_t*)impl_->d_vtab, impl_->vcap, (LsElem*)impl_->d_ls, impl_->ecap); break;
    }
    rt::stream_sync(s);
    bool hn, hp; uint32_t st;
    std::vector<uint64_t> w(W);
    int64_t rec = -1;
    peek(&hn, &hp, &st, w.data(), &rec);
    check_status(st);
    if (kmer_out) { words_to_ascii(w.data(), k, W, kmer_out); kmer_out[k] = 0; }
    if (rec_out) *rec_out = rec;
}

}  // namespace ldbg
