// Runtime shim: the ONE place where kernels meet the HIP runtime.
//
// Product build (hipcc, gfx950): thin inline wrappers over hipMalloc / hipMemcpyAsync /
// hipLaunchKernelGGL.  There is no CPU fallback in the product library: without a device every
// allocation or launch fails with LDBG_ERR_HIP.
//
// LDBG_HOSTSIM build (plain g++, tests only — tests/hostsim/): the same kernel sources are
// compiled as ordinary C++ and a "launch" runs the kernel body once per simulated thread.  This
// exists so that the kernel logic can be unit-tested in the CPU-only CI container; it is never
// linked into libldbg.so and never shipped.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/ldbg.h"
#include "ctx_host.h"   // StatusError

#ifndef LDBG_HOSTSIM
#include <hip/hip_runtime.h>
#define LDBG_KERNEL __global__
#define LDBG_WAVE_KERNEL __global__ __launch_bounds__(64)   // launched one wavefront per workgroup: no register cap
#define LDBG_WAVE_KERNEL_N(n) __global__ __launch_bounds__(n)
#define LDBG_DEV __device__ __forceinline__
#define LDBG_HOSTDEV __host__ __device__ __forceinline__
// a pointer known to be into HBM (a strand's pointers pass through lane broadcasts, which hide that from the compiler:
// it would emit flat_ accesses, which also wait on the LDS counter)
#if defined(__HIP_DEVICE_COMPILE__)
#define LDBG_GLOBAL(T, p) ((__attribute__((address_space(1))) T*)(p))
#define LDBG_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))       // likewise for a pointer into the workgroup's LDS
#else
#define LDBG_GLOBAL(T, p) ((T*)(p))
#define LDBG_LDS(T, p) ((T*)(p))
#endif

namespace ldbg {
namespace rt {

inline void check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw StatusError(LDBG_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
inline int device_count() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
inline void set_device(int d) { check(hipSetDevice(d), "hipSetDevice"); }
inline int cu_count(int d) {            // compute units of the device (256 on an MI355X in SPX mode; fewer on a partition)
    int n = 0;
    check(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d), "hipDeviceGetAttribute");
    return n > 0 ? n : 256;
}
// Device blocks that come and go with every batch (a dozen per walk batch: status words, counters, offsets ...) are kept in free lists by
// size class instead of going back to the runtime: hipMalloc / hipFree cost tens of microseconds each and hipFree waits for the whole device.
// tmalloc / tfree are that path; a block may be handed to tfree only AFTER the host has waited for the stream that last used it (hipFree
// would have waited by itself; a cached block is handed out again at once).  dmalloc / dfree stay the plain runtime calls, and a block from
// tmalloc may be given to either.  Blocks above 64 MB and anything beyond 1 GB held are not kept; a failing hipMalloc empties the lists first.
struct BlockCache {
    std::mutex m;
    std::unordered_map<void*, std::pair<int, size_t>> live;            // blocks handed out by tmalloc: device, class size
    std::map<std::pair<int, size_t>, std::vector<void*>> free_;
    size_t held = 0;
    static size_t size_class(size_t n) {                                // 2^k or 1.5 x 2^k, at least 4 KB
        size_t c = 4096;
        while (c < n) { if (c + c / 2 >= n) return c + c / 2; c *= 2; }
        return c;
    }
    void release_all() {
        std::lock_guard<std::mutex> g(m);
        for (auto& kv : free_) for (void* p : kv.second) (void)hipFree(p);
        free_.clear(); held = 0;
    }
};
inline BlockCache& block_cache() { static BlockCache c; return c; }
inline void* dmalloc(size_t n) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, n ? n : 1);
    if (e != hipSuccess) { (void)hipGetLastError(); block_cache().release_all(); e = hipMalloc(&p, n ? n : 1); }
    check(e, "hipMalloc");
    return p;
}
inline void dfree(void* p) {
    if (!p) return;
    { BlockCache& c = block_cache(); std::lock_guard<std::mutex> g(c.m); c.live.erase(p); }
    (void)hipFree(p);
}
inline void* tmalloc(size_t n) {
    BlockCache& c = block_cache();
    const size_t cls = BlockCache::size_class(n);
    if (cls > ((size_t)64 << 20)) return dmalloc(n);
    int dev = 0;
    check(hipGetDevice(&dev), "hipGetDevice");
    {
        std::lock_guard<std::mutex> g(c.m);
        auto it = c.free_.find({dev, cls});
        if (it != c.free_.end() && !it->second.empty()) {
            void* p = it->second.back();
            it->second.pop_back();
            c.held -= cls;
            c.live[p] = {dev, cls};
            return p;
        }
    }
    void* p = dmalloc(cls);
    std::lock_guard<std::mutex> g(c.m);
    c.live[p] = {dev, cls};
    return p;
}
inline void tfree(void* p) {
    if (!p) return;
    BlockCache& c = block_cache();
    {
        std::lock_guard<std::mutex> g(c.m);
        auto it = c.live.find(p);
        if (it != c.live.end() && c.held + it->second.second <= ((size_t)1 << 30)) {
            c.free_[it->second].push_back(p);
            c.held += it->second.second;
            c.live.erase(it);
            return;
        }
        if (it != c.live.end()) c.live.erase(it);
    }
    (void)hipFree(p);
}
inline void* hmalloc_pinned(size_t n) { void* p = nullptr; check(hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault), "hipHostMalloc"); return p; }
inline void hfree_pinned(void* p) { if (p) (void)hipHostFree(p); }
// is p page-locked host memory the runtime knows (ldbg_host_alloc, hipHostMalloc, hipHostRegister)?  Copies to it run at the bus rate.
inline bool host_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
typedef hipStream_t stream_t;
inline stream_t stream_create() { hipStream_t s; check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate"); return s; }
inline void stream_destroy(stream_t s) { if (s) (void)hipStreamDestroy(s); }
inline void stream_sync(stream_t s) { check(hipStreamSynchronize(s), "hipStreamSynchronize"); }
inline void h2d(void* d, const void* h, size_t n, stream_t s) { if (n) check(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s), "hipMemcpyAsync H2D"); }
inline void d2h(void* h, const void* d, size_t n, stream_t s) { if (n) check(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s), "hipMemcpyAsync D2H"); }
inline void d2d(void* d, const void* s_, size_t n, stream_t s) { if (n) check(hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync D2D"); }
inline void dmemset(void* d, int v, size_t n, stream_t s) { if (n) check(hipMemsetAsync(d, v, n, s), "hipMemsetAsync"); }
inline void mem_info(size_t* free_b, size_t* total_b) { check(hipMemGetInfo(free_b, total_b), "hipMemGetInfo"); }
inline void launch_check(const char* name) { check(hipGetLastError(), name); }

struct Event {
    hipEvent_t e = nullptr;
    Event() { check(hipEventCreate(&e), "hipEventCreate"); }
    ~Event() { if (e) (void)hipEventDestroy(e); }
    Event(Event&& o) noexcept : e(o.e) { o.e = nullptr; }
    Event& operator=(Event&& o) noexcept { if (this != &o) { if (e) (void)hipEventDestroy(e); e = o.e; o.e = nullptr; } return *this; }
    Event(const Event&) = delete;
    Event& operator=(const Event&) = delete;
    void record(stream_t s) { check(hipEventRecord(e, s), "hipEventRecord"); }
    void wait() { check(hipEventSynchronize(e), "hipEventSynchronize"); }
    static float elapsed_ms(Event& a, Event& b) {
        check(hipEventSynchronize(b.e), "hipEventSynchronize");
        float ms = 0;
        check(hipEventElapsedTime(&ms, a.e, b.e), "hipEventElapsedTime");
        return ms;
    }
};

}  // namespace rt
}  // namespace ldbg

#define LDBG_LAUNCH(kernel, grid, block, stream, ...)                               \
    do {                                                                            \
        hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, stream, __VA_ARGS__); \
        ::ldbg::rt::launch_check(#kernel);                                          \
    } while (0)

namespace ldbg {
LDBG_DEV int64_t global_tid() { return (int64_t)blockIdx.x * blockDim.x + threadIdx.x; }
LDBG_DEV int64_t global_nthreads() { return (int64_t)gridDim.x * blockDim.x; }
LDBG_DEV unsigned long long atomic_add_u64(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }
LDBG_DEV unsigned atomic_add_u32(unsigned* p, unsigned v) { return atomicAdd(p, v); }
LDBG_DEV unsigned atomic_min_u32(unsigned* p, unsigned v) { return atomicMin(p, v); }
LDBG_DEV unsigned atomic_or_u32(unsigned* p, unsigned v) { return atomicOr(p, v); }
LDBG_DEV unsigned long long atomic_min_u64(unsigned long long* p, unsigned long long v) { return atomicMin(p, v); }
LDBG_DEV unsigned long long atomic_cas_u64(unsigned long long* p, unsigned long long cmp, unsigned long long v) { return atomicCAS(p, cmp, v); }
LDBG_DEV unsigned long long atomic_exch_u64(unsigned long long* p, unsigned long long v) { return atomicExch(p, v); }
// wavefront primitives (64 lanes on gfx950); kernels that use them are launched with 64-thread blocks
LDBG_DEV int wave_size() { return blockDim.x < 64u ? (int)blockDim.x : 64; }   // a (possibly partial) wavefront; workgroups of several wavefronts index them by global_tid() / wave_size()
LDBG_DEV int wave_lane() { return (int)(threadIdx.x & 63u); }
LDBG_DEV unsigned long long wave_ballot(bool p) { return __ballot(p ? 1 : 0); }
// broadcast from a lane every lane agrees on (src is wave-uniform at every call site: it comes from a ballot): v_readlane,
// not a cross-lane shuffle through the LDS crossbar
LDBG_DEV uint32_t wave_bcast_u32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src)); }
LDBG_DEV uint64_t wave_bcast_u64(uint64_t v, int src) {
    uint32_t lo = wave_bcast_u32((uint32_t)v, src), hi = wave_bcast_u32((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
// value of ANY lane (src differs from lane to lane: a permute through the LDS crossbar, ds_bpermute)
LDBG_DEV uint32_t wave_shfl_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src & 63, 64); }
LDBG_DEV uint64_t wave_shfl_u64(uint64_t v, int src) { return ((uint64_t)wave_shfl_u32((uint32_t)(v >> 32), src) << 32) | wave_shfl_u32((uint32_t)v, src); }
LDBG_DEV void wave_fence() { __threadfence_block(); }
LDBG_DEV void device_fence() { __threadfence(); }
LDBG_DEV uint64_t wave_shfl_xor_u64(uint64_t v, int m) {
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}
// butterfly reductions: every lane receives the result
LDBG_DEV uint64_t wave_min_u64(uint64_t v) { for (int m = wave_size() >> 1; m > 0; m >>= 1) { uint64_t o = wave_shfl_xor_u64(v, m); v = o < v ? o : v; } return v; }
LDBG_DEV uint64_t wave_max_u64(uint64_t v) { for (int m = wave_size() >> 1; m > 0; m >>= 1) { uint64_t o = wave_shfl_xor_u64(v, m); v = o > v ? o : v; } return v; }
LDBG_DEV int wave_count_below(unsigned long long ballot) { return __builtin_popcountll(ballot & ((1ull << wave_lane()) - 1ull)); }
// inclusive prefix sum over the lanes of a wavefront
LDBG_DEV uint32_t wave_incl_scan_u32(uint32_t v) {
    const int lane = wave_lane();
    for (int d = 1; d < wave_size(); d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64); if (lane >= d) v += o; }
    return v;
}
}  // namespace ldbg

#else  // ------------------------------------------------------------------ LDBG_HOSTSIM (tests only)
#define LDBG_KERNEL static
#define LDBG_WAVE_KERNEL static
#define LDBG_WAVE_KERNEL_N(n) static
#define LDBG_DEV inline
#define LDBG_HOSTDEV inline
#define LDBG_GLOBAL(T, p) ((T*)(p))
#define LDBG_LDS(T, p) ((T*)(p))
#ifndef __forceinline__
#define __forceinline__
#endif

namespace ldbg {
namespace rt {
inline int device_count() { return 1; }
inline void set_device(int) {}
inline int cu_count(int) { return 256; }
// LDBG_HOSTSIM_POISON=1: "device" memory comes back filled with 0xAB instead of zeros and the simulated LDS is filled likewise when a
// wavefront starts — hipMalloc and LDS hand out whatever the last user left, and code that relies on zeros there must fail HERE
inline bool poison() { static const bool on = getenv("LDBG_HOSTSIM_POISON") != nullptr; return on; }
inline void* dmalloc(size_t n) {
    if (!poison()) return calloc(n ? n : 1, 1);
    void* p = malloc(n ? n : 1);
    if (p) memset(p, 0xAB, n ? n : 1);
    return p;
}
inline void dfree(void* p) { free(p); }
inline void* tmalloc(size_t n) { return dmalloc(n); }
inline void tfree(void* p) { free(p); }
inline void* hmalloc_pinned(size_t n) { return malloc(n ? n : 1); }
inline void hfree_pinned(void* p) { free(p); }
inline bool host_is_pinned(const void*) { return getenv("LDBG_HOSTSIM_PAGEABLE") == nullptr; }     // (the test hook sends copies through the staging path)
typedef void* stream_t;
inline stream_t stream_create() { return nullptr; }
inline void stream_destroy(stream_t) {}
inline void stream_sync(stream_t) {}
inline void h2d(void* d, const void* h, size_t n, stream_t) { if (n) memcpy(d, h, n); }
inline void d2h(void* h, const void* d, size_t n, stream_t) { if (n) memcpy(h, d, n); }
inline void d2d(void* d, const void* s, size_t n, stream_t) { if (n) memmove(d, s, n); }
inline void dmemset(void* d, int v, size_t n, stream_t) { if (n) memset(d, v, n); }
inline void mem_info(size_t* free_b, size_t* total_b) {      // LDBG_HOSTSIM_MEM_MB: pretend to be a small device (tests of the batch-splitting paths)
    size_t mb = 2048;
    if (const char* ev = getenv("LDBG_HOSTSIM_MEM_MB")) mb = (size_t)atoll(ev);
    *free_b = mb << 20; *total_b = mb << 20;
}
struct Event {
    void record(stream_t) {}
    void wait() {}
    static float elapsed_ms(Event&, Event&) { return 0.0f; }
};
}  // namespace rt
// ---- simulated wavefronts.  LDBG_HOSTSIM_LANES=1 (default): a "wave" is a single lane, every thread runs to completion in turn.
// LDBG_HOSTSIM_LANES=N (N = 64 matches the device): a kernel launched one wavefront per workgroup (block <= 64) runs its lanes in
// LOCK STEP — one fibre per lane; a wavefront primitive (ballot, broadcast, permute, scan) is a barrier at which every live lane deposits
// its operand and then reads the others' — so that the wave-cooperative code (lscoop.h, strand.h: table regrowth, walk.cpp: expansion,
// image.cpp: bucketing) is executed on the CPU as it is on the device.  Lanes that have left the kernel count as inactive.  A lane
// that arrives at a DIFFERENT primitive than its neighbours is a divergence bug: the simulation aborts and says so.
}  // namespace ldbg
#include <execinfo.h>
#include <ucontext.h>
#include <stdio.h>
#include <functional>
#include <vector>
namespace ldbg {
namespace sim {
struct Wave {
    int lanes = 1, cur = 0, live = 0, arrived = 0;
    unsigned long long gen = 0;
    bool active = false;
    ucontext_t sched;
    ucontext_t ctx[64];
    std::vector<char> stacks[64];
    uint8_t done[64];
    int kind[64];
    void* site[64];
    uint64_t x[2][64];
    uint64_t y[2][64];
    int64_t tid0 = 0, nthreads = 1;
    std::function<void()>* body = nullptr;
};
inline Wave& wave() { static thread_local Wave w; return w; }
inline int& lanes_setting() { static int n = [] { const char* e = getenv("LDBG_HOSTSIM_LANES"); int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 64 ? 64 : v); }(); return n; }
inline int lanes_env() { return lanes_setting(); }
inline void yield() { Wave& w = wave(); swapcontext(&w.ctx[w.cur], &w.sched); }
// deposit (a, b), wait until every live lane has; returns the buffer to read
inline int collective(int kind, uint64_t a, uint64_t b = 0) {
    Wave& w = wave();
    const int buf = (int)(w.gen & 1ull);
    const unsigned long long my_gen = w.gen;
    w.x[buf][w.cur] = a; w.y[buf][w.cur] = b; w.kind[w.cur] = kind; w.site[w.cur] = __builtin_return_address(0);
    w.arrived++;
    if (w.arrived == w.live) {
        for (int l = 0; l < w.lanes; l++)
            if (!w.done[l] && w.kind[l] != kind) {
                fprintf(stderr, "[hostsim] wavefront divergence: lane %d is at primitive %d (%p), lane %d at %d (%p); the arriving lane's stack:\n", w.cur, kind, w.site[w.cur], l, w.kind[l], w.site[l]);
                void* bt[32];
                backtrace_symbols_fd(bt, backtrace(bt, 32), 2);
                abort();
            }
        w.arrived = 0; w.gen++;
    } else {
        while (w.gen == my_gen) yield();
    }
    return buf;
}
inline void trampoline() {
    Wave& w = wave();
    (*w.body)();
    w.done[w.cur] = 1;
    w.live--;
    if (w.live > 0 && w.arrived == w.live) {       // the lanes still at a barrier were waiting for this one only
        const int kd = [&] { for (int l = 0; l < w.lanes; l++) if (!w.done[l]) return w.kind[l]; return 0; }();
        for (int l = 0; l < w.lanes; l++)
            if (!w.done[l] && w.kind[l] != kd) { fprintf(stderr, "[hostsim] wavefront divergence at a lane's exit\n"); abort(); }
        w.arrived = 0; w.gen++;
    }
    swapcontext(&w.ctx[w.cur], &w.sched);
}
// one wavefront of `lanes` lanes in lock step; body() is the kernel call (it reads its thread index through sim_idx())
void run_wave(int lanes, int64_t tid0, int64_t nthreads, std::function<void()>& body);
}  // namespace sim
struct SimIdx { int64_t tid, nthreads; };
inline SimIdx& sim_idx() { static thread_local SimIdx s{0, 1}; return s; }
inline int64_t global_tid() { return sim_idx().tid; }
inline int64_t global_nthreads() { return sim_idx().nthreads; }
inline void sim::run_wave(int lanes, int64_t tid0, int64_t nthreads, std::function<void()>& body) {
    Wave& w = wave();
    w.lanes = lanes; w.live = lanes; w.arrived = 0; w.gen = 0; w.active = true; w.body = &body; w.tid0 = tid0; w.nthreads = nthreads;
    for (int l = 0; l < lanes; l++) {
        if (w.stacks[l].empty()) w.stacks[l].resize((size_t)1 << 20);
        w.done[l] = 0; w.kind[l] = 0;
        getcontext(&w.ctx[l]);
        w.ctx[l].uc_stack.ss_sp = w.stacks[l].data();
        w.ctx[l].uc_stack.ss_size = w.stacks[l].size();
        w.ctx[l].uc_link = &w.sched;
        makecontext(&w.ctx[l], (void (*)())sim::trampoline, 0);
    }
    while (w.live > 0) {
        for (int l = 0; l < lanes; l++) {
            if (w.done[l]) continue;
            w.cur = l;
            sim_idx().tid = tid0 + l; sim_idx().nthreads = nthreads;
            swapcontext(&w.sched, &w.ctx[l]);
        }
    }
    w.active = false; w.lanes = 1; w.cur = 0;
}
inline unsigned long long atomic_add_u64(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
inline unsigned atomic_add_u32(unsigned* p, unsigned v) { unsigned o = *p; *p += v; return o; }
inline unsigned atomic_min_u32(unsigned* p, unsigned v) { unsigned o = *p; if (v < o) *p = v; return o; }
inline unsigned atomic_or_u32(unsigned* p, unsigned v) { unsigned o = *p; *p |= v; return o; }
inline unsigned long long atomic_min_u64(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v < o) *p = v; return o; }
inline unsigned long long atomic_cas_u64(unsigned long long* p, unsigned long long cmp, unsigned long long v) { unsigned long long o = *p; if (o == cmp) *p = v; return o; }
inline unsigned long long atomic_exch_u64(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = v; return o; }
inline int wave_size() { return sim::wave().active ? sim::wave().lanes : 1; }
inline int wave_lane() { return sim::wave().active ? sim::wave().cur : 0; }
inline unsigned long long wave_ballot(bool p) {
    sim::Wave& w = sim::wave();
    if (!w.active) return p ? 1ull : 0ull;
    const int b = sim::collective(1, p ? 1ull : 0ull);
    unsigned long long m = 0;
    for (int l = 0; l < w.lanes; l++) if (!w.done[l] && w.x[b][l]) m |= 1ull << l;
    return m;
}
// value of lane `src` (any lane: a permute; the device code uses v_readlane where src is uniform and ds_bpermute where it is not)
inline uint64_t wave_shfl_u64(uint64_t v, int src) {
    sim::Wave& w = sim::wave();
    if (!w.active) return v;
    const int b = sim::collective(2, v);
    return w.x[b][src & (w.lanes - 1)];
}
inline uint32_t wave_shfl_u32(uint32_t v, int src) { return (uint32_t)wave_shfl_u64(v, src); }
inline uint32_t wave_bcast_u32(uint32_t v, int src) { return (uint32_t)wave_shfl_u64(v, src); }
inline uint64_t wave_bcast_u64(uint64_t v, int src) { return wave_shfl_u64(v, src); }
inline uint64_t wave_shfl_xor_u64(uint64_t v, int m) { return wave_shfl_u64(v, wave_lane() ^ m); }
// a lane's write followed by another lane's read of the same place is ordered on the device by the lock step itself (the fence makes the
// write visible); the fibres of the simulation run one after the other between two primitives, so the fence is a barrier here: every
// lane has done its writes before any lane goes on to read
inline void wave_fence() { if (sim::wave().active) (void)sim::collective(3, 0ull); }
inline void device_fence() {}
inline uint64_t wave_min_u64(uint64_t v) { for (int m = wave_size() >> 1; m > 0; m >>= 1) { uint64_t o = wave_shfl_xor_u64(v, m); v = o < v ? o : v; } return v; }
inline uint64_t wave_max_u64(uint64_t v) { for (int m = wave_size() >> 1; m > 0; m >>= 1) { uint64_t o = wave_shfl_xor_u64(v, m); v = o > v ? o : v; } return v; }
inline int wave_count_below(unsigned long long ballot) { return __builtin_popcountll(ballot & ((1ull << wave_lane()) - 1ull)); }
inline uint32_t wave_incl_scan_u32(uint32_t v) {
    sim::Wave& w = sim::wave();
    if (!w.active) return v;
    const int b = sim::collective(3, v);
    uint32_t s = 0;
    for (int l = 0; l <= w.cur; l++) if (!w.done[l]) s += (uint32_t)w.x[b][l];
    return s;
}
}  // namespace ldbg

// "launch": LDBG_HOSTSIM_LANES=1, or a kernel of wider workgroups (none of those uses a wavefront primitive): every simulated thread
// runs to completion in turn.  Otherwise: wavefront after wavefront, the lanes of each in lock step.
#define LDBG_LAUNCH(kernel, grid, block, stream, ...)                      \
    do {                                                                   \
        int64_t nt__ = (int64_t)(grid) * (int64_t)(block);                 \
        if (nt__ > 4096) nt__ = 4096;                                      \
        const int lanes__ = ::ldbg::sim::lanes_env() > 1 && (int64_t)(block) <= 64 ? (int)std::min<int64_t>((int64_t)(block), (int64_t)::ldbg::sim::lanes_env()) : 1; \
        if (lanes__ > 1) {                                                 \
            nt__ = (nt__ / lanes__) * lanes__; if (nt__ < lanes__) nt__ = lanes__; \
            std::function<void()> body__ = [&]() { kernel(__VA_ARGS__); }; \
            for (int64_t t__ = 0; t__ < nt__; t__ += lanes__) ::ldbg::sim::run_wave(lanes__, t__, nt__, body__); \
            ::ldbg::sim_idx().tid = 0; ::ldbg::sim_idx().nthreads = 1;     \
            break;                                                         \
        }                                                                  \
        ::ldbg::sim_idx().nthreads = nt__;                                 \
        for (int64_t t__ = 0; t__ < nt__; t__++) {                         \
            ::ldbg::sim_idx().tid = t__;                                   \
            kernel(__VA_ARGS__);                                           \
        }                                                                  \
    } while (0)
#endif
