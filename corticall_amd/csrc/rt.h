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

#include <string>

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
inline void* dmalloc(size_t n) { void* p = nullptr; check(hipMalloc(&p, n ? n : 1), "hipMalloc"); return p; }
inline void dfree(void* p) { if (p) (void)hipFree(p); }
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
LDBG_DEV int wave_size() { return (int)blockDim.x; }   // wave kernels run one (possibly partial) wavefront per workgroup
LDBG_DEV int wave_lane() { return (int)(threadIdx.x & 63u); }
LDBG_DEV unsigned long long wave_ballot(bool p) { return __ballot(p ? 1 : 0); }
// broadcast from a lane every lane agrees on (src is wave-uniform at every call site: it comes from a ballot): v_readlane,
// not a cross-lane shuffle through the LDS crossbar
LDBG_DEV uint32_t wave_bcast_u32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src)); }
LDBG_DEV uint64_t wave_bcast_u64(uint64_t v, int src) {
    uint32_t lo = wave_bcast_u32((uint32_t)v, src), hi = wave_bcast_u32((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
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
inline void* dmalloc(size_t n) { return calloc(n ? n : 1, 1); }
inline void dfree(void* p) { free(p); }
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
struct SimIdx { int64_t tid, nthreads; };
inline SimIdx& sim_idx() { static thread_local SimIdx s{0, 1}; return s; }
inline int64_t global_tid() { return sim_idx().tid; }
inline int64_t global_nthreads() { return sim_idx().nthreads; }
inline unsigned long long atomic_add_u64(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
inline unsigned atomic_add_u32(unsigned* p, unsigned v) { unsigned o = *p; *p += v; return o; }
inline unsigned atomic_min_u32(unsigned* p, unsigned v) { unsigned o = *p; if (v < o) *p = v; return o; }
inline unsigned atomic_or_u32(unsigned* p, unsigned v) { unsigned o = *p; *p |= v; return o; }
inline unsigned long long atomic_min_u64(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v < o) *p = v; return o; }
inline unsigned long long atomic_cas_u64(unsigned long long* p, unsigned long long cmp, unsigned long long v) { unsigned long long o = *p; if (o == cmp) *p = v; return o; }
inline unsigned long long atomic_exch_u64(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = v; return o; }
// a simulated "wave" is a single lane
inline int wave_size() { return 1; }
inline int wave_lane() { return 0; }
inline unsigned long long wave_ballot(bool p) { return p ? 1ull : 0ull; }
inline uint32_t wave_bcast_u32(uint32_t v, int) { return v; }
inline uint64_t wave_bcast_u64(uint64_t v, int) { return v; }
inline void wave_fence() {}
inline void device_fence() {}
inline uint64_t wave_min_u64(uint64_t v) { return v; }
inline uint64_t wave_max_u64(uint64_t v) { return v; }
inline int wave_count_below(unsigned long long) { return 0; }
inline uint32_t wave_incl_scan_u32(uint32_t v) { return v; }
}  // namespace ldbg

// sequential "launch": every simulated thread runs to completion in turn.  Kernels must therefore
// not wait on other threads (none of ours do: walks are independent units).
#define LDBG_LAUNCH(kernel, grid, block, stream, ...)                      \
    do {                                                                   \
        int64_t nt__ = (int64_t)(grid) * (int64_t)(block);                 \
        if (nt__ > 4096) nt__ = 4096;                                      \
        ::ldbg::sim_idx().nthreads = nt__;                                 \
        for (int64_t t__ = 0; t__ < nt__; t__++) {                         \
            ::ldbg::sim_idx().tid = t__;                                   \
            kernel(__VA_ARGS__);                                           \
        }                                                                  \
    } while (0)
#endif
