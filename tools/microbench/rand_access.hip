// Microbenchmark (measurement aid, not product): dependent random 8-byte reads from HBM on MI355X as a
// function of the footprint they are spread over.  Answers "what does a pointer-chasing lane pay per access
// when 100k lanes each own a private table of R bytes" (visited tables) vs "one shared table of F bytes".
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void chase(const uint64_t* base, uint64_t region_words, uint64_t stride_words, int steps, uint64_t* out, int rmw) {
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t* p = (uint64_t*)base + tid * stride_words;
    uint64_t x = tid * 0x9E3779B97F4A7C15ull + 12345;
    uint64_t acc = 0;
    for (int i = 0; i < steps; i++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull + acc;
        uint64_t idx = (x >> 20) % region_words;
        uint64_t v = p[idx];
        if (rmw) p[idx] = v + 1;
        acc += v & 1;          // dependent chain
    }
    out[tid] = acc;
}

int main() {
    const int lanes = 1600 * 64;   // ~100k lanes, 1 wave per block
    uint64_t* out; CK(hipMalloc(&out, lanes * 8));
    size_t total = (size_t)96 << 30;
    uint64_t* buf; CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 0, total));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int steps = 2000;
    printf("mode,region_bytes,footprint_GB,rmw,ns_per_access_per_lane,G_accesses_per_s\n");
    for (int rmw = 0; rmw < 2; rmw++) {
        // private regions per lane
        for (size_t R : {(size_t)32 << 10, (size_t)128 << 10, (size_t)512 << 10, (size_t)900 << 10}) {
            hipLaunchKernelGGL(chase, dim3(lanes / 64), dim3(64), 0, 0, buf, R / 8, R / 8, 200, out, rmw); CK(hipDeviceSynchronize());
            hipEventRecord(a); hipLaunchKernelGGL(chase, dim3(lanes / 64), dim3(64), 0, 0, buf, R / 8, R / 8, steps, out, rmw); hipEventRecord(b); CK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("private,%zu,%.1f,%d,%.1f,%.2f\n", R, (double)R * lanes / (1 << 30), rmw, ms * 1e6 / steps, (double)lanes * steps / (ms * 1e-3) / 1e9);
        }
        // one shared region
        for (size_t F : {(size_t)1 << 30, (size_t)8 << 30, (size_t)32 << 30, (size_t)96 << 30}) {
            hipLaunchKernelGGL(chase, dim3(lanes / 64), dim3(64), 0, 0, buf, F / 8, 0, 200, out, rmw); CK(hipDeviceSynchronize());
            hipEventRecord(a); hipLaunchKernelGGL(chase, dim3(lanes / 64), dim3(64), 0, 0, buf, F / 8, 0, steps, out, rmw); hipEventRecord(b); CK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("shared,%zu,%.1f,%d,%.1f,%.2f\n", F, (double)F / (1 << 30), rmw, ms * 1e6 / steps, (double)lanes * steps / (ms * 1e-3) / 1e9);
        }
    }
    // fewer lanes: latency of a lone chain
    for (int nl : {64, 6400}) {
        size_t F = (size_t)32 << 30;
        hipEventRecord(a); hipLaunchKernelGGL(chase, dim3(nl / 64), dim3(64), 0, 0, buf, F / 8, 0, steps, out, 0); hipEventRecord(b); CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("shared_lanes%d,%zu,%.1f,0,%.1f,%.2f\n", nl, F, (double)F / (1 << 30), ms * 1e6 / steps, (double)nl * steps / (ms * 1e-3) / 1e9);
    }
    return 0;
}
