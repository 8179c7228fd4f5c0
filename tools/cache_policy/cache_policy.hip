// cache_policy.hip — does any of the eight sc0 / sc1 / nt combinations of global_load_dwordx4 read a 2 GiB stream faster
// than the plain "nt" the library uses (clang's __builtin_nontemporal_load)?  Same access pattern as the fused kernel:
// 256 workgroups x 256 threads, 8 x 16-byte loads per lane in flight, grid-strided.  Average of 20 launches, hipEvents.
//   hipcc --offload-arch=gfx950 -O3 -o cache_policy cache_policy.hip && ./cache_policy
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <vector>
#include <algorithm>

typedef double dvec2 __attribute__((ext_vector_type(2)));

template <int POLICY>
__device__ __forceinline__ dvec2 load16(const dvec2* p) {
    dvec2 v;
    if (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int POLICY>
__global__ __launch_bounds__(256) void probe(const dvec2* __restrict__ src, long n, double* __restrict__ out) {
    constexpr int U = 8;
    double acc = 0.0;
    const long stride = (long)gridDim.x * 256;
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c + (U - 1) * stride < n; c += (long)U * stride) {
        dvec2 x[U];
#pragma unroll
        for (int k = 0; k < U; ++k) x[k] = load16<POLICY>(src + c + (long)k * stride);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < U; ++k) acc += x[k].x + x[k].y;
    }
    if (acc == 1.2345e300) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int POLICY>
double run(const dvec2* src, long n, double* out, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe<POLICY>, dim3(grid), dim3(256), 0, 0, src, n, out);
    double total = 0.0;
    const int reps = 20;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<POLICY>, dim3(grid), dim3(256), 0, 0, src, n, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total += ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 16.0 * n / (total / reps * 1e-3) / 1e9;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const long n = (2l << 30) / 16;
    dvec2* src; double* out;
    if (hipMalloc(&src, n * 16) != hipSuccess || hipMalloc(&out, 8l << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 0, n * 16);
    const char* names[8] = {"(none)", "sc0", "sc1", "sc0 sc1", "nt", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
    for (int wgcu : {1, 2}) {
        const int grid = prop.multiProcessorCount * wgcu;
        double g[8];
        for (int round = 0; round < 2; ++round) {        // two interleaved rounds
            g[0] = run<0>(src, n, out, grid); g[1] = run<1>(src, n, out, grid); g[2] = run<2>(src, n, out, grid); g[3] = run<3>(src, n, out, grid);
            g[4] = run<4>(src, n, out, grid); g[5] = run<5>(src, n, out, grid); g[6] = run<6>(src, n, out, grid); g[7] = run<7>(src, n, out, grid);
            for (int p = 0; p < 8; ++p) printf("wg/cu %d round %d  %-12s %7.0f GB/s\n", wgcu, round, names[p], g[p]);
        }
    }
    return 0;
}
