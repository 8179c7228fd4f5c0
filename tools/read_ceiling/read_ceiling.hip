// read_ceiling.hip — how fast can ANY kernel read 2 GiB once on this device?  Sweeps workgroup size, loads in flight,
// workgroups per CU, access pattern and cache policy of a kernel that does nothing but load and add.
//   hipcc --offload-arch=gfx950 -O3 -o read_ceiling read_ceiling.hip && ./read_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double dvec2 __attribute__((ext_vector_type(2)));

template <int T, int U, int NT, int BLOCKED>
__global__ __launch_bounds__(T) void probe(const dvec2* __restrict__ src, long n, double* __restrict__ out) {
    double acc[U];
#pragma unroll
    for (int k = 0; k < U; ++k) acc[k] = 0.0;
    if (BLOCKED) {
        // each workgroup owns one contiguous span; inside it the U loads of a step are T chunks apart
        const long per = (n + gridDim.x - 1) / gridDim.x;
        const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
        for (long c = lo + threadIdx.x; c < hi; c += (long)U * T) {
            dvec2 x[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const long i = c + (long)k * T;
                x[k] = i < hi ? (NT ? __builtin_nontemporal_load(src + i) : src[i]) : dvec2{0.0, 0.0};
            }
#pragma unroll
            for (int k = 0; k < U; ++k) acc[k] += x[k].x + x[k].y;
        }
    } else {
        const long stride = (long)gridDim.x * T;
        for (long c = (long)blockIdx.x * T + threadIdx.x; c < n; c += (long)U * stride) {
            dvec2 x[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const long i = c + (long)k * stride;
                x[k] = i < n ? (NT ? __builtin_nontemporal_load(src + i) : src[i]) : dvec2{0.0, 0.0};
            }
#pragma unroll
            for (int k = 0; k < U; ++k) acc[k] += x[k].x + x[k].y;
        }
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < U; ++k) s += acc[k];
    if (s == 1.2345e300) out[blockIdx.x * T + threadIdx.x] = s;      // keeps the loads alive, practically never stores
}

struct Result { double gbs; int T, U, nt, blocked, wgcu; };

template <int T, int U, int NT, int BLOCKED>
void run(const dvec2* src, long n, double* out, int n_cu, std::vector<Result>& res) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgcu : {1, 2, 4}) {
        if (wgcu * T > 2048) continue;
        const int grid = n_cu * wgcu;
        hipLaunchKernelGGL((probe<T, U, NT, BLOCKED>), dim3(grid), dim3(T), 0, 0, src, n, out);
        float best = 1e30f;
        for (int r = 0; r < 8; ++r) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((probe<T, U, NT, BLOCKED>), dim3(grid), dim3(T), 0, 0, src, n, out);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms);
        }
        res.push_back({16.0 * n / (best * 1e-3) / 1e9, T, U, NT, BLOCKED, wgcu});
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int T, int U>
void run_tu(const dvec2* src, long n, double* out, int n_cu, std::vector<Result>& res) {
    run<T, U, 1, 0>(src, n, out, n_cu, res);
    run<T, U, 1, 1>(src, n, out, n_cu, res);
    run<T, U, 0, 0>(src, n, out, n_cu, res);
    run<T, U, 0, 1>(src, n, out, n_cu, res);
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int n_cu = prop.multiProcessorCount;
    const long n = (2l << 30) / 16;              // 2 GiB of 16-byte chunks
    dvec2* src; double* out;
    if (hipMalloc(&src, n * 16) != hipSuccess || hipMalloc(&out, (size_t)n_cu * 4 * 1024 * 8) != hipSuccess) return 1;
    hipMemset(src, 0, n * 16);
    std::vector<Result> res;
    run_tu<256, 4>(src, n, out, n_cu, res);
    run_tu<256, 8>(src, n, out, n_cu, res);
    run_tu<256, 16>(src, n, out, n_cu, res);
    run_tu<512, 4>(src, n, out, n_cu, res);
    run_tu<512, 8>(src, n, out, n_cu, res);
    run_tu<1024, 4>(src, n, out, n_cu, res);
    run_tu<1024, 8>(src, n, out, n_cu, res);
    std::sort(res.begin(), res.end(), [](const Result& a, const Result& b) { return a.gbs > b.gbs; });
    printf("%s, %d CUs; best-of-8 single launches over 2 GiB\n", prop.name, n_cu);
    printf("%8s %5s %3s %3s %8s %6s\n", "GB/s", "T", "U", "nt", "pattern", "WG/CU");
    for (size_t i = 0; i < res.size(); ++i)
        if (i < 12 || i + 4 >= res.size())
            printf("%8.0f %5d %3d %3d %8s %6d\n", res[i].gbs, res[i].T, res[i].U, res[i].nt, res[i].blocked ? "blocked" : "strided", res[i].wgcu);
    hipFree(src); hipFree(out);
    return 0;
}
