// graph_probe.hip — does a hipGraph shorten a chain of small DEPENDENT kernels (the shape of a CG iteration's tail:
// slab reduce -> step, plus a gated streaming kernel) compared with plain stream launches issued ahead of the GPU?
//   hipcc --offload-arch=gfx950 -O3 -o graph_probe graph_probe.hip && ./graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_wide(const double* in, double* out, int n) {          // 256 workgroups, trivial work (a gated streaming kernel)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0;
}
__global__ void k_reduce(const double* in, double* out, int n) {        // 128 workgroups
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 0.5 + in[(i + 1) % n] * 0.5;
}
__global__ void k_step(const double* in, double* out, int n) {          // one workgroup of 1024
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = in[i] - 1.0;
}

int main() {
    const int n = 4096, iters = 3000, per_graph = 10;
    double *a, *b, *c;
    CK(hipMalloc(&a, 65536 * 8)); CK(hipMalloc(&b, 65536 * 8)); CK(hipMalloc(&c, 65536 * 8));
    CK(hipMemset(a, 0, 65536 * 8));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto iteration = [&](hipStream_t st) {
        hipLaunchKernelGGL(k_wide, dim3(256), dim3(256), 0, st, (const double*)a, b, 65536);
        hipLaunchKernelGGL(k_reduce, dim3(128), dim3(256), 0, st, (const double*)b, c, n);
        hipLaunchKernelGGL(k_step, dim3(1), dim3(1024), 0, st, (const double*)c, a, n);
    };
    for (int i = 0; i < 100; ++i) iteration(s);
    CK(hipStreamSynchronize(s));

    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) iteration(s);
    auto t1 = std::chrono::steady_clock::now();            // host done enqueueing
    CK(hipStreamSynchronize(s));
    auto t2 = std::chrono::steady_clock::now();
    const double enq = std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
    const double tot = std::chrono::duration<double, std::micro>(t2 - t0).count() / iters;
    printf("stream launches : %.2f us per iteration (3 dependent kernels) on the device, %.2f us of host enqueue time\n", tot, enq);

    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < per_graph; ++i) iteration(s);
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(exec, s));
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters / per_graph; ++i) CK(hipGraphLaunch(exec, s));
    t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    t2 = std::chrono::steady_clock::now();
    printf("hipGraph (%d iterations per graph): %.2f us per iteration on the device, %.2f us of host enqueue time\n", per_graph,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / iters, std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
    return 0;
}
