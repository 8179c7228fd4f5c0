// grid_sync_probe.hip — what does one grid-wide barrier cost on this device (cooperative launch, one workgroup per CU)?
// Input for the "one cooperative kernel per CG iteration" idea in DESIGN.md §5 (docs/design_history_r1_r2.md §7): such a kernel needs 3 barriers per iteration.
//   hipcc --offload-arch=gfx950 -O3 -o grid_sync_probe grid_sync_probe.hip && ./grid_sync_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>

namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void barrier_loop(double* buf, int rounds) {
    cg::grid_group grid = cg::this_grid();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) buf[blockIdx.x] = acc + r;          // a little cross-workgroup traffic per round
        grid.sync();
        acc += buf[(blockIdx.x + 1) % gridDim.x];
    }
    if (tid == 0) buf[gridDim.x] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    if (!prop.cooperativeLaunch) { printf("cooperative launch not supported\n"); return 0; }
    double* buf;
    CK(hipMalloc(&buf, 4096 * sizeof(double)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int threads : {256, 1024}) {
        int grid = prop.multiProcessorCount;
        for (int rounds : {0, 1000}) {
            void* args[] = {&buf, &rounds};
            CK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(barrier_loop), dim3(grid), dim3(threads), args, 0, 0));   // warm-up
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            CK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(barrier_loop), dim3(grid), dim3(threads), args, 0, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%d workgroups x %4d threads, %4d grid barriers: %.1f us total%s\n", grid, threads, rounds, 1e3 * ms,
                   rounds ? "" : " (launch only)");
            if (rounds) printf("   -> %.2f us per barrier round\n", 1e3 * ms / rounds);
        }
    }
    return 0;
}
