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

// The same loop with a hand-written barrier instead of cooperative_groups: one agent-scope counter, every workgroup adds 1 and spins
// (s_sleep) until the counter reaches gridDim.x * (round + 1); release / acquire at agent scope around it so that the word written
// before the barrier is visible to the workgroup on another XCD after it.  Bounded spin: a barrier that does not complete within
// ~50 ms raises *fail and every workgroup leaves (no hang, whatever the scheduler does).  Launched as a plain kernel with one
// workgroup per CU, so all of them are resident.
__global__ void custom_barrier_loop(double* buf, unsigned* counter, int* fail, int rounds) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) __hip_atomic_store(&buf[blockIdx.x], acc + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = gridDim.x * (unsigned)(r + 1);
            long long spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 2000000 || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
        if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        acc += __hip_atomic_load(&buf[(blockIdx.x + 1) % gridDim.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    // hand-written barrier (plain launch)
    unsigned* counter; int* fail;
    CK(hipMalloc(&counter, 64)); CK(hipMalloc(&fail, 64));
    for (int threads : {256, 512}) {
        const int grid = prop.multiProcessorCount;
        for (int rounds : {0, 1000}) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemsetAsync(counter, 0, 64, 0)); CK(hipMemsetAsync(fail, 0, 64, 0));
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(custom_barrier_loop, dim3(grid), dim3(threads), 0, 0, buf, counter, fail, rounds);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                if (rep == 0) continue;                              // warm-up
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                int hfail = 0; double hacc = 0.0;
                CK(hipMemcpy(&hfail, fail, sizeof(int), hipMemcpyDeviceToHost));
                CK(hipMemcpy(&hacc, buf + grid, sizeof(double), hipMemcpyDeviceToHost));
                // every round adds the neighbour's (acc + r): acc_R = sum over rounds, identical in all workgroups
                printf("hand-written barrier: %d workgroups x %4d threads, %4d barriers: %.1f us total%s%s (check value %.0f)\n", grid, threads, rounds,
                       1e3 * ms, rounds ? "" : " (launch only)", hfail ? "  ** TIMED OUT **" : "", hacc);
                if (rounds && !hfail) printf("   -> %.2f us per barrier round\n", 1e3 * ms / rounds);
            }
        }
    }
    return 0;
}
