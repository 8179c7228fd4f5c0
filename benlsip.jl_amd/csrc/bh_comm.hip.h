// bh_comm.hip.h — one-shot all-reduce over peer-mapped buffers (xGMI, or one device shared by several processes)
// Part of the single translation unit of bh_api.hip (see bh_kernels.hip.h for the layout and design notes).
//
// the only exchange on the path is the sum over ranks of an n-vector (32 KiB at n = 4096) per J'·t —
// latency-bound, not link-bound.  A ring all-reduce pays 2(N-1) hops for a payload that crosses one xGMI link in 0.2 us;
// here every rank PUSHES its partial straight into all N inboxes (one per rank, each peer-mapped through hipIpc) and
// then sums the N partials it has received IN RANK ORDER: one hop, and bit-identical results on every rank by construction.
// The exchange is fused into the slab reduction that produces the partial (no separate collective launch) and, unlike a
// host-enqueued RCCL call, it sits INSIDE the device-side `state->done` gate: over-launched CG iterations exchange nothing.
//
// Inbox of one rank (fine-grained / uncached device memory, zeroed before the handles are published):
//     slots [2 parities][kMaxPeers][cap]      doubles   partial vectors, written by the peers
//     flags [2 parities][kMaxPeers][nblk_cap] u64       sequence number of the exchange whose block has landed
// Exchanges are numbered by a DEVICE-side counter that only advances when an exchange really runs (the ranks skip the
// same gated launches, so their counters agree).  Two parities suffice: rank r overwrites a parity-p slot in exchange
// e+2 only after its exchange e+1 has completed, i.e. after every peer has pushed e+1 — which a peer does only after its
// own kernel for exchange e (where it read the parity-p slot) has finished (same in-order stream).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bh_reduce.hip.h"

namespace bh {

constexpr int kMaxPeers = 8;
constexpr int kPeerBlockChunks = 16;         // 16-byte chunks per workgroup of the exchange kernel (= reduce_partials_kernel)

struct PeerArgs {
    double* slots[kMaxPeers];                // slots[p]: base of rank p's inbox slots (own rank: the local pointer)
    unsigned long long* flags[kMaxPeers];    // flags[p]: base of rank p's inbox flags
    double* scal[kMaxPeers];                 // scal[p]:  base of rank p's inbox scalars ([2 parities][kMaxPeers][nblk_cap])
    unsigned long long* seq;                 // local: number of exchanges executed so far + 1
    unsigned* arrive;                        // local: workgroups of the running exchange that have finished
    unsigned long long* err;                 // host-mapped: nonzero once an exchange has timed out (written only; the host reads it)
    unsigned* dead;                          // local copy of "an exchange has timed out" (a load from host memory costs a PCIe round trip)
    int rank, nranks;
    int64_t cap;                             // doubles per slot
    int nblk_cap;                            // flags per (parity, rank)
    unsigned long long timeout_ticks;        // wall_clock64 ticks (100 MHz) a workgroup waits for its peers
};

__device__ __forceinline__ void sys_store_f64x2(double* p, double2 v) {
    __hip_atomic_store(p, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double2 sys_load_f64x2(const double* p) {
    double2 v;
    v.x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return v;
}

// out[0 .. 2*nchunks) = sum over ranks (rank order) of this rank's vector, where this rank's vector is
//   G > 0 : the fixed-order sum of the G partial rows written by row_stream_kernel (as reduce_partials_kernel), or
//   G == 0: `out` itself (in-place all-reduce of a vector that already exists: scalars, column-panel results).
// Block = 256 threads = 16 chunks x 16 row-lanes; grid = ceil(nchunks / 16); nchunks <= cap / 2.
__global__ __launch_bounds__(256) void reduce_exchange_kernel(const double* __restrict__ partials, int64_t ld, int nchunks, int G,
                                                              double* out, const CgState* state, PeerArgs pa) {
    if (state != nullptr && state->done) return;
    __shared__ double2 sm[16][17];
    __shared__ int s_timeout;
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * kPeerBlockChunks + cl;
    const bool valid = c < nchunks;
    const int64_t ld2 = ld >> 1;
    const unsigned long long seq = *pa.seq;                 // written by the previous exchange kernel (stream order)
    const int par = (int)(seq & 1ull);
    if (threadIdx.x == 0) s_timeout = 0;

    // ---- this rank's vector --------------------------------------------------------------------------------------------
    double2 acc = make_double2(0.0, 0.0);
    if (G > 0) {
        const double2* P2 = reinterpret_cast<const double2*>(partials);
        const int cc = min(c, nchunks - 1);    // out-of-range threads load a valid chunk and drop the result
        SlabBatch sb;
        sb.issue(P2, ld2, cc, rl, G);
        acc = sb.fold(P2, ld2, cc, rl, G);
        sm[rl][cl] = acc;
        __syncthreads();
        if (rl == 0) {
            double2 t = sm[0][cl];
#pragma unroll
            for (int r = 1; r < 16; ++r) { t.x += sm[r][cl].x; t.y += sm[r][cl].y; }
            sm[16 - 1][cl] = t;       // row 15 is free again: every rl == 0 thread has read its whole column
        }
        __syncthreads();
        acc = sm[15][cl];
    } else if (valid) {
        acc = reinterpret_cast<const double2*>(out)[c];
    }

    // ---- push: thread (rl = peer, cl = chunk) stores this rank's chunk into peer rl's inbox -------------------------------
    const int64_t slot_stride = pa.cap;                                         // doubles per (parity, rank) slot
    const int64_t my_slot = ((int64_t)par * kMaxPeers + pa.rank) * slot_stride;
    // The payload goes out WRITE-THROUGH (system-scope stores: nothing stays dirty in this XCD's L2, so no L2 write-back
    // fence is needed); every storing wave waits until its stores have been acknowledged, the workgroup meets, and only then
    // is the flag raised (cdna_hip_programming.md Guideline 16, form R1).
    if (rl < pa.nranks && valid) sys_store_f64x2(pa.slots[rl] + my_slot + 2 * (int64_t)c, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (rl < pa.nranks && cl == 0) {
        unsigned long long* f = pa.flags[rl] + ((int64_t)par * kMaxPeers + pa.rank) * pa.nblk_cap + blockIdx.x;
        __hip_atomic_store(f, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }

    // ---- wait for block `blockIdx.x` of every rank, then sum in rank order --------------------------------------------------
    if (rl < pa.nranks && cl == 0) {
        const unsigned long long* f = pa.flags[pa.rank] + ((int64_t)par * kMaxPeers + rl) * pa.nblk_cap + blockIdx.x;
        const unsigned long long t0 = wall_clock64();
        // once an exchange has timed out the transport is considered dead: later exchanges do not wait again (the host has
        // been told through *pa.err and fails every call until the peer path is switched off)
        const bool dead = __hip_atomic_load(pa.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        if (dead) s_timeout = 1;
        while (!dead && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > pa.timeout_ticks) {     // a peer never arrived: report, never hang the device
                s_timeout = 1;
                __hip_atomic_store(pa.err, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(pa.dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    // every load of the payload below is a system-scope load into registers (it bypasses this CU's L1 and the XCD's L2), so
    // the flag needs no cache-invalidating acquire behind it; the barrier orders the other waves' loads after the poll
    asm volatile("" ::: "memory");
    __syncthreads();
    double2 got = make_double2(0.0, 0.0);
    if (rl < pa.nranks && valid)
        got = sys_load_f64x2(pa.slots[pa.rank] + ((int64_t)par * kMaxPeers + rl) * slot_stride + 2 * (int64_t)c);
    sm[rl][cl] = got;
    __syncthreads();
    if (rl == 0 && valid) {
        double2 t = sm[0][cl];
        for (int r = 1; r < pa.nranks; ++r) { t.x += sm[r][cl].x; t.y += sm[r][cl].y; }   // rank order: identical bits everywhere
        if (s_timeout) { t.x = __longlong_as_double(0x7ff8000000000000ll); t.y = t.x; }
        reinterpret_cast<double2*>(out)[c] = t;
    }

    // ---- the last workgroup to finish advances the exchange counter ---------------------------------------------------------
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(pa.arrive, 1u) + 1u;
        if (done == gridDim.x) {
            *pa.arrive = 0u;
            *pa.seq = seq + 1ull;
        }
    }
}

}  // namespace bh
