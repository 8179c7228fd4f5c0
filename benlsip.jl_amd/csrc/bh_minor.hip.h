// bh_minor.hip.h — the host-side steps of the reference's minor loop (inner_step, src/basic_tralcnlss.jl:430-458) that sit
// between two projected_cg calls, on the device: active-set identification and growth, reduced-gradient norm, model value.
// Part of the single translation unit of bh_api.hip (see bh_kernels.hip.h for the layout and design notes).
// with these, s, g_minor, w and the active set stay in HBM from the Cauchy step to the end of
// the minor loop; only scalars (counts, norms, status) cross PCIe.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bh_reduce.hip.h"
#include "bh_cg.hip.h"

namespace bh {

// counts[] written by active_update_kernel / canon_mask_kernel
enum { AU_AT_BOUND = 0, AU_NEW = 1, AU_FIXED = 2, AU_BRANCH = 3 };

// Exclusive scan of one int per thread over a 1024-thread workgroup (wave prefix by DPP-free shuffles, then across waves).
__device__ __forceinline__ int block_exclusive_scan_1024(int v, int* total, int* wave_sums /* 16 ints of LDS */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    __syncthreads();                       // wave_sums may still be read from a previous use
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < CG_T / 64; ++w) {
        const int s = wave_sums[w];
        if (w < wave) base += s;
        tot += s;
    }
    *total = tot;
    return base + incl - v;
}

// After a minor iterate — src/basic_tralcnlss.jl:439-453:
//     active_indx = active_bounds(lincons, x, s, delta)          src/polyhedral_constraints.jl:219-237
//     if mA + |active_indx| <= n:  add_active!(lincons, chol_aat, active_indx)      poly:252-261   (branch 0)
//     else:                        active_bounds!(lincons, x+s, chol_aat)           poly:203-215   (branch 1)
// fixflag: the device-side active set (>= 0 fixed, -1 free).  newidx: the variables fixed by THIS call, in index order (the
// Gram downdate sums their columns in that order: bit-reproducible).  One workgroup; each thread owns a contiguous range.
__global__ __launch_bounds__(CG_T) void active_update_kernel(const double* __restrict__ x, const double* __restrict__ s,
                                                             const double* __restrict__ xlow, const double* __restrict__ xupp,
                                                             double delta, double atol, int n, int n_pad, int mA, int* __restrict__ fixflag,
                                                             int* __restrict__ newidx, int* __restrict__ counts) {
    __shared__ int wave_sums[CG_T / 64];
    const int per = (n + CG_T - 1) / CG_T;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    int n_at = 0, n_new = 0;
    for (int i = lo; i < hi; ++i) {
        const double xi = x[i], si = s[i];
        const double sl = fmax(__dsub_rn(xlow[i], xi), -delta);                   // poly:226
        const double su = fmin(__dsub_rn(xupp[i], xi), delta);                    // poly:227
        const bool at = (__dsub_rn(si, sl) <= atol) || (__dsub_rn(su, si) <= atol);   // poly:231
        n_at += at ? 1 : 0;
        n_new += (at && fixflag[i] < 0) ? 1 : 0;
    }
    int tot_at = 0, tot_new = 0;
    (void)block_exclusive_scan_1024(n_at, &tot_at, wave_sums);
    int pos = block_exclusive_scan_1024(n_new, &tot_new, wave_sums);
    const int branch = (mA + tot_at <= n) ? 0 : 1;                                // :441
    int n_fix = 0;
    if (branch == 0) {
        for (int i = lo; i < hi; ++i) {
            const double xi = x[i], si = s[i];
            const double sl = fmax(__dsub_rn(xlow[i], xi), -delta);
            const double su = fmin(__dsub_rn(xupp[i], xi), delta);
            const bool at = (__dsub_rn(si, sl) <= atol) || (__dsub_rn(su, si) <= atol);
            if (at && fixflag[i] < 0) { fixflag[i] = 0; newidx[pos++] = i; }      // poly:258
            n_fix += fixflag[i] >= 0 ? 1 : 0;
        }
    } else {
        for (int i = lo; i < hi; ++i) {
            const double xs = __dadd_rn(x[i], s[i]);                              // active_bounds!(lincons, x+s, ...)  :452
            const bool f = (__dsub_rn(xs, xlow[i]) <= atol) || (__dsub_rn(xupp[i], xs) <= atol);   // poly:211
            fixflag[i] = f ? 0 : -1;
            n_fix += f ? 1 : 0;
        }
        tot_new = 0;
    }
    for (int i = n + (int)threadIdx.x; i < n_pad; i += CG_T) fixflag[i] = -1;
    int tot_fix = 0;
    (void)block_exclusive_scan_1024(n_fix, &tot_fix, wave_sums);
    if (threadIdx.x == 0) { counts[AU_AT_BOUND] = tot_at; counts[AU_NEW] = tot_new; counts[AU_FIXED] = tot_fix; counts[AU_BRANCH] = branch; }
}

// Canonical bookkeeping of a device-side mask: fixrank[i] = rank among the fixed variables (or -1), fixidx[k] = index of the
// k-th fixed variable, counts[AU_FIXED] = their number, chunks = Julia's BitVector.chunks image (bit i%64 of word i/64).
__global__ __launch_bounds__(CG_T) void canon_mask_kernel(int* __restrict__ fixrank, int* __restrict__ fixidx, int n, int n_pad,
                                                          unsigned long long* __restrict__ chunks, int* __restrict__ counts) {
    __shared__ int wave_sums[CG_T / 64];
    const int per = (n + CG_T - 1) / CG_T;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += fixrank[i] >= 0 ? 1 : 0;
    int total = 0;
    int pos = block_exclusive_scan_1024(cnt, &total, wave_sums);
    for (int i = lo; i < hi; ++i)
        if (fixrank[i] >= 0) { fixrank[i] = pos; fixidx[pos] = i; ++pos; }
    for (int i = n + (int)threadIdx.x; i < n_pad; i += CG_T) fixrank[i] = -1;
    __syncthreads();                                           // (global writes of this workgroup are visible to it after the barrier)
    if (chunks != nullptr) {
        const int nwords = (n + 63) >> 6;
        for (int wd = threadIdx.x; wd < nwords; wd += CG_T) {
            unsigned long long bits = 0ull;
            for (int b = 0; b < 64; ++b) {
                const int i = 64 * wd + b;
                if (i < n && fixrank[i] >= 0) bits |= 1ull << b;
            }
            chunks[wd] = bits;
        }
    }
    if (threadIdx.x == 0) counts[AU_FIXED] = total;
}

// fixflag[i] = 0 where bit i of the BitVector image is set (fixed), -1 elsewhere (free, and the padding): the host pushes n/8
// bytes instead of an expanded int array; canon_mask_kernel then numbers the fixed variables.
__global__ __launch_bounds__(256) void flags_from_chunks_kernel(const unsigned long long* __restrict__ chunks, int* __restrict__ fixflag, int n, int n_pad) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_pad; i += gridDim.x * 256)
        fixflag[i] = (i < n && ((chunks[i >> 6] >> (i & 63)) & 1ull)) ? 0 : -1;
}

// M <- M - sum_j a_j a_j', j over the variables that have just become fixed (newidx[0 .. counts[AU_NEW]), index order):
// A_free A_free' after add_active!(indices), without touching the other n - |new| columns of A.
__global__ __launch_bounds__(256) void gram_downdate_list_kernel(double* __restrict__ M, const double* __restrict__ A, int64_t ldA, int mA,
                                                                 const int* __restrict__ newidx, const int* __restrict__ counts) {
    const int n_new = counts[AU_NEW];
    if (n_new <= 0 || counts[AU_BRANCH] != 0) return;
    const int64_t total = (int64_t)mA * mA;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % mA), k = (int)(e / mA);
        if (i < k) continue;
        double m = M[e];
        for (int j = 0; j < n_new; ++j) {
            const int ind = newidx[j];
            m = fma(-A[(int64_t)i * ldA + ind], A[(int64_t)k * ldA + ind], m);
        }
        M[e] = m;
    }
}

// out[0] = ||v||_2 (norm_reduced_gradient's norm(reduced_g), src/basic_tralcnlss.jl:873-874).  One workgroup.
__global__ __launch_bounds__(CG_T) void vec_norm_kernel(const double* __restrict__ v, int n, double* __restrict__ out) {
    __shared__ double scratch[CG_T / 64];
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += CG_T) acc[0] = fma(v[i], v[i], acc[0]);
    block_reduce<CG_T, 1>(acc, scratch, OpSum(), 0.0);
    if (threadIdx.x == 0) out[0] = sqrt(acc[0]);
}

// out[0] = ||mask(v)||_2: the box-constrained projection (fixed components zero) and the norm in one launch.  Same loop and
// reduction as vec_norm_kernel over the masked vector (a masked entry contributes fma(0, 0, acc) = acc): identical bits.
__global__ __launch_bounds__(CG_T) void vec_norm_masked_kernel(const double* __restrict__ v, const int* __restrict__ fixrank, int n,
                                                               double* __restrict__ out) {
    __shared__ double scratch[CG_T / 64];
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += CG_T) {
        const double vi = (fixrank != nullptr && fixrank[i] >= 0) ? 0.0 : v[i];
        acc[0] = fma(vi, vi, acc[0]);
    }
    block_reduce<CG_T, 1>(acc, scratch, OpSum(), 0.0);
    if (threadIdx.x == 0) out[0] = sqrt(acc[0]);
}

// out[0] = dot(a, b).  One workgroup.
__global__ __launch_bounds__(CG_T) void vec_dot_kernel(const double* __restrict__ a, const double* __restrict__ b, int n, double* __restrict__ out) {
    __shared__ double scratch[CG_T / 64];
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += CG_T) acc[0] = fma(a[i], b[i], acc[0]);
    block_reduce<CG_T, 1>(acc, scratch, OpSum(), 0.0);
    if (threadIdx.x == 0) out[0] = acc[0];
}

// model_reduction = g.s + s'Hs / 2 (src/basic_tralcnlss.jl:458) from a g_minor that holds H*s + g:  out[0] = s.(g_minor - g) = s'Hs,
// out[1] = g.s — two dot products instead of a J v sweep.
__global__ __launch_bounds__(CG_T) void model_from_gminor_kernel(const double* __restrict__ g, const double* __restrict__ sv,
                                                                 const double* __restrict__ gm, int n, double* __restrict__ out) {
    __shared__ double scratch[2 * (CG_T / 64)];
    double acc[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < n; i += CG_T) {
        const double si = sv[i], gi = g[i];
        acc[0] = fma(si, __dsub_rn(gm[i], gi), acc[0]);
        acc[1] = fma(gi, si, acc[1]);
    }
    block_reduce<CG_T, 2>(acc, scratch, OpSum(), 0.0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = acc[1]; }
}

}  // namespace bh
