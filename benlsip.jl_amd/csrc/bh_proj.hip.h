// bh_proj.hip.h — null-space projection: left_mul / left_mul_tr, Gram (VALU and fp64 MFMA), Cholesky (small, blocked, downdate), triangular solves
// Part of the single translation unit of bh_api.hip (see bh_kernels.hip.h for the layout and design notes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "bh_reduce.hip.h"
#include "bh_cg.hip.h"

namespace bh {

// ------------------------------------------------------------------------------------------
// Projection kernels (src/polyhedral_constraints.jl:72-136).
// ------------------------------------------------------------------------------------------
struct ProjArgs {
    const double* A;        // row-major mA x ldA image of lineq (NULL when mA == 0)
    int64_t ldA;
    int mA, n, nfix, mpp;   // mpp = order of the factor in use (mA + nfix augmented, mA reduced)
    const int* fixrank;     // n   (-1 free)
    const int* fixidx;      // nfix
    const double* L;        // mpp x mpp column-major, lower triangle valid
    double* tw;             // mpp workspace
    const CgState* state;   // NULL, or skip unless (!done && need_proj)
    int reduced;            // 1: reduced form  v_free = r_free - A_free'(A_free A_free')^{-1} A_free r_free, v_fix = 0
    // four-kernel general-constraint CG iteration (bh_cgfuse.hip.h):
    const double* tpart;    // trsv_small_kernel: tw[i] = sum over tpart_nblk workgroups of tpart[b*mA + i] (A_free r partials) first
    int tpart_nblk;
    double* rvpart;         // proj_left_mul_tr_kernel: this workgroup's partial of r.v
    int fused_j;            // > 0: launch of iteration fused_j — skipped iff the loop stopped at or before it (CgState::stop_at)
    // three-kernel general-constraint CG iteration (proj_apply_linv_kernel):
    const double* W;        // [Linv | Linv'] of tri_inv_small_kernel (2 x 64 x 64, zero outside the triangle and beyond mA)
    double* vvpart;         // INIT: this workgroup's partial of v.v (tol_cg = kappa2*||v||, :710)
    double* p_out;          // INIT: p_1 = -v (:708)
    int nch_pad;            // chunks per padded vector (ld / 2): the kernel keeps [n, ld) of its outputs at zero
    const double* Mgram;    // A_free A_free' (lower triangle, column-major mA x mA) for one step of iterative refinement of y; NULL: none
};

__device__ __forceinline__ bool proj_skip(const CgState* st) { return st != nullptr && (st->done || !st->need_proj); }
__device__ __forceinline__ bool proj_skip(const ProjArgs& a) {
    if (a.fused_j > 0) return a.state->stop_at != 0 && a.fused_j >= a.state->stop_at;
    return proj_skip(a.state);
}

// Box-only projection: v = fixed ? 0 : r.
__global__ __launch_bounds__(256) void proj_mask_kernel(const double* __restrict__ r, double* __restrict__ v, const int* fixrank, int n,
                                                        const CgState* st) {
    if (proj_skip(st)) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        v[i] = (fixrank != nullptr && fixrank[i] >= 0) ? 0.0 : r[i];
}

// One thread's share of a row of A times x.  Eight chunks per thread and batch: all their loads go out together (clamped
// indices; MASK is a template parameter so that no branch stands between a load and the next one), the fmas follow in chunk
// order — the row is 8 chunks per thread at n = 4096, i.e. one memory round trip instead of eight.
template <bool MASK>
__device__ __forceinline__ double left_mul_row(const double2* __restrict__ rp, const double2* __restrict__ x2, const int2* __restrict__ f2, int nch) {
    double acc = 0.0;
    for (int c0 = threadIdx.x; c0 < nch; c0 += 8 * 256) {
        double2 av[8], xv[8];
        int2 f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = min(c0 + 256 * k, nch - 1);
            av[k] = rp[c];
            xv[k] = x2[c];
            f[k] = MASK ? f2[c] : make_int2(-1, -1);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (c0 + 256 * k < nch) {
                const double x0 = MASK ? keep_if_free(xv[k].x, f[k].x) : xv[k].x, x1 = MASK ? keep_if_free(xv[k].y, f[k].y) : xv[k].y;
                acc = fma(av[k].x, x0, acc);
                acc = fma(av[k].y, x1, acc);
            }
        }
    }
    return acc;
}

// left_mul: tw[0:mA] = A x (one workgroup per row: 4 waves share the row, fixed-order combine), and in the augmented
// form tw[mA+k] = x[fixidx[k]] (:86-98).  Reduced form: the fixed components of x are masked out (A_free x_free).
// grid = mA + ceil(nfix/256) blocks of 256 (gather blocks only in the augmented form).
__device__ __forceinline__ void proj_left_mul_body(const ProjArgs& a, const double* __restrict__ x, int block) {
    __shared__ double scratch[4];
    if (block < a.mA) {
        const int row = block;
        const double2* rp = reinterpret_cast<const double2*>(a.A + (int64_t)row * a.ldA);
        const double2* x2 = reinterpret_cast<const double2*>(x);
        const int2* f2 = reinterpret_cast<const int2*>(a.fixrank);
        const int nch = (int)(a.ldA >> 1);
        const bool mask = a.reduced && a.fixrank != nullptr;
        double acc[1] = {0.0};
        acc[0] = mask ? left_mul_row<true>(rp, x2, f2, nch) : left_mul_row<false>(rp, x2, f2, nch);
        block_reduce<256, 1>(acc, scratch, OpSum(), 0.0);
        if (threadIdx.x == 0) a.tw[row] = acc[0];
    } else if (!a.reduced) {
        const int k = (block - a.mA) * 256 + threadIdx.x;
        if (k < a.nfix) a.tw[a.mA + k] = x[a.fixidx[k]];
    }
}
__global__ __launch_bounds__(256) void proj_left_mul_kernel(ProjArgs a, const double* __restrict__ x) {
    if (proj_skip(a.state)) return;
    proj_left_mul_body(a, x, (int)blockIdx.x);
}

// out = r - left_mul_tr(tw)   (:72-84, :116, :134);  with SUBTRACT=false: out = left_mul_tr(tw).
// Block = 64 chunks x RG row groups (rows i = rg, rg+RG, ...), combined through LDS in fixed order; grid = ceil(nch/64).
// RG = 4 (256 threads) up to 64 rows, RG = 16 (1024 threads) above: the grid is only nch/64 workgroups, so many rows per
// thread serialise (33.8 us at mA = 512 with RG = 4).
template <bool SUBTRACT, int RG>
__device__ __forceinline__ void proj_left_mul_tr_body(const ProjArgs& a, const double* __restrict__ r, double* __restrict__ out, int block) {
    __shared__ double2 sm[RG][64];
    const int nch = (a.n + 1) >> 1;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = block * 64 + cl;
    double2 acc = make_double2(0.0, 0.0);
    if (c < nch) {
        const double2* A2 = reinterpret_cast<const double2*>(a.A);
        const int64_t ld2 = a.ldA >> 1;
        // sixteen rows per batch in flight (all of them for mA <= 64), fmas in row order
        for (int i0 = rg; i0 < a.mA; i0 += 16 * RG) {
            double wi[16];
            double2 av[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = min(i0 + RG * k, a.mA - 1);
                wi[k] = a.tw[i];
                av[k] = A2[(int64_t)i * ld2 + c];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (i0 + RG * k < a.mA) {
                    acc.x = fma(wi[k], av[k].x, acc.x);
                    acc.y = fma(wi[k], av[k].y, acc.y);
                }
            }
        }
    }
    sm[rg][cl] = acc;
    __syncthreads();
    if (rg != 0) return;
    if (SUBTRACT && a.rvpart != nullptr && a.reduced) {
        // four-kernel CG iteration: v = r - A_free'y on the free variables, and this workgroup's share of r.v (:743) — the whole
        // of wave 0 stays together for the reduction
        double2 t = make_double2(0.0, 0.0);
#pragma unroll
        for (int g4 = 0; g4 < RG; g4 += 4) {
            const double qx = (sm[g4][cl].x + sm[g4 + 1][cl].x) + (sm[g4 + 2][cl].x + sm[g4 + 3][cl].x);
            const double qy = (sm[g4][cl].y + sm[g4 + 1][cl].y) + (sm[g4 + 2][cl].y + sm[g4 + 3][cl].y);
            t.x = (g4 == 0) ? qx : t.x + qx;
            t.y = (g4 == 0) ? qy : t.y + qy;
        }
        double rv = 0.0;
        if (c < nch) {
            const int j0 = 2 * c, j1 = 2 * c + 1;
            int k0 = -1, k1 = -1;
            if (a.fixrank != nullptr) { k0 = a.fixrank[j0]; if (j1 < a.n) k1 = a.fixrank[j1]; }
            const double r0 = r[j0], r1 = (j1 < a.n) ? r[j1] : 0.0;
            const double v0 = (k0 >= 0) ? 0.0 : r0 - t.x;
            const double v1 = (k1 >= 0 || j1 >= a.n) ? 0.0 : r1 - t.y;
            out[j0] = v0;
            if (j1 < a.n) out[j1] = v1;
            rv = fma(r1, v1, r0 * v0);
        }
        rv = wave_sum(rv);
        if (cl == 0) a.rvpart[block] = rv;
        return;
    }
    if (c >= nch) return;
    acc = make_double2(0.0, 0.0);
#pragma unroll
    for (int g4 = 0; g4 < RG; g4 += 4) {         // groups of four, each combined as (0+1)+(2+3), then added in order
        const double qx = (sm[g4][cl].x + sm[g4 + 1][cl].x) + (sm[g4 + 2][cl].x + sm[g4 + 3][cl].x);
        const double qy = (sm[g4][cl].y + sm[g4 + 1][cl].y) + (sm[g4 + 2][cl].y + sm[g4 + 3][cl].y);
        acc.x = (g4 == 0) ? qx : acc.x + qx;
        acc.y = (g4 == 0) ? qy : acc.y + qy;
    }
    const int j0 = 2 * c, j1 = 2 * c + 1;
    int k0 = -1, k1 = -1;
    if (a.fixrank != nullptr) { k0 = a.fixrank[j0]; if (j1 < a.n) k1 = a.fixrank[j1]; }
    if (a.reduced) {
        // fixed components of the projection are exactly zero
        if (SUBTRACT) {
            out[j0] = (k0 >= 0) ? 0.0 : r[j0] - acc.x;
            if (j1 < a.n) out[j1] = (k1 >= 0) ? 0.0 : r[j1] - acc.y;
        } else {
            out[j0] = (k0 >= 0) ? 0.0 : acc.x;
            if (j1 < a.n) out[j1] = (k1 >= 0) ? 0.0 : acc.y;
        }
        return;
    }
    if (k0 >= 0) acc.x += a.tw[a.mA + k0];
    if (k1 >= 0) acc.y += a.tw[a.mA + k1];
    if (SUBTRACT) {
        out[j0] = r[j0] - acc.x;
        if (j1 < a.n) out[j1] = r[j1] - acc.y;
    } else {
        out[j0] = acc.x;
        if (j1 < a.n) out[j1] = acc.y;
    }
}
template <bool SUBTRACT, int RG>
__global__ __launch_bounds__(64 * RG) void proj_left_mul_tr_kernel(ProjArgs a, const double* __restrict__ r, double* __restrict__ out) {
    if (proj_skip(a)) return;
    proj_left_mul_tr_body<SUBTRACT, RG>(a, r, out, (int)blockIdx.x);
}

// Reduced-form factor, built on the device whenever the active set changes (bh_proj_set_active):
//   M = A_free A_free'  (lower triangle, column-major mA x mA): one wave per entry (i >= k).
__global__ __launch_bounds__(256) void gram_free_kernel(const double* __restrict__ A, int64_t ldA, int mA, const int* __restrict__ fixrank,
                                                        double* __restrict__ M) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);     // packed lower-triangular index
    const int64_t total = (int64_t)mA * (mA + 1) / 2;
    if (e >= total) return;
    // e = i*(i+1)/2 + k, 0 <= k <= i
    int i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while ((int64_t)(i + 1) * (i + 2) / 2 <= e) ++i;
    while ((int64_t)i * (i + 1) / 2 > e) --i;
    const int k = (int)(e - (int64_t)i * (i + 1) / 2);
    const double2* ri = reinterpret_cast<const double2*>(A + (int64_t)i * ldA);
    const double2* rk = reinterpret_cast<const double2*>(A + (int64_t)k * ldA);
    const int2* f2 = reinterpret_cast<const int2*>(fixrank);
    const int nch = (int)(ldA >> 1);
    double acc = 0.0;
    for (int c = lane; c < nch; c += 64) {
        const double2 x = ri[c], y = rk[c];
        int2 f = make_int2(-1, -1);
        if (fixrank != nullptr) f = f2[c];
        if (f.x < 0) acc = fma(x.x, y.x, acc);
        if (f.y < 0) acc = fma(x.y, y.y, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) M[i + (int64_t)k * mA] = acc;
}

// M = A_free A_free' on the matrix cores: the one genuinely GEMM-shaped product around the hot path (mA x n x mA, fp64).
// One workgroup (16 waves) per 16 x 16 lower tile; v_mfma_f64_16x16x4_f64 with A_op[i][k] = Af[16 ti + i][c + k],
// B_op[k][j] = Af[16 tk + j][c + k]  (lane l holds i or j = l & 15 and k = l >> 4; C/D: col = l & 15, row = (l >> 4) + 4 reg).
// The k index is permuted so that lane group l >> 4 owns 4 CONSECUTIVE columns per 16-column super-step (one 32-byte load
// per lane and operand, 128 contiguous bytes per matrix row); the 16 waves split the super-steps and are combined through
// LDS in fixed order (bit-reproducible).  Fixed variables are masked out of the A operand (A_free = A with those columns 0).
typedef double dvec4 __attribute__((ext_vector_type(4)));
constexpr int GRAM_T = 1024;     // 16 waves split the k range of one tile (a tile has only mA-independent work: n/16 super-steps)
__global__ __launch_bounds__(GRAM_T) void gram_free_mfma_kernel(const double* __restrict__ A, int64_t ldA, int mA,
                                                                const int* __restrict__ fixrank, double* __restrict__ M) {
    constexpr int NW = GRAM_T / 64;
    __shared__ double red[NW][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // packed lower-triangular tile index -> (ti, tk), ti >= tk
    const int e = blockIdx.x;
    int ti = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= e) ++ti;
    while (ti * (ti + 1) / 2 > e) --ti;
    const int tk = e - ti * (ti + 1) / 2;
    const int ri = 16 * ti + (lane & 15), rk = 16 * tk + (lane & 15), kq = lane >> 4;
    const bool vi = ri < mA, vk = rk < mA;
    const double* pa = A + (int64_t)(vi ? ri : 0) * ldA + 4 * kq;
    const double* pb = A + (int64_t)(vk ? rk : 0) * ldA + 4 * kq;
    const int nsuper = (int)(ldA >> 4);          // 16 columns per super-step (ldA is a multiple of 16)
    dvec4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};   // two independent accumulation chains
#pragma unroll 4
    for (int sidx = wave; sidx < nsuper; sidx += NW) {
        const int64_t c = (int64_t)sidx * 16;
        double2 a01 = make_double2(0.0, 0.0), a23 = a01, b01 = a01, b23 = a01;
        if (vi) { a01 = *reinterpret_cast<const double2*>(pa + c); a23 = *reinterpret_cast<const double2*>(pa + c + 2); }
        if (vk) { b01 = *reinterpret_cast<const double2*>(pb + c); b23 = *reinterpret_cast<const double2*>(pb + c + 2); }
        if (fixrank != nullptr) {
            const int4 f = *reinterpret_cast<const int4*>(fixrank + c + 4 * kq);
            if (f.x >= 0) a01.x = 0.0;
            if (f.y >= 0) a01.y = 0.0;
            if (f.z >= 0) a23.x = 0.0;
            if (f.w >= 0) a23.y = 0.0;
        }
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.x, b01.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.y, b01.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.x, b23.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.y, b23.y, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc0[r] + acc1[r];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w][lane][r];
        const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tk + (lane & 15);
        if (row < mA && col <= row) M[row + (int64_t)col * mA] = t;
    }
}

// In-place lower Cholesky of the mA x mA matrix M (column-major, lower triangle), single workgroup, right-looking.
// info[0] = 0 on success, else 1 + index of the first non-positive pivot (the reference's PosDefException).
__global__ __launch_bounds__(CG_T) void chol_lower_kernel(const double* __restrict__ Msrc, double* __restrict__ M, int m, int* info,
                                                          const CgState* gate) {
    if (gate != nullptr && gate->done) return;
    __shared__ double s_piv;
    const int tid = threadIdx.x;
    if (tid == 0) info[0] = 0;
    for (int64_t e = tid; e < (int64_t)m * m; e += CG_T) M[e] = Msrc[e];
    __syncthreads();
    for (int j = 0; j < m; ++j) {
        if (tid == 0) {
            const double d = M[j + (int64_t)j * m];
            if (!(d > 0.0) && info[0] == 0) info[0] = j + 1;
            s_piv = sqrt(d);
        }
        __syncthreads();
        const double piv = s_piv;
        for (int i = j + tid; i < m; i += CG_T) M[i + (int64_t)j * m] = (i == j) ? piv : M[i + (int64_t)j * m] / piv;
        __syncthreads();
        // trailing update of the lower triangle: M[i][k] -= L[i][j]*L[k][j], j < k <= i.  32 x 32 thread tiles over the
        // lower triangle (tx along i: coalesced in the column-major matrix; no integer division per element).
        const int rem = m - j - 1;
        const int tx = tid & 31, ty = tid >> 5;
        const double* colj = M + (int64_t)j * m + (j + 1);
        for (int kb = 0; kb < rem; kb += 32) {
            const int kk = kb + ty;
            const double lkj = (kk < rem) ? colj[kk] : 0.0;
            for (int ib = kb; ib < rem; ib += 32) {
                const int ii = ib + tx;
                if (ii < rem && kk < rem && ii >= kk) {
                    double* e = M + (int64_t)(j + 1 + kk) * m + (j + 1 + ii);
                    *e = fma(-colj[ii], lkj, *e);
                }
            }
        }
        __syncthreads();
    }
}

// Reduced-form factor for mA <= 64: right-looking Cholesky on 256 threads.  lane = row, wave w owns the 16-column panel
// [16w, 16w+16) of that row in REGISTERS (statically indexed: the column loop is unrolled per panel); each step the
// owning wave publishes column j through a double-buffered 64-entry LDS vector (one barrier per step) and every wave
// applies the rank-one update to its panel.  (History: fully unrolled one-wave register version 100 us, instruction-fetch
// bound; one-wave LDS loops 66-170 us, latency bound.)  Writes L (lower, column-major m x m) and the reciprocal diagonal
// dinv[m] right after the matrix (dst + m*m), which turns the substitutions' divisions into multiplications.
//   Strided form: src/dst are the top-left corners of an nb x nb (nb <= 64) block inside matrices with leading dimensions
//   ld_src / ld_dst (in-place allowed); dinv_out receives the reciprocal diagonal; pivot failures are reported as
//   info_base + column + 1.  reset_info: write 0 on success (stand-alone use) — blocked/in-loop callers only ever raise it.
// down_A != NULL: the Gram matrix is first downdated by column `down_ind` of A (the variable that has just become fixed:
// A_free A_free' after add_active!, M <- M - a a' on the lower triangle, written back to Msrc) — gram_downdate_kernel's arithmetic,
// without a launch of its own.
__device__ __forceinline__ void chol_small_body(double* Msrc, int64_t ld_src, double* M, int64_t ld_dst, int m, double* dinv_out, int* info,
                                                int info_base, int reset_info, const double* __restrict__ down_A, int64_t down_ldA,
                                                int down_ind) {
    __shared__ __attribute__((aligned(16))) double colbuf[2][4][64];
    __shared__ double s_down[64];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool down = down_A != nullptr && down_ind >= 0;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int k = 16 * wave + c;
        a[c] = (lane < m && k < m && k <= lane) ? Msrc[lane + (int64_t)k * ld_src] : 0.0;
    }
    if (down && wave == 0) s_down[lane] = (lane < m) ? down_A[(int64_t)lane * down_ldA + down_ind] : 0.0;
    if (tid == 0) s_bad = 0;
    double dinv_mine = 0.0;
    int buf = 0;
    __syncthreads();
    if (down) {
        const double ai = s_down[lane];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int k = 16 * wave + c;
            if (lane < m && k < m && k <= lane) {
                a[c] = fma(-ai, s_down[k], a[c]);
                Msrc[lane + (int64_t)k * ld_src] = a[c];
            }
        }
    }
    // FOUR columns per workgroup barrier: the wave that owns columns j0 .. j0+3 factors them one after the other — each column's
    // rank-one update goes into its own panel at once (wave-synchronous: no barrier inside a wave) — and publishes all four;
    // the waves to the right then apply the four updates in column order.  Every entry still receives its updates in
    // increasing j, so L is the same bits as with one barrier per column; 16 barriers instead of 64 (37 -> ~20 us at m = 64).
    for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int j0 = 16 * p + 4 * blk;
            if (j0 < m) {                                   // uniform
                if (wave == p) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int jj = 4 * blk + b, j = j0 + b;
                        if (j < m) {                        // uniform
                            const double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[jj]), j),
                                                                __builtin_amdgcn_readlane(__double2loint(a[jj]), j));
                            // one reciprocal square root instead of sqrt + 64 divisions: the dependent fp64 chain per step is the
                            // cost.  v_rsq_f64 seed + two Newton steps (y <- y(1.5 - 0.5 x y^2)): full fp64 accuracy for the normal,
                            // positive pivots of an SPD matrix without the library rsqrt's range handling
                            double rinv = __builtin_amdgcn_rsq(piv);
                            rinv = rinv * fma(-0.5 * piv * rinv, rinv, 1.5);
                            rinv = rinv * fma(-0.5 * piv * rinv, rinv, 1.5);
                            if (!(piv > 0.0) && lane == 0 && s_bad == 0) s_bad = j + 1;
                            double lij = 0.0;
                            if (lane == j) { lij = piv * rinv; dinv_mine = rinv; }
                            else if (lane > j) lij = a[jj] * rinv;
                            a[jj] = lij;
                            colbuf[buf][b][lane] = lij;
                            // own panel, right away (this wave reads back what it has just written: LDS is in order within a wave)
                            // Only the columns right of j (static: c > jj), and no row mask: an entry above the diagonal (row <
                            // column) picks up garbage that nothing reads — a column's own step zeroes the lanes above its pivot and
                            // only k <= lane is stored.  The single wave working here is what the other three wait for: its
                            // instruction count per column is the critical path of the kernel.
                            const double* cb = &colbuf[buf][b][16 * wave];
#pragma unroll
                            for (int c = 0; c < 16; ++c)
                                if (c > jj) a[c] = fma(-lij, cb[c], a[c]);
                        }
                    }
                }
                __syncthreads();
                if (wave > p) {                             // panels to the right of the four columns (wave-uniform)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (j0 + b < m) {                   // uniform
                            const double lij = colbuf[buf][b][lane];
                            const double* cb = &colbuf[buf][b][16 * wave];     // the 16 l_kj of this panel: contiguous, broadcast reads
#pragma unroll
                            for (int c = 0; c < 16; ++c) a[c] = fma(-lij, cb[c], a[c]);      // (entries above the diagonal: never read)
                        }
                    }
                }
                buf ^= 1;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int k = 16 * wave + c;
        if (lane < m && k < m && k <= lane) M[lane + (int64_t)k * ld_dst] = a[c];
    }
    if (dinv_out != nullptr && lane < m && wave == (lane >> 4)) dinv_out[lane] = dinv_mine;
    __syncthreads();
    if (tid == 0) {
        if (s_bad != 0) info[0] = info_base + s_bad;
        else if (reset_info) info[0] = 0;
    }
}
__global__ __launch_bounds__(256) void chol_small_kernel(const double* Msrc, int64_t ld_src, double* M, int64_t ld_dst, int m,
                                                         double* dinv_out, int* info, int info_base, int reset_info,
                                                         const CgState* gate) {
    if (gate != nullptr && gate->done) return;
    chol_small_body(const_cast<double*>(Msrc), ld_src, M, ld_dst, m, dinv_out, info, info_base, reset_info, nullptr, 0, -1);
}
// ---- blocked Cholesky for m > 64: potrf (chol_small_kernel on the 64 x 64 diagonal block) / trsm / syrk per panel ------
// Copy the lower triangle (gate-aware) so that the factorisation can run in place on dst.
__global__ __launch_bounds__(256) void copy_lower_kernel(const double* __restrict__ src, double* __restrict__ dst, int m, int* info,
                                                         const CgState* gate) {
    if (gate != nullptr && gate->done) return;
    const int64_t total = (int64_t)m * m;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % m), k = (int)(e / m);
        if (i >= k) dst[e] = src[e];
    }
    if (gate == nullptr && blockIdx.x == 0 && threadIdx.x == 0) info[0] = 0;
}

// L21 <- A21 L11^{-T}: rows [k0+nb, m) of the panel [k0, k0+nb).  One thread per row; the row's entries live in a
// lane-private LDS column (xs[c][tid], conflict-free), L11 (lower, row-contiguous copy) and its reciprocal diagonal in LDS
// as wave-wide broadcasts: no global traffic inside the dependent chain (the first version re-read the row's earlier
// columns from global memory at every step: 77 us per panel).  A register-resident row needs the 2016-step substitution
// fully unrolled, which the compiler turned into a 15 KiB scratch array instead.
// Dynamic LDS: xs[64][TRSM_T] | l11[64][64] | di[64].
constexpr int TRSM_T = 128;
constexpr size_t kTrsmLdsBytes = (size_t)(64 * TRSM_T + 64 * 64 + 64) * sizeof(double);
__global__ __launch_bounds__(TRSM_T) void chol_trsm_kernel(double* __restrict__ M, int m, int k0, int nb, const double* __restrict__ dinv,
                                                           const CgState* gate) {
    if (gate != nullptr && gate->done) return;
    extern __shared__ __attribute__((aligned(16))) double trsm_smem[];
    double* xs = trsm_smem;                       // [64][TRSM_T]
    double* l11 = trsm_smem + 64 * TRSM_T;        // [64][64], row j = L11[j, :]
    double* di = l11 + 64 * 64;
    const int tid = threadIdx.x;
    for (int e = tid; e < 64 * 64; e += TRSM_T) {
        const int i = e & 63, k = e >> 6;         // consecutive threads read down a column: coalesced; zero outside the block
        l11[i * 64 + k] = (i < nb && k < nb && i > k) ? M[(k0 + i) + (int64_t)(k0 + k) * m] : 0.0;   // STRICTLY lower: the
    }                                                                                                    // diagonal enters through di
    if (tid < nb) di[tid] = dinv[tid];
    const int r = k0 + nb + blockIdx.x * TRSM_T + tid;
    const bool live = r < m;
    if (live) {
        // 16 independent loads in flight per step (a rolled loop waits for each load before its LDS store)
        for (int j0 = 0; j0 < nb; j0 += 16) {
            double t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = (j0 + u < nb) ? M[r + (int64_t)(k0 + j0 + u) * m] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) xs[(j0 + u) * TRSM_T + tid] = t[u];
        }
    }
    __syncthreads();
    if (!live) return;
    // Column j of the row: x_j = (a_j - sum_{c<j} x_c L11[j,c]) / L11[j,j].  The sum runs over whole blocks of 8 columns
    // (the LDS copy of L11 is strictly lower, so the padding terms c >= j add exact zeros), two independent accumulators, and the next block's
    // 16 LDS operands are fetched while the current block is accumulated.
    for (int j = 0; j < nb; ++j) {
        const double* lj = l11 + j * 64;
        const int nblk = (j + 7) >> 3;
        double a0 = xs[j * TRSM_T + tid], a1 = 0.0;
        double xv[8], lv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { xv[u] = xs[u * TRSM_T + tid]; lv[u] = lj[u]; }
        for (int b = 0; b < nblk; ++b) {
            double xn[8], ln[8];
            const int cn = (b + 1 < nblk) ? 8 * (b + 1) : 0;        // the last prefetch re-reads block 0 (harmless, unused)
#pragma unroll
            for (int u = 0; u < 8; ++u) { xn[u] = xs[(cn + u) * TRSM_T + tid]; ln[u] = lj[cn + u]; }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                a0 = fma(-xv[u], lv[u], a0);
                a1 = fma(-xv[u + 1], lv[u + 1], a1);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { xv[u] = xn[u]; lv[u] = ln[u]; }
        }
        const double acc = (a0 + a1) * di[j];
        xs[j * TRSM_T + tid] = acc;
        M[r + (int64_t)(k0 + j) * m] = acc;
    }
}

// A22 <- A22 - L21 L21' (lower triangle only): 16 x 16 thread tiles, each thread one element, nb-long dot product of two
// rows of the panel (column-major: consecutive threads along i read consecutive addresses).
__global__ __launch_bounds__(256) void chol_syrk_kernel(double* __restrict__ M, int m, int k0, int nb, const CgState* gate) {
    if (gate != nullptr && gate->done) return;
    const int base = k0 + nb;
    const int i = base + blockIdx.x * 16 + (threadIdx.x & 15);
    const int k = base + blockIdx.y * 16 + (threadIdx.x >> 4);
    if (blockIdx.y > blockIdx.x || i >= m || k >= m || i < k) return;
    double acc = 0.0;
#pragma unroll 8
    for (int c = 0; c < nb; ++c) acc = fma(M[i + (int64_t)(k0 + c) * m], M[k + (int64_t)(k0 + c) * m], acc);
    M[i + (int64_t)k * m] -= acc;
}

// Rank-one Cholesky DOWNDATE: L L' <- L L' - a a' with a = column `ind` (state->status) of A — what add_active! does to
// A_free A_free' when one more variable becomes fixed.  O(m^2) instead of refactoring (O(m^3)); hyperbolic rotations:
// s = a_k / l_kk, c = sqrt(1 - s^2), l_kk <- c l_kk, l_ik <- (l_ik - s a_i)/c, a_i <- c a_i - s l_ik.  A non-positive 1 - s^2
// (the downdated matrix is no longer positive definite) raises info.  Used for m > 64 only (option chol_downdate = 1): up to
// 64 rows refactoring the downdated Gram matrix costs the same and does not accumulate error (round 3: on the pinned operands of
// tests/golden/cauchy_events.json, mA = 2 and A_free A_free' close to singular, 40 downdates in a row left a factor whose
// projections were no longer in null(A); the one-wave register kernel for m <= 64 went with that finding).
// One workgroup, L in global memory (column k is contiguous), a in LDS.
__global__ __launch_bounds__(CG_T) void chol_downdate_kernel(double* __restrict__ L, const double* __restrict__ A, int64_t ldA, int m,
                                                             int* info, const CgState* st) {
    if (st->done) return;
    const int ind = st->status;
    if (ind < 0) return;
    extern __shared__ __attribute__((aligned(16))) double a_sh[];      // m doubles
    __shared__ double s_c, s_s, s_rc;
    const int tid = threadIdx.x;
    for (int i = tid; i < m; i += CG_T) a_sh[i] = A[(int64_t)i * ldA + ind];
    __syncthreads();
    for (int k = 0; k < m; ++k) {
        double* colk = L + (int64_t)k * m;
        if (tid == 0) {
            const double lkk = colk[k];
            const double sn = a_sh[k] / lkk;
            const double t = fma(-sn, sn, 1.0);
            if (!(t > 0.0) && info[0] == 0) info[0] = k + 1;
            const double c = sqrt(t);
            colk[k] = c * lkk;
            s_c = c; s_s = sn; s_rc = 1.0 / c;
        }
        __syncthreads();
        const double c = s_c, sn = s_s, rc = s_rc;
        for (int i = k + 1 + tid; i < m; i += CG_T) {
            const double lik = (colk[i] - sn * a_sh[i]) * rc;
            a_sh[i] = fma(c, a_sh[i], -sn * lik);
            colk[i] = lik;
        }
        __syncthreads();
    }
}

// tw <- L' \ (L \ tw) for m <= 64 (reduced form).  256 threads stage L into an LDS tile (all loads in flight at once,
// row stride 65: conflict-free both row- and column-wise); wave 0 then runs the 2 x m dependent steps
// (readlane + LDS read + fma) with the reciprocal diagonal from chol_small_kernel.
__device__ __forceinline__ void trsv_small_body(const ProjArgs& a) {
    __shared__ double t[64 * 65];
    __shared__ double tq[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = a.mpp;
    if (a.tpart != nullptr) {
        // right-hand side from the per-workgroup partials of A_free r: wave q folds a quarter of the workgroups in order
        const int per = (a.tpart_nblk + 3) / 4;
        const int b0 = wave * per, b1 = min(a.tpart_nblk, b0 + per);
        double acc = 0.0;
        if (lane < m) {
            int b = b0;
            for (; b + 8 <= b1; b += 8) {           // eight loads in flight, then the adds in index order
                double x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = a.tpart[(int64_t)(b + k) * m + lane];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += x[k];
            }
            for (; b < b1; ++b) acc += a.tpart[(int64_t)b * m + lane];
        }
        tq[wave][lane] = acc;
    }
    const double* __restrict__ L = a.L;
    // (the diagonal and, without partials, the right-hand side are asked for here with the tile, not after the barrier)
    const double di_early = (lane < m) ? L[(int64_t)m * m + lane] : 0.0;
    const double tw_early = (a.tpart == nullptr && lane < m) ? a.tw[lane] : 0.0;
    double tmp[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int k = 16 * wave + c;
        tmp[c] = (lane < m && k < m && k <= lane) ? L[lane + (int64_t)k * m] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) t[lane * 65 + 16 * wave + c] = tmp[c];
    __syncthreads();
    if (wave != 0) return;
    const double di = di_early;
    double xi = 0.0;
    if (lane < m) xi = (a.tpart != nullptr) ? ((tq[0][lane] + tq[1][lane]) + (tq[2][lane] + tq[3][lane])) : tw_early;
#pragma unroll 8
    for (int j = 0; j < m; ++j) {                           // forward: L y = t
        const double lij = t[lane * 65 + j];
        if (lane == j) xi = xi * di;
        const double xj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xi), j),
                                           __builtin_amdgcn_readlane(__double2loint(xi), j));
        if (lane > j) xi = fma(-lij, xj, xi);
    }
#pragma unroll 8
    for (int j = m - 1; j >= 0; --j) {                      // backward: L' w = y   (L[j][i] = t[j*65 + i], consecutive lanes)
        const double lji = t[j * 65 + lane];
        if (lane == j) xi = xi * di;
        const double xj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xi), j),
                                           __builtin_amdgcn_readlane(__double2loint(xi), j));
        if (lane < j) xi = fma(-lji, xj, xi);
    }
    if (lane < m) a.tw[lane] = xi;
}
__global__ __launch_bounds__(256) void trsv_small_kernel(ProjArgs a) {
    if (proj_skip(a)) return;
    trsv_small_body(a);
}
// A Cauchy pass with linear equalities (mA <= 64, reduced form), everything that one workgroup does between two decisions, in ONE
// launch: the Gram matrix downdated by the column of the variable fixed at the last breakpoint and refactored from scratch (as the
// reference rebuilds its factor at every breakpoint, :631), the right-hand side t = A_free(-g) — the value left_mul formed for the
// previous active set (t_fresh, computed next to the row kernel of the previous pass) minus the one column that left it, so every t
// carries exactly one update on top of a fresh sum —, and the two triangular solves: y in a.tw.
__global__ __launch_bounds__(256) void cauchy_factor_solve_kernel(double* Mgram, double* L, int m, int* info, ProjArgs a,
                                                                 const double* __restrict__ r, const double* __restrict__ t_fresh,
                                                                 const CgState* st) {
    if (st->done) return;
    const int ind = st->status;
    double t_mine = 0.0;
    if ((int)threadIdx.x < m) t_mine = fma(-a.A[(int64_t)threadIdx.x * a.ldA + ind], r[ind], t_fresh[threadIdx.x]);
    chol_small_body(Mgram, m, L, m, m, L + (int64_t)m * m, info, 0, 0, a.A, a.ldA, ind);
    if ((int)threadIdx.x < m) a.tw[threadIdx.x] = t_mine;
    __threadfence_block();
    __syncthreads();                                       // L, its reciprocal diagonal and t are in memory for the whole workgroup
    trsv_small_body(a);
}

// Explicit inverse of the reduced-form factor, m <= 64:  W[0 .. 4096) = Linv column-major (W[k*64 + i] = Linv[i][k]),
// W[4096 .. 8192) = Linv' column-major (W[4096 + i*64 + k] = Linv[i][k]); zero above the diagonal and beyond m.  With it
// y = (A_free A_free')^{-1} t = Linv'(Linv t) is two 64-term dot products per entry that EVERY workgroup of the kernel consuming
// y can afford to repeat — the single-workgroup, 128-step dependent triangular solve (trsv_small_kernel, 8.6 us per CG
// iteration) leaves the loop.  Built only when the factor has changed since the last projected_cg that wanted it.
// One workgroup of 256 threads, everything in LDS, by block recursion on  inv([A 0; B C]) = [Ai 0; -Ci B Ai  Ci]:
//   level 1  the four 16 x 16 diagonal blocks, one column per lane (16 lanes of each wave), substitution fully unrolled in registers;
//   level 2  two merges to 32 x 32 (two 16^3 products each), level 3 one merge to 64 x 64 (two 32^3 products): every thread
//            owns fixed entries of the product and accumulates them in index order (bit-reproducible).
// (History: lane = row with v_readlane broadcasts, 16 columns per wave in registers — 36 us: 64 steps x 16 columns of
// readlane + fma per wave, instruction-issue bound.)  m < 64 is padded with the identity.
__global__ __launch_bounds__(256) void tri_inv_small_kernel(const double* __restrict__ L, int m, double* __restrict__ W) {
    __shared__ double Ls[64][65], Xs[64][65], Ts[32][33], ds[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 4096; e += 256) {
        const int i = e & 63, k = e >> 6;
        Ls[i][k] = (i < m && k <= i) ? L[i + (int64_t)k * m] : (i == k ? 1.0 : 0.0);
        Xs[i][k] = 0.0;
    }
    if (tid < 64) ds[tid] = (tid < m) ? L[(int64_t)m * m + tid] : 1.0;          // reciprocal diagonal (chol_small_kernel)
    __syncthreads();
    if (lane < 16) {                                                             // level 1: block `wave`, column `lane` of it
        const int base = 16 * wave;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double acc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int l = 0; l < i; ++l) acc = fma(-Ls[base + i][base + l], x[l], acc);
            x[i] = acc * ds[base + i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) Xs[base + i][base + lane] = x[i];
    }
    __syncthreads();
    // merge<H>(base, t): X21 = -X22 (L21 X11) for the diagonal blocks [base, base+H) and [base+H, base+2H).  8 H threads per
    // merge: thread t owns column c = t % H and the H/8 rows r0, r0 + 8, ... of it (independent accumulators; the sums run
    // over the whole block — the zeros above the diagonals of X11 and X22 add nothing — so the loops have fixed trip counts).
    auto merge = [&](auto Htag, const int base, const int t, const int trow0) {
        constexpr int H = decltype(Htag)::value, NR = H / 8;
        const int R0 = base + H, C0 = base, c = t % H, r0 = t / H;
        double acc[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) acc[q] = 0.0;
#pragma unroll 8
        for (int l = 0; l < H; ++l) {                                            // T = L21 X11
            const double x = Xs[C0 + l][C0 + c];
#pragma unroll
            for (int q = 0; q < NR; ++q) acc[q] = fma(Ls[R0 + r0 + 8 * q][C0 + l], x, acc[q]);
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) Ts[trow0 + r0 + 8 * q][c] = acc[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NR; ++q) acc[q] = 0.0;
#pragma unroll 8
        for (int l = 0; l < H; ++l) {                                            // X21 = -X22 T
            const double tv = Ts[trow0 + l][c];
#pragma unroll
            for (int q = 0; q < NR; ++q) acc[q] = fma(Xs[R0 + r0 + 8 * q][R0 + l], tv, acc[q]);
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) Xs[R0 + r0 + 8 * q][C0 + c] = -acc[q];
        __syncthreads();
    };
    merge(std::integral_constant<int, 16>(), 32 * (tid >> 7), tid & 127, 16 * (tid >> 7));   // level 2: two merges side by side
    merge(std::integral_constant<int, 32>(), 0, tid, 0);                                       // level 3
    for (int e = tid; e < 4096; e += 256) {
        const int i = e & 63, k = e >> 6;
        W[k * 64 + i] = (i < m && k < m) ? Xs[i][k] : 0.0;                       // Linv, column-major
    }
    for (int e = tid; e < 4096; e += 256) {
        const int k = e & 63, i = e >> 6;
        W[4096 + i * 64 + k] = (i < m && k < m) ? Xs[i][k] : 0.0;                // Linv', column-major
    }
}

// Third kernel of the three-kernel general-constraint CG iteration (reduced projection form, mA <= 64):
//   t = sum of the per-workgroup partials of A_free r that cg_reduce_update_kernel<GEN> left        (left_mul, poly:86-98)
//   y = Linv'(Linv t)                                                                               (the two solves, poly:132-133)
//   v = r_free - A_free' y on the free variables, 0 on the fixed ones; this workgroup's partial of r.v   (poly:134, :743)
// Every workgroup forms t and y itself (same operands, same order: same bits everywhere), then its 64 chunks of v.
// Block = 64 chunks x 4 row groups (as proj_left_mul_tr_kernel<true, 4>); grid = ceil(nch_pad / 64).
// INIT (before the first H*p, :705-710): r = g_minor, tpart = the one "partial" proj_left_mul_kernel wrote; also stores
// p_1 = -v and the partials of v.v.
template <bool INIT>
__global__ __launch_bounds__(256) void proj_apply_linv_kernel(ProjArgs a, const double* __restrict__ r, double* __restrict__ out) {
    if (!INIT && a.state->stop_at != 0 && a.fused_j >= a.state->stop_at) return;
    __shared__ double tq[4][64];
    __shared__ double ts[64], us[64], ys[64];
    __shared__ double2 sm[4][64];
    const int tid = threadIdx.x, cl = tid & 63, rg = tid >> 6, m = a.mA;
    const int c = blockIdx.x * 64 + cl;
    const int cc = min(c, a.nch_pad - 1);
    // ---- every load first; the partials of A_free r lead (they are consumed first and loads return in order) ------------------
    // right-hand side: wave q folds a quarter of the partial blocks, in order; 32 loads in flight per batch (all of them at
    // n = 4096: 128 blocks), clamped indices so that no branch stands between the loads
    const int per = (a.tpart_nblk + 3) / 4;
    const int b0 = rg * per, b1 = min(a.tpart_nblk, b0 + per);
    const int col = min(cl, m - 1);
    double x0[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) x0[k] = a.tpart[(int64_t)max(min(b0 + k, b1 - 1), 0) * m + col];
    double w1[16], w2[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        w1[k] = a.W[(16 * rg + k) * 64 + cl];               // Linv[cl][16 rg + k]
        w2[k] = a.W[4096 + (16 * rg + k) * 64 + cl];        // Linv[16 rg + k][cl]
    }
    const double2* A2 = reinterpret_cast<const double2*>(a.A);
    const int64_t ld2 = a.ldA >> 1;
    double2 av[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) av[k] = A2[(int64_t)min(rg + 4 * k, m - 1) * ld2 + cc];
    const double2 rk = reinterpret_cast<const double2*>(r)[cc];
    int2 fr = make_int2(-1, -1);
    if (a.fixrank != nullptr) fr = reinterpret_cast<const int2*>(a.fixrank)[cc];
    // row cl of the (symmetric) Gram matrix, entries 16 rg .. 16 rg + 15, for the refinement step below
    const bool refine = a.Mgram != nullptr;
    double mg[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int kk = min(16 * rg + k, m - 1), ii = min(cl, m - 1);
        mg[k] = refine ? a.Mgram[(ii >= kk) ? (ii + (int64_t)kk * m) : (kk + (int64_t)ii * m)] : 0.0;
    }
    {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (b0 + k < b1) acc += x0[k];
        for (int b = b0 + 32; b < b1; b += 32) {            // more than 128 partial blocks (n > 4096)
            double x[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) x[k] = a.tpart[(int64_t)min(b + k, b1 - 1) * m + col];
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (b + k < b1) acc += x[k];
        }
        tq[rg][cl] = (cl < m) ? acc : 0.0;
    }
    __syncthreads();
    if (rg == 0) ts[cl] = (tq[0][cl] + tq[1][cl]) + (tq[2][cl] + tq[3][cl]);
    __syncthreads();
    // ---- u = Linv t -------------------------------------------------------------------------------------------------------------
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fma(w1[k], ts[16 * rg + k], acc);
    tq[rg][cl] = acc;
    __syncthreads();
    if (rg == 0) us[cl] = (tq[0][cl] + tq[1][cl]) + (tq[2][cl] + tq[3][cl]);
    __syncthreads();
    // ---- y = Linv' u ------------------------------------------------------------------------------------------------------------
    acc = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fma(w2[k], us[16 * rg + k], acc);
    tq[rg][cl] = acc;
    __syncthreads();
    if (rg == 0) ys[cl] = (tq[0][cl] + tq[1][cl]) + (tq[2][cl] + tq[3][cl]);
    __syncthreads();
    if (refine) {
        // ---- one step of iterative refinement: rho = t - M y, y += Linv'(Linv rho).  The explicit inverse leaves a residual of the
        //      normal equations of order cond(M) eps (measured: |A_free v| 100 - 10^5 x that of the triangular solves once
        //      cond(A_free A_free') reaches 10^9 - 10^15, tests/manual/illcond_probe.py); the correction brings A_free v back to the
        //      level of the reference's two substitutions.  Three more 64-term dot products per entry.
        acc = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (16 * rg + k < m) acc = fma(mg[k], ys[16 * rg + k], acc);
        tq[rg][cl] = acc;
        __syncthreads();
        if (rg == 0) ts[cl] = (cl < m) ? ts[cl] - ((tq[0][cl] + tq[1][cl]) + (tq[2][cl] + tq[3][cl])) : 0.0;
        __syncthreads();
        acc = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fma(w1[k], ts[16 * rg + k], acc);
        tq[rg][cl] = acc;
        __syncthreads();
        if (rg == 0) us[cl] = (tq[0][cl] + tq[1][cl]) + (tq[2][cl] + tq[3][cl]);
        __syncthreads();
        acc = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fma(w2[k], us[16 * rg + k], acc);
        tq[rg][cl] = acc;
        __syncthreads();
        if (rg == 0) ys[cl] += (tq[0][cl] + tq[1][cl]) + (tq[2][cl] + tq[3][cl]);
        __syncthreads();
    }
    // ---- left_mul_tr on this workgroup's chunks (rows rg, rg + 4, ...: the order of proj_left_mul_tr_kernel<true, 4>) -------------
    double2 z = make_double2(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = rg + 4 * k;
        if (i < m) {
            const double yi = ys[i];
            z.x = fma(yi, av[k].x, z.x);
            z.y = fma(yi, av[k].y, z.y);
        }
    }
    sm[rg][cl] = z;
    __syncthreads();
    if (rg != 0) return;
    const double tx = (sm[0][cl].x + sm[1][cl].x) + (sm[2][cl].x + sm[3][cl].x);
    const double ty = (sm[0][cl].y + sm[1][cl].y) + (sm[2][cl].y + sm[3][cl].y);
    double rv = 0.0, vv = 0.0;
    if (c < a.nch_pad) {
        const int j0 = 2 * c, j1 = 2 * c + 1;
        const double v0 = (fr.x >= 0 || j0 >= a.n) ? 0.0 : rk.x - tx;
        const double v1 = (fr.y >= 0 || j1 >= a.n) ? 0.0 : rk.y - ty;
        reinterpret_cast<double2*>(out)[c] = make_double2(v0, v1);
        if (INIT) reinterpret_cast<double2*>(a.p_out)[c] = make_double2(-v0, -v1);
        const double r0 = (j0 < a.n) ? rk.x : 0.0, r1 = (j1 < a.n) ? rk.y : 0.0;
        rv = fma(r1, v1, r0 * v0);
        vv = fma(v1, v1, v0 * v0);
    }
    rv = wave_sum(rv);
    if (INIT) vv = wave_sum(vv);
    if (cl == 0) {
        a.rvpart[blockIdx.x] = rv;
        if (INIT) a.vvpart[blockIdx.x] = vv;
    }
}

// tw <- L' \ (L \ tw)   (:114-115, :132-133).  Single workgroup, 64-wide blocked substitution.
//  * diagonal block: staged through LDS (padded tile, conflict-free row- and column-wise), solved by wave 0 with
//    v_readlane broadcasts; the reciprocals of its diagonal are formed by all lanes at once BEFORE the dependent chain
//    (one multiply per step in the chain instead of a division);
//  * forward trailing update x[i] -= sum_jj L[i, j0+jj] x[j0+jj]: the 64 columns are split into SPLIT slices so that
//    SPLIT x (rows left) threads work, partial sums combined in a fixed order (SPLIT = 4 when the LDS holds the partials,
//    else 1);
//  * backward update x[j0+c] -= sum_{k >= j0+nb} L[k, j0+c] x[k]: all 64 columns at once, 16 lanes per column
//    (contiguous 128-byte pieces of the column), combined by a 16-lane butterfly;
//  * the next diagonal tile is loaded into registers before the trailing update and written to LDS after it, so its
//    latency overlaps the update.
// Dynamic LDS: x[m2] | tile[64*65] | part[SPLIT == 4 ? 4*m2 : 0]   (m2 = m rounded up to even).
__global__ __launch_bounds__(CG_T) void trsv_pair_kernel(ProjArgs a, int split) {
    if (proj_skip(a.state)) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int m = a.mpp;
    const int m2 = (m + 1) & ~1;
    double* x = smem;
    double* tile = smem + m2;                // [64][65]
    double* part = tile + 64 * 65;           // [4][m2] when split == 4
    const double* __restrict__ L = a.L;
    const int64_t ld = m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = (m + 63) / 64;

    double treg[4];
    auto tile_fetch = [&](int j0) {          // 4096 entries, 4 per thread; zero outside the lower triangle / the block
        const int nb = min(64, m - j0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * CG_T, rr = e & 63, cc = e >> 6;
            treg[q] = (rr < nb && cc < nb && rr >= cc) ? L[(j0 + rr) + (int64_t)(j0 + cc) * ld] : 0.0;
        }
    };
    auto tile_store = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * CG_T, rr = e & 63, cc = e >> 6;
            tile[rr * 65 + cc] = treg[q];
        }
    };

    for (int i = tid; i < m; i += CG_T) x[i] = a.tw[i];
    tile_fetch(0);
    tile_store();
    __syncthreads();

    // ---- forward: L y = t ----
    for (int b = 0; b < nblk; ++b) {
        const int j0 = b * 64;
        const int nb = min(64, m - j0);
        if (wave == 0) {
            // lane i owns unknown j0+i; column jj of the block is tile[i*65 + jj]
            const double dii = tile[lane * 65 + lane];
            const double di = (lane < nb) ? 1.0 / dii : 0.0;
            double xi = (lane < nb) ? x[j0 + lane] : 0.0;
#pragma unroll 8
            for (int jj = 0; jj < nb; ++jj) {
                const double lij = tile[lane * 65 + jj];
                if (lane == jj) xi = xi * di;
                const double xj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xi), jj),
                                                   __builtin_amdgcn_readlane(__double2loint(xi), jj));
                if (lane > jj) xi = fma(-lij, xj, xi);
            }
            if (lane < nb) x[j0 + lane] = xi;
        }
        __syncthreads();                     // x[j0 .. j0+nb) final; wave 0 is done with the tile
        const int i0 = j0 + nb, rem = m - i0;
        if (b + 1 < nblk) tile_fetch(i0);    // in flight during the update
        if (rem > 0) {
            if (split == 4) {
                const int per = (nb + 3) >> 2;
                for (int w = tid; w < 4 * rem; w += CG_T) {
                    const int q = w / rem, i = i0 + (w - q * rem);
                    const int jlo = q * per, jhi = min(nb, jlo + per);
                    double acc = 0.0;
#pragma unroll 8
                    for (int jj = jlo; jj < jhi; ++jj) acc = fma(L[i + (int64_t)(j0 + jj) * ld], x[j0 + jj], acc);
                    part[q * m2 + i] = acc;
                }
                __syncthreads();
                for (int i = i0 + tid; i < m; i += CG_T)
                    x[i] -= (part[i] + part[m2 + i]) + (part[2 * m2 + i] + part[3 * m2 + i]);
            } else {
                for (int i = i0 + tid; i < m; i += CG_T) {
                    double acc = 0.0;
                    for (int jj = 0; jj < nb; ++jj) acc = fma(L[i + (int64_t)(j0 + jj) * ld], x[j0 + jj], acc);
                    x[i] -= acc;
                }
            }
        }
        if (b + 1 < nblk) tile_store();
        __syncthreads();
    }

    // ---- backward: L' w = y ----   (the tile in LDS is the LAST diagonal block: exactly what the first step needs)
    for (int b = nblk - 1; b >= 0; --b) {
        const int j0 = b * 64;
        const int nb = min(64, m - j0);
        if (wave == 0) {
            // lane i owns unknown j0+i and needs L[jj, i] for jj > i: tile[jj*65 + i] (consecutive lanes, conflict-free)
            const double dii = tile[lane * 65 + lane];
            const double di = (lane < nb) ? 1.0 / dii : 0.0;
            double xi = (lane < nb) ? x[j0 + lane] : 0.0;
#pragma unroll 8
            for (int jj = nb - 1; jj >= 0; --jj) {
                const double lji = tile[jj * 65 + lane];
                if (lane == jj) xi = xi * di;
                const double xj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xi), jj),
                                                   __builtin_amdgcn_readlane(__double2loint(xi), jj));
                if (lane < jj) xi = fma(-lji, xj, xi);
            }
            if (lane < nb) x[j0 + lane] = xi;
        }
        __syncthreads();
        if (b > 0) {
            // the block above: its 64 columns each need sum_{k >= j0} L[k, c] x[k]  (everything below it is final now)
            const int p0 = j0 - 64;
            tile_fetch(p0);
            const int c = tid >> 4, ks = tid & 15;          // 64 columns x 16 lanes
            double acc = 0.0;
            const double* col = L + (int64_t)(p0 + c) * ld;
#pragma unroll 8
            for (int k = j0 + ks; k < m; k += 16) acc = fma(col[k], x[k], acc);
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 16);
            __syncthreads();                 // every thread has read x before lane 0 of each group updates it
            if (ks == 0) x[p0 + c] -= acc;
            tile_store();
            __syncthreads();
        }
    }

    for (int i = tid; i < m; i += CG_T) a.tw[i] = x[i];
}

}  // namespace bh
