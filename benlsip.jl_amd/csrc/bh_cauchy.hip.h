// bh_cauchy.hip.h — device-resident cauchy_step (breakpoint search, active-set growth)
// Part of the single translation unit of bh_api.hip (see bh_kernels.hip.h for the layout and design notes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bh_reduce.hip.h"
#include "bh_cg.hip.h"
#include "bh_proj.hip.h"   // dvec4

namespace bh {

// ------------------------------------------------------------------------------------------
// cauchy_step on the device — src/basic_tralcnlss.jl:574-639 with next_breakpoint (:536-562) and the initial
// active_bounds! (src/polyhedral_constraints.jl:203-215).  SURVEY.md §8 "next" row f-3.
// The loop state reuses CgState (so the row-stream / projection kernels can gate on ->done / ->need_proj):
//   rtv = phi_p, pHp = phi_pp, gamma = theta, alpha = delta_t, iter = nb_fix, max_iter = n - mA (nmm),
//   status = index fixed at the last breakpoint (-1: none), approx_solved = min_found, neg_curvature = 1 when no
//   breakpoint exists (the reference would index fixvars[-1]), pad = breakpoints taken, n_hmul = H*d products.
// ------------------------------------------------------------------------------------------
constexpr int CA_T = 512;        // threads of cauchy_advance_kernel / cauchy_fused_kernel

// Loop state of the one-kernel-per-breakpoint form (cauchy_fused_kernel): two records, launch k reads [(k + 1) & 1] and workgroup 0
// writes [k & 1] — no workgroup reads a word its own launch writes.
struct CauchyPass { int done, nfix, n_hmul, breakpoints, err, pad0, pad1, pad2; };

struct CauchyArgs {
    CgState* st;
    const double* x; const double* g; const double* xlow; const double* xupp;
    double* negg; double* d; const double* Hd; double* s; double* dl; double* du;
    int* fixrank;
    int n, n_pad, nmm;
    double delta, atol;
    int box;                // mA == 0: the projection is a mask, so it is maintained in place (d[ind] = 0 when ind becomes fixed)
    unsigned long long* mirror; unsigned tag;
    // image-space search (box constraints): s'Hd and d'Hd arrive as per-workgroup partial sums over the rows of J (cauchy_image_kernel)
    const double* img_part; int img_G;
    // one-kernel-per-breakpoint form: "fixed before launch k" marks and the first loop-state record (NULL: not used)
    int* fixpass; CauchyPass* pp0;
};

// Image-space form of the box-constrained search.  With box constraints a breakpoint only zeroes one component of d, so in the
// row space of J~ = [J; sqrt-weighted C] the two vectors  t_d = J~ d  and  t_s = J~ s_c  follow by rank-one updates
//     t_s += theta t_d ,   t_d -= d_ind J~[:, ind]        (one COLUMN of J: d + q strided loads, 4 MiB of traffic at config 3)
// and the search's scalars are  d'Hd = sum_i w_i t_d,i^2  and  s_c'Hd = sum_i w_i t_s,i t_d,i  (:610-611, :634-635) — the same
// numbers as dot(d, H*d) and dot(s_c, H*d), rounded differently, without the sweep over J that H*d costs per breakpoint (:633):
// one J v sweep for t_d at the start, then ~10 us per breakpoint instead of ~317 us.
struct CauchyImgArgs {
    const CgState* st;
    const double* J; int64_t ld; int64_t nrows, d_rows; double mu;
    double* td; double* ts;
    double* part;           // [2][gridDim.x]: partial sums of w t_s t_d and of w t_d^2
    int first;              // pass 0: t_d = J d has just been formed by the J v kernel, t_s = 0: no update, only the sums
};

// The same with linear equalities (reduced projection form, small mA).  d = P(-g) = -D g - D A'y changes in every free component
// per breakpoint (D = mask of the free variables, y = (A_free A_free')^{-1} A_free(-g), left in ProjArgs::tw by the projection
// kernels), but in the row space it is  t_d = J~ d = -a - B y  with  a = J~ D g  (rows)  and  B = J~ D A'  (rows x mA, stored
// column by column), and fixing variable `ind` is a one-column update of both:  a -= g_ind J~[:, ind],  B -= J~[:, ind] A[:, ind]'.
// a and B cost 1 + mA J v sweeps at the start; afterwards a breakpoint costs rows x (mA + 3) doubles of traffic instead of a sweep.
struct CauchyImgGenArgs {
    CauchyImgArgs b;
    double* a;              // rows
    double* B;              // mA x rows_cap (column j at B + j * rows_cap)
    int64_t rows_cap;
    int mA;
    const double* A; int64_t ldA;     // row-major mA x ldA image of lineq
    const double* tw;       // y (mA), written by the projection of this pass
    const double* g;
};

// Few equalities (mA <= 16): one row per thread, the columns of B in a serial loop — 8.2 us per pass at mA = 8 against 10.3 us for the
// tiled form below, whose barriers and idle waves cost more than its loads in flight gain there.
__device__ __forceinline__ void cauchy_image_gen_rows_body(const CauchyImgGenArgs& ga, int block, int nblocks) {
    const CauchyImgArgs& a = ga.b;
    const CgState* st = a.st;
    __shared__ double scratch[2 * 4];
    __shared__ double s_acol[64], s_y[64];
    const int ind = st->status;
    const double theta = st->gamma;
    const bool upd = !a.first && ind >= 0;
    if ((int)threadIdx.x < ga.mA) {
        s_acol[threadIdx.x] = upd ? ga.A[(int64_t)threadIdx.x * ga.ldA + ind] : 0.0;
        s_y[threadIdx.x] = ga.tw[threadIdx.x];
    }
    const double g_ind = upd ? ga.g[ind] : 0.0;
    __syncthreads();
    double acc[2] = {0.0, 0.0};
    for (int64_t i = (int64_t)block * 256 + threadIdx.x; i < a.nrows; i += (int64_t)nblocks * 256) {
        double ts = 0.0;
        if (!a.first) ts = __dadd_rn(a.ts[i], __dmul_rn(theta, a.td[i]));        // s_c += theta d  with the PREVIOUS d   (:628)
        const double col = upd ? a.J[i * a.ld + ind] : 0.0;
        double ai = ga.a[i];
        if (upd) { ai = __dsub_rn(ai, __dmul_rn(g_ind, col)); ga.a[i] = ai; }
        double td = -ai;
        for (int j = 0; j < ga.mA; ++j) {
            double bij = ga.B[(int64_t)j * ga.rows_cap + i];
            if (upd) { bij = __dsub_rn(bij, __dmul_rn(col, s_acol[j])); ga.B[(int64_t)j * ga.rows_cap + i] = bij; }
            td = fma(-bij, s_y[j], td);
        }
        a.td[i] = td;
        a.ts[i] = ts;
        const double w = (i < a.d_rows) ? 1.0 : a.mu;
        acc[0] = fma(w * ts, td, acc[0]);
        acc[1] = fma(w * td, td, acc[1]);
    }
    block_reduce<256, 2>(acc, scratch, OpSum(), 0.0);
    if (threadIdx.x == 0) { a.part[block] = acc[0]; a.part[nblocks + block] = acc[1]; }
}
__global__ __launch_bounds__(256) void cauchy_image_gen_rows_kernel(CauchyImgGenArgs ga) {
    if (ga.b.st->done) return;
    cauchy_image_gen_rows_body(ga, (int)blockIdx.x, (int)gridDim.x);
}

// A workgroup works on tiles of 64 rows: wave w owns columns [w cpw, (w+1) cpw) of B (cpw = ceil(mA / 4) <= 16) for all 64 rows, so
// up to 16 independent 512-byte loads per wave are in flight (one row per thread and a serial loop over mA columns left the
// 67 MB this pass moves at mA = 64 at 2.5 TB/s); the column entry of J — the expensive strided load — is fetched once per row by
// wave 0 and shared through LDS, the four partial dot products meet there too.  Fixed order of accumulation: bit-reproducible.
__device__ __forceinline__ void cauchy_image_gen_body(const CauchyImgGenArgs& ga, int block, int nblocks) {
    const CauchyImgArgs& a = ga.b;
    const CgState* st = a.st;
    __shared__ double s_acol[64], s_y[64], s_col[64], s_dot[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ind = st->status;
    const double theta = st->gamma;
    const bool upd = !a.first && ind >= 0;
    if ((int)threadIdx.x < ga.mA) {
        s_acol[threadIdx.x] = upd ? ga.A[(int64_t)threadIdx.x * ga.ldA + ind] : 0.0;
        s_y[threadIdx.x] = ga.tw[threadIdx.x];
    }
    const double g_ind = upd ? ga.g[ind] : 0.0;
    const int cpw = (ga.mA + 3) >> 2, j0 = wave * cpw;
    double acc[2] = {0.0, 0.0};
    for (int64_t tile = block; tile * 64 < a.nrows; tile += nblocks) {
        const int64_t i = tile * 64 + lane;
        const bool vrow = i < a.nrows;
        const int64_t ic = vrow ? i : a.nrows - 1;
        // this wave's entries of B for the tile: all loads out before any is used
        double b[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int jj = min(j0 + k, ga.mA - 1);
            b[k] = (k < cpw) ? ga.B[(int64_t)jj * ga.rows_cap + ic] : 0.0;       // (k < cpw is uniform over the launch)
        }
        double ai = 0.0, ts = 0.0;
        if (wave == 0) {
            ai = ga.a[ic];
            if (!a.first) ts = __dadd_rn(a.ts[ic], __dmul_rn(theta, a.td[ic]));  // s_c += theta d  with the PREVIOUS d   (:628)
            const double col = upd ? a.J[ic * a.ld + ind] : 0.0;
            s_col[lane] = col;
            if (upd) { ai = __dsub_rn(ai, __dmul_rn(g_ind, col)); if (vrow) ga.a[i] = ai; }
        }
        __syncthreads();                                              // s_col (and, first tile, s_acol / s_y) are in place
        const double col = s_col[lane];
        double dot = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int jj = j0 + k;
            if (k < cpw && jj < ga.mA) {
                double bij = b[k];
                if (upd) { bij = __dsub_rn(bij, __dmul_rn(col, s_acol[jj])); if (vrow) ga.B[(int64_t)jj * ga.rows_cap + i] = bij; }
                dot = fma(bij, s_y[jj], dot);
            }
        }
        s_dot[wave][lane] = dot;
        __syncthreads();
        if (wave == 0 && vrow) {
            double td = -ai;                                          // t_d = -a - B y
            td = __dsub_rn(td, s_dot[0][lane]); td = __dsub_rn(td, s_dot[1][lane]);
            td = __dsub_rn(td, s_dot[2][lane]); td = __dsub_rn(td, s_dot[3][lane]);
            a.td[i] = td;
            a.ts[i] = ts;
            const double w = (i < a.d_rows) ? 1.0 : a.mu;
            acc[0] = fma(w * ts, td, acc[0]);
            acc[1] = fma(w * td, td, acc[1]);
        }
        __syncthreads();                                              // wave 0 is done with s_dot before the next tile overwrites it
    }
    if (wave == 0) {
        acc[0] = wave_sum(acc[0]); acc[1] = wave_sum(acc[1]);
        if (lane == 0) { a.part[block] = acc[0]; a.part[nblocks + block] = acc[1]; }
    }
}
__global__ __launch_bounds__(256) void cauchy_image_gen_kernel(CauchyImgGenArgs ga) {
    if (ga.b.st->done) return;
    cauchy_image_gen_body(ga, (int)blockIdx.x, (int)gridDim.x);
}
// The row kernel and d = P(-g) = -g_free - A_free'y need the same y and nothing of each other: one launch, the first `row_blocks`
// workgroups take the rows, the others 64 chunks of d each (proj_left_mul_tr_kernel<true, 4>'s arithmetic).
// A third group of mA workgroups forms t_fresh = A_free(-g) for the CURRENT active set (left_mul, one row of A each): the next
// pass's right-hand side is this value minus the column of the variable the decision in between fixes (cauchy_factor_solve_kernel).
__global__ __launch_bounds__(256) void cauchy_gen_rows_and_d_kernel(CauchyImgGenArgs ga, int tiled, int row_blocks, int d_blocks, ProjArgs pa,
                                                                   const double* __restrict__ r, double* __restrict__ d_out,
                                                                   double* __restrict__ t_fresh) {
    if (ga.b.st->done) return;
    const int b = (int)blockIdx.x;
    if (b < row_blocks) {
        if (tiled) cauchy_image_gen_body(ga, b, row_blocks);
        else cauchy_image_gen_rows_body(ga, b, row_blocks);
    } else if (b < row_blocks + d_blocks) {
        proj_left_mul_tr_body<true, 4>(pa, r, d_out, b - row_blocks);
    } else {
        ProjArgs pf = pa;
        pf.tw = t_fresh;
        proj_left_mul_body(pf, r, b - row_blocks - d_blocks);
    }
}

// B = J~ D A'  (rows x mA, column j at B + j * rows_cap) on the matrix cores: the one place on this path where a tile of J meets a
// dense GEMM (M = rows of J, N = mA <= 64, K = n) — 1 + mA J v sweeps became two.  v_mfma_f64_16x16x4_f64; a workgroup of four waves
// owns 128 rows of J (a wave 32 of them: two 16-row tiles x four 16-column output tiles); the K index is permuted as in
// gram_free_mfma_kernel (lane group l >> 4 owns 4 consecutive columns of every 16-column super-step: one 32-byte load per lane and
// operand, 128 contiguous bytes per matrix row).  The masked 64 x 16 tile of A of a super-step is fetched ONCE per workgroup (wave
// w loads output tile w's 16 rows) and handed to the other waves through a double-buffered LDS slot — read by every wave
// straight from L2 it was 4/5 of the kernel's traffic (10 GiB through L2 for 2 GiB of J: 1.4 ms at config-5 size).
// Fixed order of accumulation: bit-reproducible.
__global__ __launch_bounds__(256) void image_b_mfma_kernel(const double* __restrict__ J, int64_t ld, int64_t nrows, const double* __restrict__ A,
                                                           int64_t ldA, int mA, const int* __restrict__ fixrank, double* __restrict__ B,
                                                           int64_t rows_cap) {
    __shared__ double2 atile[2][4][2][64];            // [buffer][output tile][column pair 01 / 23][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t strip = ((int64_t)blockIdx.x * 4 + wave) * 32;         // (waves past the last row still take part in the barriers)
    const int ri = lane & 15, kq = lane >> 4;
    const double* pj[2];
    bool vrow[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        vrow[m] = strip + 16 * m + ri < nrows;
        pj[m] = J + (vrow[m] ? strip + 16 * m + ri : 0) * ld + 4 * kq;
    }
    const bool va = 16 * wave + ri < mA;               // this wave fetches rows 16 wave .. 16 wave + 15 of A (output tile `wave`)
    const double* pa = A + (int64_t)(va ? 16 * wave + ri : 0) * ldA + 4 * kq;
    dvec4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = dvec4{0.0, 0.0, 0.0, 0.0};
    const int nsuper = (int)(ld >> 4);                 // ld (= ldA) is a multiple of 16; the padding columns of both images are zero
    auto fetch_a = [&](int sidx, double2& a01, double2& a23) {
        const int64_t c = (int64_t)sidx * 16;
        a01 = a23 = make_double2(0.0, 0.0);
        if (va) { a01 = *reinterpret_cast<const double2*>(pa + c); a23 = *reinterpret_cast<const double2*>(pa + c + 2); }
        if (fixrank != nullptr) {
            const int4 f = *reinterpret_cast<const int4*>(fixrank + c + 4 * kq);
            if (f.x >= 0) a01.x = 0.0;
            if (f.y >= 0) a01.y = 0.0;
            if (f.z >= 0) a23.x = 0.0;
            if (f.w >= 0) a23.y = 0.0;
        }
    };
    auto fetch_j = [&](int sidx, double2 (&j01)[2], double2 (&j23)[2]) {
        const int64_t c = (int64_t)sidx * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            j01[m] = j23[m] = make_double2(0.0, 0.0);
            if (vrow[m]) { j01[m] = *reinterpret_cast<const double2*>(pj[m] + c); j23[m] = *reinterpret_cast<const double2*>(pj[m] + c + 2); }
        }
    };
    double2 na01, na23, nj01[2], nj23[2];
    fetch_a(0, na01, na23);
    fetch_j(0, nj01, nj23);
    atile[0][wave][0][lane] = na01; atile[0][wave][1][lane] = na23;
    __syncthreads();
    for (int sidx = 0; sidx < nsuper; ++sidx) {
        const int buf = sidx & 1;
        double2 j01[2], j23[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) { j01[m] = nj01[m]; j23[m] = nj23[m]; }
        if (sidx + 1 < nsuper) {                                           // next super-step's operands: in flight during the MFMAs
            fetch_a(sidx + 1, na01, na23);
            fetch_j(sidx + 1, nj01, nj23);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double2 a01 = atile[buf][t][0][lane], a23 = atile[buf][t][1][lane];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(j01[m].x, a01.x, acc[m][t], 0, 0, 0);
                acc[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(j01[m].y, a01.y, acc[m][t], 0, 0, 0);
                acc[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(j23[m].x, a23.x, acc[m][t], 0, 0, 0);
                acc[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(j23[m].y, a23.y, acc[m][t], 0, 0, 0);
            }
        }
        if (sidx + 1 < nsuper) { atile[buf ^ 1][wave][0][lane] = na01; atile[buf ^ 1][wave][1][lane] = na23; }
        __syncthreads();                                                   // one barrier per super-step (the slot written is the other one)
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = 16 * t + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = strip + 16 * m + (lane >> 4) + 4 * r;
                if (col < mA && row < nrows) B[(int64_t)col * rows_cap + row] = acc[m][t][r];
            }
        }
}

// Several ranks: this rank's two sums (its rows of J) in scal[0..1], ready for the all-reduce that precedes the advance kernel.
__global__ __launch_bounds__(64) void cauchy_image_sum_kernel(const double* __restrict__ part, int G, double* __restrict__ scal, const CgState* st) {
    if (st->done) return;
    const double a = wave_fixed_sum(part, G), b = wave_fixed_sum(part + G, G);
    if (threadIdx.x == 0) { scal[0] = a; scal[1] = b; }
}

__global__ __launch_bounds__(256) void cauchy_image_kernel(CauchyImgArgs a) {
    const CgState* st = a.st;
    __shared__ double scratch[2 * 4];
    // (the first row's t_d, t_s are asked for together with the loop state — the column entry of J needs `ind` and follows; at
    // config 3 a thread owns one row, so the pass is two dependent memory round trips instead of three)
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t ic = min(i0, a.nrows - 1 > 0 ? a.nrows - 1 : (int64_t)0);
    const double td0 = a.td[ic], ts0 = a.ts[ic];
    const int done_in = st->done;
    const int ind = st->status;                 // the variable fixed at the previous breakpoint, its step and its old d component
    const double theta = st->gamma, dind = st->beta;
    if (done_in) return;
    double acc[2] = {0.0, 0.0};
    for (int64_t i = i0; i < a.nrows; i += (int64_t)gridDim.x * 256) {
        double td = (i == i0) ? td0 : a.td[i], ts = 0.0;
        if (!a.first) {
            ts = __dadd_rn((i == i0) ? ts0 : a.ts[i], __dmul_rn(theta, td));    // s_c += theta d          (:628)
            td = __dsub_rn(td, __dmul_rn(dind, a.J[i * a.ld + ind]));            // d[ind] = 0              (:632, box)
            a.td[i] = td;
        }
        a.ts[i] = ts;
        const double w = (i < a.d_rows) ? 1.0 : a.mu;
        acc[0] = fma(w * ts, td, acc[0]);
        acc[1] = fma(w * td, td, acc[1]);
    }
    block_reduce<256, 2>(acc, scratch, OpSum(), 0.0);
    if (threadIdx.x == 0) { a.part[blockIdx.x] = acc[0]; a.part[gridDim.x + blockIdx.x] = acc[1]; }
}

// progress word of the search: tag | error flag | done | breakpoints taken | passes run
__device__ __forceinline__ void publish_cauchy_word(const CauchyArgs& a, int err, int done, int breakpoints, int passes) {
    if (a.mirror == nullptr) return;
    const unsigned long long wv = ((unsigned long long)(a.tag & 0xffffu) << 48) | ((unsigned long long)(err & 0xf) << 44) |
                                  ((unsigned long long)(done & 0xf) << 40) | ((unsigned long long)(breakpoints & 0xfffff) << 20) |
                                  (unsigned long long)(passes & 0xfffff);
    __hip_atomic_store(a.mirror, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// :587-603: s_c = 0; active_bounds!(lincons, x); -g; d_u = min(xupp - x, delta); d_l = max(xlow - x, -delta).
__global__ __launch_bounds__(CG_T) void cauchy_init_kernel(CauchyArgs a) {
    __shared__ double scratch[CG_T / 64];
    double cnt[1] = {0.0};
    for (int i = a.n + threadIdx.x; i < a.n_pad; i += CG_T) { a.negg[i] = 0.0; a.d[i] = 0.0; a.s[i] = 0.0; a.fixrank[i] = -1; }
    for (int i = threadIdx.x; i < a.n; i += CG_T) {
        const double xi = a.x[i];
        const bool act = (__dsub_rn(xi, a.xlow[i]) <= a.atol) || (__dsub_rn(a.xupp[i], xi) <= a.atol);   // poly:211
        a.fixrank[i] = act ? 0 : -1;
        if (a.fixpass != nullptr) a.fixpass[i] = act ? -1 : 0x7fffffff;
        cnt[0] += act ? 1.0 : 0.0;
        a.negg[i] = -a.g[i];
        if (a.box) a.d[i] = act ? 0.0 : -a.g[i];                // d = projection(lincons, -g) for box constraints (:592)
        a.s[i] = 0.0;
        a.du[i] = fmin(__dsub_rn(a.xupp[i], xi), a.delta);     // :602
        a.dl[i] = fmax(__dsub_rn(a.xlow[i], xi), -a.delta);    // :603
    }
    block_reduce<CG_T, 1>(cnt, scratch, OpSum(), 0.0);
    if (threadIdx.x == 0) {
        CgState* st = a.st;
        st->rtv = 0.0; st->pHp = 0.0; st->gamma = 0.0; st->alpha = 0.0; st->beta = 0.0; st->tol_cg = 0.0;
        st->iter = (int)cnt[0]; st->max_iter = a.nmm;
        st->approx_solved = 0; st->outside_region = 0; st->neg_curvature = 0;
        st->done = 0; st->status = -1; st->n_hmul = 0; st->need_proj = 1; st->pad = 0;
        if (a.pp0 != nullptr) { CauchyPass z{}; z.nfix = (int)cnt[0]; *a.pp0 = z; }
    }
}

// One pass: phi_p, phi_pp for the current (d, Hd) (:610-611 / :634-635), the while test (:615), next_breakpoint (:617),
// the three-way branch (:620-636) including s_c update and the active-set growth of add_active! (poly:240-249).
// Latency-bound (one workgroup of CA_T = 512 threads: eight waves leave each 256 VGPRs, so the 56 operands of a thread stay in registers —
// the 1024-thread shape of the CG step kernels spilled here, and a kernel that touches scratch pays for it at every dispatch): every load of the pass — the loop state, the row-space partial sums, the
// first batch of vector elements — is issued before anything is used (also before the `done` gate: the addresses are valid either
// way), and the three sums and the arg-min share one barrier.
__global__ __launch_bounds__(CA_T) void cauchy_advance_kernel(CauchyArgs a) {
    constexpr int NW = CA_T / 64;
    __shared__ double scratch[4 * NW];
    __shared__ int iscratch[NW];
    CgState* st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    const bool img = a.img_part != nullptr;      // s'Hd and d'Hd come from the row-space partial sums: Hd is not read
    // ---- loads first ---------------------------------------------------------------------------------------------------------
    const int done_in = st->done;
    const int nfix = st->iter;
    const int n_hmul_in = st->n_hmul, pad_in = st->pad;          // (read here, not at the tail: thread 0's last stores wait for nothing)
    LaneBatch<8> b0, b1;
    if (img) { b0.issue(a.img_part, a.img_G); b1.issue(a.img_part + a.img_G, a.img_G); }
    // Eight elements per thread and batch (all of them for n <= 4096): their seven operands are requested together and the pass
    // runs from registers, one memory round trip instead of eight; the first batch's d and s are kept for the update below.
    constexpr int E = 8;
    double dv[E], hv[E], sv[E], gv[E], lv[E], uv[E];
    int fv[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int i = min(tid + k * CA_T, a.n - 1);
        dv[k] = a.d[i]; hv[k] = img ? 0.0 : a.Hd[i]; sv[k] = a.s[i]; gv[k] = a.g[i]; lv[k] = a.dl[i]; uv[k] = a.du[i]; fv[k] = a.fixrank[i];
    }
    if (done_in) return;
    double x3[3] = {0.0, 0.0, 0.0};              // s'Hd, d'Hd, g'd
    double th = INF;
    int ind = 0x7fffffff;
    double d0[E], s0[E];
    for (int base = 0; base < a.n; base += E * CA_T) {
        if (base > 0) {
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const int i = min(base + tid + k * CA_T, a.n - 1);
                dv[k] = a.d[i]; hv[k] = img ? 0.0 : a.Hd[i]; sv[k] = a.s[i]; gv[k] = a.g[i]; lv[k] = a.dl[i]; uv[k] = a.du[i]; fv[k] = a.fixrank[i];
            }
        }
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int i = base + tid + k * CA_T;
            if (base == 0) { d0[k] = dv[k]; s0[k] = sv[k]; }
            if (i >= a.n) continue;
            const double di = dv[k], hdi = hv[k], si = sv[k];
            x3[0] = fma(si, hdi, x3[0]);
            x3[1] = fma(di, hdi, x3[1]);
            x3[2] = fma(gv[k], di, x3[2]);
            if (fv[k] < 0) {                                          // :547
                double t = INF;
                if (di < 0.0) t = __ddiv_rn(__dsub_rn(lv[k], si), di);       // :549
                else if (di > 0.0) t = __ddiv_rn(__dsub_rn(uv[k], si), di);  // :551
                if (t < th) { th = t; ind = i; }                      // strict <: first minimiser in index order (:555)
            }
        }
    }
    // ---- the three sums (block_reduce's order: wave butterfly, then the waves in ascending order) and the arg-min with the
    //      smallest index among equal thetas, behind one pair of barriers
#pragma unroll
    for (int q = 0; q < 3; ++q) x3[q] = wave_reduce(x3[q], OpSum());
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double t2 = __shfl_xor(th, off);
        const int i2 = __shfl_xor(ind, off);
        if (t2 < th || (t2 == th && i2 < ind)) { th = t2; ind = i2; }
    }
    if (lane == 0) {
        scratch[wave] = x3[0]; scratch[NW + wave] = x3[1]; scratch[2 * NW + wave] = x3[2];
        scratch[3 * NW + wave] = th; iscratch[wave] = ind;
    }
    __syncthreads();
    double sums[2] = {0.0, 0.0}, gd = 0.0;
    for (int w = 0; w < NW; ++w) { sums[0] += scratch[w]; sums[1] += scratch[NW + w]; gd += scratch[2 * NW + w]; }
    th = scratch[3 * NW]; ind = iscratch[0];
    for (int w = 1; w < NW; ++w) {
        const double t2 = scratch[3 * NW + w];
        const int i2 = iscratch[w];
        if (t2 < th || (t2 == th && i2 < ind)) { th = t2; ind = i2; }
    }
    if (img) { sums[0] = wave_sum(b0.fold_sum(a.img_part, a.img_G)); sums[1] = wave_sum(b1.fold_sum(a.img_part + a.img_G, a.img_G)); }
    if (ind == 0x7fffffff) ind = -1;                              // :544

    const double phi_p = __dadd_rn(sums[0], gd);                  // :610 / :634
    const double phi_pp = sums[1];                                // :611 / :635
    int done = 0, min_found = 0, err = 0, advance = 0;
    double step = 0.0;
    const double delta_t = (phi_pp > 0.0) ? __ddiv_rn(-phi_p, phi_pp) : 0.0;     // :618
    if (!(nfix < a.nmm)) {                                        // :615
        done = 1;
    } else if (phi_p >= 0.0) {                                    // :620
        min_found = 1; done = 1;
    } else if (phi_p < 0.0 && phi_pp > 0.0 && delta_t < th) {     // :622
        step = delta_t; min_found = 1; done = 1;                  // :625
    } else {                                                      // :627
        if (ind < 0) { err = 1; done = 1; }
        else { step = th; advance = 1; }                          // :628
    }
    if (step != 0.0 || advance) {
#pragma unroll
        for (int k = 0; k < E; ++k) {                                 // the first batch: from registers
            const int i = tid + k * CA_T;
            if (i >= a.n) continue;
            a.s[i] = __dadd_rn(s0[k], __dmul_rn(step, d0[k]));
            // box constraints: projection!(lincons, -g, d) after add_active!(ind) only zeroes d[ind] (:632) — done by the
            // thread that owns the element, after it has used the old value (kept in st->beta for the image-space update)
            if (advance && a.box && i == ind) { st->beta = d0[k]; a.d[i] = 0.0; }
        }
        for (int i = E * CA_T + tid; i < a.n; i += CA_T) {
            const double di = a.d[i];
            a.s[i] = __dadd_rn(a.s[i], __dmul_rn(step, di));
            if (advance && a.box && i == ind) { st->beta = di; a.d[i] = 0.0; }
        }
    }
    if (tid == 0) {
        st->rtv = phi_p; st->pHp = phi_pp; st->gamma = th; st->alpha = delta_t;
        st->n_hmul = n_hmul_in + 1;
        st->approx_solved = min_found; st->neg_curvature = err;
        if (advance) {
            a.fixrank[ind] = 0;                                   // add_active!: fixvars[ind] = true (poly:246)
            st->iter = nfix + 1; st->status = ind; st->pad = pad_in + 1;
        }
        st->done = done; st->need_proj = done ? 0 : 1;
        publish_cauchy_word(a, err, done, pad_in + (advance ? 1 : 0), n_hmul_in + 1);
    }
}

// ONE kernel per breakpoint (box constraints, one rank, row-space form).  The two-kernel pass above is
//     cauchy_image_kernel (rows: apply the last breakpoint, leave partial sums)  ->  cauchy_advance_kernel (one workgroup: decide)
// and each kernel boundary costs more than the work behind it.  Here the decision moves into the prologue of the row kernel:
// launch k >= 1 has EVERY workgroup redo decision k-1 for the whole vector (same operands, same order, same bits: 144 KiB from L2),
// then apply it to its own rows and leave the partial sums for launch k+1.  Nothing a launch writes is read by the same launch:
//   s            ping-pong (read [(k+1)&1], write [k&1]; the 64-element group i>>6 is stored by workgroup (i>>6) mod G);
//   d            not stored at all: with box constraints d_i = -g_i while i is free, 0 afterwards (:592, :632);
//   fixed flags  fixpass[i] = launch that fixed i (-1: active from the start, INT_MAX: free); launch k treats i as fixed iff
//                fixpass[i] < k, so the mark workgroup 0 sets in launch k (value k) reads as "free" in launch k either way;
//   loop state   two CauchyPass records; a gated launch copies the record forward so that `done` survives over-launching;
//   partial sums ping-pong.
// Launch 0 is cauchy_image_kernel(first = 1): the sums of t_d = J~ d_0 only.  Decisions = launches - 1.
struct CauchyFusedArgs {
    CauchyPass* pp; int k;
    const double* g; const double* dl; const double* du;
    double* sbuf[2];
    int* fixpass; int* fixrank;
    int n, nmm;
    const double* J; int64_t ld; int64_t nrows, d_rows; double mu;
    double* td; double* ts;
    const double* part_in; int Gin;          // [2][Gin] from the previous launch
    double* part_out;                        // [2][gridDim.x]
    unsigned long long* mirror; unsigned tag;
};

__global__ __launch_bounds__(CA_T) void cauchy_fused_kernel(CauchyFusedArgs a) {
    constexpr int NW = CA_T / 64;
    constexpr int E = 8;
    __shared__ double scratch[3 * NW];
    __shared__ int iscratch[NW];
    __shared__ double rscratch[2 * NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    const CauchyPass* in = a.pp + ((a.k + 1) & 1);
    CauchyPass* out = a.pp + (a.k & 1);
    const double* __restrict__ s_in = a.sbuf[(a.k + 1) & 1];
    double* __restrict__ s_out = a.sbuf[a.k & 1];
    // ---- every load that does not depend on the decision goes out first ----------------------------------------------------------
    const int done_in = in->done, nfix = in->nfix, nh_in = in->n_hmul, bp_in = in->breakpoints, err_in = in->err;
    LaneBatch<8> b0, b1;
    b0.issue(a.part_in, a.Gin); b1.issue(a.part_in + a.Gin, a.Gin);
    const int64_t r0 = (int64_t)blockIdx.x * CA_T + tid;
    const int64_t rc = min(r0, a.nrows > 0 ? a.nrows - 1 : (int64_t)0);
    const double td0 = a.td[rc], ts0 = a.ts[rc];
    double sv[E], gv[E], lv[E], uv[E];
    int fv[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int i = min(tid + k * CA_T, a.n - 1);
        sv[k] = s_in[i]; gv[k] = a.g[i]; lv[k] = a.dl[i]; uv[k] = a.du[i]; fv[k] = a.fixpass[i];
    }
    if (done_in) {                                   // over-launched: hand the final record on, touch nothing else
        if (blockIdx.x == 0 && tid == 0) { CauchyPass z = *in; *out = z; }
        return;
    }
    // ---- decision k-1, by every workgroup (cauchy_advance_kernel's arithmetic, element for element) ---------------------------
    double gd = 0.0, th = INF, dbest = 0.0;
    int ind = 0x7fffffff;
    double s0[E], d0[E];
    for (int base = 0; base < a.n; base += E * CA_T) {
        if (base > 0) {
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const int i = min(base + tid + k * CA_T, a.n - 1);
                sv[k] = s_in[i]; gv[k] = a.g[i]; lv[k] = a.dl[i]; uv[k] = a.du[i]; fv[k] = a.fixpass[i];
            }
        }
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int i = base + tid + k * CA_T;
            const bool fixed = fv[k] < a.k;
            const double di = fixed ? 0.0 : -gv[k];                   // d = projection(lincons, -g), box constraints (:592 / :632)
            if (base == 0) { s0[k] = sv[k]; d0[k] = di; }
            if (i >= a.n) continue;
            const double si = sv[k];
            gd = fma(gv[k], di, gd);
            if (!fixed) {                                             // :547
                double t = INF;
                if (di < 0.0) t = __ddiv_rn(__dsub_rn(lv[k], si), di);       // :549
                else if (di > 0.0) t = __ddiv_rn(__dsub_rn(uv[k], si), di);  // :551
                if (t < th) { th = t; ind = i; dbest = di; }          // strict <: first minimiser in index order (:555)
            }
        }
    }
    gd = wave_reduce(gd, OpSum());
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double t2 = __shfl_xor(th, off), d2 = __shfl_xor(dbest, off);
        const int i2 = __shfl_xor(ind, off);
        if (t2 < th || (t2 == th && i2 < ind)) { th = t2; ind = i2; dbest = d2; }
    }
    if (lane == 0) { scratch[wave] = gd; scratch[NW + wave] = th; scratch[2 * NW + wave] = dbest; iscratch[wave] = ind; }
    __syncthreads();
    gd = 0.0;
    for (int w = 0; w < NW; ++w) gd += scratch[w];
    th = scratch[NW]; dbest = scratch[2 * NW]; ind = iscratch[0];
    for (int w = 1; w < NW; ++w) {
        const double t2 = scratch[NW + w];
        const int i2 = iscratch[w];
        if (t2 < th || (t2 == th && i2 < ind)) { th = t2; ind = i2; dbest = scratch[2 * NW + w]; }
    }
    const double shd = wave_sum(b0.fold_sum(a.part_in, a.Gin)), dhd = wave_sum(b1.fold_sum(a.part_in + a.Gin, a.Gin));
    if (ind == 0x7fffffff) ind = -1;                              // :544
    const double phi_p = __dadd_rn(shd, gd);                      // :610 / :634
    const double phi_pp = dhd;                                    // :611 / :635
    int done = 0, err = 0, advance = 0;
    double step = 0.0;
    const double delta_t = (phi_pp > 0.0) ? __ddiv_rn(-phi_p, phi_pp) : 0.0;     // :618
    if (!(nfix < a.nmm)) {                                        // :615
        done = 1;
    } else if (phi_p >= 0.0) {                                    // :620
        done = 1;
    } else if (phi_p < 0.0 && phi_pp > 0.0 && delta_t < th) {     // :622
        step = delta_t; done = 1;                                 // :625
    } else {                                                      // :627
        if (ind < 0) { err = 1; done = 1; }
        else { step = th; advance = 1; }                          // :628
    }
    // ---- s_c += step d (:625 / :628): the groups of 64 elements this workgroup owns ----------------------------------------------
    const bool move = (step != 0.0 || advance);
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int i = tid + k * CA_T;
        if (i < a.n && ((i >> 6) % (int)gridDim.x) == (int)blockIdx.x)
            s_out[i] = move ? __dadd_rn(s0[k], __dmul_rn(step, d0[k])) : s0[k];
    }
    for (int i = E * CA_T + tid; i < a.n; i += CA_T) {
        if (((i >> 6) % (int)gridDim.x) != (int)blockIdx.x) continue;
        const double di = (a.fixpass[i] < a.k) ? 0.0 : -a.g[i];
        const double si = s_in[i];
        s_out[i] = move ? __dadd_rn(si, __dmul_rn(step, di)) : si;
    }
    if (blockIdx.x == 0 && tid == 0) {
        CauchyPass z{};
        z.done = done; z.nfix = nfix + advance; z.n_hmul = nh_in + 1; z.breakpoints = bp_in + advance; z.err = err_in | err;
        *out = z;
        if (advance) { a.fixpass[ind] = a.k; a.fixrank[ind] = 0; }    // add_active!: fixvars[ind] = true (poly:246)
        if (a.mirror != nullptr) {
            const unsigned long long wv = ((unsigned long long)(a.tag & 0xffffu) << 48) | ((unsigned long long)(z.err & 0xf) << 44) |
                                          ((unsigned long long)(done & 0xf) << 40) | ((unsigned long long)(z.breakpoints & 0xfffff) << 20) |
                                          (unsigned long long)(z.n_hmul & 0xfffff);
            __hip_atomic_store(a.mirror, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (done) return;
    // ---- this workgroup's rows: t_s += theta t_d, t_d -= d_ind J~[:, ind], and the partial sums for the next decision -------------
    double acc[2] = {0.0, 0.0};
    for (int64_t i = r0; i < a.nrows; i += (int64_t)gridDim.x * CA_T) {
        double td = (i == r0) ? td0 : a.td[i];
        const double ts = __dadd_rn((i == r0) ? ts0 : a.ts[i], __dmul_rn(step, td));   // s_c += theta d     (:628)
        td = __dsub_rn(td, __dmul_rn(dbest, a.J[i * a.ld + ind]));                   // d[ind] = 0         (:632, box)
        a.td[i] = td;
        a.ts[i] = ts;
        const double w = (i < a.d_rows) ? 1.0 : a.mu;
        acc[0] = fma(w * ts, td, acc[0]);
        acc[1] = fma(w * td, td, acc[1]);
    }
    block_reduce<CA_T, 2>(acc, rscratch, OpSum(), 0.0);
    if (tid == 0) { a.part_out[blockIdx.x] = acc[0]; a.part_out[gridDim.x + blockIdx.x] = acc[1]; }
}

// M <- M - a a',  a = column `ind` of A (the variable that just became fixed):  A_free A_free' after add_active!.
__global__ __launch_bounds__(256) void gram_downdate_kernel(double* __restrict__ M, const double* __restrict__ A, int64_t ldA, int mA,
                                                            const CgState* st) {
    if (st->done) return;
    const int ind = st->status;
    if (ind < 0) return;
    const int64_t total = (int64_t)mA * mA;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % mA), k = (int)(e / mA);
        if (i >= k) M[e] = fma(-A[(int64_t)i * ldA + ind], A[(int64_t)k * ldA + ind], M[e]);
    }
}

}  // namespace bh
