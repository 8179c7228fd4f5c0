// bh_kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels for the BEnlsip hot path.
//
// Data layout in HBM: the Jacobian block of this rank is stored ROW-MAJOR,
// rows padded to a multiple of 16 doubles (128 B): Jd[i*ld + j].  The host hands J over
// column-major (Julia); bh_hess_create transposes it once on the device.  Rows [d, d+q) of
// the same image hold C, so H*p = J'(Jp) + C'(mu C p) is ONE sweep with a per-row weight.
// Why row-major: n (<= 16384) is small enough that a whole row of J lives in the registers
// of one workgroup (n/T doubles per lane), so J'(Jp) needs ONE read of J: the row is dotted
// with p (wave64 DPP + permlane reduction, cross-wave through LDS), then scaled by that dot
// and accumulated into a register-resident slice of z — 8*d*n bytes instead of 16*d*n.
//
// Reference call sites: src/basic_tralcnlss.jl:92-106 (vthv, *), :690-764 (projected_cg),
// :793-809 (factor_to_boundary); src/polyhedral_constraints.jl:72-136 (left_mul, left_mul_tr,
// projection_nullspace!, projection_subspace!).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bh {

typedef double dvec2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// wave64 reductions: DPP inside a 16-lane row, v_permlane16_swap / v_permlane32_swap across
// rows (gfx950).  Butterfly form: every lane ends with the same bits (a+b == b+a).
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

struct OpSum { __device__ __forceinline__ double operator()(double a, double b) const { return a + b; } };
// Julia's min(): NaN-propagating (src/basic_tralcnlss.jl:803,805 use min(gamma, ...)).
struct OpMinNan {
    __device__ __forceinline__ double operator()(double a, double b) const {
        return (a != a) ? a : ((b != b) ? b : (a < b ? a : b));
    }
};

template <class Op>
__device__ __forceinline__ double wave_reduce(double x, Op op) {
    x = op(x, dpp_mov_f64<0xB1>(x));   // quad_perm [1,0,3,2]   lane ^ 1
    x = op(x, dpp_mov_f64<0x4E>(x));   // quad_perm [2,3,0,1]   lane ^ 2
    x = op(x, dpp_mov_f64<0x141>(x));  // row_half_mirror       7 - lane (mod 8)
    x = op(x, dpp_mov_f64<0x140>(x));  // row_mirror            15 - lane (mod 16)
    {   // rows 0<->1, 2<->3
        unsigned lo = __double2loint(x), hi = __double2hiint(x);
        auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        x = op(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
    }
    {   // halves 0<->1
        unsigned lo = __double2loint(x), hi = __double2hiint(x);
        auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        x = op(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
    }
    return x;
}
__device__ __forceinline__ double wave_sum(double x) { return wave_reduce(x, OpSum()); }
__device__ __forceinline__ double wave_min(double x) { return wave_reduce(x, OpMinNan()); }

// Block-wide reduction of NV values at once; fixed combination order -> bit-reproducible.
// `scratch` holds NV * (T/64) doubles.  Every thread returns the same totals.
template <int T, int NV, class Op>
__device__ __forceinline__ void block_reduce(double (&x)[NV], double* scratch, Op op, double identity) {
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = wave_reduce(x[i], op);
    if (NW == 1) return;
    __syncthreads();   // scratch may still be read from a previous use
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[i * NW + wave] = x[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double t = identity;
        for (int w = 0; w < NW; ++w) t = op(t, scratch[i * NW + w]);
        x[i] = t;
    }
}

// ------------------------------------------------------------------------------------------
// Device-resident CG state (one per bh_pcg call).  Mirrors the locals of projected_cg,
// src/basic_tralcnlss.jl:702-718.
// ------------------------------------------------------------------------------------------
struct CgState {
    double rtv, tol_cg, pHp, alpha, gamma, beta;
    int iter;        // reference `iter` (starts at 1, :713)
    int max_iter;    // 2*(n - mA - nfix), :714
    int approx_solved, outside_region, neg_curvature;   // :716-718
    int done;        // loop condition :720 is false
    int status;      // BH_CG_*
    int n_hmul;      // H*p products performed
    int need_proj;   // general path: step_a decided to continue -> projection + step_b run
    int pad;
};

// ------------------------------------------------------------------------------------------
// Row-streaming kernel: J·v, J'·u and the fused single-read J'(W ∘ (J v)).
//   T   threads per workgroup, CPT 16-byte chunks (2 doubles) per thread per row, R rows per step.
//   A workgroup owns row groups g = blockIdx.x, +gridDim.x, ... (the whole grid marches through
//   HBM together, like a copy); the next group's loads are issued before the current group's
//   reduction so they stay in flight across the barrier.
// ------------------------------------------------------------------------------------------
enum { MODE_JV = 0, MODE_JTV = 1, MODE_FUSED = 2 };

struct RowStreamArgs {
    const double* J;        // row-major image, (nrows) x ld
    int64_t ld;             // doubles per row (multiple of 16)
    int64_t nrows;          // rows swept by this launch
    int64_t d_rows;         // rows [0,d_rows) have weight 1, rows >= d_rows weight mu (the C block)
    int nchunks;            // ld / 2
    const double* v;        // n_pad doubles (JV, FUSED)
    const double* u;        // nrows doubles (JTV)
    double* t_out;          // nrows doubles or NULL (JV)
    double* partials;       // gridDim.x x ld  (JTV, FUSED)
    double* sq_partials;    // gridDim.x or NULL (JV: sum_i weight_i * t_i^2, for vthv)
    double mu;
    const CgState* state;   // NULL, or skip the launch when state->done
    int reverse;            // sweep the row groups last-to-first (ping-pong order keeps the tail of J in the Infinity Cache)
    int accumulate;         // JV: t_out += (column panels of a wide J are swept one launch each)
    int weighted_u;         // JTV: coefficient u[row] * (row < d_rows ? 1 : mu)  (second pass of the two-pass H*p)
    int negate;             // JV/FUSED: use -mask(v) instead of v (first CG iteration: p0 = -P(g) for box constraints, :706-708)
    const int* negmask;     // fixrank (>= 0: fixed -> 0) or NULL, with negate
};

// NT: J is read exactly once per launch -> non-temporal loads (global_load_dwordx4 ... nt): measured +10 % (6.39 -> 7.05 TB/s).
// PF: 1 = issue the next row group's loads before reducing the current one (two register buffers); 0 = one buffer, latency
// hidden by several co-resident workgroups instead.
template <int T, int CPT, int R, int MODE, int NT = 1, int PF = 1>
__global__ __launch_bounds__(T) void row_stream_kernel(RowStreamArgs a) {
    if (a.state != nullptr && a.state->done) return;
    constexpr int NW = T / 64;
    __shared__ double red[2][R][NW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t ld2 = a.ld >> 1;   // row stride in double2
    const double2* __restrict__ J2 = reinterpret_cast<const double2*>(a.J);

    bool act[CPT];
    double2 vv[CPT], zz[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = tid + k * T;
        act[k] = c < a.nchunks;
        vv[k] = make_double2(0.0, 0.0);
        zz[k] = make_double2(0.0, 0.0);
        if (MODE != MODE_JTV && act[k]) {
            vv[k] = reinterpret_cast<const double2*>(a.v)[c];
            if (a.negate) {
                int2 f = make_int2(-1, -1);
                if (a.negmask != nullptr) f = reinterpret_cast<const int2*>(a.negmask)[c];
                vv[k].x = (f.x >= 0) ? 0.0 : -vv[k].x;
                vv[k].y = (f.y >= 0) ? 0.0 : -vv[k].y;
            }
        }
    }

    const int64_t ngroups = (a.nrows + R - 1) / R;
    const int64_t G = gridDim.x;
    double sq_acc = 0.0;
    int buf = 0;

    double2 A[R][CPT], B[R][CPT];

    auto load_group = [&](double2 (&dst)[R][CPT], int64_t grp) {
        if (a.reverse) grp = ngroups - 1 - grp;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = grp * R + r;
            const bool rv = row < a.nrows;
            const double2* rp = J2 + (rv ? row : 0) * ld2;
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                dst[r][k] = make_double2(0.0, 0.0);
                if (rv && act[k]) {
                    if (NT) {
                        const dvec2 t = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(rp + tid + k * T));
                        dst[r][k] = make_double2(t.x, t.y);
                    } else {
                        dst[r][k] = rp[tid + k * T];
                    }
                }
            }
        }
    };

    auto process = [&](double2 (&X)[R][CPT], int64_t grp) {
        if (a.reverse) grp = ngroups - 1 - grp;
        double s[R];
        if (MODE != MODE_JTV) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    acc = fma(X[r][k].x, vv[k].x, acc);
                    acc = fma(X[r][k].y, vv[k].y, acc);
                }
                s[r] = wave_sum(acc);
            }
            if (NW > 1) {
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < R; ++r) red[buf][r][wave] = s[r];
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    double t = 0.0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) t += red[buf][r][w];
                    s[r] = t;
                }
                buf ^= 1;
            }
        }
        if (MODE == MODE_JV && a.t_out != nullptr) {
            // the R results of the group leave in ONE store instruction (lanes 0..R-1 of wave 0, R*8 contiguous bytes)
            double mine = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (tid == r) mine = s[r];
            const int64_t row = grp * R + tid;
            if (tid < R && row < a.nrows) a.t_out[row] = a.accumulate ? a.t_out[row] + mine : mine;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = grp * R + r;
            const bool rv = row < a.nrows;
            if (MODE == MODE_JV) {
                if (rv) {
                    const double wgt = (row < a.d_rows) ? 1.0 : a.mu;
                    sq_acc = fma(wgt * s[r], s[r], sq_acc);
                }
            } else {
                double coef;
                if (MODE == MODE_JTV) coef = rv ? (a.weighted_u && row >= a.d_rows ? a.mu * a.u[row] : a.u[row]) : 0.0;
                else coef = (row < a.d_rows) ? s[r] : a.mu * s[r];
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    zz[k].x = fma(coef, X[r][k].x, zz[k].x);
                    zz[k].y = fma(coef, X[r][k].y, zz[k].y);
                }
            }
        }
    };

    int64_t g = blockIdx.x;
    if (!PF) {
        for (; g < ngroups; g += G) {
            load_group(A, g);
            process(A, g);
        }
    } else if (g < ngroups) {
        load_group(A, g);
        while (true) {
            int64_t gn = g + G;
            if (gn < ngroups) load_group(B, gn);
            process(A, g);
            if (gn >= ngroups) break;
            g = gn;
            gn = g + G;
            if (gn < ngroups) load_group(A, gn);
            process(B, g);
            if (gn >= ngroups) break;
            g = gn;
        }
    }

    if (MODE == MODE_JV) {
        if (a.sq_partials != nullptr && tid == 0) a.sq_partials[blockIdx.x] = sq_acc;
    } else {
        double2* out = reinterpret_cast<double2*>(a.partials) + (int64_t)blockIdx.x * ld2;
#pragma unroll
        for (int k = 0; k < CPT; ++k)
            if (act[k]) out[tid + k * T] = zz[k];
    }
}

// Sum the G partial rows written by row_stream_kernel into out (fixed order).
// Block = 256 threads = 16 chunks x 16 row-lanes; grid = ceil(nchunks/16).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ partials, int64_t ld,
                                                              int nchunks, int G, double* __restrict__ out,
                                                              const CgState* state) {
    if (state != nullptr && state->done) return;
    __shared__ double2 sm[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int64_t ld2 = ld >> 1;
    const double2* P2 = reinterpret_cast<const double2*>(partials);
    double2 acc = make_double2(0.0, 0.0);
    if (c < nchunks) {
        int g = rl;
        for (; g + 48 < G; g += 64) {
            const double2 x0 = P2[(int64_t)g * ld2 + c];
            const double2 x1 = P2[(int64_t)(g + 16) * ld2 + c];
            const double2 x2 = P2[(int64_t)(g + 32) * ld2 + c];
            const double2 x3 = P2[(int64_t)(g + 48) * ld2 + c];
            acc.x += x0.x; acc.y += x0.y;
            acc.x += x1.x; acc.y += x1.y;
            acc.x += x2.x; acc.y += x2.y;
            acc.x += x3.x; acc.y += x3.y;
        }
        for (; g < G; g += 16) {
            const double2 x0 = P2[(int64_t)g * ld2 + c];
            acc.x += x0.x; acc.y += x0.y;
        }
    }
    sm[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < nchunks) {
        double2 t = sm[0][cl];
#pragma unroll
        for (int r = 1; r < 16; ++r) { t.x += sm[r][cl].x; t.y += sm[r][cl].y; }
        reinterpret_cast<double2*>(out)[c] = t;
    }
}

// Sum m doubles (single workgroup) into out[0]; used for the vthv scalar.
__global__ __launch_bounds__(256) void reduce_scalar_kernel(const double* __restrict__ x, int m, double* out) {
    __shared__ double scratch[4];
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < m; i += 256) acc[0] += x[i];
    block_reduce<256, 1>(acc, scratch, OpSum(), 0.0);
    if (threadIdx.x == 0) out[0] = acc[0];
}

// sum_i w_i t_i^2 with w_i = 1 (i < d_rows) or mu: vthv for a J swept in column panels.  Single workgroup.
__global__ __launch_bounds__(1024) void weighted_sqsum_kernel(const double* __restrict__ t, int64_t nrows, int64_t d_rows, double mu,
                                                               double* __restrict__ out) {
    __shared__ double scratch[1024 / 64];
    double acc[1] = {0.0};
    for (int64_t i = threadIdx.x; i < nrows; i += 1024) {
        const double ti = t[i];
        acc[0] = fma((i < d_rows) ? ti : mu * ti, ti, acc[0]);
    }
    block_reduce<1024, 1>(acc, scratch, OpSum(), 0.0);
    if (threadIdx.x == 0) out[0] = acc[0];
}

// Column-major (host layout, leading dimension lds) -> row-major padded image.  32x32 tiles via LDS.
__global__ __launch_bounds__(256) void transpose_cm_to_rm_kernel(const double* __restrict__ src, int64_t lds_, int64_t rows,
                                                                 int64_t cols, double* __restrict__ dst, int64_t ldd) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int64_t r0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t c = c0 + ty + 8 * k, r = r0 + tx;
        tile[ty + 8 * k][tx] = (r < rows && c < cols) ? src[r + c * lds_] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < rows && c < ldd) dst[r * ldd + c] = (c < cols) ? tile[tx][ty + 8 * k] : 0.0;
    }
}

// Synthetic Jacobian of SURVEY.md §8(d), generated in place (row-major, padded columns = 0).
__device__ __forceinline__ double splitmix_uniform(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return __dsub_rn(__dmul_rn((double)(z >> 11), 2.0 / 9007199254740992.0), 1.0);
}

__global__ __launch_bounds__(256) void synth_fill_kernel(double* __restrict__ dst, int64_t ldd, int64_t rows, int64_t n,
                                                         int64_t row0, int64_t d_total, uint64_t seed,
                                                         const double* __restrict__ colscale, double divisor) {
    const int64_t total = rows * ldd;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t i = idx / ldd, j = idx - i * ldd;
        double val = 0.0;
        if (j < n) {
            val = __ddiv_rn(splitmix_uniform(seed, (uint64_t)(row0 + i) + (uint64_t)j * (uint64_t)d_total), divisor);
            if (colscale != nullptr) val = __dmul_rn(val, colscale[j]);
        }
        dst[idx] = val;
    }
}

// ------------------------------------------------------------------------------------------
// CG vector kernels (single workgroup of 1024 threads: n-vectors are 32 KiB at n = 4096, the
// whole step is latency- not bandwidth-bound; one workgroup avoids any grid-level exchange).
// Element-wise updates use separately rounded mul/add like the reference's broadcasts
// (src/basic_tralcnlss.jl:729,737,739,740,745); dots use fma like BLAS ddot.
// ------------------------------------------------------------------------------------------
constexpr int CG_T = 1024;

struct CgArgs {
    CgState* st;
    double* w; double* r; double* v; double* p;
    const double* Hp;
    const double* g;        // init only
    const double* wl; const double* wu;
    const int* fixrank;     // -1 free, else rank among fixed variables (NULL = nothing fixed)
    int n;
    int n_pad;              // length of the workspace vectors; init zeroes [n, n_pad) (the workspace is reused across calls)
    int w_in_ws;            // w points into the padded workspace (else: caller's buffer of exactly n doubles)
    int max_iter;
    double kappa2, atol_neg, atol_f2b;
    double* trace; int trace_cap;
    double* hw;                   // NULL, or H*w accumulated alongside w (hw += step*Hp): lets minor_iterate's linesearch form
                                  // w'Hw = w.hw without another sweep over J (src/basic_tralcnlss.jl:775 calls vthv(H,w))
    unsigned long long* mirror;   // host-mapped word the host polls instead of copying CgState back (NULL: none)
    unsigned tag;                 // per-call tag stored in the mirror's top 16 bits
};

__device__ __forceinline__ double f2b_term(double p, double w, double wl, double wu, double atol) {
    // src/basic_tralcnlss.jl:802-806
    double g = __longlong_as_double(0x7ff0000000000000ll);   // +Inf
    if (p <= -atol) g = __ddiv_rn(__dsub_rn(wl, w), p);
    else if (p >= atol) g = __ddiv_rn(__dsub_rn(wu, w), p);
    return g;
}

__device__ __forceinline__ int cg_final_status(const CgState* st) {
    // src/basic_tralcnlss.jl:753-761
    if (st->approx_solved) return 0;
    if (st->outside_region) return 1;
    if (st->neg_curvature) return 2;
    if (st->iter == st->max_iter) return 3;
    return 4;
}

// One 8-byte system-scope store to host-mapped memory: [tag:16 | status:4 | done:4 | iter:20 | n_hmul:20].  A single
// naturally aligned word cannot tear, so the host needs no ordering beyond reading it.
__device__ __forceinline__ void publish_state(const CgArgs& a, const CgState* st) {
    if (a.mirror == nullptr) return;
    const unsigned long long wv = ((unsigned long long)(a.tag & 0xffffu) << 48) | ((unsigned long long)(st->status & 0xf) << 44) |
                                  ((unsigned long long)(st->done & 0xf) << 40) | ((unsigned long long)(st->iter & 0xfffff) << 20) |
                                  (unsigned long long)(st->n_hmul & 0xfffff);
    __hip_atomic_store(a.mirror, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// w = 0; r = g  (:702-705).  With BOX: v = mask(r), then the tail of cg_init_finish.
template <bool BOX>
__global__ __launch_bounds__(CG_T) void cg_init_kernel(CgArgs a) {
    __shared__ double scratch[2 * (CG_T / 64)];
    double acc[2] = {0.0, 0.0};
    for (int i = a.n + threadIdx.x; i < a.n_pad; i += CG_T) {     // stale padding from an earlier, larger problem
        a.r[i] = 0.0; a.v[i] = 0.0; a.p[i] = 0.0;
        if (a.w_in_ws) a.w[i] = 0.0;
        if (a.hw != nullptr) a.hw[i] = 0.0;
    }
    for (int i = threadIdx.x; i < a.n; i += CG_T) {
        const double ri = a.g[i];
        a.r[i] = ri;
        a.w[i] = 0.0;
        if (a.hw != nullptr) a.hw[i] = 0.0;
        if (BOX) {
            const double vi = (a.fixrank != nullptr && a.fixrank[i] >= 0) ? 0.0 : ri;
            a.v[i] = vi;
            a.p[i] = -vi;
            acc[0] = fma(ri, vi, acc[0]);
            acc[1] = fma(vi, vi, acc[1]);
        }
    }
    if (BOX) {
        block_reduce<CG_T, 2>(acc, scratch, OpSum(), 0.0);
        if (threadIdx.x == 0) {
            CgState* st = a.st;
            st->rtv = acc[0];                       // :707
            st->tol_cg = a.kappa2 * sqrt(acc[1]);   // :710
            st->pHp = 0.0; st->alpha = 0.0; st->gamma = 0.0; st->beta = 0.0;
            st->iter = 1; st->max_iter = a.max_iter;
            st->approx_solved = 0; st->outside_region = 0; st->neg_curvature = 0;
            st->n_hmul = 0; st->need_proj = 0;
            st->done = (1 <= a.max_iter) ? 0 : 1;   // :720
            st->status = cg_final_status(st);
            publish_state(a, st);
        }
    } else if (threadIdx.x == 0) {
        a.st->done = 0; a.st->need_proj = 1;
    }
}

// General path, after v = P(r):  rtv = r.v ; p = -v ; tol_cg = kappa2*||v||  (:707-710).
__global__ __launch_bounds__(CG_T) void cg_init_finish_kernel(CgArgs a) {
    __shared__ double scratch[2 * (CG_T / 64)];
    double acc[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < a.n; i += CG_T) {
        const double ri = a.r[i], vi = a.v[i];
        a.p[i] = -vi;
        acc[0] = fma(ri, vi, acc[0]);
        acc[1] = fma(vi, vi, acc[1]);
    }
    block_reduce<CG_T, 2>(acc, scratch, OpSum(), 0.0);
    if (threadIdx.x == 0) {
        CgState* st = a.st;
        st->rtv = acc[0];
        st->tol_cg = a.kappa2 * sqrt(acc[1]);
        st->pHp = 0.0; st->alpha = 0.0; st->gamma = 0.0; st->beta = 0.0;
        st->iter = 1; st->max_iter = a.max_iter;
        st->approx_solved = 0; st->outside_region = 0; st->neg_curvature = 0;
        st->n_hmul = 0; st->need_proj = 0;
        st->done = (1 <= a.max_iter) ? 0 : 1;
        st->status = cg_final_status(st);
        publish_state(a, st);
    }
}

// One pass of the loop body, src/basic_tralcnlss.jl:722-750.
//   PHASE 0 (box, fused): everything, projection = mask.
//   PHASE 1 (general, step_a): pHp, gamma, branch, w/r update; sets need_proj.
//   PHASE 2 (general, step_b): rtv_next, beta, p, exit test (after v = P(r)).
template <int PHASE>
__global__ __launch_bounds__(CG_T) void cg_step_kernel(CgArgs a) {
    __shared__ double scratch[2 * (CG_T / 64)];
    __shared__ int s_continue;
    CgState* st = a.st;
    if (st->done) return;
    const int tid = threadIdx.x;
    const double INF = __longlong_as_double(0x7ff0000000000000ll);

    double alpha = 0.0, rtv = st->rtv;

    if (PHASE != 2) {
        // pHp = dot(p,Hp) (:723) and gamma = factor_to_boundary(p,w,w_l,w_u) (:728,:734)
        double sum[1] = {0.0};
        double gmin[1] = {INF};
        OpMinNan opmin;
        for (int i = tid; i < a.n; i += CG_T) {
            const double pi = a.p[i];
            sum[0] = fma(pi, a.Hp[i], sum[0]);
            gmin[0] = opmin(gmin[0], f2b_term(pi, a.w[i], a.wl[i], a.wu[i], a.atol_f2b));
        }
        block_reduce<CG_T, 1>(sum, scratch, OpSum(), 0.0);
        block_reduce<CG_T, 1>(gmin, scratch, opmin, INF);
        const double pHp = sum[0], gamma = gmin[0];

        int cont = 0;        // 1: CG update (:739-748) follows
        double step = 0.0;   // multiple of p added to w
        int neg = 0, outside = 0;
        if (pHp <= a.atol_neg) {                    // :725
            neg = 1;
            if (fabs(pHp) > a.atol_neg) step = gamma;   // :727-729
            else step = 0.0;
        } else {
            alpha = __ddiv_rn(rtv, pHp);            // :733  (rtv == dot(r,v) bit for bit: deterministic dot)
            outside = alpha > gamma;                // :735
            if (outside) step = gamma;              // :737
            else { step = alpha; cont = 1; }        // :739
        }
        const bool add_w = !(neg && !(fabs(pHp) > a.atol_neg));
        if (add_w) {
            for (int i = tid; i < a.n; i += CG_T) a.w[i] = __dadd_rn(a.w[i], __dmul_rn(step, a.p[i]));
            if (a.hw != nullptr)
                for (int i = tid; i < a.n; i += CG_T) a.hw[i] = __dadd_rn(a.hw[i], __dmul_rn(step, a.Hp[i]));
        }
        if (tid == 0) {
            st->pHp = pHp; st->gamma = gamma; st->alpha = (pHp <= a.atol_neg) ? __longlong_as_double(0x7ff8000000000000ll) : alpha;
            st->n_hmul += 1;
            st->neg_curvature = neg; st->outside_region = outside;
            if (!cont) {
                st->done = 1; st->need_proj = 0;
                st->status = cg_final_status(st);
                if (a.trace != nullptr && st->n_hmul <= a.trace_cap) {
                    double* row = a.trace + 4 * (int64_t)(st->n_hmul - 1);
                    row[0] = pHp; row[1] = st->alpha; row[2] = (neg && !add_w) ? __longlong_as_double(0x7ff8000000000000ll) : gamma; row[3] = rtv;
                }
                publish_state(a, st);
            } else {
                st->need_proj = 1;
            }
            s_continue = cont;
        }
        __syncthreads();
        if (!s_continue) return;
        // r .+= alpha*Hp  (:740)
        if (PHASE == 0) {
            double acc[1] = {0.0};
            for (int i = tid; i < a.n; i += CG_T) {
                const double ri = __dadd_rn(a.r[i], __dmul_rn(alpha, a.Hp[i]));
                a.r[i] = ri;
                const double vi = (a.fixrank != nullptr && a.fixrank[i] >= 0) ? 0.0 : ri;   // projection!, box case (:741)
                a.v[i] = vi;
                acc[0] = fma(ri, vi, acc[0]);       // :743
            }
            block_reduce<CG_T, 1>(acc, scratch, OpSum(), 0.0);
            const double rtv_next = acc[0];
            const double beta = __ddiv_rn(rtv_next, rtv);       // :744
            for (int i = tid; i < a.n; i += CG_T)
                a.p[i] = __dadd_rn(-a.v[i], __dmul_rn(beta, a.p[i]));   // :745
            if (tid == 0) {
                st->beta = beta; st->rtv = rtv_next;            // :746
                st->approx_solved = fabs(rtv_next) < st->tol_cg;   // :747
                st->iter += 1;                                  // :748
                st->need_proj = 0;
                if (st->approx_solved || st->iter > st->max_iter) { st->done = 1; st->status = cg_final_status(st); }
                if (a.trace != nullptr && st->n_hmul <= a.trace_cap) {
                    double* row = a.trace + 4 * (int64_t)(st->n_hmul - 1);
                    row[0] = st->pHp; row[1] = alpha; row[2] = st->gamma; row[3] = rtv_next;
                }
                publish_state(a, st);
            }
        } else {
            for (int i = tid; i < a.n; i += CG_T) a.r[i] = __dadd_rn(a.r[i], __dmul_rn(alpha, a.Hp[i]));
        }
    } else {
        if (!st->need_proj) return;
        alpha = st->alpha;
        double acc[1] = {0.0};
        for (int i = tid; i < a.n; i += CG_T) acc[0] = fma(a.r[i], a.v[i], acc[0]);
        block_reduce<CG_T, 1>(acc, scratch, OpSum(), 0.0);
        const double rtv_next = acc[0];
        const double beta = __ddiv_rn(rtv_next, rtv);
        for (int i = tid; i < a.n; i += CG_T) a.p[i] = __dadd_rn(-a.v[i], __dmul_rn(beta, a.p[i]));
        if (tid == 0) {
            st->beta = beta; st->rtv = rtv_next;
            st->approx_solved = fabs(rtv_next) < st->tol_cg;
            st->iter += 1;
            st->need_proj = 0;
            if (st->approx_solved || st->iter > st->max_iter) { st->done = 1; st->status = cg_final_status(st); }
            if (a.trace != nullptr && st->n_hmul <= a.trace_cap) {
                double* row = a.trace + 4 * (int64_t)(st->n_hmul - 1);
                row[0] = st->pHp; row[1] = alpha; row[2] = st->gamma; row[3] = rtv_next;
            }
            publish_state(a, st);
        }
    }
}

// Register-resident forms of cg_step_kernel<PHASE> for n <= 2*CG_T*CH: every element a thread owns is loaded ONCE with
// 16-byte loads that are all in flight together, the loop body (src/basic_tralcnlss.jl:722-750) runs out of registers
// with at most two block reductions, and results are stored once.  One HBM/L2 round trip instead of four.
//   PHASE 0: box constraints, everything fused (projection = mask).
//   PHASE 1: general constraints, step_a (pHp, gamma, branch, w and r updates; sets need_proj).
//   PHASE 2: general constraints, step_b after v = P(r) (rtv_next, beta, p, exit test).
//   FIRST (PHASE 0 only): the first pass also does the initialisation of projected_cg (:702-718: w = 0, r = g,
//   v = P(r), rtv, p = -v, tol_cg) — no separate init kernel; the preceding H*p launch forms p0 = -mask(g) on the fly.
template <int CH, int PHASE, bool FIRST = false>
__global__ __launch_bounds__(CG_T) void cg_step_reg_kernel(CgArgs a) {
    constexpr int NW = CG_T / 64;
    __shared__ double scratch[4 * NW];
    CgState* st = a.st;
    if (!FIRST && st->done) return;
    if (PHASE == 2 && !st->need_proj) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    const double QNAN = __longlong_as_double(0x7ff8000000000000ll);
    const int nch = (a.n + 1) >> 1;
    double rtv = FIRST ? 0.0 : st->rtv, tol_cg = FIRST ? 0.0 : st->tol_cg;
    const int iter0 = FIRST ? 1 : st->iter, max_iter = FIRST ? a.max_iter : st->max_iter, n_hmul0 = FIRST ? 0 : st->n_hmul;

    bool act[CH];
    double2 p[CH], hp[CH], w[CH], wl[CH], wu[CH], r[CH], v[CH];
    int2 fr[CH];
    if (FIRST) {   // stale padding from an earlier, larger problem (the workspace is shared by all calls)
        for (int i = a.n + tid; i < a.n_pad; i += CG_T) {
            a.r[i] = 0.0; a.v[i] = 0.0; a.p[i] = 0.0;
            if (a.w_in_ws) a.w[i] = 0.0;
            if (a.hw != nullptr) a.hw[i] = 0.0;
        }
    }
    double2 hw[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int c = tid + k * CG_T;
        act[k] = c < nch;
        hw[k] = make_double2(0.0, 0.0);
        if (PHASE != 2 && !FIRST && a.hw != nullptr && act[k]) hw[k] = reinterpret_cast<const double2*>(a.hw)[c];
        p[k] = hp[k] = w[k] = wl[k] = wu[k] = r[k] = v[k] = make_double2(0.0, 0.0);
        fr[k] = make_int2(-1, -1);
        if (act[k]) {
            if (FIRST) {
                r[k] = reinterpret_cast<const double2*>(a.g)[c];          // r = g_minor (:705)
                if ((2 * c + 1) >= a.n) r[k].y = 0.0;                     // odd n: never trust the element past the end
            } else {
                p[k] = reinterpret_cast<const double2*>(a.p)[c];
                r[k] = reinterpret_cast<const double2*>(a.r)[c];
            }
            if (PHASE != 2) {
                hp[k] = reinterpret_cast<const double2*>(a.Hp)[c];
                if (!FIRST) w[k] = reinterpret_cast<const double2*>(a.w)[c];   // w = 0 (:702)
                wl[k] = reinterpret_cast<const double2*>(a.wl)[c];
                wu[k] = reinterpret_cast<const double2*>(a.wu)[c];
            } else {
                v[k] = reinterpret_cast<const double2*>(a.v)[c];
            }
            if (PHASE == 0 && a.fixrank != nullptr) fr[k] = reinterpret_cast<const int2*>(a.fixrank)[c];
        }
    }

    int cont = 0, neg = 0, outside = 0;
    double pHp = 0.0, gamma = INF, step = 0.0, alpha = QNAN;
    bool add_w = true;
    if (PHASE != 2) {
        // pHp = dot(p,Hp) (:723); gamma = factor_to_boundary(p,w,w_l,w_u) (:728,:734).  Padding elements are zeros: no effect.
        OpMinNan opmin;
        double sum = 0.0, gmin = INF, rtv0 = 0.0, vv0 = 0.0;
        if (FIRST) {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                if (a.fixrank != nullptr && act[k]) fr[k] = reinterpret_cast<const int2*>(a.fixrank)[tid + k * CG_T];
                const double v0x = (fr[k].x >= 0) ? 0.0 : r[k].x, v0y = (fr[k].y >= 0) ? 0.0 : r[k].y;   // v = P(r) (:706)
                rtv0 = fma(r[k].x, v0x, rtv0); rtv0 = fma(r[k].y, v0y, rtv0);                            // :707
                vv0 = fma(v0x, v0x, vv0); vv0 = fma(v0y, v0y, vv0);                                      // :710
                p[k].x = -v0x; p[k].y = -v0y;                                                            // :708
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            sum = fma(p[k].x, hp[k].x, sum);
            sum = fma(p[k].y, hp[k].y, sum);
            gmin = opmin(gmin, f2b_term(p[k].x, w[k].x, wl[k].x, wu[k].x, a.atol_f2b));
            gmin = opmin(gmin, f2b_term(p[k].y, w[k].y, wl[k].y, wu[k].y, a.atol_f2b));
        }
        sum = wave_sum(sum);
        gmin = wave_min(gmin);
        if (FIRST) { rtv0 = wave_sum(rtv0); vv0 = wave_sum(vv0); }
        if (lane == 0) {
            scratch[wave] = sum; scratch[NW + wave] = gmin;
            if (FIRST) { scratch[2 * NW + wave] = rtv0; scratch[3 * NW + wave] = vv0; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NW; ++i) { pHp += scratch[i]; gamma = opmin(gamma, scratch[NW + i]); }
        if (FIRST) {
            rtv = 0.0;
            double vv = 0.0;
#pragma unroll
            for (int i = 0; i < NW; ++i) { rtv += scratch[2 * NW + i]; vv += scratch[3 * NW + i]; }
            tol_cg = a.kappa2 * sqrt(vv);               // :710
        }
        __syncthreads();   // scratch is reused below

        if (pHp <= a.atol_neg) {                        // :725
            neg = 1;
            if (fabs(pHp) > a.atol_neg) step = gamma;   // :727-729
            else add_w = false;
        } else {
            alpha = __ddiv_rn(rtv, pHp);                // :733  (rtv == dot(r,v) bit for bit: deterministic dot)
            outside = alpha > gamma;                    // :735
            if (outside) step = gamma;                  // :737
            else { step = alpha; cont = 1; }            // :739
        }
        if (add_w) {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                w[k].x = __dadd_rn(w[k].x, __dmul_rn(step, p[k].x));
                w[k].y = __dadd_rn(w[k].y, __dmul_rn(step, p[k].y));
                hw[k].x = __dadd_rn(hw[k].x, __dmul_rn(step, hp[k].x));    // H*w rides along (a.hw)
                hw[k].y = __dadd_rn(hw[k].y, __dmul_rn(step, hp[k].y));
            }
        }
        if (cont) {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                r[k].x = __dadd_rn(r[k].x, __dmul_rn(alpha, hp[k].x));     // :740
                r[k].y = __dadd_rn(r[k].y, __dmul_rn(alpha, hp[k].y));
                if (PHASE == 0) {
                    v[k].x = (fr[k].x >= 0) ? 0.0 : r[k].x;                // projection!, box case (:741)
                    v[k].y = (fr[k].y >= 0) ? 0.0 : r[k].y;
                }
            }
        }
    } else {
        cont = 1;
        pHp = st->pHp; gamma = st->gamma; alpha = st->alpha;
    }

    double rtv_next = rtv, beta = 0.0;
    if (cont && PHASE != 1) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            acc = fma(r[k].x, v[k].x, acc);                                // :743
            acc = fma(r[k].y, v[k].y, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) scratch[wave] = acc;
        __syncthreads();
        rtv_next = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) rtv_next += scratch[i];
        beta = __ddiv_rn(rtv_next, rtv);                                   // :744
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            p[k].x = __dadd_rn(-v[k].x, __dmul_rn(beta, p[k].x));          // :745
            p[k].y = __dadd_rn(-v[k].y, __dmul_rn(beta, p[k].y));
        }
    }
    // stores (never beyond n: w += Inf*0 would poison the padding)
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int c = tid + k * CG_T;
        if (!act[k]) continue;
        const bool full = (2 * c + 1) < a.n;
        if (full) {
            if (PHASE != 2 && (add_w || FIRST)) reinterpret_cast<double2*>(a.w)[c] = w[k];
            if (PHASE != 2 && (add_w || FIRST) && a.hw != nullptr) reinterpret_cast<double2*>(a.hw)[c] = hw[k];
            if (cont) {
                if (PHASE != 2) reinterpret_cast<double2*>(a.r)[c] = r[k];
                if (PHASE == 0) reinterpret_cast<double2*>(a.v)[c] = v[k];
                if (PHASE != 1) reinterpret_cast<double2*>(a.p)[c] = p[k];
            }
        } else {
            if (PHASE != 2 && (add_w || FIRST)) a.w[2 * c] = w[k].x;
            if (PHASE != 2 && (add_w || FIRST) && a.hw != nullptr) a.hw[2 * c] = hw[k].x;
            if (cont) {
                if (PHASE != 2) a.r[2 * c] = r[k].x;
                if (PHASE == 0) a.v[2 * c] = v[k].x;
                if (PHASE != 1) a.p[2 * c] = p[k].x;
            }
        }
    }
    if (tid == 0) {
        const int n_hmul = (PHASE == 2) ? n_hmul0 : n_hmul0 + 1;
        bool write_trace = false;
        if (FIRST) {
            st->rtv = rtv; st->tol_cg = tol_cg; st->beta = 0.0;
            st->iter = 1; st->max_iter = max_iter; st->approx_solved = 0; st->done = 0; st->status = 4;
        }
        if (PHASE != 2) {
            st->pHp = pHp; st->gamma = gamma; st->alpha = alpha; st->n_hmul = n_hmul;
            st->neg_curvature = neg; st->outside_region = outside;
            st->need_proj = (PHASE == 1 && cont) ? 1 : 0;
            if (!cont) {
                st->done = 1;
                st->status = cg_final_status(st);
                write_trace = true;
            }
        }
        if (cont && PHASE != 1) {
            st->beta = beta; st->rtv = rtv_next;                           // :746
            st->approx_solved = fabs(rtv_next) < tol_cg;                   // :747
            st->iter = iter0 + 1;                                          // :748
            st->need_proj = 0;
            if (st->approx_solved || st->iter > max_iter) { st->done = 1; st->status = cg_final_status(st); }
            write_trace = true;
        }
        if (write_trace && a.trace != nullptr && n_hmul <= a.trace_cap) {
            double* row = a.trace + 4 * (int64_t)(n_hmul - 1);
            row[0] = pHp; row[1] = alpha; row[2] = (PHASE != 2 && neg && !add_w) ? QNAN : gamma; row[3] = rtv_next;
        }
        if (write_trace) publish_state(a, st);     // an iteration (or the whole loop) has completed
    }
}

// ------------------------------------------------------------------------------------------
// Callers of projected_cg on the device.
// ------------------------------------------------------------------------------------------
// The w_l / w_u construction of minor_iterate — src/basic_tralcnlss.jl:660-665: +-Inf on the free variables,
// min(xupp - (x+s), delta) / max(xlow - (x+s), -delta) on the fixed ones.
__global__ __launch_bounds__(256) void step_bounds_kernel(const double* __restrict__ x, const double* __restrict__ s,
                                                          const double* __restrict__ xlow, const double* __restrict__ xupp,
                                                          const int* __restrict__ fixrank, double delta, int n,
                                                          double* __restrict__ wl, double* __restrict__ wu) {
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        double lo = -INF, hi = INF;
        if (fixrank != nullptr && fixrank[i] >= 0) {
            const double xm = __dadd_rn(x[i], s[i]);           // x_minor = x + s  (:660)
            hi = fmin(__dsub_rn(xupp[i], xm), delta);          // :664
            lo = fmax(__dsub_rn(xlow[i], xm), -delta);         // :665
        }
        wl[i] = lo;
        wu[i] = hi;
    }
}

// linesearch — src/basic_tralcnlss.jl:766-791, given wHw = vthv(H,w) in wHw[0]; optionally scales w by alpha in place
// (minor_iterate :670-671).  out[0] = alpha.  Single workgroup.
__global__ __launch_bounds__(CG_T) void linesearch_kernel(const double* __restrict__ g, double* __restrict__ w,
                                                          const double* __restrict__ wl, const double* __restrict__ wu,
                                                          const int* __restrict__ fixrank, const double* __restrict__ wHw_p,
                                                          const double* __restrict__ hw, int n, int scale_w, double* __restrict__ out) {
    __shared__ double scratch[2 * (CG_T / 64)];
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    OpMinNan opmin;
    double gw[1] = {0.0}, amin[1] = {INF}, whw[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += CG_T) {
        const double wi = w[i];
        gw[0] = fma(g[i], wi, gw[0]);
        if (hw != nullptr) whw[0] = fma(wi, hw[i], whw[0]);      // w'Hw from the H*w the CG loop accumulated
        if (fixrank == nullptr || fixrank[i] < 0) {              // :781
            if (wi < 0.0) amin[0] = opmin(amin[0], __ddiv_rn(wl[i], wi));      // :783
            else if (wi > 0.0) amin[0] = opmin(amin[0], __ddiv_rn(wu[i], wi)); // :785
        }
    }
    block_reduce<CG_T, 1>(gw, scratch, OpSum(), 0.0);
    block_reduce<CG_T, 1>(amin, scratch, opmin, INF);
    if (hw != nullptr) block_reduce<CG_T, 1>(whw, scratch, OpSum(), 0.0);
    const double wHw = (hw != nullptr) ? whw[0] : wHw_p[0];
    const double alpha_opt = (wHw > 0.0) ? __ddiv_rn(-gw[0], wHw) : INF;      // :776
    const double alpha = opmin(alpha_opt, amin[0]);                            // :790
    if (scale_w)
        for (int i = threadIdx.x; i < n; i += CG_T) w[i] = __dmul_rn(alpha, w[i]);   // :671
    if (threadIdx.x == 0) out[0] = alpha;
}

// out = a + b (g_minor = H*s + g, src/basic_tralcnlss.jl:412,:437)
__global__ __launch_bounds__(256) void vec_add_kernel(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = __dadd_rn(a[i], b[i]);
}

// ------------------------------------------------------------------------------------------
// cauchy_step on the device — src/basic_tralcnlss.jl:574-639 with next_breakpoint (:536-562) and the initial
// active_bounds! (src/polyhedral_constraints.jl:203-215).  SURVEY.md §8 "next" row f-3.
// The loop state reuses CgState (so the row-stream / projection kernels can gate on ->done / ->need_proj):
//   rtv = phi_p, pHp = phi_pp, gamma = theta, alpha = delta_t, iter = nb_fix, max_iter = n - mA (nmm),
//   status = index fixed at the last breakpoint (-1: none), approx_solved = min_found, neg_curvature = 1 when no
//   breakpoint exists (the reference would index fixvars[-1]), pad = breakpoints taken, n_hmul = H*d products.
// ------------------------------------------------------------------------------------------
struct CauchyArgs {
    CgState* st;
    const double* x; const double* g; const double* xlow; const double* xupp;
    double* negg; double* d; const double* Hd; double* s; double* dl; double* du;
    int* fixrank;
    int n, n_pad, nmm;
    double delta, atol;
    int box;                // mA == 0: the projection is a mask, so it is maintained in place (d[ind] = 0 when ind becomes fixed)
    unsigned long long* mirror; unsigned tag;
};

__device__ __forceinline__ void publish_cauchy(const CauchyArgs& a, const CgState* st) {
    if (a.mirror == nullptr) return;
    const unsigned long long wv = ((unsigned long long)(a.tag & 0xffffu) << 48) | ((unsigned long long)(st->neg_curvature & 0xf) << 44) |
                                  ((unsigned long long)(st->done & 0xf) << 40) | ((unsigned long long)(st->pad & 0xfffff) << 20) |
                                  (unsigned long long)(st->n_hmul & 0xfffff);
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
    }
}

// One pass: phi_p, phi_pp for the current (d, Hd) (:610-611 / :634-635), the while test (:615), next_breakpoint (:617),
// the three-way branch (:620-636) including s_c update and the active-set growth of add_active! (poly:240-249).
__global__ __launch_bounds__(CG_T) void cauchy_advance_kernel(CauchyArgs a) {
    constexpr int NW = CG_T / 64;
    __shared__ double scratch[2 * NW];
    __shared__ int iscratch[NW];
    CgState* st = a.st;
    if (st->done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    double sums[2] = {0.0, 0.0};
    double gd[1] = {0.0};
    double th = INF;
    int ind = 0x7fffffff;
    for (int i = tid; i < a.n; i += CG_T) {
        const double di = a.d[i], hdi = a.Hd[i], si = a.s[i];
        sums[0] = fma(si, hdi, sums[0]);
        sums[1] = fma(di, hdi, sums[1]);
        gd[0] = fma(a.g[i], di, gd[0]);
        if (a.fixrank[i] < 0) {                                   // :547
            double t = INF;
            if (di < 0.0) t = __ddiv_rn(__dsub_rn(a.dl[i], si), di);       // :549
            else if (di > 0.0) t = __ddiv_rn(__dsub_rn(a.du[i], si), di);  // :551
            if (t < th) { th = t; ind = i; }                      // strict <: first minimiser in index order (:555)
        }
    }
    block_reduce<CG_T, 2>(sums, scratch, OpSum(), 0.0);
    block_reduce<CG_T, 1>(gd, scratch, OpSum(), 0.0);
    // arg-min with the smallest index among equal thetas
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double t2 = __shfl_xor(th, off);
        const int i2 = __shfl_xor(ind, off);
        if (t2 < th || (t2 == th && i2 < ind)) { th = t2; ind = i2; }
    }
    __syncthreads();
    if (lane == 0) { scratch[wave] = th; iscratch[wave] = ind; }
    __syncthreads();
    th = scratch[0]; ind = iscratch[0];
    for (int w = 1; w < NW; ++w) {
        const double t2 = scratch[w];
        const int i2 = iscratch[w];
        if (t2 < th || (t2 == th && i2 < ind)) { th = t2; ind = i2; }
    }
    if (ind == 0x7fffffff) ind = -1;                              // :544

    const double phi_p = __dadd_rn(sums[0], gd[0]);               // :610 / :634
    const double phi_pp = sums[1];                                // :611 / :635
    const int nfix = st->iter;
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
    if (step != 0.0 || advance)
        for (int i = tid; i < a.n; i += CG_T) {
            a.s[i] = __dadd_rn(a.s[i], __dmul_rn(step, a.d[i]));
            // box constraints: projection!(lincons, -g, d) after add_active!(ind) only zeroes d[ind] (:632) — done by the
            // thread that owns the element, after it has used the old value
            if (advance && a.box && i == ind) a.d[i] = 0.0;
        }
    if (tid == 0) {
        st->rtv = phi_p; st->pHp = phi_pp; st->gamma = th; st->alpha = delta_t;
        st->n_hmul += 1;
        st->approx_solved = min_found; st->neg_curvature = err;
        if (advance) {
            a.fixrank[ind] = 0;                                   // add_active!: fixvars[ind] = true (poly:246)
            st->iter = nfix + 1; st->status = ind; st->pad += 1;
        }
        st->done = done; st->need_proj = done ? 0 : 1;
        publish_cauchy(a, st);
    }
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

// Stand-alone factor_to_boundary (tests).
__global__ __launch_bounds__(CG_T) void f2b_kernel(const double* p, const double* w, const double* wl, const double* wu,
                                                   int n, double atol, double* out) {
    __shared__ double scratch[CG_T / 64];
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    double gmin[1] = {INF};
    OpMinNan opmin;
    for (int i = threadIdx.x; i < n; i += CG_T) gmin[0] = opmin(gmin[0], f2b_term(p[i], w[i], wl[i], wu[i], atol));
    block_reduce<CG_T, 1>(gmin, scratch, opmin, INF);
    if (threadIdx.x == 0) out[0] = gmin[0];
}

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
};

__device__ __forceinline__ bool proj_skip(const CgState* st) { return st != nullptr && (st->done || !st->need_proj); }

// Box-only projection: v = fixed ? 0 : r.
__global__ __launch_bounds__(256) void proj_mask_kernel(const double* __restrict__ r, double* __restrict__ v, const int* fixrank, int n,
                                                        const CgState* st) {
    if (proj_skip(st)) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        v[i] = (fixrank != nullptr && fixrank[i] >= 0) ? 0.0 : r[i];
}

// left_mul: tw[0:mA] = A x (one workgroup per row: 4 waves share the row, fixed-order combine), and in the augmented
// form tw[mA+k] = x[fixidx[k]] (:86-98).  Reduced form: the fixed components of x are masked out (A_free x_free).
// grid = mA + ceil(nfix/256) blocks of 256 (gather blocks only in the augmented form).
__global__ __launch_bounds__(256) void proj_left_mul_kernel(ProjArgs a, const double* __restrict__ x) {
    if (proj_skip(a.state)) return;
    __shared__ double scratch[4];
    if ((int)blockIdx.x < a.mA) {
        const int row = blockIdx.x;
        const double2* rp = reinterpret_cast<const double2*>(a.A + (int64_t)row * a.ldA);
        const double2* x2 = reinterpret_cast<const double2*>(x);
        const int2* f2 = reinterpret_cast<const int2*>(a.fixrank);
        const int nch = (int)(a.ldA >> 1);
        const bool mask = a.reduced && a.fixrank != nullptr;
        double acc[1] = {0.0};
        for (int c = threadIdx.x; c < nch; c += 256) {
            const double2 av = rp[c];
            double2 xv = x2[c];
            if (mask) {
                const int2 f = f2[c];
                if (f.x >= 0) xv.x = 0.0;
                if (f.y >= 0) xv.y = 0.0;
            }
            acc[0] = fma(av.x, xv.x, acc[0]);
            acc[0] = fma(av.y, xv.y, acc[0]);
        }
        block_reduce<256, 1>(acc, scratch, OpSum(), 0.0);
        if (threadIdx.x == 0) a.tw[row] = acc[0];
    } else if (!a.reduced) {
        const int k = ((int)blockIdx.x - a.mA) * 256 + threadIdx.x;
        if (k < a.nfix) a.tw[a.mA + k] = x[a.fixidx[k]];
    }
}

// out = r - left_mul_tr(tw)   (:72-84, :116, :134);  with SUBTRACT=false: out = left_mul_tr(tw).
// Block = 64 chunks x 4 row groups (rows i = rg, rg+4, ...), combined through LDS in fixed order; grid = ceil(nch/64).
template <bool SUBTRACT>
__global__ __launch_bounds__(256) void proj_left_mul_tr_kernel(ProjArgs a, const double* __restrict__ r, double* __restrict__ out) {
    if (proj_skip(a.state)) return;
    __shared__ double2 sm[4][64];
    const int nch = (a.n + 1) >> 1;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double2 acc = make_double2(0.0, 0.0);
    if (c < nch) {
        const double2* A2 = reinterpret_cast<const double2*>(a.A);
        const int64_t ld2 = a.ldA >> 1;
        for (int i = rg; i < a.mA; i += 4) {
            const double wi = a.tw[i];
            const double2 av = A2[(int64_t)i * ld2 + c];
            acc.x = fma(wi, av.x, acc.x);
            acc.y = fma(wi, av.y, acc.y);
        }
    }
    sm[rg][cl] = acc;
    __syncthreads();
    if (rg != 0 || c >= nch) return;
    acc.x = (sm[0][cl].x + sm[1][cl].x) + (sm[2][cl].x + sm[3][cl].x);
    acc.y = (sm[0][cl].y + sm[1][cl].y) + (sm[2][cl].y + sm[3][cl].y);
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
__global__ __launch_bounds__(256) void chol_small_kernel(const double* Msrc, int64_t ld_src, double* M, int64_t ld_dst, int m,
                                                         double* dinv_out, int* info, int info_base, int reset_info,
                                                         const CgState* gate) {
    if (gate != nullptr && gate->done) return;
    __shared__ __attribute__((aligned(16))) double colbuf[2][64];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int k = 16 * wave + c;
        a[c] = (lane < m && k < m && k <= lane) ? Msrc[lane + (int64_t)k * ld_src] : 0.0;
    }
    if (tid == 0) s_bad = 0;
    double dinv_mine = 0.0;
    int buf = 0;
    __syncthreads();
    for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = 16 * p + jj;
            if (j < m) {                                    // uniform
                if (wave == p) {
                    const double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[jj]), j),
                                                        __builtin_amdgcn_readlane(__double2loint(a[jj]), j));
                    // one reciprocal square root instead of sqrt + 64 divisions: the dependent fp64 chain per step is the cost
                    // v_rsq_f64 seed + two Newton steps (y <- y(1.5 - 0.5 x y^2)): full fp64 accuracy for the normal, positive
                    // pivots of an SPD matrix without the library rsqrt's range handling (the pivot chain is the critical path)
                    double rinv = __builtin_amdgcn_rsq(piv);
                    rinv = rinv * fma(-0.5 * piv * rinv, rinv, 1.5);
                    rinv = rinv * fma(-0.5 * piv * rinv, rinv, 1.5);
                    if (!(piv > 0.0) && lane == 0 && s_bad == 0) s_bad = j + 1;
                    double lij = 0.0;
                    if (lane == j) { lij = piv * rinv; dinv_mine = rinv; }
                    else if (lane > j) lij = a[jj] * rinv;
                    a[jj] = lij;
                    colbuf[buf][lane] = lij;
                }
                __syncthreads();
                if (16 * wave + 15 > j) {                           // this wave's panel has columns right of j (wave-uniform)
                    const double lij = colbuf[buf][lane];
                    const double* cb = &colbuf[buf][16 * wave];     // the 16 l_kj of this panel: contiguous, broadcast reads
                    // branch-free: rows above the diagonal and columns <= j get a zero coefficient (columns >= m hold zeros)
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const int k = 16 * wave + c;
                        const double lkj = (k > j) ? cb[c] : 0.0;
                        const double li = (lane >= k) ? lij : 0.0;
                        a[c] = fma(-li, lkj, a[c]);
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

// L21 <- A21 L11^{-T}: rows [r0, m) of the panel [k0, k0+nb).  One thread per row; L11 (lower, nb x nb) and its reciprocal
// diagonal in LDS; column j of the row is finished before column j+1 (own earlier columns are re-read from global memory).
__global__ __launch_bounds__(256) void chol_trsm_kernel(double* __restrict__ M, int m, int k0, int nb, const double* __restrict__ dinv,
                                                        const CgState* gate) {
    if (gate != nullptr && gate->done) return;
    __shared__ double l11[64 * 65];
    __shared__ double di[64];
    for (int e = threadIdx.x; e < nb * nb; e += 256) {
        const int i = e % nb, k = e / nb;
        l11[i * 65 + k] = (i >= k) ? M[(k0 + i) + (int64_t)(k0 + k) * m] : 0.0;
    }
    if ((int)threadIdx.x < nb) di[threadIdx.x] = dinv[threadIdx.x];
    __syncthreads();
    const int r = k0 + nb + blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    for (int j = 0; j < nb; ++j) {
        double acc = M[r + (int64_t)(k0 + j) * m];
#pragma unroll 8
        for (int c = 0; c < j; ++c) acc = fma(-M[r + (int64_t)(k0 + c) * m], l11[j * 65 + c], acc);
        M[r + (int64_t)(k0 + j) * m] = acc * di[j];
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
// A_free A_free' when one more variable becomes fixed.  O(m^2) instead of refactoring (O(m^3)); hyperbolic rotations in
// the reciprocal-diagonal form: s = a_k / l_kk, c = sqrt(1 - s^2), l_kk <- c l_kk, l_ik <- (l_ik - s a_i)/c,
// a_i <- c a_i - s l_ik.  m <= 64: one wave, row i of L in lane i's registers, 64 unrolled steps of two v_readlane
// broadcasts + one rsqrt.  A non-positive 1 - s^2 (the downdated matrix is no longer positive definite) raises info.
__global__ __launch_bounds__(64) void chol_downdate_small_kernel(double* __restrict__ L, const double* __restrict__ A, int64_t ldA, int m,
                                                                 int* info, const CgState* st) {
    if (st->done) return;
    const int ind = st->status;
    if (ind < 0) return;
    const int lane = threadIdx.x;
    double row[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) row[k] = (lane < m && k <= lane && k < m) ? L[lane + (int64_t)k * m] : 0.0;
    double dinv = (lane < m) ? L[(int64_t)m * m + lane] : 0.0;
    double a = (lane < m) ? A[(int64_t)lane * ldA + ind] : 0.0;
    int bad = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        if (k < m) {
            const double ak = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), k), __builtin_amdgcn_readlane(__double2loint(a), k));
            const double dk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dinv), k),
                                               __builtin_amdgcn_readlane(__double2loint(dinv), k));
            const double sn = ak * dk;
            const double t = fma(-sn, sn, 1.0);
            if (!(t > 0.0) && bad == 0) bad = k + 1;
            double rc = __builtin_amdgcn_rsq(t);  // 1/c: hardware seed + two Newton steps (t is a normal number in (0, 1])
            rc = rc * fma(-0.5 * t * rc, rc, 1.5);
            rc = rc * fma(-0.5 * t * rc, rc, 1.5);
            const double c = t * rc;
            if (lane == k) { row[k] = row[k] * c; dinv = dinv * rc; }
            else if (lane > k) {
                const double lik = (row[k] - sn * a) * rc;
                a = fma(c, a, -sn * lik);
                row[k] = lik;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 64; ++k)
        if (lane < m && k <= lane && k < m) L[lane + (int64_t)k * m] = row[k];
    if (lane < m) L[(int64_t)m * m + lane] = dinv;
    if (lane == 0 && bad != 0) info[0] = bad;
}

// The same for any m: one workgroup, L in global memory (column k is contiguous), a in LDS.
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
__global__ __launch_bounds__(256) void trsv_small_kernel(ProjArgs a) {
    if (proj_skip(a.state)) return;
    __shared__ double t[64 * 65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = a.mpp;
    const double* __restrict__ L = a.L;
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
    const double di = (lane < m) ? L[(int64_t)m * m + lane] : 0.0;
    double xi = (lane < m) ? a.tw[lane] : 0.0;
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

// tw <- L' \ (L \ tw)   (:114-115, :132-133).  Single workgroup, 64-wide blocked substitution;
// the diagonal block is staged through LDS (coalesced column reads, conflict-free padded tile) and
// solved by one wave with v_readlane broadcasts; trailing updates use all 16 waves.
// Dynamic LDS: mpp doubles (the vector) + 64*65 doubles (tile).
__global__ __launch_bounds__(CG_T) void trsv_pair_kernel(ProjArgs a) {
    if (proj_skip(a.state)) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int m = a.mpp;
    double* x = smem;
    double* tile = smem + ((m + 1) & ~1);    // [64][65]
    const double* __restrict__ L = a.L;
    const int64_t ld = m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < m; i += CG_T) x[i] = a.tw[i];
    __syncthreads();

    // ---- forward: L y = t ----
    for (int j0 = 0; j0 < m; j0 += 64) {
        const int nb = min(64, m - j0);
        for (int e = tid; e < 64 * 64; e += CG_T) {
            const int rr = e & 63, cc = e >> 6;
            tile[rr * 65 + cc] = (rr < nb && cc < nb && rr >= cc) ? L[(j0 + rr) + (int64_t)(j0 + cc) * ld] : 0.0;
        }
        __syncthreads();
        if (wave == 0) {
            // lane i owns unknown j0+i; column jj of the block is tile[i*65 + jj] (conflict-free: stride 65)
            double xi = (lane < nb) ? x[j0 + lane] : 0.0;
#pragma unroll 8
            for (int jj = 0; jj < nb; ++jj) {
                const double ljj = tile[jj * 65 + jj];
                const double lij = tile[lane * 65 + jj];
                const double xs = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xi), jj),
                                                   __builtin_amdgcn_readlane(__double2loint(xi), jj));
                const double xj = xs / ljj;
                if (lane == jj) xi = xj;
                else if (lane > jj) xi = fma(-lij, xj, xi);
            }
            if (lane < nb) x[j0 + lane] = xi;
        }
        __syncthreads();
        for (int i = j0 + nb + tid; i < m; i += CG_T) {
            double acc = 0.0;
            for (int jj = 0; jj < nb; ++jj) acc = fma(L[i + (int64_t)(j0 + jj) * ld], x[j0 + jj], acc);
            x[i] -= acc;
        }
        __syncthreads();
    }

    // ---- backward: L' w = y ----
    const int nblk = (m + 63) / 64;
    for (int b = nblk - 1; b >= 0; --b) {
        const int j0 = b * 64;
        const int nb = min(64, m - j0);
        // x[j0+c] -= sum_{k >= j0+nb} L[k, j0+c] * x[k]   (column segments are contiguous: wave per column)
        for (int c = wave; c < nb; c += CG_T / 64) {
            double acc = 0.0;
            const double* col = L + (int64_t)(j0 + c) * ld;
            for (int k = j0 + nb + lane; k < m; k += 64) acc = fma(col[k], x[k], acc);
            acc = wave_sum(acc);
            if (lane == 0) x[j0 + c] -= acc;
        }
        for (int e = tid; e < 64 * 64; e += CG_T) {
            const int rr = e & 63, cc = e >> 6;
            tile[rr * 65 + cc] = (rr < nb && cc < nb && rr >= cc) ? L[(j0 + rr) + (int64_t)(j0 + cc) * ld] : 0.0;
        }
        __syncthreads();
        if (wave == 0) {
            // lane i owns unknown j0+i and needs L[jj, i] for jj > i: tile[jj*65 + i] (consecutive lanes, conflict-free)
            double xi = (lane < nb) ? x[j0 + lane] : 0.0;
#pragma unroll 8
            for (int jj = nb - 1; jj >= 0; --jj) {
                const double ljj = tile[jj * 65 + jj];
                const double lji = tile[jj * 65 + lane];
                const double xs = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xi), jj),
                                                   __builtin_amdgcn_readlane(__double2loint(xi), jj));
                const double xj = xs / ljj;
                if (lane == jj) xi = xj;
                else if (lane < jj) xi = fma(-lji, xj, xi);
            }
            if (lane < nb) x[j0 + lane] = xi;
        }
        __syncthreads();
    }

    for (int i = tid; i < m; i += CG_T) a.tw[i] = x[i];
}

// Self-test of the wave reduction network (bh_selftest): out[wave] = sum, out[16 + wave] = min.
__global__ __launch_bounds__(256) void selftest_wave_kernel(const double* in, double* out) {
    const double x = in[threadIdx.x];
    const double s = wave_sum(x), mn = wave_min(x);
    out[threadIdx.x] = s;
    out[256 + threadIdx.x] = mn;
}

}  // namespace bh
