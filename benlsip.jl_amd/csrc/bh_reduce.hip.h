// bh_reduce.hip.h — wave64 / workgroup reductions (DPP + permlane swaps) and the device-resident loop state
// Part of the single translation unit of bh_api.hip (see bh_kernels.hip.h for the layout and design notes).
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
// Sum over the 16 lanes of one DPP row (lanes 16k .. 16k+15): the first four steps of the butterfly.
__device__ __forceinline__ double row16_sum(double x) {
    x = x + dpp_mov_f64<0xB1>(x);
    x = x + dpp_mov_f64<0x4E>(x);
    x = x + dpp_mov_f64<0x141>(x);
    x = x + dpp_mov_f64<0x140>(x);
    return x;
}
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
    // Tie log: how close the loop's branch scalars came to their thresholds in this call.
    double min_margin;   // smallest relative distance |a - b| / max(|a|, |b|) of any branch test to its threshold
    int margin_kind;     // which test that was (TIE_*)
    int margin_at;       // ... at which H*p product
    int tie_flags;       // TIE_* bits of the tests that came within kTieRel of their threshold
    int tie_first;       // first H*p product at which that happened (0: never)
    // Two-kernel box iteration (bh_cgfuse.hip.h): the iteration at which the loop stopped (0: still running).  Kernels of
    // iteration j > stop_at return at once; kernels of iteration j == stop_at never look at a word their own launch writes.
    int stop_at;
    int pad2;
    // CGP = 3 (two-kernel iteration over RCCL): r.v after k updates at rtv_pp[k & 1] — the launch of iteration j reads the slot its
    // own workgroup 0 does not write
    double rtv_pp[2];
};

// Sum / min of m doubles by ONE wave in an order that does not depend on the workgroup shape: lane l folds x[l], x[l+64], ...
// in index order, then the fixed butterfly.  Every wave (of any kernel) that calls it on the same data gets the same bits.
// The loads of the first 64*B entries go out together (issue), the adds follow in index order (fold): one memory round trip
// instead of one per 64 entries — these reductions sit on the critical path of latency-bound kernels.  A caller with several
// reductions issues all of them before folding any.
template <int B>
struct LaneBatch {
    double t[B];
    // (unconditional loads from clamped indices: a load under the same predicate as its add invites the compiler to fuse the
    // two and wait for the data on the spot, which serialises the batch again)
    __device__ __forceinline__ void issue(const double* __restrict__ x, int m) {
        const int lane = threadIdx.x & 63, last = max(m - 1, 0);      // x holds at least one addressable element
#pragma unroll
        for (int k = 0; k < B; ++k) t[k] = x[min(lane + 64 * k, last)];
    }
    __device__ __forceinline__ double fold_sum(const double* __restrict__ x, int m) const {
        const int lane = threadIdx.x & 63;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < B; ++k) if (lane + 64 * k < m) acc += t[k];
        for (int i = lane + 64 * B; i < m; i += 64) acc += x[i];
        return acc;
    }
    __device__ __forceinline__ double fold_min(const double* __restrict__ x, int m) const {
        const int lane = threadIdx.x & 63;
        OpMinNan op;
        double acc = __longlong_as_double(0x7ff0000000000000ll);
#pragma unroll
        for (int k = 0; k < B; ++k) if (lane + 64 * k < m) acc = op(acc, t[k]);
        for (int i = lane + 64 * B; i < m; i += 64) acc = op(acc, x[i]);
        return acc;
    }
};
__device__ __forceinline__ double wave_fixed_sum(const double* __restrict__ x, int m) {
    LaneBatch<8> b;
    b.issue(x, m);
    return wave_sum(b.fold_sum(x, m));
}
__device__ __forceinline__ double wave_fixed_min(const double* __restrict__ x, int m) {
    LaneBatch<8> b;
    b.issue(x, m);
    return wave_min(b.fold_min(x, m));
}

// Column sums of the partial slabs a row-stream launch left: thread (rl, c) of a 16 x 16 arrangement folds slab rows rl, rl+16,
// rl+32, ... of 16-byte chunk c in ascending row order.  As above the loads go out 16 at a time (256 slab rows per round trip).
struct SlabBatch {
    double2 x[16];
    __device__ __forceinline__ void issue(const double2* __restrict__ P2, int64_t ld2, int c, int rl, int G) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = P2[(int64_t)min(rl + 16 * k, G - 1) * ld2 + c];      // G >= 1; clamped as in LaneBatch
    }
    __device__ __forceinline__ double2 fold(const double2* __restrict__ P2, int64_t ld2, int c, int rl, int G) const {
        double2 acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (rl + 16 * k < G) { acc.x += x[k].x; acc.y += x[k].y; }
        for (int g0 = rl + 256; g0 < G; g0 += 256) {            // more than 256 slab rows (several workgroups per CU): batch by batch
            double2 y[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) y[k] = P2[(int64_t)min(g0 + 16 * k, G - 1) * ld2 + c];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (g0 + 16 * k < G) { acc.x += y[k].x; acc.y += y[k].y; }
        }
        return acc;
    }
};

// x where the variable is free (rank < 0), +0.0 where it is fixed — as a bit mask instead of a compare + select: a compare on a
// freshly loaded word is scheduled right behind its load (with the wait), which serialises a batch of loads again.
__device__ __forceinline__ double keep_if_free(double x, int rank) {
    const long long keep = (long long)(rank >> 31);            // rank < 0: all ones; rank >= 0: zero
    return __longlong_as_double(__double_as_longlong(x) & keep);
}

// Branch tests of projected_cg whose outcome can flip under rounding (src/basic_tralcnlss.jl:725, :727, :735, :747).
enum { TIE_NEGCURV = 1, TIE_NEGCURV_ABS = 2, TIE_BOUND = 4, TIE_TOL = 8 };
constexpr double kTieRel = 1e-10;

__device__ __forceinline__ double rel_margin(double a, double b) {
    const double m = fmax(fabs(a), fabs(b));
    if (!(m < __longlong_as_double(0x7ff0000000000000ll))) return (a == b) ? 0.0 : 1.0;   // an infinite operand: far, unless both are the same infinity
    return m > 0.0 ? fabs(a - b) / m : 0.0;
}
__device__ __forceinline__ void tie_reset(CgState* st) {
    st->min_margin = __longlong_as_double(0x7ff0000000000000ll);
    st->margin_kind = 0; st->margin_at = 0; st->tie_flags = 0; st->tie_first = 0;
}
__device__ __forceinline__ void tie_note(CgState* st, int kind, double margin, int n_hmul) {
    if (margin < st->min_margin) { st->min_margin = margin; st->margin_kind = kind; st->margin_at = n_hmul; }
    if (margin <= kTieRel) { st->tie_flags |= kind; if (st->tie_first == 0) st->tie_first = n_hmul; }
}
// After step_a's branch (:725-739): pHp against tol_negcurve, |pHp| against it again, alpha against gamma.
__device__ __forceinline__ void tie_note_step_a(CgState* st, double pHp, double atol_neg, double alpha, double gamma, int n_hmul) {
    tie_note(st, TIE_NEGCURV, rel_margin(pHp, atol_neg), n_hmul);
    if (pHp <= atol_neg) tie_note(st, TIE_NEGCURV_ABS, rel_margin(fabs(pHp), atol_neg), n_hmul);
    else tie_note(st, TIE_BOUND, rel_margin(alpha, gamma), n_hmul);
}

// The same log kept in registers by the two-kernel iteration: its kernels load the five words up front together with everything
// else they need (the log lives in CgState, last written by the PREVIOUS launch), note in registers and store once — the
// memory-resident form above costs a dependent load per field on the single thread that commits an iteration, and that thread's
// workgroup is on the critical path of its launch.
struct TieRegs {
    double min_margin;
    int margin_kind, margin_at, tie_flags, tie_first;
    __device__ __forceinline__ void load(const CgState* st) {
        min_margin = st->min_margin; margin_kind = st->margin_kind; margin_at = st->margin_at;
        tie_flags = st->tie_flags; tie_first = st->tie_first;
    }
    __device__ __forceinline__ void note(int kind, double margin, int n_hmul) {
        if (margin < min_margin) { min_margin = margin; margin_kind = kind; margin_at = n_hmul; }
        if (margin <= kTieRel) { tie_flags |= kind; if (tie_first == 0) tie_first = n_hmul; }
    }
    __device__ __forceinline__ void note_step_a(double pHp, double atol_neg, double alpha, double gamma, int n_hmul) {
        note(TIE_NEGCURV, rel_margin(pHp, atol_neg), n_hmul);
        if (pHp <= atol_neg) note(TIE_NEGCURV_ABS, rel_margin(fabs(pHp), atol_neg), n_hmul);
        else note(TIE_BOUND, rel_margin(alpha, gamma), n_hmul);
    }
    __device__ __forceinline__ void store(CgState* st) const {
        st->min_margin = min_margin; st->margin_kind = margin_kind; st->margin_at = margin_at;
        st->tie_flags = tie_flags; st->tie_first = tie_first;
    }
};

// Self-test of the wave reduction network (bh_selftest): out[wave] = sum, out[16 + wave] = min.
__global__ __launch_bounds__(256) void selftest_wave_kernel(const double* in, double* out) {
    const double x = in[threadIdx.x];
    const double s = wave_sum(x), mn = wave_min(x);
    out[threadIdx.x] = s;
    out[256 + threadIdx.x] = mn;
}

}  // namespace bh
