// bh_cg.hip.h — projected_cg loop kernels, linesearch / minor_iterate helpers, factor_to_boundary
// Part of the single translation unit of bh_api.hip (see bh_kernels.hip.h for the layout and design notes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bh_reduce.hip.h"

namespace bh {

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
    if (st->done) {   // the tie log of the finished call: read by the host after it has drained the stream
        const unsigned long long tw = ((unsigned long long)(st->margin_kind & 0xf) << 48) | ((unsigned long long)(st->tie_flags & 0xff) << 40) |
                                      ((unsigned long long)(st->tie_first & 0xfffff) << 20) | (unsigned long long)(st->margin_at & 0xfffff);
        __hip_atomic_store(a.mirror + 1, tw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.mirror + 2, (unsigned long long)__double_as_longlong(st->min_margin), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(a.mirror, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The same word (and, when done, the tie log) from values the caller holds in registers.
__device__ __forceinline__ void publish_word(unsigned long long* mirror, unsigned tag, int status, int done, int iter, int n_hmul,
                                             const TieRegs& tr) {
    if (mirror == nullptr) return;
    const unsigned long long wv = ((unsigned long long)(tag & 0xffffu) << 48) | ((unsigned long long)(status & 0xf) << 44) |
                                  ((unsigned long long)(done & 0xf) << 40) | ((unsigned long long)(iter & 0xfffff) << 20) |
                                  (unsigned long long)(n_hmul & 0xfffff);
    if (done) {
        const unsigned long long tw = ((unsigned long long)(tr.margin_kind & 0xf) << 48) | ((unsigned long long)(tr.tie_flags & 0xff) << 40) |
                                      ((unsigned long long)(tr.tie_first & 0xfffff) << 20) | (unsigned long long)(tr.margin_at & 0xfffff);
        __hip_atomic_store(mirror + 1, tw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(mirror + 2, (unsigned long long)__double_as_longlong(tr.min_margin), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(mirror, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// cg_final_status (:753-761) on register values
__device__ __forceinline__ int cg_status_of(int approx_solved, int outside_region, int neg_curvature, int iter, int max_iter) {
    if (approx_solved) return 0;
    if (outside_region) return 1;
    if (neg_curvature) return 2;
    if (iter == max_iter) return 3;
    return 4;
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
            tie_reset(st);
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
        st->n_hmul = 0; st->need_proj = 0; st->stop_at = 0;
        tie_reset(st);
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
            tie_note_step_a(st, pHp, a.atol_neg, alpha, gamma, st->n_hmul);
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
                tie_note(st, TIE_TOL, rel_margin(fabs(rtv_next), st->tol_cg), st->n_hmul);
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
            tie_note(st, TIE_TOL, rel_margin(fabs(rtv_next), st->tol_cg), st->n_hmul);
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
    // the tie log travels in registers (loaded here with everything else, stored once by the committing thread): kept in memory
    // it costs that thread a dependent load per field at the very end of a single-workgroup kernel
    TieRegs tr;
    if (FIRST) { tr.min_margin = __longlong_as_double(0x7ff0000000000000ll); tr.margin_kind = 0; tr.margin_at = 0; tr.tie_flags = 0; tr.tie_first = 0; }
    else tr.load(st);

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
        int done = 0, status = 4, iter = iter0;       // (status of a progress word: the host reads it of a finished loop only)
        if (FIRST) {
            st->rtv = rtv; st->tol_cg = tol_cg; st->beta = 0.0;
            st->iter = 1; st->max_iter = max_iter; st->approx_solved = 0; st->done = 0; st->status = 4;
        }
        if (PHASE != 2) {
            st->pHp = pHp; st->gamma = gamma; st->alpha = alpha; st->n_hmul = n_hmul;
            st->neg_curvature = neg; st->outside_region = outside;
            tr.note_step_a(pHp, a.atol_neg, alpha, gamma, n_hmul);
            st->need_proj = (PHASE == 1 && cont) ? 1 : 0;
            if (!cont) {
                // approx_solved is 0 here: a loop that had met :747 would have stopped before this pass
                done = 1;
                status = cg_status_of(0, outside, neg, iter0, max_iter);
                write_trace = true;
            }
        }
        if (cont && PHASE != 1) {
            const int approx = fabs(rtv_next) < tol_cg ? 1 : 0;            // :747
            st->beta = beta; st->rtv = rtv_next;                           // :746
            st->approx_solved = approx;
            tr.note(TIE_TOL, rel_margin(fabs(rtv_next), tol_cg), n_hmul);
            iter = iter0 + 1;                                              // :748
            st->iter = iter;
            st->need_proj = 0;
            // (the pass continued: neither outside_region nor neg_curvature is set — step_a of this iteration cleared them)
            if (approx || iter > max_iter) { done = 1; status = cg_status_of(approx, 0, 0, iter, max_iter); }
            write_trace = true;
        }
        if (done) { st->done = 1; st->status = status; }
        tr.store(st);
        if (write_trace && a.trace != nullptr && n_hmul <= a.trace_cap) {
            double* row = a.trace + 4 * (int64_t)(n_hmul - 1);
            row[0] = pHp; row[1] = alpha; row[2] = (PHASE != 2 && neg && !add_w) ? QNAN : gamma; row[3] = rtv_next;
        }
        if (write_trace) publish_word(a.mirror, a.tag, status, done, iter, n_hmul, tr);     // an iteration (or the whole loop) has completed
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
                                                          double* __restrict__ hw, int n, int scale_w, double* __restrict__ out) {
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
        for (int i = threadIdx.x; i < n; i += CG_T) {
            w[i] = __dmul_rn(alpha, w[i]);   // :671
            if (hw != nullptr) hw[i] = __dmul_rn(alpha, hw[i]);      // stays H*w for the scaled w (bh_step_accumulate_dev: g_minor += H*w)
        }
    if (threadIdx.x == 0) out[0] = alpha;
}

// out = a + b (g_minor = H*s + g, src/basic_tralcnlss.jl:412,:437)
__global__ __launch_bounds__(256) void vec_add_kernel(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = __dadd_rn(a[i], b[i]);
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

}  // namespace bh
