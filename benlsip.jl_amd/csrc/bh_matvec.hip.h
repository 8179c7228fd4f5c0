// bh_matvec.hip.h — row-streaming kernels for J v, J'u and the fused single-read J'(W.(J v)); slab reduction; upload transpose; synthetic J
// Part of the single translation unit of bh_api.hip (see bh_kernels.hip.h for the layout and design notes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bh_reduce.hip.h"
#include "bh_cg.hip.h"

namespace bh {

// ------------------------------------------------------------------------------------------
// Row-streaming kernel: J·v, J'·u and the fused single-read J'(W ∘ (J v)).
//   T   threads per workgroup, CPT 16-byte chunks (2 doubles) per thread per row, R rows per step.
//   A workgroup owns row groups g = blockIdx.x, +gridDim.x, ... (the whole grid marches through
//   HBM together, like a copy); the next group's loads are issued before the current group's
//   reduction so they stay in flight across the barrier.
// ------------------------------------------------------------------------------------------
enum { MODE_JV = 0, MODE_JTV = 1, MODE_FUSED = 2 };

// Two-kernel box-constrained CG iteration: the H*p launch of iteration j forms p_j itself
// (CGP = 1) from what cg_reduce_update_kernel(j-1) left behind, and decides the loop's exit test on the way in.
struct CgFuse {
    CgState* st;
    int j;                  // iteration of this launch, 1-based (the reference's `iter` while the H*p of :722 runs)
    int init_done;          // general constraints: 1 = :702-718 was done by the init kernels (p_1 = -P(g) is in memory, CgState is set);
                            // 2 = v, p_1 and the partials of r.v / v.v are in memory, workgroup 0 of launch 1 sets CgState from them
    int vv_off;             // init_done == 2: the v.v partials follow the r.v partials at this offset
    int n, max_iter;
    const double* vvec;     // v = P(r) after iteration j-1                (j >= 2)
    const double* p_old;    // p_{j-1}                                     (j >= 2)
    double* p_new;          // p_j: every workgroup stores the chunks it owns (run c / CPT of consecutive chunks belongs to workgroup (c / CPT) % gridDim.x)
    const double* rvpart;   // partial sums of r.v written by cg_reduce_update_kernel(j-1)
    int nrv;
    const double* w;        // w after iteration j-1 (j == 1: w = 0)
    const double* wl; const double* wu;
    double* sqpart;         // [gridDim.x]  sum over this workgroup's rows of weight_i * (J p)_i^2   ->  pHp
    double* gpart;          // [gridDim.x]  min of the factor_to_boundary terms of the chunks this workgroup owns   ->  gamma
    double kappa2, atol_f2b;
    double* trace; int trace_cap;
    unsigned long long* mirror; unsigned tag;
    // CGP = 3 (several ranks over RCCL: two kernels + the collective per iteration): the launch of iteration j ALSO performs the
    // vector update of iteration j-1 (:725-748) that cg_reduce_update_kernel does on one rank — every workgroup for the whole
    // vector, redundantly (the vectors are 32 KiB: L2 hits), so that no kernel stands between the all-reduce and the next H*p
    const double* gpart_in; // the gamma partials the launch of iteration j-1 left (ping-pong with gpart: this launch writes the other one)
    const double* Hp;       // all-reduced H*p of iteration j-1 (ld doubles) followed by the all-reduced p'Hp at [sq_index]
    int sq_index;
    const double* r_old;    // r before the update of iteration j-1   (ping-pong: everybody reads it, the owners write r_new)
    double* r_new;
    double* w_rw;           // w, updated in place by the workgroup that owns the chunk (nobody else reads it)
    double* hw;             // H*w accumulated next to w, or NULL
    const double* g;        // j == 1: r_1 = g_minor (:705)
    const int* fixrank;     // v = mask(r)
    double atol_neg;
};

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
    CgFuse cf;              // CGP = 1 only
};

// NT: J is read exactly once per launch -> non-temporal loads (global_load_dwordx4 ... nt): measured +10 % (6.39 -> 7.05 TB/s).
// PF: 1 = issue the next row group's loads before reducing the current one (two register buffers); 0 = one buffer, latency
// hidden by several co-resident workgroups instead.
// VL: 1 = each lane parks its slice of v in LDS (dynamic, nchunks x 16 bytes, lane-private slots: no barrier, no bank
// conflicts) instead of registers.  For 8192 < n <= 16384 the two row buffers and the z accumulators of the fused mode
// fill the register file on their own; the 128 KiB of LDS a CU has left over hold v.
// CGP: 1 = MODE_FUSED launch of the two-kernel box CG iteration: the prologue below forms p_j (and takes the loop's exit test)
// while the first row group is already on its way from HBM.  2 = the same code as a SEPARATE symbol for the launch the host
// expects to find the loop finished (its prediction: the previous call's iteration count): no prefetch before the exit test,
// and — the reason for a symbol of its own — its ~2 us dispatches do not dilute the streaming kernel's per-kernel average in a
// rocprofv3 --stats summary.  Either symbol does the right thing if the prediction is wrong.
template <int T, int CPT, int R, int MODE, int NT = 1, int PF = 1, int VL = 0, int CGP = 0>
__global__ __launch_bounds__(T) void row_stream_kernel(RowStreamArgs a) {
    if (!CGP && a.state != nullptr && a.state->done) return;
    // the loop stopped before this iteration (CGP = 3: the launch that FINDS the stop is j = stop_at + 1 and writes stop_at itself —
    // every one of its workgroups must still get through its prologue, whose owner stores complete w)
    if (CGP && a.cf.j > 1 && a.cf.st->stop_at != 0 && a.cf.j > a.cf.st->stop_at + (CGP == 3 ? 1 : 0)) return;
    constexpr int NW = T / 64;
    __shared__ double red[2][R][NW];
    __shared__ double pro[2][NW];                                       // CGP, iteration 1, workgroup 0 only
    extern __shared__ __attribute__((aligned(16))) double2 v_lds[];     // VL only: [CPT][T]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t ld2 = a.ld >> 1;   // row stride in double2
    const double2* __restrict__ J2 = reinterpret_cast<const double2*>(a.J);
    const int64_t ngroups = (a.nrows + R - 1) / R;
    const int64_t G = gridDim.x;

    bool act[CPT];
    double2 vv[CPT], zz[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) act[k] = (tid + k * T) < a.nchunks;

    double2 A[R][CPT], B[R][CPT];
    double sq_acc = 0.0;
    int buf = 0;

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
                    const double2 vk = VL ? v_lds[k * T + tid] : vv[k];
                    acc = fma(X[r][k].x, vk.x, acc);
                    acc = fma(X[r][k].y, vk.y, acc);
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
                else {
                    coef = (row < a.d_rows) ? s[r] : a.mu * s[r];
                    if (CGP && rv) sq_acc = fma(coef, s[r], sq_acc);          // p'Hp = sum_i weight_i (J p)_i^2
                }
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    zz[k].x = fma(coef, X[r][k].x, zz[k].x);
                    zz[k].y = fma(coef, X[r][k].y, zz[k].y);
                }
            }
        }
    };

    int64_t g = blockIdx.x;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        vv[k] = make_double2(0.0, 0.0);
        zz[k] = make_double2(0.0, 0.0);
    }
    if (!CGP || a.cf.j == 1) {
        // the first row group does not depend on the vector: its loads go out first
        if (PF && g < ngroups) load_group(A, g);
        // (all loads first, the masking afterwards: a compare next to its load makes the compiler wait for each chunk in turn)
        int2 nm[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = tid + k * T;
            nm[k] = make_int2(-1, -1);
            if (MODE != MODE_JTV && act[k]) {
                vv[k] = reinterpret_cast<const double2*>(a.v)[c];
                if (a.negate && a.negmask != nullptr) nm[k] = reinterpret_cast<const int2*>(a.negmask)[c];
            }
        }
        if (MODE != MODE_JTV && a.negate) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                vv[k].x = (nm[k].x >= 0) ? 0.0 : -vv[k].x;
                vv[k].y = (nm[k].y >= 0) ? 0.0 : -vv[k].y;
            }
        }
    }
    if (CGP == 3) {
        // ---- several ranks over RCCL: update of iteration j-1 + exit test + p_j, every workgroup for the whole vector -------------
        const CgFuse& f = a.cf;
        CgState* st = f.st;
        const double QNAN = __longlong_as_double(0x7ff8000000000000ll);
        OpMinNan opmin;
        double2 wnew[CPT];                       // w after the update, for the chunks this workgroup owns (factor_to_boundary below)
        bool own[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = tid + k * T;
            own[k] = act[k] && ((c / CPT) % (int)G) == (int)blockIdx.x;
            wnew[k] = make_double2(0.0, 0.0);
        }
        if (f.j == 1) {
            // :702-718: vv = -mask(g) was formed above; r_1 = g and w = 0 are stored by the owners, CgState is set by workgroup 0
            double rtv0 = 0.0;
#pragma unroll
            for (int k = 0; k < CPT; ++k) { rtv0 = fma(vv[k].x, vv[k].x, rtv0); rtv0 = fma(vv[k].y, vv[k].y, rtv0); }
            rtv0 = wave_sum(rtv0);
            if (lane == 0) pro[0][wave] = rtv0;
            __syncthreads();
            double t = 0.0;
            for (int w2 = 0; w2 < NW; ++w2) t += pro[0][w2];
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                if (!own[k]) continue;
                const int c = tid + k * T;
                double2 gk = reinterpret_cast<const double2*>(f.g)[c];
                if (2 * c >= f.n) gk.x = 0.0;
                if (2 * c + 1 >= f.n) gk.y = 0.0;
                reinterpret_cast<double2*>(f.r_new)[c] = gk;
                reinterpret_cast<double2*>(f.w_rw)[c] = make_double2(0.0, 0.0);
                if (f.hw != nullptr) reinterpret_cast<double2*>(f.hw)[c] = make_double2(0.0, 0.0);
            }
            if (blockIdx.x == 0 && tid == 0) {
                st->rtv = t; st->rtv_pp[0] = t; st->rtv_pp[1] = 0.0;
                st->tol_cg = f.kappa2 * sqrt(t);
                st->pHp = 0.0; st->alpha = 0.0; st->gamma = 0.0; st->beta = 0.0;
                st->iter = 1; st->max_iter = f.max_iter;
                st->approx_solved = 0; st->outside_region = 0; st->neg_curvature = 0;
                st->n_hmul = 0; st->need_proj = 0; st->done = 0; st->status = 4; st->stop_at = 0;
                tie_reset(st);
            }
            __syncthreads();                                   // pro[] is reused below
        } else {
            double2 hp[CPT], po[CPT], rk[CPT];
            int2 fr[CPT];
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                hp[k] = po[k] = rk[k] = make_double2(0.0, 0.0);
                fr[k] = make_int2(-1, -1);
                if (act[k]) {
                    const int c = tid + k * T;
                    hp[k] = reinterpret_cast<const double2*>(f.Hp)[c];
                    po[k] = reinterpret_cast<const double2*>(f.p_old)[c];
                    rk[k] = reinterpret_cast<const double2*>(f.r_old)[c];
                    if (f.fixrank != nullptr) fr[k] = reinterpret_cast<const int2*>(f.fixrank)[c];
                }
            }
            const double pHp = f.Hp[f.sq_index];                               // :723, summed over the ranks by the all-reduce
            const double rtv = st->rtv_pp[f.j & 1], tol_cg = st->tol_cg;       // r.v entering iteration j-1
            TieRegs tr;
            tr.load(st);
            const double gamma = wave_fixed_min(f.gpart_in, (int)G);           // :728 / :734, left by the launch of iteration j-1
            int cont = 0, neg = 0, outside = 0;
            double step = 0.0, alpha = QNAN;
            bool add_w = true;
            if (pHp <= f.atol_neg) {                        // :725
                neg = 1;
                if (fabs(pHp) > f.atol_neg) step = gamma;   // :727-729
                else add_w = false;
            } else {
                alpha = __ddiv_rn(rtv, pHp);                // :733
                outside = alpha > gamma;                    // :735
                if (outside) step = gamma;                  // :737
                else { step = alpha; cont = 1; }            // :739
            }
            // r += alpha Hp (:740), v = mask(r) (:741), r.v (:743) — the whole vector in every workgroup, same order everywhere
            double part = 0.0;
            double2 vk[CPT];
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int c = tid + k * T;
                vk[k] = make_double2(0.0, 0.0);
                if (cont) {
                    rk[k].x = __dadd_rn(rk[k].x, __dmul_rn(alpha, hp[k].x));
                    rk[k].y = __dadd_rn(rk[k].y, __dmul_rn(alpha, hp[k].y));
                    if (!act[k] || 2 * c >= f.n) rk[k].x = 0.0;               // padding stays exactly zero (Inf * 0 would poison it)
                    if (!act[k] || 2 * c + 1 >= f.n) rk[k].y = 0.0;
                    vk[k].x = (fr[k].x >= 0) ? 0.0 : rk[k].x;
                    vk[k].y = (fr[k].y >= 0) ? 0.0 : rk[k].y;
                    part = fma(rk[k].y, vk[k].y, fma(rk[k].x, vk[k].x, part));
                }
            }
            part = wave_sum(part);
            if (lane == 0) pro[0][wave] = part;
            __syncthreads();
            double rtv_next = 0.0;
            for (int w2 = 0; w2 < NW; ++w2) rtv_next += pro[0][w2];
            const bool solved = cont && fabs(rtv_next) < tol_cg;               // :747
            const bool stop = !cont || solved || f.j > f.max_iter;             // :720 with iter = j after :748
            const double beta = __ddiv_rn(rtv_next, rtv);                      // :744
            // the owners store w (+= step p, :729 / :737 / :739), H*w and the new r
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                if (!own[k]) continue;
                const int c = tid + k * T;
                double2 wk = reinterpret_cast<const double2*>(f.w_rw)[c];
                if (add_w) {
                    wk.x = __dadd_rn(wk.x, __dmul_rn(step, po[k].x));
                    wk.y = __dadd_rn(wk.y, __dmul_rn(step, po[k].y));
                    if (2 * c >= f.n) wk.x = 0.0;
                    if (2 * c + 1 >= f.n) wk.y = 0.0;
                    reinterpret_cast<double2*>(f.w_rw)[c] = wk;
                    if (f.hw != nullptr) {
                        double2 hwk = reinterpret_cast<const double2*>(f.hw)[c];
                        hwk.x = __dadd_rn(hwk.x, __dmul_rn(step, hp[k].x));
                        hwk.y = __dadd_rn(hwk.y, __dmul_rn(step, hp[k].y));
                        if (2 * c >= f.n) hwk.x = 0.0;
                        if (2 * c + 1 >= f.n) hwk.y = 0.0;
                        reinterpret_cast<double2*>(f.hw)[c] = hwk;
                    }
                }
                wnew[k] = wk;
                reinterpret_cast<double2*>(f.r_new)[c] = rk[k];
            }
            if (blockIdx.x == 0 && tid == 0) {
                const int it = f.j - 1;                                        // the iteration being completed
                tr.note_step_a(pHp, f.atol_neg, alpha, gamma, it);
                if (cont) tr.note(TIE_TOL, rel_margin(fabs(rtv_next), tol_cg), it);
                tr.store(st);
                if (f.trace != nullptr && it <= f.trace_cap) {
                    double* row = f.trace + 4 * (int64_t)(it - 1);
                    row[0] = pHp; row[1] = alpha; row[2] = (neg && !add_w) ? QNAN : gamma; row[3] = cont ? rtv_next : rtv;
                }
                st->pHp = pHp; st->gamma = gamma; st->alpha = alpha; st->n_hmul = it; st->beta = beta;
                st->neg_curvature = neg; st->outside_region = outside; st->need_proj = 0;
                st->rtv = cont ? rtv_next : rtv; st->rtv_pp[(f.j - 1) & 1] = rtv_next;
                st->iter = cont ? f.j : it;                                    // :748
                int status = 4;
                if (stop) {
                    status = cont ? cg_status_of(solved ? 1 : 0, 0, 0, f.j, f.max_iter) : cg_status_of(0, outside, neg, it, f.max_iter);
                    st->approx_solved = solved ? 1 : 0; st->done = 1; st->stop_at = it; st->status = status;
                }
                publish_word(f.mirror, f.tag, status, stop ? 1 : 0, cont ? f.j : it, it, tr);
            }
            if (stop) return;
            if (PF && g < ngroups) load_group(A, g);                           // the stream starts here (no prefetch before the exit test)
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                vv[k].x = __dadd_rn(-vk[k].x, __dmul_rn(beta, po[k].x));       // :745
                vv[k].y = __dadd_rn(-vk[k].y, __dmul_rn(beta, po[k].y));
            }
            __syncthreads();                                                   // pro[] is reused below
        }
        // the owners store p_j and fold their factor_to_boundary terms (:734 / :728) into this workgroup's partial of gamma
        double gm = __longlong_as_double(0x7ff0000000000000ll);
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            if (!own[k]) continue;
            const int c = tid + k * T;
            reinterpret_cast<double2*>(f.p_new)[c] = vv[k];
            const double2 lo = reinterpret_cast<const double2*>(f.wl)[c], hi = reinterpret_cast<const double2*>(f.wu)[c];
            if (2 * c < f.n) gm = opmin(gm, f2b_term(vv[k].x, wnew[k].x, lo.x, hi.x, f.atol_f2b));
            if (2 * c + 1 < f.n) gm = opmin(gm, f2b_term(vv[k].y, wnew[k].y, lo.y, hi.y, f.atol_f2b));
        }
        gm = wave_min(gm);
        if (NW > 1) {
            if (lane == 0) pro[1][wave] = gm;
            __syncthreads();
            gm = pro[1][0];
            for (int w2 = 1; w2 < NW; ++w2) gm = opmin(gm, pro[1][w2]);
        }
        if (tid == 0) f.gpart[blockIdx.x] = gm;
    }
    if (CGP && CGP != 3) {
        const CgFuse& f = a.cf;
        CgState* st = f.st;
        if (f.j == 1) {
            // projected_cg's initialisation (:702-718) belongs to workgroup 0: r = g, v = P(r) = mask(g), p = -v (formed above),
            // rtv = r.v, tol_cg = kappa2*||v||, iter = 1, all flags down
            if (blockIdx.x == 0 && f.init_done != 1) {
                double t = 0.0, vsq = 0.0;
                if (f.init_done == 2) {
                    // general constraints, three-kernel iteration: v = P(g), p_1 = -v are in memory (proj_apply_linv_kernel<INIT>),
                    // which also left the partials of r.v and of v.v (every wave gets the same sums)
                    t = wave_fixed_sum(f.rvpart, f.nrv);
                    vsq = wave_fixed_sum(f.rvpart + f.vv_off, f.nrv);
                } else {
                    double rtv0 = 0.0;
#pragma unroll
                    for (int k = 0; k < CPT; ++k) {        // vv = -mask(g):  r.v = v.v = sum of squares of the free components
                        rtv0 = fma(vv[k].x, vv[k].x, rtv0); rtv0 = fma(vv[k].y, vv[k].y, rtv0);
                    }
                    rtv0 = wave_sum(rtv0);
                    if (lane == 0) pro[0][wave] = rtv0;
                    __syncthreads();
                    for (int w2 = 0; w2 < NW; ++w2) t += pro[0][w2];
                    vsq = t;
                }
                if (tid == 0) {
                    st->rtv = t;                                   // :707  (r.v; box: v = mask(r))
                    st->tol_cg = f.kappa2 * sqrt(vsq);             // :710
                    st->pHp = 0.0; st->alpha = 0.0; st->gamma = 0.0; st->beta = 0.0;
                    st->iter = 1; st->max_iter = f.max_iter;
                    st->approx_solved = 0; st->outside_region = 0; st->neg_curvature = 0;
                    st->n_hmul = 0; st->need_proj = 0; st->done = 0; st->status = 4; st->stop_at = 0;
                    tie_reset(st);
                }
                __syncthreads();                                   // pro[] is reused below
            }
        } else {
            // CGP = 1 (expected to go on): the first row group of J goes out FIRST — loads return in order, so the prologue's own
            // operands (v, p, the r.v partials: L2 hits) arrive right behind it and the prologue's arithmetic runs while nothing
            // else is outstanding; issued the other way round the stream would start one memory round trip later.  If the loop
            // turns out to have stopped, the 128 KiB this workgroup asked for are simply dropped.
            if (CGP != 2 && PF && g < ngroups) load_group(A, g);
            double2 vk[CPT], po[CPT];
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                vk[k] = po[k] = make_double2(0.0, 0.0);
                if (act[k]) {
                    vk[k] = reinterpret_cast<const double2*>(f.vvec)[tid + k * T];
                    po[k] = reinterpret_cast<const double2*>(f.p_old)[tid + k * T];
                }
            }
            const double rtv = st->rtv, tol_cg = st->tol_cg;
            TieRegs tr;                                                        // (uniform loads, issued with the rest)
            tr.load(st);
            const int outside_prev = st->outside_region, neg_prev = st->neg_curvature;
            const double rtv_next = wave_fixed_sum(f.rvpart, f.nrv);          // :743, same bits in every wave of every workgroup
            const bool solved = fabs(rtv_next) < tol_cg;                       // :747
            const bool stop = solved || f.j > f.max_iter;                      // :720 with iter = j after :748
            const double beta = __ddiv_rn(rtv_next, rtv);                      // :744
            if (blockIdx.x == 0 && tid == 0) {
                // everything below works on registers: this thread's workgroup starts streaming only when it is through
                tr.note(TIE_TOL, rel_margin(fabs(rtv_next), tol_cg), f.j - 1);
                tr.store(st);
                if (f.trace != nullptr && f.j - 1 <= f.trace_cap) f.trace[4 * (int64_t)(f.j - 2) + 3] = rtv_next;
                st->iter = f.j;                        // :748 (nobody reads it back: the kernels count iterations by launch)
                int status = 4;
                if (stop) {
                    status = cg_status_of(solved ? 1 : 0, outside_prev, neg_prev, f.j, f.max_iter);
                    st->beta = beta; st->rtv = rtv_next; st->approx_solved = solved ? 1 : 0;
                    st->done = 1; st->stop_at = f.j - 1;
                    st->status = status;
                }
                // "stopped after iteration j-1", or "iter = j: iteration j is streaming" (n_hmul = j - 1 either way)
                publish_word(f.mirror, f.tag, status, stop ? 1 : 0, f.j, f.j - 1, tr);
            }
            if (stop) return;
            if (CGP == 2 && PF && g < ngroups) load_group(A, g);              // the prediction was wrong: carry on
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                vv[k].x = __dadd_rn(-vk[k].x, __dmul_rn(beta, po[k].x));       // :745
                vv[k].y = __dadd_rn(-vk[k].y, __dmul_rn(beta, po[k].y));
            }
        }
        // the chunks this workgroup owns: store p_j; the workgroup's share of gamma = factor_to_boundary(p, w, w_l, w_u)
        // (:734 / :728).  Ownership goes by RUNS of CPT consecutive chunks (run r belongs to workgroup r % gridDim.x): a
        // workgroup's chunks then sit in CPT neighbouring lanes of one k — one round trip for their w, w_l, w_u.  (Dealt out
        // chunk by chunk, c % gridDim.x, one lane owned a chunk in every k and walked through CPT dependent round trips
        // while the rest of the workgroup waited at the barrier below: ~4 us per launch at n = 4096.)
        OpMinNan opmin;
        double gm = __longlong_as_double(0x7ff0000000000000ll);
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = tid + k * T;
            if (!act[k] || ((c / CPT) % (int)G) != (int)blockIdx.x) continue;
            reinterpret_cast<double2*>(f.p_new)[c] = vv[k];
            double2 wk = make_double2(0.0, 0.0);
            if (f.j > 1) wk = reinterpret_cast<const double2*>(f.w)[c];
            const double2 lo = reinterpret_cast<const double2*>(f.wl)[c], hi = reinterpret_cast<const double2*>(f.wu)[c];
            if (2 * c < f.n) gm = opmin(gm, f2b_term(vv[k].x, wk.x, lo.x, hi.x, f.atol_f2b));
            if (2 * c + 1 < f.n) gm = opmin(gm, f2b_term(vv[k].y, wk.y, lo.y, hi.y, f.atol_f2b));
        }
        gm = wave_min(gm);
        if (NW > 1) {
            if (lane == 0) pro[1][wave] = gm;
            __syncthreads();
            gm = pro[1][0];
            for (int w2 = 1; w2 < NW; ++w2) gm = opmin(gm, pro[1][w2]);
        }
        if (tid == 0) f.gpart[blockIdx.x] = gm;
    }
    if (VL && MODE != MODE_JTV) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) v_lds[k * T + tid] = vv[k];     // read back only by this lane
    }

    if (!PF) {
        for (; g < ngroups; g += G) {
            load_group(A, g);
            process(A, g);
        }
    } else if (g < ngroups) {
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
        if (CGP && tid == 0) a.cf.sqpart[blockIdx.x] = sq_acc;
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
    const int cc = min(c, nchunks - 1);        // out-of-range threads load a valid chunk and drop the result (no branch around the loads)
    SlabBatch sb;
    sb.issue(P2, ld2, cc, rl, G);
    const double2 acc = sb.fold(P2, ld2, cc, rl, G);
    sm[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < nchunks) {
        double2 t = sm[0][cl];
#pragma unroll
        for (int r = 1; r < 16; ++r) { t.x += sm[r][cl].x; t.y += sm[r][cl].y; }
        reinterpret_cast<double2*>(out)[c] = t;
    }
}

// The same for the RCCL form of the two-kernel CG iteration: gated on CgState::stop_at, and block 0 also folds this rank's
// partials of p'Hp (sum_i w_i (Jp)_i^2) into out[sq_index], the slot behind the vector that rides through the all-reduce.
__global__ __launch_bounds__(256) void reduce_partials_sq_kernel(const double* __restrict__ partials, int64_t ld, int nchunks, int G,
                                                                 double* __restrict__ out, const double* __restrict__ sqpart, int sq_index,
                                                                 const CgState* st, int j) {
    if (st->stop_at != 0 && j > st->stop_at) return;
    __shared__ double2 sm[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int64_t ld2 = ld >> 1;
    const double2* P2 = reinterpret_cast<const double2*>(partials);
    const int cc = min(c, nchunks - 1);
    SlabBatch sb;
    sb.issue(P2, ld2, cc, rl, G);
    double sq = 0.0;
    if (blockIdx.x == 0) sq = wave_fixed_sum(sqpart, G);
    const double2 acc = sb.fold(P2, ld2, cc, rl, G);
    sm[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < nchunks) {
        double2 t = sm[0][cl];
#pragma unroll
        for (int r = 1; r < 16; ++r) { t.x += sm[r][cl].x; t.y += sm[r][cl].y; }
        reinterpret_cast<double2*>(out)[c] = t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[sq_index] = sq; out[sq_index + 1] = 0.0; }
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
// wcols: destination columns written per row, counted from dst (>= cols; columns [cols, wcols) are zero padding) — a
// column chunk of a larger image writes only its own columns (bh_hess_create_async), a whole image passes wcols = ldd.
__global__ __launch_bounds__(256) void transpose_cm_to_rm_kernel(const double* __restrict__ src, int64_t lds_, int64_t rows,
                                                                 int64_t cols, double* __restrict__ dst, int64_t ldd, int64_t wcols) {
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
        if (r < rows && c < wcols) dst[r * ldd + c] = (c < cols) ? tile[tx][ty + 8 * k] : 0.0;
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

// Read-only stream probe: the practical HBM read ceiling on the SAME image the row-stream kernels sweep (bh_time_kernel
// kinds 3..6).  Nothing but 16-byte non-temporal loads (8 independent ones in flight per lane) and one add per load; one
// partial per workgroup is written so the loads cannot be elided.  What this kernel reaches is what any single-read
// kernel can reach on this device; the fused kernel is quoted against it in bench.py ("read_probe").
__global__ __launch_bounds__(256) void read_probe_kernel(const double* __restrict__ J, int64_t nchunks_total, double* __restrict__ out) {
    __shared__ double scratch[256 / 64];
    const dvec2* __restrict__ src = reinterpret_cast<const dvec2*>(J);
    const int64_t stride = (int64_t)gridDim.x * 256;
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; c + 7 * stride < nchunks_total; c += 8 * stride) {
        dvec2 x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = __builtin_nontemporal_load(src + c + k * stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += x[k].x + x[k].y;
    }
    for (; c < nchunks_total; c += stride) {
        const dvec2 x = __builtin_nontemporal_load(src + c);
        acc[0] += x.x + x.y;
    }
    double tot[1] = {((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))};
    block_reduce<256, 1>(tot, scratch, OpSum(), 0.0);
    if (threadIdx.x == 0) out[blockIdx.x] = tot[0];
}


}  // namespace bh
