// bh_cgfuse.hip.h — second kernel of the two-kernel box-constrained CG iteration
// Part of the single translation unit of bh_api.hip (see bh_kernels.hip.h for the layout and design notes).
//
// One iteration of projected_cg (src/basic_tralcnlss.jl:722-750) used to be three launches: H*p stream, slab reduction,
// single-workgroup step kernel.  For box constraints (projection = mask) it is two:
//   row_stream_kernel<..., CGP = 1>(j)   forms p_j = -v + beta p_{j-1} on the fly (beta and the exit test |r.v| < tol_cg come
//                                         from the partial sums the update kernel of iteration j-1 left), streams J once,
//                                         leaves the per-workgroup slabs of J'(W.(J p)), the partial sums of p'Hp = sum_i
//                                         w_i (Jp)_i^2 and the factor_to_boundary terms of p_j;
//   cg_reduce_update_kernel(j)            128 workgroups: each recomputes the iteration's scalars (pHp, gamma, r.v, alpha and the
//                                         branch of :725-739 — identical bits everywhere: shape-independent wave reductions),
//                                         sums ITS 32 columns of the slabs and updates ITS 32 entries of w, H*w, r, v at once,
//                                         leaving its partial of the next r.v.
// No workgroup ever reads a word its own launch writes: scalars that change are either recomputed from partials or read
// from CgState fields that only the PREVIOUS launch wrote; the exit is signalled through CgState::stop_at (iteration
// number), which gates launches of later iterations only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bh_reduce.hip.h"
#include "bh_cg.hip.h"
#include "bh_comm.hip.h"

namespace bh {

struct CgUpdArgs {
    CgState* st;
    int j;                          // iteration (1-based) = number of the H*p product being consumed
    const double* partials; int64_t ld; int nchunks; int G;      // slabs of the preceding row_stream launch
    int Gs, Gq;                     // how many slabs / p'Hp partials to sum: G, or 1 when an all-reduced H*p with its p'Hp slot takes their place (RCCL)
    const double* sqpart;           // [Gq]
    const double* gpart;            // [G]
    const double* rvpart_in;        // [nrv] partials of r.v from iteration j-1   (j >= 2)
    double* rvpart_out;             // [gridDim.x]
    int nrv;
    const double* p;                // p_j
    double* w; double* hw;          // hw may be NULL
    double* r; const double* g;     // j == 1: r = g_minor (:705), w = 0 (:702)
    double* v;
    const int* fixrank;             // NULL: nothing fixed
    int n;
    double atol_neg;
    double* trace; int trace_cap;
    unsigned long long* mirror; unsigned tag;
    // GEN (linear equalities, reduced projection form): no v / r.v here — this kernel leaves the per-workgroup partials of
    // A_free r, trsv_small_kernel sums and solves, proj_left_mul_tr_kernel forms v = P(r) and the partials of r.v
    const double* A; int64_t ldA; int mA;
    double* tpart;                  // [gridDim.x][mA]
    int init_in_memory;             // j == 1: r = g and w = 0 are already in memory (init kernels), not taken from g
};

// grid = ceil(nchunks / 16) workgroups of 256 threads = 16 chunks x 16 slab lanes (as reduce_partials_kernel).
// PEER: several ranks over the peer-buffer transport (bh_comm.hip.h) — the workgroup's 32 columns of this rank's slab sum and
// this rank's share of pHp are pushed into every inbox, and the sums over ranks (rank order: identical bits everywhere) take
// their place before the update.  The exchange is inside the stop_at gate by construction.
template <bool GEN, bool PEER>
__global__ __launch_bounds__(256) void cg_reduce_update_kernel(CgUpdArgs a, PeerArgs pa) {
    CgState* st = a.st;
    if (st->stop_at != 0 && a.j > st->stop_at) return;
    __shared__ double2 sm[16][17];
    __shared__ double rvs[16];
    __shared__ double2 rsm[16];                  // GEN: the masked new r of this workgroup's 16 chunks
    const int tid = threadIdx.x, cl = tid & 15, rl = tid >> 4;
    const int c = blockIdx.x * 16 + cl;
    const bool valid = c < a.nchunks;
    const int64_t ld2 = a.ld >> 1;
    const double QNAN = __longlong_as_double(0x7ff8000000000000ll);

    // ---- every load the kernel needs goes out first: the iteration's partial sums, this workgroup's slab rows, its vector
    //      entries.  The kernel is latency-bound (a few hundred bytes per thread); issued one reduction at a time the same
    //      loads cost ~15 dependent memory round trips, issued together ~2.
    const bool upd = (rl == 0) && valid;                       // the 16 threads that update this workgroup's 16 chunks
    const bool from_g = (a.j == 1 && !a.init_in_memory);      // r = g_minor (:705), w = 0 (:702)
    double2 pk = make_double2(0.0, 0.0), wk = pk, hwk = pk, rk = pk;
    int2 fr = make_int2(-1, -1);
    if (upd) {
        pk = reinterpret_cast<const double2*>(a.p)[c];
        if (from_g) {
            rk = reinterpret_cast<const double2*>(a.g)[c];
        } else {
            rk = reinterpret_cast<const double2*>(a.r)[c];
            wk = reinterpret_cast<const double2*>(a.w)[c];
            if (a.hw != nullptr) hwk = reinterpret_cast<const double2*>(a.hw)[c];
        }
    }
    TieRegs tr;                                                // (uniform: every thread holds the log, one of them commits it)
    tr.load(st);
    const int max_iter = st->max_iter;
    const double rtv_state = st->rtv;                          // j == 1: written by the H*p launch
    LaneBatch<8> b_sq, b_gp, b_rv;
    b_sq.issue(a.sqpart, a.Gq);
    b_gp.issue(a.gpart, a.G);
    b_rv.issue(a.rvpart_in, a.nrv);                            // (j == 1: addressable, not used)
    const double2* P2 = reinterpret_cast<const double2*>(a.partials);
    const int cc = min(c, a.nchunks - 1);                      // out-of-range threads load a valid chunk and drop the result:
    SlabBatch sb;                                              // no branch around the loads (see LaneBatch::issue)
    sb.issue(P2, ld2, cc, rl, a.Gs);
    // GEN: this thread's entries of A for the partials of A_free r below (rows rl, rl + 16, ...; up to 64 rows), asked for now
    double2 arow[4];
    if (GEN) {
        const double2* A2 = reinterpret_cast<const double2*>(a.A);
        const int64_t ldA2 = a.ldA >> 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) arow[k] = A2[(int64_t)min(rl + 16 * k, a.mA - 1) * ldA2 + cc];
    }
    if (upd && a.fixrank != nullptr) fr = reinterpret_cast<const int2*>(a.fixrank)[c];   // last: its compare is scheduled next to it

    // ---- this workgroup's 32 columns of Hp = sum of the slabs (fixed order) --------------------------------------------
    const double2 acc = sb.fold(P2, ld2, cc, rl, a.Gs);
    sm[rl][cl] = acc;

    // ---- the iteration's scalars, recomputed by every wave from the partials (identical bits everywhere) -----------------
    double pHp = wave_sum(b_sq.fold_sum(a.sqpart, a.Gq));                               // :723 (this rank's rows)
    if (PEER) {
        __shared__ int s_timeout;
        __shared__ double2 xs[kMaxPeers][16];
        __shared__ double ps[kMaxPeers];
        if (tid == 0) s_timeout = 0;
        __syncthreads();                                    // sm[][] is complete
        // this rank's slab sum for the workgroup's 16 chunks
        if (rl == 0) {
            double2 t = sm[0][cl];
#pragma unroll
            for (int q = 1; q < 16; ++q) { t.x += sm[q][cl].x; t.y += sm[q][cl].y; }
            sm[15][cl] = t;
        }
        __syncthreads();
        const double2 mine = sm[15][cl];
        const unsigned long long seq = *pa.seq;
        const int par = (int)(seq & 1ull);
        const int64_t slot = ((int64_t)par * kMaxPeers + pa.rank) * pa.cap;
        const int64_t fidx = ((int64_t)par * kMaxPeers + pa.rank) * pa.nblk_cap + blockIdx.x;
        // push (write-through), drain, meet, raise the flags — as reduce_exchange_kernel
        if (rl < pa.nranks && valid) sys_store_f64x2(pa.slots[rl] + slot + 2 * (int64_t)c, mine);
        if (rl < pa.nranks && cl == 0) __hip_atomic_store(pa.scal[rl] + fidx, pHp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (rl < pa.nranks && cl == 0) __hip_atomic_store(pa.flags[rl] + fidx, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (rl < pa.nranks && cl == 0) {
            const unsigned long long* f = pa.flags[pa.rank] + ((int64_t)par * kMaxPeers + rl) * pa.nblk_cap + blockIdx.x;
            const unsigned long long t0 = wall_clock64();
            const bool dead = __hip_atomic_load(pa.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (dead) s_timeout = 1;
            while (!dead && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > pa.timeout_ticks) {
                    s_timeout = 1;
                    __hip_atomic_store(pa.err, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pa.dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        asm volatile("" ::: "memory");
        __syncthreads();
        if (rl < pa.nranks) {
            double2 got = make_double2(0.0, 0.0);
            if (valid) got = sys_load_f64x2(pa.slots[pa.rank] + ((int64_t)par * kMaxPeers + rl) * pa.cap + 2 * (int64_t)c);
            xs[rl][cl] = got;
            if (cl == 0) ps[rl] = __hip_atomic_load(pa.scal[pa.rank] + ((int64_t)par * kMaxPeers + rl) * pa.nblk_cap + blockIdx.x,
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        // sums over ranks in rank order; sm[0] then holds the global H*p columns and rows 1.. are zero for the code below
        double2 t = xs[0][cl];
        double ph = ps[0];
        for (int r = 1; r < pa.nranks; ++r) { t.x += xs[r][cl].x; t.y += xs[r][cl].y; ph += ps[r]; }
        if (s_timeout) { ph = __longlong_as_double(0x7ff8000000000000ll); }
        pHp = ph;
        __syncthreads();
        sm[rl][cl] = (rl == 0) ? t : make_double2(0.0, 0.0);
        // the last workgroup to get here advances the exchange counter (nobody reads it again in this launch)
        if (tid == 0) {
            const unsigned done = atomicAdd(pa.arrive, 1u) + 1u;
            if (done == gridDim.x) { *pa.arrive = 0u; *pa.seq = seq + 1ull; }
        }
    }
    const double gamma = wave_min(b_gp.fold_min(a.gpart, a.G));                        // :728 / :734
    const double rtv = (a.j == 1) ? rtv_state : wave_sum(b_rv.fold_sum(a.rvpart_in, a.nrv));   // :732
    int cont = 0, neg = 0, outside = 0;
    double step = 0.0, alpha = QNAN;
    bool add_w = true;
    if (pHp <= a.atol_neg) {                        // :725
        neg = 1;
        if (fabs(pHp) > a.atol_neg) step = gamma;   // :727-729
        else add_w = false;
    } else {
        alpha = __ddiv_rn(rtv, pHp);                // :733
        outside = alpha > gamma;                    // :735
        if (outside) step = gamma;                  // :737
        else { step = alpha; cont = 1; }            // :739
    }
    __syncthreads();

    // ---- update this workgroup's entries ----------------------------------------------------------------------------------
    double rv_part = 0.0;
    if (rl == 0 && valid) {
        double2 hp = sm[0][cl];
#pragma unroll
        for (int q = 1; q < 16; ++q) { hp.x += sm[q][cl].x; hp.y += sm[q][cl].y; }
        // Every vector here is either the caller's buffer with n == 2*nchunks or a zero-padded workspace vector: whole 16-byte
        // chunks are always addressable.  Elements at or beyond n are kept at exactly 0 (never updated: Inf*0 would poison them).
        const bool e0 = 2 * c < a.n, e1 = 2 * c + 1 < a.n;
        if (add_w) {
            wk.x = __dadd_rn(wk.x, __dmul_rn(step, pk.x));           // :729 / :737 / :739
            wk.y = __dadd_rn(wk.y, __dmul_rn(step, pk.y));
            hwk.x = __dadd_rn(hwk.x, __dmul_rn(step, hp.x));
            hwk.y = __dadd_rn(hwk.y, __dmul_rn(step, hp.y));
        }
        double2 vk = make_double2(0.0, 0.0);
        if (cont) {
            rk.x = __dadd_rn(rk.x, __dmul_rn(alpha, hp.x));          // :740
            rk.y = __dadd_rn(rk.y, __dmul_rn(alpha, hp.y));
            vk.x = (fr.x >= 0) ? 0.0 : rk.x;                         // projection!, box case (:741)
            vk.y = (fr.y >= 0) ? 0.0 : rk.y;
        }
        if (!e0) { wk.x = 0.0; hwk.x = 0.0; rk.x = 0.0; vk.x = 0.0; }
        if (!e1) { wk.y = 0.0; hwk.y = 0.0; rk.y = 0.0; vk.y = 0.0; }
        if (cont && !GEN) rv_part = fma(rk.y, vk.y, rk.x * vk.x);    // :743 (this chunk)
        if (add_w || a.j == 1) {
            reinterpret_cast<double2*>(a.w)[c] = wk;
            if (a.hw != nullptr) reinterpret_cast<double2*>(a.hw)[c] = hwk;
        }
        if (cont || a.j == 1) {
            reinterpret_cast<double2*>(a.r)[c] = rk;
            if (!GEN) reinterpret_cast<double2*>(a.v)[c] = vk;
        }
        if (GEN) rsm[cl] = vk;                                       // mask(r_new): what A_free multiplies
    } else if (GEN && rl == 0) {
        rsm[cl] = make_double2(0.0, 0.0);
    }
    if (!GEN) {
        if (rl == 0) rvs[cl] = rv_part;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += rvs[q];
            a.rvpart_out[blockIdx.x] = t;
        }
    } else {
        // left_mul, reduced form (src/polyhedral_constraints.jl:86-98 with the fixed components masked out): this workgroup's
        // share of A_free r for every row of A — thread (rl, cl) takes rows rl, rl+16, ... and chunk cl; the 16 chunk-lanes of a
        // row are one DPP row
        __syncthreads();
        if (cont) {
            const double2 rm = rsm[cl];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = rl + 16 * k;
                if (i < a.mA) {                                   // (uniform per 16-lane row)
                    double prod = valid ? fma(arow[k].y, rm.y, arow[k].x * rm.x) : 0.0;
                    prod = row16_sum(prod);
                    if (cl == 0) a.tpart[(int64_t)blockIdx.x * a.mA + i] = prod;
                }
            }
        }
    }

    // ---- workgroup 0 commits the iteration ----------------------------------------------------------------------------------
    if (blockIdx.x == 0 && tid == 0) {
        st->pHp = pHp; st->gamma = gamma; st->alpha = alpha; st->n_hmul = a.j; st->rtv = rtv;
        st->neg_curvature = neg; st->outside_region = outside; st->need_proj = 0; st->iter = a.j;
        tr.note_step_a(pHp, a.atol_neg, alpha, gamma, a.j);
        tr.store(st);
        if (a.trace != nullptr && a.j <= a.trace_cap) {
            double* row = a.trace + 4 * (int64_t)(a.j - 1);
            row[0] = pHp; row[1] = alpha; row[2] = (neg && !add_w) ? QNAN : gamma; row[3] = rtv;   // [3]: r.v after :746 follows in the next launch
        }
        int status = 4;                          // (a progress word: the host reads the status of a finished loop only)
        if (!cont) {
            status = cg_status_of(0, outside, neg, a.j, max_iter);
            st->approx_solved = 0; st->done = 1; st->stop_at = a.j;
            st->status = status;
        }
        publish_word(a.mirror, a.tag, status, cont ? 0 : 1, a.j, a.j, tr);      // progress (n_hmul = j) or the final state
    }
}

}  // namespace bh
