/*
 * benlsip_hip.h — C ABI of the MI355X (gfx950) backend for BEnlsip.jl's
 * trust-region subproblem hot path.
 *
 * The reference (pure Julia, /root/reference) has no FFI; its seam is Julia
 * multiple dispatch on two concrete types and one function.
 * Every entry point below names the reference method it replaces
 * (path:line relative to the reference root).  julia/BEnlsipHIP.jl holds the
 * `ccall` stubs a maintainer would load next to the unchanged package
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - every function returns int32: 0 = BH_OK, negative = error (bh_strerror);
 *     nothing throws, nothing aborts.
 *   - host pointers are BORROWED for the duration of the call (Julia's GC pins
 *     ccall array arguments exactly that long); anything kept is copied.
 *   - matrices handed over by the host are COLUMN-MAJOR (Julia) with an explicit
 *     leading dimension; indices are 0-based on the C side.
 *   - the only arithmetic type is IEEE fp64.
 *   - one process drives one GPU; handles are not thread-safe (the reference is
 *     single-threaded); every export that takes or returns HOST data is
 *     synchronous unless its name ends _async.
 *   - functions with the suffix _dev take DEVICE pointers (vectors already
 *     resident in HBM; used by bench.py and by callers that keep s, g, w on the
 *     device between calls).  Their device-side results are ordered on the
 *     library stream: they return once the host has what it is owed (status
 *     words, scalars), not necessarily after the stream has drained — see the
 *     option "final_sync" (1 restores a full drain per call).
 *   - multi-GPU: rows of J are sharded over ranks (one process per GPU); each
 *     rank passes its own row block to bh_hess_create*.  After bh_comm_init every
 *     J'·(…) product ends in ONE RCCL all-reduce of n doubles.
 *     Call bh_comm_init BEFORE creating handles (rank 0 alone applies the replicated C rows).
 */
#ifndef BENLSIP_HIP_H
#define BENLSIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------- */
#define BH_OK                 0
#define BH_ERR_INVALID_ARG   -1   /* NULL handle/pointer, negative size, bad leading dimension */
#define BH_ERR_NOT_INIT      -2   /* bh_init has not been called */
#define BH_ERR_HIP           -3   /* a HIP runtime call failed (bh_last_error_detail) */
#define BH_ERR_RCCL          -4   /* an RCCL call failed or librccl could not be loaded */
#define BH_ERR_PRECONDITION  -5   /* reference @assert violated: mpp <= n, mA < mpp when fixed, ... */
#define BH_ERR_SHAPE         -6   /* handle shapes disagree (H.n != P.n, ...) */
#define BH_ERR_NO_DEVICE     -7   /* no gfx950 device visible */
#define BH_ERR_UNSUPPORTED   -8

/* ---- CG_status — src/basic_tralcnlss.jl:12, plus Julia's `nothing` ------ */
#define BH_CG_SOLVED              0
#define BH_CG_BOUND_HIT           1
#define BH_CG_NEGATIVE_CURVATURE  2
#define BH_CG_MAX_ITER_REACHED    3
#define BH_CG_NONE                4   /* reference returns `nothing` */

/* ---- bh_init flags ------------------------------------------------------ */
#define BH_FLAG_PROFILE   1   /* record hipEvents around every H*p launch (bh_stats) */

typedef struct bh_hess bh_hess;   /* device image of AlHessian  — src/basic_tralcnlss.jl:6-10 */
typedef struct bh_proj bh_proj;   /* device image of MixedConstraints — src/polyhedral_constraints.jl:1-7 */

typedef struct bh_stats_t {
    int64_t n_hmul;          /* H*p products executed (fused J'(Jp) launches)            */
    int64_t n_jv;            /* stand-alone J·v launches                                 */
    int64_t n_jtv;           /* stand-alone J'·u launches                                */
    int64_t n_proj;          /* projections applied                                      */
    int64_t n_pcg;           /* bh_pcg calls                                             */
    int64_t n_cg_iter;       /* CG iterations over all bh_pcg calls                      */
    int64_t n_allreduce;     /* RCCL all-reduces issued                                  */
    double  hmul_ms;         /* sum of hipEvent durations of the H*p kernel (BH_FLAG_PROFILE) */
    int64_t hmul_timed;      /* number of H*p launches contributing to hmul_ms           */
    double  bytes_per_hmul;  /* algorithmic bytes of one H*p launch on this rank (8*(d+q)*n + 16*n) */
    /* library-wide (all handles), since bh_init: host <-> device copies the library has issued — vectors staged for the
     * host-pointer entry points, masks, factors, traces, J uploads; 8-byte scalars coming back through pinned memory are
     * included.  The *_dev entry points move none of the n-vectors they work on (tests check the difference of these counters). */
    int64_t h2d_bytes, d2h_bytes, h2d_calls, d2h_calls;
    /* launch shape of the last bh_pcg on this handle: kernels per CG iteration in the fused forms, the collective not counted
     * (box constraints 2, on one rank and over either transport; equalities 3 through the explicit factor inverse — 4 with
     * cg_fused = 2 — on one rank and over the peer buffers, one more over RCCL: the slab reduction in front of ncclAllReduce);
     * 0 = the separate-kernel forms (three kernels for box constraints, seven with equalities; option "cg_fused"). */
    int64_t cg_kernels;
} bh_stats_t;

/* ---- library / device --------------------------------------------------- */

/* Select HIP device `device` for this process, create the stream and workspaces. */
int32_t bh_init(int32_t device, int32_t flags);
int32_t bh_shutdown(void);
/* Use a caller-owned hipStream_t (e.g. torch's current stream) for all launches; NULL restores the library stream. */
int32_t bh_set_stream(void* hip_stream);
int32_t bh_synchronize(void);
const char* bh_strerror(int32_t code);
const char* bh_last_error_detail(void);
/* Name, CU count and arch string of the selected device (diagnostics). */
int32_t bh_device_info(char* name_out, int64_t name_cap, int32_t* n_cu, char* arch_out, int64_t arch_cap);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) --------------------- */
#define BH_UNIQUE_ID_BYTES 128
/* rank 0 creates the id, the host runtime (torch.distributed, MPI, a Julia Distributed channel) broadcasts it. */
int32_t bh_comm_unique_id(void* id_out /* BH_UNIQUE_ID_BYTES */);
/* Two interchangeable transports for the one exchange on the path (the sum over ranks of an n-vector per J'·), chosen by the
 * environment variable BH_COMM when the communicator is created:
 *   "rccl" (default)  ncclAllReduce(n doubles, sum) on the library stream.
 *   "ipc"             one-shot exchange over peer-mapped buffers (hipIpc; ranks of ONE node, <= 8): every rank pushes its
 *                     partial into all inboxes and sums what it received in rank order — one hop instead of a ring, fused
 *                     into the kernel that reduces the per-workgroup slabs, bit-identical on all ranks by construction, and
 *                     skipped on the device by over-launched (already finished) CG iterations.  Needs no librccl.
 *   "both"            both are brought up; bh_set_option("comm_path", 0|1) switches (default RCCL).
 * Both return BH_ERR_PRECONDITION while any bh_hess handle is alive: a handle records at creation whether THIS rank applies
 * the replicated C rows (rank 0 does), so the rank must not change under it.  Order: bh_init, bh_comm_init, create handles,
 * ..., destroy handles, bh_comm_destroy.  (nranks == 1 creates no communicator and is always accepted.)
 * BH_RCCL_LIB (environment) names the librccl to load; default: the copy already mapped into the process, else the system one.
 * With the peer-buffer transport bh_comm_init ends with a REACHABILITY CHECK: one real exchange of a known vector through every
 * mapped inbox (timeout BH_PEER_ECHO_TIMEOUT_S, default 5 s).  If it fails on any rank — the mappings opened but a peer's stores
 * do not land — it fails on EVERY rank (BH_ERR_RCCL, "reachability check failed"), no communicator is left behind, and the caller
 * may call bh_comm_init again with BH_COMM=rccl.  (BH_PEER_TIMEOUT_S, default 20 s: how long a later exchange waits for a peer
 * before it poisons its result with NaN and raises the transport's error state.  BH_PEER_ECHO_SKIP_RANK=r: test hook, rank r
 * stays away from the check.) */
int32_t bh_comm_init(int32_t rank, int32_t nranks, const void* id /* BH_UNIQUE_ID_BYTES */);
int32_t bh_comm_destroy(void);
int32_t bh_comm_info(int32_t* rank, int32_t* nranks);

/* ---- AlHessian ---------------------------------------------------------- */

/* Replaces the constructor AlHessian(Jx,Cx,mu) at src/basic_tralcnlss.jl:46,84: uploads this
 * rank's row block of J (d x n, column-major, leading dimension ldJ >= d) and C (q x n; may be
 * NULL when q == 0; replicated on every rank) and lays them out for the kernels. */
int32_t bh_hess_create(bh_hess** out, const double* J, int64_t d, int64_t n, int64_t ldJ,
                       const double* C, int64_t q, int64_t ldC, double mu);
/* The same for a Jacobian that already lives in HBM (a device-side jac_res): J_dev is a DEVICE pointer,
 * column-major d x n with leading dimension ldJ; only the transpose into the kernels' layout happens (no PCIe traffic for J).
 * C (q x n) stays a host pointer. */
int32_t bh_hess_create_dev(bh_hess** out, const double* J_dev, int64_t d, int64_t n, int64_t ldJ,
                           const double* C, int64_t q, int64_t ldC, double mu);
/* Asynchronous ingest: like bh_hess_create, but returns as soon as the upload has been started.  A worker
 * thread streams J in column chunks (option "upload_chunk_mb", default 64 MiB) through two device staging buffers: the PCIe
 * copy of chunk k+1 overlaps the device transpose of chunk k, and both overlap whatever the caller does next on the host
 * (the reference evaluates the residuals and the constraints of the next point there, src/basic_tralcnlss.jl:352).
 * J must stay valid and unchanged until bh_hess_wait — or the first use of the handle, which waits implicitly — returns;
 * C is small and is copied before the call returns.  With a communicator every rank must call bh_hess_wait (or make its
 * first use of the handle) at the same point of its call sequence. */
int32_t bh_hess_create_async(bh_hess** out, const double* J, int64_t d, int64_t n, int64_t ldJ,
                             const double* C, int64_t q, int64_t ldC, double mu);
int32_t bh_hess_wait(bh_hess* H);
/* Benchmark constructor: rows [row0, row0+d) of the d_total x n synthetic Jacobian of SURVEY.md §8(d),
 * element (i,j) = u(seed, i + j*d_total)/sqrt(d_total) * (colscale ? colscale[j] : 1), generated in HBM. */
int32_t bh_hess_create_synthetic(bh_hess** out, int64_t d, int64_t n, int64_t row0, int64_t d_total,
                                 uint64_t seed, const double* colscale /* n or NULL */, double mu);
int32_t bh_hess_set_mu(bh_hess* H, double mu);
int32_t bh_hess_destroy(bh_hess* H);
int32_t bh_hess_shape(const bh_hess* H, int64_t* d, int64_t* n, int64_t* q);

/* Base.:*(H::AlHessian, v) — src/basic_tralcnlss.jl:102-106:  out = J'(Jv) + C'(mu C v). */
int32_t bh_hmul(bh_hess* H, const double* v, double* out_n);
/* vthv(H,v) — src/basic_tralcnlss.jl:92-96:  ||Jv||^2 + mu ||Cv||^2. */
int32_t bh_vthv(bh_hess* H, const double* v, double* out_scalar);
/* H.J*v — src/basic_tralcnlss.jl:93,103 (this rank's d rows). */
int32_t bh_jv(bh_hess* H, const double* v, double* out_d);
/* H.J'*u — src/basic_tralcnlss.jl:105 and g = Jx'*rx at :45,:74,:893 (u = this rank's d rows; result all-reduced). */
int32_t bh_jtv(bh_hess* H, const double* u, double* out_n);
/* Device-pointer forms (no PCIe traffic). */
int32_t bh_hmul_dev(bh_hess* H, const double* v_dev, double* out_n_dev);
int32_t bh_jv_dev(bh_hess* H, const double* v_dev, double* out_d_dev);
int32_t bh_jtv_dev(bh_hess* H, const double* u_dev, double* out_n_dev);

/* ---- MixedConstraints --------------------------------------------------- */

/* Replaces MixedConstraints(A, chol_aat; l, u) — src/polyhedral_constraints.jl:9-18: uploads lineq = A (mA x n,
 * column-major, ldA >= mA; A may be NULL when mA == 0).  xlow/xupp stay on the Julia side (only the callers use them). */
int32_t bh_proj_create(bh_proj** out, const double* A, int64_t mA, int64_t n, int64_t ldA);
/* Must be called after every change of lincons.fixvars / lincons.chol (active_bounds!, add_active!, update_chol! —
 * src/polyhedral_constraints.jl:62-68,203-261).  fix_chunks = BitVector.chunks (bit i%64 of word i/64 <=> variable i fixed).
 * L = lincons.chol factor, mpp x mpp column-major, ONLY the lower triangle (i >= j) is read;
 * mpp must equal mA + popcount(fix) (or mA when nothing is fixed).  With mA == 0 the factor is the identity and L may be NULL. */
int32_t bh_proj_set_active(bh_proj* P, const uint64_t* fix_chunks, int64_t n,
                           const double* L, int64_t mpp, int64_t ldL);
int32_t bh_proj_destroy(bh_proj* P);
int32_t bh_proj_shape(const bh_proj* P, int64_t* mA, int64_t* n, int64_t* n_fixed);
/* projection!(lincons, r, v) / projection(lincons, r) — src/polyhedral_constraints.jl:150-170
 * (projection_nullspace! :104-118 when nothing is fixed, projection_subspace! :120-136 otherwise). */
int32_t bh_project(bh_proj* P, const double* r, double* v_out);
int32_t bh_project_dev(bh_proj* P, const double* r_dev, double* v_out_dev);
/* left_mul(lincons, x) — src/polyhedral_constraints.jl:86-98:  out (mpp) = [A x ; x[fixvars]]. */
int32_t bh_left_mul(bh_proj* P, const double* x, double* out_mpp);
/* left_mul_tr(lincons, y) — src/polyhedral_constraints.jl:72-84:  out (n) = A'y[1:mA] (+ scatter y[mA+1:end]). */
int32_t bh_left_mul_tr(bh_proj* P, const double* y, double* out_n);

/* ---- projected_cg -------------------------------------------------------- */

/* projected_cg(g_minor, H, w_l, w_u, lincons, kappa2; atol) — src/basic_tralcnlss.jl:690-764, with
 * factor_to_boundary — :793-809 — evaluated on the device.  The whole loop (H*p, dots, updates, projection, branch
 * scalars, exit test) is device-resident; the host only polls a done flag.
 *   atol_negcurv = sqrt(eps) (:697),  atol_f2b = 1e-10 (:798).
 *   status: BH_CG_*;  iters: the reference's `iter` variable at exit (starts at 1);
 *   trace (optional, may be NULL): row k = {pHp, alpha, gamma, rtv} of the k-th H*p product, trace_cap rows.
 *   n_hmul (optional): number of H*p products performed. */
int32_t bh_pcg(bh_hess* H, bh_proj* P, const double* g_minor, const double* w_l, const double* w_u,
               double kappa2, double atol_negcurv, double atol_f2b,
               double* w_out, int32_t* status, int32_t* iters,
               double* trace, int64_t trace_cap, int32_t* n_hmul);
int32_t bh_pcg_dev(bh_hess* H, bh_proj* P, const double* g_minor_dev, const double* w_l_dev, const double* w_u_dev,
                   double kappa2, double atol_negcurv, double atol_f2b,
                   double* w_out_dev, int32_t* status, int32_t* iters,
                   double* trace, int64_t trace_cap, int32_t* n_hmul);
/* Tie log of the LAST projected_cg run on this handle (bh_pcg, bh_pcg_dev, bh_minor_iterate) — SURVEY.md §8(c): the loop's
 * branches (src/basic_tralcnlss.jl:725 pHp <= tol_negcurve, :727 |pHp| > tol, :735 alpha > gamma, :747 |rtv| < tol_cg) decide
 * status and iteration count, so a scalar within rounding of its threshold may legitimately flip against another
 * implementation.  tie_flags: BH_TIE_* bits of the tests that came within 1e-10 (relative) of their threshold;
 * first_tie_hmul: the H*p product at which that first happened (0 = never); min_margin / _kind / _hmul: the smallest
 * relative distance |a-b|/max(|a|,|b|) any of those tests had in the call, which test, and at which product. */
#define BH_TIE_NEGCURV      1   /* :725 */
#define BH_TIE_NEGCURV_ABS  2   /* :727 */
#define BH_TIE_BOUND        4   /* :735 */
#define BH_TIE_TOL          8   /* :747 */
int32_t bh_pcg_tie_info(const bh_hess* H, int32_t* tie_flags, int32_t* first_tie_hmul, double* min_margin,
                        int32_t* min_margin_kind, int32_t* min_margin_hmul);

/* ---- callers of projected_cg, device-resident (SURVEY.md §8 a9, a10 and "next" row f-2) ----------------- */

/* minor_iterate(x, s, g_model, H, lincons, delta, kappa2) — src/basic_tralcnlss.jl:649-675: builds w_l/w_u exactly as :660-665
 * (only the FIXED variables get finite bounds), runs projected_cg (:667) and, unless the status is
 * negative_curvature, linesearch and w .= alpha*w (:669-672) — one call, no intermediate PCIe round trips.
 * xlow/xupp = lincons.xlow/xupp.  alpha_out (optional) receives the line-search factor (NaN when it was skipped).
 * The line search's w'Hw is taken from H*w accumulated inside the CG loop (option "ls_from_cg", default 1) instead of a
 * separate vthv(H,w) sweep; same value up to rounding. */
int32_t bh_minor_iterate(bh_hess* H, bh_proj* P, const double* x, const double* s, const double* g_model,
                         const double* xlow, const double* xupp, double delta, double kappa2,
                         double atol_negcurv, double atol_f2b, double* w_out, int32_t* status, int32_t* iters,
                         int32_t* n_hmul, double* alpha_out);
/* linesearch(g_model, H, w, w_l, w_u, lincons.fixvars) — src/basic_tralcnlss.jl:766-791. */
int32_t bh_linesearch(bh_hess* H, bh_proj* P, const double* g_model, const double* w, const double* w_l, const double* w_u,
                      double* alpha_out);
/* cauchy_step(x, g, H, chol_aat, lincons, delta) — src/basic_tralcnlss.jl:574-639 (with next_breakpoint :536-562 and the
 * initial active_bounds!, src/polyhedral_constraints.jl:203-215), device-resident ("next" row f-3).  Per breakpoint: one
 * H*d, one projection, and a rank-one Gram downdate + mA x mA Cholesky on the device in place of the reference's O(p^3)
 * add_active! -> cholesky_aug_aat rebuild.  Needs the reduced projection form (default).  On return the handle holds the
 * final active set; fix_chunks_out (ceil(n/64) words, optional) receives it in BitVector.chunks layout so the caller can
 * update lincons.fixvars (and its own factor, if it still needs one). */
int32_t bh_cauchy_step(bh_hess* H, bh_proj* P, const double* x, const double* g, const double* xlow, const double* xupp,
                       double delta, double* s_out, uint64_t* fix_chunks_out, int32_t* n_breakpoints, int32_t* n_hmul);
/* g = Jx'*rx + Cx'*y_bar — src/basic_tralcnlss.jl:45 (new_point), :74 (first_derivatives); r = this rank's d rows, y_bar has q entries. */
int32_t bh_grad(bh_hess* H, const double* r, const double* ybar, double* g_out);
/* dot(rx,rx) of the augmented-Lagrangian value mx = 0.5*dot(rx,rx) + dot(y,cx) + 0.5*mu*dot(cx,cx) — src/basic_tralcnlss.jl:44
 * (new_point) and :58 (evaluate_al).  r = this rank's d rows of the residual; the ranks' partial sums are all-reduced in rank
 * order, so out is the GLOBAL squared norm, bit-identical on every rank (with one rank: plain dot(r,r)).  Row-sharded callers
 * need it to keep the replicated trust-region control flow (rho = ared/pred, :353-358) in lock-step. */
int32_t bh_resid_sqnorm(const double* r, int64_t d, double* out);
/* g_minor = H*s + g — src/basic_tralcnlss.jl:412,:437. */
int32_t bh_hmul_add(bh_hess* H, const double* s, const double* g, double* out_n);

/* ---- the minor loop of inner_step with its vectors resident in HBM --------------------------------
 * src/basic_tralcnlss.jl:410-458: s = cauchy_step(...); g_minor = H*s+g; while ...: w = minor_iterate(...); s .+= w;
 * g_minor = H*s+g; active_bounds / add_active!; norm_reduced_gradient x 2; ...; model_reduction.  Every *_dev entry point takes
 * and returns DEVICE pointers for the n-vectors (x, s, g, g_minor, w, xlow, xupp); only scalars, counts and the n/8-byte
 * BitVector image of the active set cross PCIe (bh_stats' h2d/d2h counters let a caller check that).  A caller that owns the
 * loop (julia/BEnlsipHIP.jl's inner_step method; tests/hip_ops.py::DeviceResidentInnerStep) uploads x, g and the bounds once
 * per trust-region iteration and downloads s once. */
int32_t bh_cauchy_step_dev(bh_hess* H, bh_proj* P, const double* x_dev, const double* g_dev, const double* xlow_dev,
                           const double* xupp_dev, double delta, double* s_out_dev, uint64_t* fix_chunks_out /* host, optional */,
                           int32_t* n_breakpoints, int32_t* n_hmul);
int32_t bh_minor_iterate_dev(bh_hess* H, bh_proj* P, const double* x_dev, const double* s_dev, const double* g_model_dev,
                             const double* xlow_dev, const double* xupp_dev, double delta, double kappa2, double atol_negcurv,
                             double atol_f2b, double* w_out_dev, int32_t* status, int32_t* iters, int32_t* n_hmul, double* alpha_out);
int32_t bh_linesearch_dev(bh_hess* H, bh_proj* P, const double* g_model_dev, const double* w_dev, const double* w_l_dev,
                          const double* w_u_dev, double* alpha_out);
/* r_dev = this rank's d residual rows in HBM (a device-side residual callback); y_bar (q entries) stays a host vector. */
int32_t bh_grad_dev(bh_hess* H, const double* r_dev, const double* ybar, double* g_out_dev);
int32_t bh_hmul_add_dev(bh_hess* H, const double* s_dev, const double* g_dev, double* out_n_dev);
/* s .+= w ; g_minor = H*s + g — src/basic_tralcnlss.jl:436-437 (s_dev is updated in place). */
int32_t bh_step_accumulate_dev(bh_hess* H, double* s_dev, const double* w_dev, const double* g_dev, double* g_minor_out_dev);
/* src/basic_tralcnlss.jl:439-453 on the device-side active set:
 *     active_indx = active_bounds(lincons, x, s, delta)                      src/polyhedral_constraints.jl:219-237 (atol = sqrt(eps))
 *     if mA + |active_indx| <= n:  add_active!(lincons, chol_aat, active_indx)    poly:252-261      -> *branch = 0
 *     else:                        active_bounds!(lincons, x+s, chol_aat)         poly:203-215      -> *branch = 1
 * Newly fixed variables leave A_free A_free' through a Gram DOWNDATE over their columns followed by the mA x mA factorisation
 * (the reference rebuilds the (mA+p) x (mA+p) augmented factor from scratch, O(p^3)).  n_at_bound = |active_indx|,
 * n_fixed = count(fixvars) afterwards; fix_chunks_out (host, optional, ceil(n/64) words) receives the new lincons.fixvars.chunks. */
int32_t bh_proj_update_active_dev(bh_proj* P, const double* x_dev, const double* s_dev, const double* xlow_dev, const double* xupp_dev,
                                  double delta, double atol, int32_t* n_at_bound, int32_t* n_fixed, int32_t* branch,
                                  uint64_t* fix_chunks_out);
/* norm_reduced_gradient(g, lincons) = norm(projection(lincons, -g)) — src/basic_tralcnlss.jl:869-875 (also criticality_measure :839). */
int32_t bh_reduced_gradient_norm_dev(bh_proj* P, const double* g_dev, double* out);
/* model_reduction = dot(g,s) + 0.5*vthv(H,s) — src/basic_tralcnlss.jl:458. */
int32_t bh_model_reduction_dev(bh_hess* H, const double* g_dev, const double* s_dev, double* out);

/* factor_to_boundary(p, w, w_l, w_u; atol) — src/basic_tralcnlss.jl:793-809, stand-alone (tests). */
int32_t bh_factor_to_boundary(const double* p, const double* w, const double* w_l, const double* w_u,
                              int64_t n, double atol, double* gamma_out);

/* ---- plumbing ------------------------------------------------------------ */
int32_t bh_dev_alloc(void** out, int64_t bytes);
int32_t bh_dev_free(void* p);
int32_t bh_dev_upload(void* dst_dev, const void* src_host, int64_t bytes);
int32_t bh_dev_download(void* dst_host, const void* src_dev, int64_t bytes);
int32_t bh_stats(bh_hess* H, bh_stats_t* out);
int32_t bh_stats_reset(bh_hess* H);
/* Tuning knobs; unknown keys return BH_ERR_INVALID_ARG.  Defaults in brackets.
 *   "proj_form"      [1] 1 = reduced mA x mA projection (factor built on the device), 0 = the reference's augmented form
 *                        (needs the caller's factor in bh_proj_set_active)
 *   "pcg_batch"      [0] CG iterations enqueued per launch-ahead batch; 0 = by problem size (1 when an H*p streams >= 100 us)
 *   "fold_init"      [1] box constraints: fold projected_cg's initialisation into the first H*p / step launches
 *   "cg_fused"       [1] shape of a CG iteration.  1 (default): box constraints in two kernels (the H*p launch forms p and takes the
 *                        exit test, one kernel reduces the slabs — exchanging them between the ranks of a peer-buffer communicator —
 *                        and updates w, r, v; over RCCL: two kernels + the collective, the update of the previous iteration
 *                        living in the prologue of the H*p launch); linear equalities (reduced form, mA <= 64, one rank) in
 *                        three kernels (H*p, reduce/update, projection with the explicit inverse of the reduced factor).
 *                        2: linear equalities in four kernels (triangular solves in a launch of their own); box as 1.
 *                        0: the round-1 shapes (three kernels box, seven with equalities; p'Hp = dot(p, H*p) exactly as the
 *                        reference forms it, where 1 and 2 form it as sum_i w_i (Jp)_i^2)
 *   "final_sync"     [0] device-pointer entry points (*_dev): 1 = always drain the library stream before returning, as the
 *                        host-pointer entry points do.  0 = return as soon as everything the HOST is owed has arrived; results
 *                        that stay in HBM are ordered on the library stream (later calls see them; bh_synchronize, or sharing
 *                        the caller's stream through bh_set_stream, orders them for anybody else):
 *                        - bh_pcg_dev: when the caller's device vectors are used in place and the loop was stopped by its exit
 *                          test (solved / iterations exhausted), w was complete before the launch that reported the stop began,
 *                          and only prologue-only launches that write nothing are still in flight (saves ~10 us per call);
 *                        - bh_minor_iterate_dev, bh_reduced_gradient_norm_dev, bh_model_reduction_dev, bh_proj_update_active_dev,
 *                          bh_cauchy_step_dev: counts, norms, alpha and the BitVector image come back through a host-mapped
 *                          mailbox page the kernels write and seal with a sequence number (no DMA per scalar, no stream
 *                          synchronize: 84 -> 28 us for bh_proj_update_active_dev, 25 -> 11 us for a reduced-gradient norm);
 *                        - bh_hmul_dev, bh_hmul_add_dev, bh_step_accumulate_dev, bh_jv_dev, bh_jtv_dev, bh_project_dev owe the
 *                          host nothing and return once their work is enqueued.
 *                        A caller's device vector is read where it lies when it needs no padding (n a multiple of 16, 16-byte
 *                        aligned); every other case, and every host-pointer entry point, behaves as before.
 *   "host_copy_kernels" [1] host-pointer entry points: the caller's vectors travel between the pinned arena and HBM by a small
 *                        copy kernel on the mapped arena instead of a DMA engine transfer, and the call ends with a mailbox seal +
 *                        poll instead of hipStreamSynchronize (bh_pcg 0.660 -> 0.652 ms, bh_minor_iterate 0.693 -> 0.665 ms,
 *                        bh_project 31 -> 21 us on the config-3 instance); 0 = DMA + synchronize (round 1)
 *   "mailbox_flush"  [0] experiment: end the host-pointer entry points with a mailbox seal + poll instead of
 *                        hipStreamSynchronize (measured slower behind a D2H DMA; docs/design_history_r1_r2.md §4)
 *   "ls_from_cg"     [1] bh_minor_iterate: w'Hw of the line search from the H*w the CG loop accumulated (0: explicit vthv)
 *   "step_from_cg"   [0] opt-in (the resident inner-step mirrors set it around their loop): bh_step_accumulate_dev called right
 *                        behind the bh_minor_iterate_dev that produced its w (same handle,
 *                        w_dev = that call's w_out_dev, g_minor_out_dev = that call's g_model_dev, which holds H*s + g as in
 *                        src/basic_tralcnlss.jl:412,:434-437): g_minor += H*w with the H*w that CG loop accumulated instead of a
 *                        fresh sweep H*(s + w) + g over J (same value, rounded differently; one H-product less per minor
 *                        iterate).  It trusts the caller's invariant that g_minor_out_dev holds H*s + g for the CURRENT s.
 *                        Under the same option bh_model_reduction_dev(H, g, s) takes s'Hs = s.(g_minor - g) from the g_minor the
 *                        library last wrote for exactly these H, s, g (bh_hmul_add_dev / bh_step_accumulate_dev) instead of a J v sweep.
 *                        0, or any other calling pattern: the explicit product.  Needs "ls_from_cg" = 1.
 *   "chol_downdate"  [0] bh_cauchy_step, per breakpoint: 0 = downdate the Gram matrix and refactor (as accurate as the reference's
 *                        from-scratch rebuild), 1 = for mA > 64 only: rank-one downdate of the factor itself (O(mA^2)), rebuilt from
 *                        scratch every 8th breakpoint (errors accumulate in between; up to 64 rows the refactoring path is as fast)
 *   "cauchy_image"   [1] bh_cauchy_step with box constraints on one rank: search in the row space of J (J d and J s_c maintained by
 *                        one-column updates; d'Hd = ||J d||^2_W, s'Hd = (J s).(J d)_W: the same numbers as dot(d, H*d), dot(s, H*d),
 *                        rounded differently): one J v sweep at the start instead of one H*d sweep per breakpoint.  0: as the reference
 *   "cauchy_image_max_ma" [64] ... and with up to this many linear equalities (0..64): there the row-space form keeps a = J D g and
 *                        B = J D A' (rows x mA) next to J d, J s_c — two sweeps over J up front (a: J v; B: "cauchy_gemm"), one
 *                        column of J per breakpoint afterwards; above this value and up to 64 rows the form is used when the
 *                        previous search on the same bh_proj took more than 4 (1 + mA) passes
 *   "cauchy_fused"   [1] that search (box constraints, one rank) with ONE kernel per breakpoint: every workgroup of the row kernel redoes the
 *                        previous pass's decision in its prologue (s_c and the loop state are ping-pong buffers); 0: two kernels per pass
 *   "cauchy_fused_grid" [0] workgroups of that kernel (0: one row per thread up to 256 workgroups; measured best — tools/scratch/cauchy_grid_sweep.py)
 *   "linv_refine"    [1] three-kernel CG iteration with linear equalities (cg_fused = 1): one step of iterative refinement behind the
 *                        explicit inverse of the factor (rho = t - A_free A_free' y, y += L^-T L^-1 rho), so that A_free v stays at the level
 *                        of the reference's two triangular solves also for ill-conditioned A_free A_free' (0: plain explicit inverse;
 *                        not applied with a single equality, where the factor is a scalar)
 *   "cauchy_gemm"    [1] B = J D A' in ONE sweep over J on the fp64 matrix cores (a tall-skinny GEMM: M = rows of J, N = mA, K = n);
 *                        0: mA J v sweeps over the masked rows of A
 *   "chol_blocked"   [1] mA > 64: blocked potrf / trsm / syrk (0: one-workgroup kernel)
 *   "gram_mfma"      [1] A_free A_free' on fp64 MFMA when mA > 96 (2: always, 0: never)
 *   "rs_variant"     [0] A/B geometries of the row-streaming kernel for 2048 < n <= 4096 (tools/kernel_ab.py)
 *   "blocks_per_cu"  [0] workgroups per CU of the row-streaming kernels (0: per-geometry default)
 *   "pingpong"       [0] alternate the sweep direction of J between consecutive H*p
 *   "blocks_per_cu" accepts 0..8 (the partial-slab buffers hold 8 workgroups per CU); anything else is BH_ERR_INVALID_ARG
 *   "comm_path"      [0 with BH_COMM=rccl|both, 1 with BH_COMM=ipc] which communicator carries the all-reduces: 0 = RCCL,
 *                        1 = the one-shot peer-buffer exchange fused into the slab reduction (needs BH_COMM=ipc or both)
 *   "profile"        [flags of bh_init] 1 = hipEvents around every profile_stride-th H*p launch (bh_stats: hmul_ms / hmul_timed)
 *   "image_pool"     [2] Jacobian images of destroyed handles kept for the next bh_hess_create* of a similar size (0..8; 0 frees at
 *                        once).  The reference builds a new AlHessian per accepted step and drops the old one: recycling the
 *                        2 GiB image avoids the driver's background scrub of freed VRAM and a ~4 ms hipMalloc per step.
 *   "upload_chunk_mb" [64] bh_hess_create_async: MiB of J per pipelined column chunk (1..4096)
 *   "profile_stride" [8] >= 1; an event pair costs ~10 us of stream time, so 1 is for short runs only (at most 512 samples per call) */
int32_t bh_set_option(const char* key, int64_t value);
/* Time `reps` back-to-back launches of one kernel class with hipEvents on the launch stream.
 * kind: 0 = fused J'(Jp), 1 = J·v, 2 = J'·u; 3..6 = read-only stream probe over the same image (nothing but 16-byte
 * non-temporal loads and adds) with 1, 2, 4, 8 workgroups per CU: the practical single-read ceiling the kernels are
 * quoted against; 7 = the all-reduce of one n-vector on the active communicator path (every rank must call it together);
 * 8 = everything an H*p does after its streaming kernel (slab reduction + all-reduce; without a communicator: the slab
 * reduction alone).  Returns the average milliseconds per launch. */
int32_t bh_time_kernel(bh_hess* H, int32_t kind, int32_t reps, double* avg_ms);
/* Device self-test of the wave64 DPP/permlane reduction network (sum and NaN-propagating min). */
int32_t bh_selftest(void);

#ifdef __cplusplus
}
#endif
#endif /* BENLSIP_HIP_H */
