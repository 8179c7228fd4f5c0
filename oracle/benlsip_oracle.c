/*
 * benlsip_oracle.c — plain-C restatement of the BEnlsip.jl hot path (second, independent CPU oracle).
 *
 * TEST INFRASTRUCTURE ONLY: checker for the HIP path and the timed CPU baseline ("port") of bench.py.  Nothing in
 * benlsip.jl_amd/ links or calls it.  Parity pinning: same as oracle/benlsip_ref.py (HS48 known answer of
 * test/structures.jl:37-58 etc.; projected_cg itself has no golden data in the reference) — the C and NumPy
 * restatements are additionally checked against each other in tests/test_oracle_c_cpu.py.
 *
 * All matrices column-major with leading dimension (Julia layout).  Citations: path:line relative to /root/reference.
 * OpenMP parallelises the two dense products exactly where the reference's BLAS threads do (dgemv N / T).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int bo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Cap the OpenMP team (medium-sized test problems run an order of magnitude slower on a 128-thread team than on 8). */
void bo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* t = J v  (dgemv 'N'), J d x n column-major.  Row blocks per thread, columns streamed with unit stride. */
static void gemv_n(const double* J, long d, long n, long ld, const double* v, double* t) {
#pragma omp parallel if (d * n > 2000000L)   /* small products: a parallel region costs more than the product */
    {
        int nt = 1, id = 0;
#ifdef _OPENMP
        nt = omp_get_num_threads(); id = omp_get_thread_num();
#endif
        long blk = (d + nt - 1) / nt;
        blk = (blk + 7) & ~7L;
        long lo = id * blk, hi = lo + blk > d ? d : lo + blk;
        if (lo < hi) {
            for (long i = lo; i < hi; ++i) t[i] = 0.0;
            for (long j = 0; j < n; ++j) {
                const double vj = v[j];
                const double* col = J + j * ld;
                for (long i = lo; i < hi; ++i) t[i] += col[i] * vj;
            }
        }
    }
}

/* z = J' t  (dgemv 'T'): one dot product per column. */
static void gemv_t(const double* J, long d, long n, long ld, const double* t, double* z) {
#pragma omp parallel for schedule(static) if (d * n > 2000000L)
    for (long j = 0; j < n; ++j) {
        const double* col = J + j * ld;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        long i = 0;
        for (; i + 3 < d; i += 4) {
            s0 += col[i] * t[i]; s1 += col[i + 1] * t[i + 1]; s2 += col[i + 2] * t[i + 2]; s3 += col[i + 3] * t[i + 3];
        }
        for (; i < d; ++i) s0 += col[i] * t[i];
        z[j] = (s0 + s1) + (s2 + s3);
    }
}

static double dot(const double* a, const double* b, long n) {
    double s = 0.0;
    for (long i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* Base.:*(H, v) — src/basic_tralcnlss.jl:102-106:  J'(Jv) + C'((mu*C) v).  work: d + 2q + n doubles. */
void bo_hmul(const double* J, long d, long n, long ldJ, const double* C, long q, long ldC, double mu, const double* v,
             double* out, double* work) {
    double* Jv = work;
    double* muCv = work + d;
    double* z2 = work + d + q;
    gemv_n(J, d, n, ldJ, v, Jv);                         /* :103 */
    for (long i = 0; i < q; ++i) {                       /* :104  (mu*C)*v */
        double s = 0.0;
        for (long j = 0; j < n; ++j) s += (mu * C[i + j * ldC]) * v[j];
        muCv[i] = s;
    }
    gemv_t(J, d, n, ldJ, Jv, out);                       /* :105 */
    if (q > 0) {
        gemv_t(C, q, n, ldC, muCv, z2);
        for (long j = 0; j < n; ++j) out[j] += z2[j];
    }
}

/* vthv(H, v) — src/basic_tralcnlss.jl:92-96. */
double bo_vthv(const double* J, long d, long n, long ldJ, const double* C, long q, long ldC, double mu, const double* v,
               double* work) {
    double* Jv = work;
    double* Cv = work + d;
    gemv_n(J, d, n, ldJ, v, Jv);
    gemv_n(C, q, n, ldC, v, Cv);
    return dot(Jv, Jv, d) + mu * dot(Cv, Cv, q);
}

/* projection!(lincons, r, v) — src/polyhedral_constraints.jl:104-136,158-170.
 * fix: n bytes (0/1); L: mpp x mpp lower factor (only i >= j read), mpp = mA + count(fix).  work: 2*mpp doubles. */
void bo_projection(const double* A, long mA, long n, long ldA, const unsigned char* fix, const double* L, long mpp, long ldL,
                   const double* r, double* v, double* work) {
    double* y = work;
    /* left_mul (:86-98) */
    for (long i = 0; i < mA; ++i) {
        double s = 0.0;
        for (long j = 0; j < n; ++j) s += A[i + j * ldA] * r[j];
        y[i] = s;
    }
    long k = mA;
    for (long j = 0; j < n; ++j) if (fix[j]) y[k++] = r[j];
    /* y = L \ t ; w = L' \ y  (:114-115, :132-133) */
    for (long i = 0; i < mpp; ++i) {
        double s = y[i];
        for (long c = 0; c < i; ++c) s -= L[i + c * ldL] * y[c];
        y[i] = s / L[i + i * ldL];
    }
    for (long i = mpp - 1; i >= 0; --i) {
        double s = y[i];
        for (long c = i + 1; c < mpp; ++c) s -= L[c + i * ldL] * y[c];
        y[i] = s / L[i + i * ldL];
    }
    /* v = r - left_mul_tr(w)  (:72-84, :116, :134) */
    k = mA;
    for (long j = 0; j < n; ++j) {
        double s = 0.0;
        for (long i = 0; i < mA; ++i) s += A[i + j * ldA] * y[i];
        if (fix[j]) s += y[k++];
        v[j] = r[j] - s;
    }
}

/* Julia's min: NaN-propagating. */
static double jl_min(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }

/* factor_to_boundary — src/basic_tralcnlss.jl:793-809. */
double bo_factor_to_boundary(const double* p, const double* w, const double* wl, const double* wu, long n, double atol) {
    double gamma = INFINITY;
    for (long i = 0; i < n; ++i) {
        if (p[i] <= -atol) gamma = jl_min(gamma, (wl[i] - w[i]) / p[i]);
        else if (p[i] >= atol) gamma = jl_min(gamma, (wu[i] - w[i]) / p[i]);
    }
    return gamma;
}

/* projected_cg — src/basic_tralcnlss.jl:690-764.  status: 0..3 = CG_status, 4 = `nothing`.  trace rows: pHp, alpha, gamma, rtv. */
int bo_projected_cg(const double* J, long d, long n, long ldJ, const double* C, long q, long ldC, double mu,
                    const double* A, long mA, long ldA, const unsigned char* fix, const double* L, long mpp, long ldL,
                    const double* g_minor, const double* w_l, const double* w_u, double kappa2, double atol, double atol_f2b,
                    double* w, int* status_out, int* iters_out, int* n_hmul_out, double* trace, long trace_cap) {
    long nfix = 0;
    for (long j = 0; j < n; ++j) nfix += fix[j] ? 1 : 0;
    double* buf = (double*)calloc((size_t)(4 * n + d + 2 * q + n + 2 * mpp + 8), sizeof(double));
    if (!buf) return -1;
    double *r = buf, *v = buf + n, *p = buf + 2 * n, *Hp = buf + 3 * n, *work = buf + 4 * n;
    double* pwork = work + d + 2 * q + n;
    for (long i = 0; i < n; ++i) { w[i] = 0.0; r[i] = g_minor[i]; }                   /* :702-705 */
    bo_projection(A, mA, n, ldA, fix, L, mpp, ldL, r, v, pwork);                      /* :706 */
    double rtv = dot(r, v, n);                                                        /* :707 */
    for (long i = 0; i < n; ++i) p[i] = -v[i];                                        /* :708 */
    const double tol_cg = kappa2 * sqrt(dot(v, v, n));                                /* :710 */
    int iter = 1;                                                                     /* :713 */
    const long max_iter = 2 * (n - mA - nfix);                                        /* :714 */
    int approx_solved = 0, neg = 0, outside = 0, n_hmul = 0;
    while (!approx_solved && !outside && !neg && iter <= max_iter) {                  /* :720 */
        bo_hmul(J, d, n, ldJ, C, q, ldC, mu, p, Hp, work);                            /* :722 */
        const double pHp = dot(p, Hp, n);                                             /* :723 */
        double alpha = NAN, gamma = NAN;
        ++n_hmul;
        if (pHp <= atol) {                                                            /* :725 */
            neg = 1;
            if (fabs(pHp) > atol) {
                gamma = bo_factor_to_boundary(p, w, w_l, w_u, n, atol_f2b);
                for (long i = 0; i < n; ++i) w[i] = w[i] + gamma * p[i];
            }
        } else {
            rtv = dot(r, v, n);                                                       /* :732 */
            alpha = rtv / pHp;
            gamma = bo_factor_to_boundary(p, w, w_l, w_u, n, atol_f2b);
            outside = alpha > gamma;
            if (outside) {
                for (long i = 0; i < n; ++i) w[i] = w[i] + gamma * p[i];              /* :737 */
            } else {
                for (long i = 0; i < n; ++i) { w[i] = w[i] + alpha * p[i]; r[i] = r[i] + alpha * Hp[i]; }   /* :739-740 */
                bo_projection(A, mA, n, ldA, fix, L, mpp, ldL, r, v, pwork);          /* :741 */
                const double rtv_next = dot(r, v, n);
                const double beta = rtv_next / rtv;
                for (long i = 0; i < n; ++i) p[i] = -v[i] + beta * p[i];              /* :745 */
                rtv = rtv_next;
                approx_solved = fabs(rtv) < tol_cg;
                ++iter;
            }
        }
        if (trace && n_hmul <= trace_cap) {
            double* row = trace + 4 * (long)(n_hmul - 1);
            row[0] = pHp; row[1] = alpha; row[2] = gamma; row[3] = rtv;
        }
    }
    int status;
    if (approx_solved) status = 0;                                                    /* :753-761 */
    else if (outside) status = 1;
    else if (neg) status = 2;
    else if (iter == max_iter) status = 3;
    else status = 4;
    *status_out = status; *iters_out = iter; *n_hmul_out = n_hmul;
    free(buf);
    return 0;
}
