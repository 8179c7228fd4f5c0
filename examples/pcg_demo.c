/*
 * pcg_demo.c — a plain-C caller of the C ABI (include/benlsip_hip.h), i.e. what Julia's `ccall` does, without Python.
 *
 *   gcc -O2 -Iinclude examples/pcg_demo.c -o pcg_demo -Lbenlsip.jl_amd/lib -lbenlsip_hip -Wl,-rpath,$PWD/benlsip.jl_amd/lib -lm
 *   ./pcg_demo problem.bin result.bin
 *
 * problem.bin (little-endian): int64 d, n, q, mA, mpp ; double mu, kappa2 ;
 *   J (d*n, column-major), C (q*n), A (mA*n), L (mpp*mpp), fix (n doubles: 0/1), g, w_l, w_u (n each).
 * result.bin: int64 status, iters, n_hmul ; double w[n], Hg[n] (= H*g), v[n] (= projection of g), vthv(H,g).
 * tests/test_c_abi_demo_gpu.py writes the problem, runs this program and checks the result against the oracle.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "benlsip_hip.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int32_t rc_ = (call);                                                                    \
        if (rc_ != BH_OK) {                                                                      \
            fprintf(stderr, "%s failed: %s (%d): %s\n", #call, bh_strerror(rc_), rc_, bh_last_error_detail()); \
            return 2;                                                                            \
        }                                                                                        \
    } while (0)

static double* read_doubles(FILE* f, int64_t count) {
    double* p = (double*)malloc(sizeof(double) * (size_t)(count > 0 ? count : 1));
    if (count > 0 && fread(p, sizeof(double), (size_t)count, f) != (size_t)count) { fprintf(stderr, "short read\n"); exit(3); }
    return p;
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("problem"); return 1; }
    int64_t hdr[5];
    double par[2];
    if (fread(hdr, sizeof(int64_t), 5, f) != 5 || fread(par, sizeof(double), 2, f) != 2) { fprintf(stderr, "bad header\n"); return 1; }
    const int64_t d = hdr[0], n = hdr[1], q = hdr[2], mA = hdr[3], mpp = hdr[4];
    const double mu = par[0], kappa2 = par[1];
    double* J = read_doubles(f, d * n);
    double* C = read_doubles(f, q * n);
    double* A = read_doubles(f, mA * n);
    double* L = read_doubles(f, mpp * mpp);
    double* fix = read_doubles(f, n);
    double* g = read_doubles(f, n);
    double* w_l = read_doubles(f, n);
    double* w_u = read_doubles(f, n);
    fclose(f);

    /* Julia BitVector.chunks layout */
    const int64_t nwords = (n + 63) / 64;
    uint64_t* chunks = (uint64_t*)calloc((size_t)nwords, sizeof(uint64_t));
    for (int64_t i = 0; i < n; ++i)
        if (fix[i] != 0.0) chunks[i >> 6] |= (uint64_t)1 << (i & 63);

    bh_hess* H = NULL;
    bh_proj* P = NULL;
    CHECK(bh_init(0, 0));
    CHECK(bh_selftest());
    CHECK(bh_hess_create(&H, J, d, n, d > 0 ? d : 1, q > 0 ? C : NULL, q, q > 0 ? q : 1, mu));       /* AlHessian(J, C, mu) */
    CHECK(bh_proj_create(&P, mA > 0 ? A : NULL, mA, n, mA > 0 ? mA : 1));                             /* MixedConstraints(A, ...) */
    CHECK(bh_proj_set_active(P, chunks, n, mpp > 0 ? L : NULL, mpp, mpp > 0 ? mpp : 1));              /* fixvars + chol */

    double* w = (double*)malloc(sizeof(double) * (size_t)n);
    double* Hg = (double*)malloc(sizeof(double) * (size_t)n);
    double* v = (double*)malloc(sizeof(double) * (size_t)n);
    double gHg = 0.0;
    int32_t status = -1, iters = 0, n_hmul = 0;
    CHECK(bh_hmul(H, g, Hg));                                                                          /* H * g */
    CHECK(bh_vthv(H, g, &gHg));                                                                        /* vthv(H, g) */
    CHECK(bh_project(P, g, v));                                                                        /* projection(lincons, g) */
    CHECK(bh_pcg(H, P, g, w_l, w_u, kappa2, 1.4901161193847656e-08, 1e-10, w, &status, &iters, NULL, 0, &n_hmul));   /* projected_cg */

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror("result"); return 1; }
    int64_t res[3] = {status, iters, n_hmul};
    fwrite(res, sizeof(int64_t), 3, o);
    fwrite(w, sizeof(double), (size_t)n, o);
    fwrite(Hg, sizeof(double), (size_t)n, o);
    fwrite(v, sizeof(double), (size_t)n, o);
    fwrite(&gHg, sizeof(double), 1, o);
    fclose(o);
    printf("status=%d iters=%d n_hmul=%d\n", status, iters, n_hmul);

    CHECK(bh_proj_destroy(P));
    CHECK(bh_hess_destroy(H));
    CHECK(bh_shutdown());
    return 0;
}
