#!/usr/bin/env python3
"""BASELINE config 5 (d=65536, n=4096, mA=64 linear equalities + p=512 active bounds -> mpp=576): time the
general projection and a projected_cg run.  The factor is built on the host exactly as cholesky_aug_aat does."""
import os
import sys
import time

import numpy as np
from scipy.linalg import solve_triangular

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402
import bench  # noqa: E402


def aug_factor(A, fix):
    L0 = np.linalg.cholesky(A @ A.T)
    G = solve_triangular(L0, A[:, fix], lower=True)
    p = G.shape[1]
    L = np.zeros((A.shape[0] + p, A.shape[0] + p))
    L[:A.shape[0], :A.shape[0]] = L0
    L[A.shape[0]:, :A.shape[0]] = G.T
    L[A.shape[0]:, A.shape[0]:] = np.linalg.cholesky(np.eye(p) - G.T @ G)
    return L


def main():
    bh.init(0)
    lib = bh._lib.lib()
    d, n, mA = 65536, 4096, 64
    syn = bh.synthetic
    H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
    x, x_l, x_u, fix = syn.box_vectors(n, fix_every=8)
    A = syn.splitmix_uniform(4, np.arange(mA * n)).reshape((mA, n), order="F")
    L = aug_factor(A, fix)
    cons = bh.MixedConstraints(A, L, fix, l=x_l, u=x_u)
    g = H.jtv(syn.residual_rows(0, d))
    w_l, w_u = syn.step_bounds(x, x_l, x_u, fix, syn.initial_tr(g))
    r = np.random.default_rng(0).standard_normal(n)
    rd, vd = bh.DeviceVector(n, r), bh.DeviceVector(n)
    for _ in range(3):
        bh._lib.check(lib.bh_project_dev(cons.handle, rd.ptr, vd.ptr), "proj")
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        lib.bh_project_dev(cons.handle, rd.ptr, vd.ptr)
    lib.bh_synchronize()
    print("projection (mA=64, p=512, mpp=576): %.1f us per call (incl. host sync)" % (1e6 * (time.perf_counter() - t0) / reps))
    v = vd.download()
    print("  |A v| = %.2e, |v[fix]| = %.2e" % (np.linalg.norm(A @ v), np.linalg.norm(v[fix])))
    dv = {k: bh.DeviceVector(n, val) for k, val in (("g", g), ("wl", w_l), ("wu", w_u))}
    dv["w"] = bh.DeviceVector(n)
    for kappa2 in (0.1, 1e-3):
        bench.run_steps(bh, H, cons, dv, kappa2, 2)
        per = []
        for _ in range(10):
            t0 = time.perf_counter()
            st, it, nh = bench.run_steps(bh, H, cons, dv, kappa2, 1)
            per.append(time.perf_counter() - t0)
        print("  per-call ms:", ["%.2f" % (1e3 * x) for x in per])
        el = sorted(per)[len(per) // 2]
        print("pcg config5 kappa2=%g: %s iters=%d n_hmul=%d  %.3f ms per subproblem, %.1f us per iteration" % (kappa2, st.name, it, nh, 1e3 * el, 1e6 * el / max(nh, 1)))


if __name__ == "__main__":
    main()
