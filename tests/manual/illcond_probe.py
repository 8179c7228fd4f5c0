"""Manual probe: projected_cg with ill-conditioned linear equalities (rows of A nearly dependent) — how far do the device's three
iteration shapes (explicit inverse of the factor / triangular solves / separate kernels) land from the oracle, and how far does the
oracle itself move when its projection solves the normal equations another way (LU instead of Cholesky)?
    python tests/manual/illcond_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import benlsip_ref as R
import benlsip_jl_amd as bh
from _util import relnorm
bh.init(0)
rng = np.random.default_rng(3)
d, n, mA = 600, 300, 12
J = rng.standard_normal((d, n)) / np.sqrt(d)
g = rng.standard_normal(n)
for eps_dep in (1e-2, 1e-4, 1e-6, 1e-7):
    A = rng.standard_normal((mA, n))
    A[1] = A[0] + eps_dep * rng.standard_normal(n)            # two nearly parallel rows
    A[5] = A[4] - A[3] + eps_dep * rng.standard_normal(n)     # a nearly dependent triple
    M = A @ A.T
    L0 = R.chol_lower(M)
    cons_o = R.make_mixed_constraints(A, L0, None, l=-np.ones(n), u=np.ones(n))
    wl, wu = -10 * np.ones(n), 10 * np.ones(n)
    Ho = R.AlHessian(J, np.zeros((0, n)), 1.0)
    w_ref, st_ref, it_ref = R.projected_cg(g, Ho, wl, wu, cons_o, 1e-3)

    def proj_lu(cons, r):                                     # the same projector, normal equations solved by LU
        y = np.linalg.solve(M, A @ r)
        return r - A.T @ y
    w_lu, st_lu, it_lu = R.projected_cg(g, Ho, wl, wu, cons_o, 1e-3, proj_fn=proj_lu)
    H = bh.AlHessian(J, None, 1.0)
    out = []
    for fused in (1, 2, 0):
        bh.set_option("cg_fused", fused)
        cons = bh.MixedConstraints(A, None, None, l=-np.ones(n), u=np.ones(n))
        w, st, info = bh.projected_cg(g, H, wl, wu, cons, 1e-3, full_output=True)
        out.append((fused, int(st), info["iters"], relnorm(w, w_ref), float(np.linalg.norm(A @ w) / (np.linalg.norm(A) * np.linalg.norm(w)))))
        cons.close()
    bh.set_option("cg_fused", 1)
    print("dependence %.0e: cond(A A') = %.1e; oracle %s %d it; oracle(LU projector) %s %d it, |w - w_ref|/|w_ref| = %.1e, |A w_ref|/|A||w| = %.1e"
          % (eps_dep, np.linalg.cond(M), st_ref.name, it_ref, st_lu.name, it_lu, relnorm(w_lu, w_ref),
             np.linalg.norm(A @ w_ref) / (np.linalg.norm(A) * np.linalg.norm(w_ref))))
    for o in out:
        print("    device cg_fused=%d: status %d, %d it, |w - w_ref|/|w_ref| = %.1e, |A w|/|A||w| = %.1e" % o)
    H.close()
