"""Manual check (not collected): n above the register-resident limit (column panels) with linear equalities, every CG iteration shape, against the oracle."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np
import benlsip_jl_amd as bh, benlsip_ref as R
from _util import relnorm, w_tolerance
bh.init(0)
rng = np.random.default_rng(2)
for (d, n, mA, nfix) in [(40, 16400, 3, 50), (25, 20011, 5, 0), (60, 17000, 0, 900), (30, 16385, 64, 10)]:
    J = rng.standard_normal((d, n)) / np.sqrt(d); A = rng.standard_normal((mA, n))
    fix = np.zeros(n, dtype=bool)
    if nfix: fix[rng.choice(n, nfix, replace=False)] = True
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix if nfix else None, l=-np.ones(n), u=np.ones(n))
    Ho = R.AlHessian(J, np.zeros((0, n)), 2.0)
    g = J.T @ rng.standard_normal(d) + 1e-3 * rng.standard_normal(n)
    w_l, w_u = R.build_step_bounds(np.where(fix, 1.0, 0.0), cons_o, 0.5 * np.linalg.norm(g))
    w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.1)
    H = bh.AlHessian(J, None, 2.0); cons = bh.MixedConstraints(A, cons_o.chol_L if mA else None, fix)
    for fused in (1, 0, 2):
        bh.set_option("cg_fused", fused)
        w, st, info = bh.projected_cg(g, H, w_l, w_u, cons, 0.1, full_output=True)
        tol = max(1e-8, w_tolerance(g, Ho, w_l, w_u, cons_o, 0.1, w_ref))
        print((d, n, mA, nfix), "fused", fused, "ok" if (int(st) == int(s_ref) and info["iters"] == it_ref and relnorm(w, w_ref) <= tol) else "MISMATCH", int(st), info["iters"], it_ref, relnorm(w, w_ref), flush=True)
    pv = bh.projection(cons, g)
    print("   projection rel", relnorm(pv, R.projection(cons_o, g)))
    H.close()
bh.set_option("cg_fused", 1)
