"""Manual probe: cauchy_step with ill-conditioned linear equalities (rows of A nearly dependent): |A s|/|A||s| and the active set of the
device's two forms against the oracle.   python tests/manual/illcond_cauchy_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import benlsip_ref as R
import benlsip_jl_amd as bh
from _util import relnorm, ReducedFormOps
bh.init(0)
rng = np.random.default_rng(5)
d, n, mA = 400, 200, 10
J = rng.standard_normal((d, n)) / np.sqrt(d)
g = rng.standard_normal(n)
xl, xu = -np.ones(n), np.ones(n)
x = np.clip(0.5 * rng.standard_normal(n), -0.9, 0.9)
feas = lambda A, s: float(np.linalg.norm(A @ s) / (np.linalg.norm(A) * max(np.linalg.norm(s), 1e-300)))
for dep in (1e-2, 1e-4, 1e-6):
    A = rng.standard_normal((mA, n))
    A[1] = A[0] + dep * rng.standard_normal(n)
    A[5] = A[4] - A[3] + dep * rng.standard_normal(n)
    L0 = R.chol_lower(A @ A.T)
    Ho = R.AlHessian(J, np.zeros((0, n)), 1.0)
    delta = 0.3 * np.linalg.norm(g)
    res = {}
    for name, ops in (("oracle", R.NumpyOps()), ("reduced-form CPU", ReducedFormOps())):
        cons = R.make_mixed_constraints(A, L0, l=xl, u=xu)
        s = R.cauchy_step(x, g, Ho, L0, cons, delta, ops)
        res[name] = (s, cons.fixvars.copy())
    H = bh.AlHessian(J, None, 1.0)
    for image in (1, 0):
        bh.set_option("cauchy_image", image)
        cons = bh.MixedConstraints(A, None, None, l=xl, u=xu)
        s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
        res["device image=%d" % image] = (s, np.asarray(cons.fixvars, dtype=bool).copy())
        cons.close()
    bh.set_option("cauchy_image", 1)
    H.close()
    s0, f0 = res["oracle"]
    print("dependence %.0e, cond(A A') = %.1e:" % (dep, np.linalg.cond(A @ A.T)))
    for name, (s, f) in res.items():
        print("    %-18s %3d active, |A s|/|A||s| = %.1e, |s - s_oracle|/|s_oracle| = %.1e, sets differ in %d" % (name, int(f.sum()), feas(A, s), relnorm(s, s0), int((f != f0).sum())))
