"""Manual probe: projected_cg with badly SCALED linear equalities (row norms of A from 1e-6 to 1e6; well-conditioned after scaling):
|A w| per row relative to |a_i||w|, and w against the oracle, for the device's iteration shapes.
    python tests/manual/rowscale_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import benlsip_ref as R
import benlsip_jl_amd as bh
from _util import relnorm
bh.init(0)
rng = np.random.default_rng(4)
d, n, mA = 600, 300, 12
J = rng.standard_normal((d, n)) / np.sqrt(d)
g = rng.standard_normal(n)
for span in (0, 3, 6):
    scale = 10.0 ** np.linspace(-span, span, mA)
    A = rng.standard_normal((mA, n)) * scale[:, None]
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), None, l=-np.ones(n), u=np.ones(n))
    wl, wu = -10 * np.ones(n), 10 * np.ones(n)
    Ho = R.AlHessian(J, np.zeros((0, n)), 1.0)
    w_ref, st_ref, it_ref = R.projected_cg(g, Ho, wl, wu, cons_o, 1e-3)
    rowfeas = lambda w: float(np.max(np.abs(A @ w) / (np.linalg.norm(A, axis=1) * np.linalg.norm(w))))
    print("row norms 1e-%d .. 1e+%d, cond(A A') = %.1e: oracle %s %d it, worst row |a_i.w|/|a_i||w| = %.1e" % (span, span, np.linalg.cond(A @ A.T), st_ref.name, it_ref, rowfeas(w_ref)))
    H = bh.AlHessian(J, None, 1.0)
    for fused in (1, 2, 0):
        bh.set_option("cg_fused", fused)
        cons = bh.MixedConstraints(A, None, None, l=-np.ones(n), u=np.ones(n))
        w, st, info = bh.projected_cg(g, H, wl, wu, cons, 1e-3, full_output=True)
        print("    device cg_fused=%d: status %d, %d it, |w - w_ref|/|w_ref| = %.1e, worst row = %.1e" % (fused, int(st), info["iters"], relnorm(w, w_ref), rowfeas(w)))
        cons.close()
    bh.set_option("cg_fused", 1)
    H.close()
