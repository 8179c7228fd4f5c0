"""Degenerate shapes (also runnable by hand; collected through tests/test_parity_gpu.py::test_degenerate_shapes_match_the_oracle): — no residual rows (d = 0), no rows at all, n = 1 and 2, every
variable fixed, mA = n - nfix (no degrees of freedom), mu = 0 — against the oracle on every CG iteration shape."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import benlsip_ref as R  # noqa: E402
from _util import relnorm  # noqa: E402

CASES = [  # d, n, q, mA, nfix, mu
    (0, 7, 2, 0, 0, 3.0), (0, 7, 0, 0, 0, 1.0), (5, 1, 0, 0, 0, 1.0), (5, 2, 1, 0, 1, 1.0), (9, 6, 0, 0, 6, 1.0),
    (9, 6, 0, 2, 4, 1.0), (9, 6, 1, 3, 0, 0.0), (30, 17, 0, 1, 16, 2.0), (1, 33, 0, 0, 0, 1.0), (64, 16, 3, 5, 2, 0.5),
]


def run(bh, fused):
    """Mismatches (as strings) of every case on the CG iteration shape `fused`."""
    rng = np.random.default_rng(3)
    bad = []
    for (d, n, q, mA, nfix, mu) in CASES:
        J = rng.standard_normal((d, n)); C = rng.standard_normal((q, n)); A = rng.standard_normal((mA, n))
        fix = np.zeros(n, dtype=bool)
        fix[rng.choice(n, nfix, replace=False)] = True
        L0 = R.chol_lower(A @ A.T)
        cons_o = R.make_mixed_constraints(A, L0, fix if nfix else None, l=-np.ones(n), u=np.ones(n))
        Ho = R.AlHessian(J, C, mu)
        g = rng.standard_normal(n)
        w_l, w_u = R.build_step_bounds(np.where(fix, 1.0, 0.0), cons_o, 0.7 * np.linalg.norm(g))
        with np.errstate(all="ignore"):
            w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.1)
            hv_ref = R.hmul(Ho, g)
            pv_ref = R.projection(cons_o, g)
        H = bh.AlHessian(J, C, mu)
        cons = bh.MixedConstraints(A, cons_o.chol_L if mA else None, fix)
        w, st, info = bh.projected_cg(g, H, w_l, w_u, cons, 0.1, full_output=True)
        ok = int(st) == int(s_ref) and info["iters"] == it_ref
        fin = np.isfinite(w_ref)
        ok = ok and np.array_equal(np.isfinite(w), fin) and (not fin.any() or relnorm(w[fin], w_ref[fin]) <= 1e-8 or np.linalg.norm(w_ref[fin]) == 0)
        ok = ok and np.linalg.norm(H * g - hv_ref) <= 1e-12 * max(np.linalg.norm(hv_ref), 1e-300) + 1e-300
        ok = ok and np.linalg.norm(bh.projection(cons, g) - pv_ref) <= 1e-10 * max(np.linalg.norm(g), 1e-300)
        ok = ok and abs(bh.vthv(H, g) - R.vthv(Ho, g)) <= 1e-12 * max(abs(R.vthv(Ho, g)), 1e-300) + 1e-300
        if not ok:
            bad.append("fused=%d case=%s: status %s/%s iters %s/%s" % (fused, (d, n, q, mA, nfix, mu), st, s_ref, info["iters"], it_ref))
        H.close(); cons.close()
    return bad


if __name__ == "__main__":
    import benlsip_jl_amd as bh
    bh.init(0)
    total = []
    for fused in (1, 0, 2):
        bh.set_option("cg_fused", fused)
        total += run(bh, fused)
    bh.set_option("cg_fused", 1)
    print("\n".join(total))
    print("edge shapes: %d cases x 3 iteration shapes, %d mismatches" % (len(CASES), len(total)))
