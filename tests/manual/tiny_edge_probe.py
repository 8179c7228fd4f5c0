"""Manual probe: the smallest shapes through the caller-level entry points (n = 1, 2; every variable on a bound; a single row of J),
device against oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import benlsip_ref as R
import benlsip_jl_amd as bh
bh.init(0)
rng = np.random.default_rng(0)
bad = 0
for n, d, mA, allact in ((1, 1, 0, False), (1, 3, 0, True), (2, 1, 0, False), (2, 2, 1, False), (3, 1, 1, False), (2, 5, 0, True), (5, 2, 2, False), (4, 4, 3, False)):
    J = rng.standard_normal((d, n)); A = rng.standard_normal((mA, n)); g = rng.standard_normal(n)
    xl, xu = -np.ones(n), np.ones(n)
    x = np.where(rng.random(n) < 0.5, -1.0, 1.0) if allact else np.clip(0.5 * rng.standard_normal(n), -0.9, 0.9)
    L0 = R.chol_lower(A @ A.T)
    Ho = R.AlHessian(J, np.zeros((0, n)), 1.0)
    H = bh.AlHessian(J, None, 1.0)
    delta = 0.7
    for image, fused in ((1, 1), (1, 0), (0, 0)):
        bh.set_option("cauchy_image", image); bh.set_option("cauchy_fused", fused)
        cons_o = R.make_mixed_constraints(A, L0, l=xl, u=xu)
        try:
            s_ref = R.cauchy_step(x, g, Ho, L0, cons_o, delta, R.NumpyOps())
        except Exception as e:
            s_ref = None
        cons = bh.MixedConstraints(A, None, None, l=xl, u=xu)
        try:
            s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
        except bh.BenlsipHipError as e:
            s = None
        if (s is None) != (s_ref is None):
            bad += 1; print("MISMATCH error behaviour", n, d, mA, allact, image, fused, s, s_ref)
        elif s is not None:
            ok = np.array_equal(np.asarray(cons.fixvars, dtype=bool), cons_o.fixvars) and np.linalg.norm(s - s_ref) <= 1e-9 * max(np.linalg.norm(s_ref), 1e-300)
            if not ok:
                bad += 1; print("MISMATCH", n, d, mA, allact, image, fused, s, s_ref, cons.fixvars, cons_o.fixvars)
        cons.close()
    bh.set_option("cauchy_image", 1); bh.set_option("cauchy_fused", 1)
    # minor iterate from a feasible configuration
    fix = np.zeros(n, dtype=bool)
    if mA < n:
        mo = R.make_mixed_constraints(A, L0, None, l=xl, u=xu)
        mc = bh.MixedConstraints(A, None, fix, l=xl, u=xu)
        x0 = np.zeros(n)
        w_ref, st_ref = R.minor_iterate(x0, np.zeros(n), g, Ho, mo, delta, 0.1)
        w, st, info = bh.minor_iterate(x0, np.zeros(n), g, H, mc, delta, 0.1, full_output=True)
        fin = np.all(np.isfinite(w_ref))
        if int(st) != int(st_ref) or (fin and np.linalg.norm(w - w_ref) > 1e-8 * max(np.linalg.norm(w_ref), 1e-300)) or (not fin and np.all(np.isfinite(w))):
            bad += 1; print("MISMATCH minor", n, d, mA, int(st), int(st_ref), w, w_ref)
        mc.close()
    H.close()
print("tiny shapes: mismatches", bad)
sys.exit(1 if bad else 0)
