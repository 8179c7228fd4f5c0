"""Manual: traces of the two-kernel box CG iteration against the three-kernel one on small instances (GPU box)."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    sys.path.insert(0, p)
import benlsip_jl_amd as bh
import benlsip_ref as R

bh.init(0)
for (d, n, q, nfix, kappa2, seed) in [(79, 23, 2, 0, 0.1, 1), (88, 21, 0, 0, 0.1, 2), (138, 42, 2, 13, 0.01, 3), (600, 300, 0, 30, 0.01, 4), (2000, 4096, 0, 512, 0.1, 5)]:
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    C = rng.standard_normal((q, n))
    fix = np.zeros(n, dtype=bool); fix[rng.choice(n, nfix, replace=False)] = True
    g = rng.standard_normal(n)
    big = np.full(n, 5.0)
    w_l, w_u = np.where(fix, 0.0, -big), np.where(fix, 0.0, big)
    H = bh.AlHessian(J, C, 3.0)
    cons = bh.MixedConstraints(np.zeros((0, n)), None, fix)
    out = {}
    for fused in (0, 1):
        bh.set_option("cg_fused", fused)
        w, st, info = bh.projected_cg(g, H, w_l, w_u, cons, kappa2, trace_cap=64, full_output=True)
        out[fused] = (w, st, info)
    Ho = R.AlHessian(J, C, 3.0)
    A = np.zeros((0, n))
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix)
    tr = R.CGTrace()
    w_ref, st_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, kappa2, trace=tr)
    print("d=%d n=%d q=%d nfix=%d: status %s/%s/%s iters %d/%d/%d  |w1-w0|/|w0| %.2e  |w0-ref| %.2e |w1-ref| %.2e" % (
        d, n, q, nfix, out[0][1].name, out[1][1].name, st_ref.name, out[0][2]["iters"], out[1][2]["iters"], it_ref,
        np.linalg.norm(out[1][0] - out[0][0]) / np.linalg.norm(out[0][0]), np.linalg.norm(out[0][0] - w_ref) / np.linalg.norm(w_ref),
        np.linalg.norm(out[1][0] - w_ref) / np.linalg.norm(w_ref)))
    t0, t1 = out[0][2]["trace"], out[1][2]["trace"]
    for k in range(min(len(t0), len(t1), 6)):
        print("   it %d  unfused %s\n          fused  %s\n          oracle %s" % (k + 1, t0[k], t1[k], np.array(tr.rows[k]) if k < len(tr.rows) else None))
