"""Manual probe: the random instance that uses most of its tolerance in test_pcg_random_instances (d=200 n=64 q=1 mA=3, seed 2): the
device's deviation from the oracle per iteration shape, next to the oracle's own movement under re-associated H*p."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import benlsip_ref as R
import benlsip_jl_amd as bh
from _util import relnorm, w_tolerance
bh.init(0)
d, n, q, mA, nfix, seed = 200, 64, 1, 3, 10, 2
rng = np.random.default_rng(seed)
J = rng.standard_normal((d, n)) / np.sqrt(d)
C = rng.standard_normal((q, n))
A = rng.standard_normal((mA, n))
L0 = R.chol_lower(A @ A.T)
fix = np.zeros(n, dtype=bool)
fix[rng.choice(n, nfix, replace=False)] = True
cons_o = R.make_mixed_constraints(A, L0, fix, l=-np.ones(n), u=np.ones(n))
x_minor = np.clip(0.3 * rng.standard_normal(n), -0.9, 0.9)
x_minor[fix] = 1.0
g = rng.standard_normal(n)
w_l, w_u = R.build_step_bounds(x_minor, cons_o, 0.1 * np.linalg.norm(g))
Ho = R.AlHessian(J, C, 10.0)
tr = R.CGTrace()
w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.1, trace=tr)
print("oracle:", s_ref.name, it_ref, "iterations; tolerance", w_tolerance(g, Ho, w_l, w_u, cons_o, 0.1, w_ref))
for row in tr.rows:
    print("   pHp %.6e alpha %.6e gamma %.6e rtv %.6e" % row)
Hm = J.T @ J + 10.0 * C.T @ C
free = ~fix
print("cond(H_free) = %.2e" % np.linalg.cond(Hm[np.ix_(free, free)]))
# oracle variants: H*p accumulated differently
for name, hm in (("H*p as (J'J + mu C'C) p", lambda H, v: Hm @ v), ("rows reversed", lambda H, v: J[::-1].T @ (J[::-1] @ v) + 10.0 * C.T @ (C @ v))):
    w2, s2, it2 = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.1, hmul_fn=hm)
    print("oracle with %s: %s %d, |w - w_ref|/|w_ref| = %.2e" % (name, s2.name, it2, relnorm(w2, w_ref)))
H = bh.AlHessian(J, C, 10.0)
for fused in (1, 2, 0):
    for form in (1, 0):
        bh.set_option("cg_fused", fused); bh.set_option("proj_form", form)
        cons = bh.MixedConstraints(A, cons_o.chol_L, fix)
        w, st, info = bh.projected_cg(g, H, w_l, w_u, cons, 0.1, trace_cap=16, full_output=True)
        print("device cg_fused=%d proj_form=%d: status %d, %d it, |w - w_ref|/|w_ref| = %.2e; last trace row rel. dev %.1e" % (
            fused, form, int(st), info["iters"], relnorm(w, w_ref), np.max(np.abs(info["trace"][:len(tr.rows)] - np.array(tr.rows)) / np.abs(np.array(tr.rows)))))
        cons.close()
bh.set_option("cg_fused", 1); bh.set_option("proj_form", 1)
