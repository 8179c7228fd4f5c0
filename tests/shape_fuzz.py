#!/usr/bin/env python3
"""One-off hunt for shape-dependent bugs: projected_cg (box, and a third of the cases with linear equalities) on random (d, n) from wide ranges — tall, wide, odd n,
n on both sides of every kernel-geometry boundary — against the NumPy oracle, on both iteration shapes.  Not a parity test
(the tolerance is the oracle's own sensitivity): prints every disagreement it cannot explain.
    python tests/shape_fuzz.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))        # (uses the oracle as the checker, so it lives under tests/, not tools/)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import benlsip_jl_amd as bh  # noqa: E402
import benlsip_ref as R  # noqa: E402
from _util import relnorm, w_tolerance  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    bh.init(0)
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    edges = [128, 512, 1024, 2048, 4096, 8192, 16384]
    t0, cases, bad = time.time(), 0, []
    while time.time() - t0 < budget:
        if rng.random() < 0.5:
            e = int(rng.choice(edges))
            n = max(2, e + int(rng.integers(-3, 4)))
        else:
            n = int(np.exp(rng.uniform(np.log(2), np.log(20000))))
        max_d = max(4, min(120000, int(6e7 // n)))
        d = int(np.exp(rng.uniform(np.log(3), np.log(max_d))))
        kappa2 = float(rng.choice([0.3, 0.1, 1e-2]))
        fused = int(rng.integers(0, 3))
        bh.set_option("cg_fused", fused)
        # a third of the cases with linear equalities (reduced projection form; mA on both sides of the 64 / 96 boundaries)
        mA = 0
        if rng.random() < 0.33 and n >= 8:
            mA = int(min(n // 3, rng.choice([1, 3, 17, 63, 64, 65, 97, 130])))
        nfix = int(rng.integers(0, max(1, (n - mA) // 3)))
        J = rng.standard_normal((d, n)) / np.sqrt(d)
        fix = np.zeros(n, dtype=bool)
        if nfix:
            fix[rng.choice(n, nfix, replace=False)] = True
        A = rng.standard_normal((mA, n))
        cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix if nfix else None, l=-np.ones(n), u=np.ones(n))
        g = J.T @ rng.standard_normal(d) + 1e-3 * rng.standard_normal(n)
        w_l, w_u = R.build_step_bounds(np.where(fix, 1.0, 0.0), cons_o, float(rng.choice([0.05, 0.5, 5.0])) * np.linalg.norm(g))
        Ho = R.AlHessian(J, np.zeros((0, n)), 2.0)
        w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, kappa2)
        H = bh.AlHessian(J, None, 2.0)
        cons = bh.MixedConstraints(A, cons_o.chol_L if mA else None, fix)
        for rep in range(2):
            w, st, info = bh.projected_cg(g, H, w_l, w_u, cons, kappa2, full_output=True)
            ok = int(st) == int(s_ref) and info["iters"] == it_ref
            if ok:
                ok = relnorm(w, w_ref) <= max(1e-6, w_tolerance(g, Ho, w_l, w_u, cons_o, kappa2, w_ref))
            if not ok:
                tol = w_tolerance(g, Ho, w_l, w_u, cons_o, kappa2, w_ref)
                bad.append((d, n, mA, nfix, kappa2, fused, rep, int(s_ref), int(st), it_ref, info["iters"], relnorm(w, w_ref), tol))
                print("MISMATCH", bad[-1], flush=True)
                break
        H.close()
        cons.close()
        cases += 1
        if cases % 25 == 0:
            print("%d cases, %d mismatches, %.0f s" % (cases, len(bad), time.time() - t0), flush=True)
    bh.set_option("cg_fused", 1)
    print("done: %d cases, %d mismatches" % (cases, len(bad)))


if __name__ == "__main__":
    main()
