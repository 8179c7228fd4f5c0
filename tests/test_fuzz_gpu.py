"""Randomised sweep of small projected_cg instances (box, linear equalities, q > 0, odd n, empty active sets, both
projection forms) against the plain-C oracle — looks for rare shape-dependent bugs rather than for rounding."""
import numpy as np
import pytest

import benlsip_oracle as BO
import benlsip_ref as R
from _util import relnorm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_projected_cg_against_c_oracle(bh, seed):
    rng = np.random.default_rng(1000 + seed)
    lib = bh._lib.lib()
    mism = []
    for case in range(40):
        n = int(rng.integers(2, 90))
        d = int(rng.integers(3 * n, 5 * n + 2))
        q = int(rng.integers(0, 3))
        mA = int(rng.integers(0, min(4, n - 1) + 1)) if rng.random() < 0.6 else 0
        nfix = int(rng.integers(0, max(1, (n - mA) // 2)))
        form = int(rng.integers(0, 2))
        lib.bh_set_option(b"proj_form", form)
        J = rng.standard_normal((d, n)) / np.sqrt(d)
        C = 0.3 * rng.standard_normal((q, n))
        A = rng.standard_normal((mA, n))
        fix = np.zeros(n, dtype=bool)
        if nfix:
            fix[rng.choice(n, nfix, replace=False)] = True
        cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix if nfix else None, l=-np.ones(n), u=np.ones(n))
        x_minor = np.clip(0.3 * rng.standard_normal(n), -0.9, 0.9)
        x_minor[fix] = np.where(rng.random(nfix) < 0.5, -1.0, 1.0)
        g = rng.standard_normal(n)
        delta = float(rng.choice([0.05, 0.5, 5.0])) * np.linalg.norm(g)
        w_l, w_u = R.build_step_bounds(x_minor, cons_o, delta)
        if rng.random() < 0.25:      # finite bounds on free variables too: exercises the bound_hit exit
            w_l = np.where(fix, w_l, -0.05)
            w_u = np.where(fix, w_u, 0.05)
        kappa2 = float(rng.choice([0.1, 1e-2]))      # loose exits: CG has not yet amplified rounding (cf. _util.pcg_sensitivity)
        mu = 1.0
        w0, s0, it0, nh0, _ = BO.projected_cg(g, J, C, mu, w_l, w_u, A, fix, cons_o.chol_L, kappa2)
        H = bh.AlHessian(J, C, mu)
        cons = bh.MixedConstraints(A, cons_o.chol_L, fix)
        w, st, info = bh.projected_cg(g, H, w_l, w_u, cons, kappa2, full_output=True)
        ok = int(st) == s0 and info["iters"] == it0 and info["n_hmul"] == nh0
        if ok and np.all(np.isfinite(w0)):
            ok = relnorm(w, w0) <= 1e-6
        if not ok:
            # rounding or bug?  Ask the second oracle: if the two CPU restatements differ from each other as much as the GPU
            # differs from them (or disagree on the exit), the instance is rounding-sensitive and proves nothing.
            w1, s1, it1 = R.projected_cg(g, R.AlHessian(J, C, mu), w_l, w_u, cons_o, kappa2)
            if (int(s1), it1) != (s0, it0) or (np.all(np.isfinite(w0)) and relnorm(w1, w0) > 1e-2 * max(relnorm(w, w0), 1e-300)):
                continue
            mism.append((case, n, d, q, mA, nfix, form, kappa2, s0, int(st), it0, info["iters"], relnorm(w, w0)))
        H.close()
        cons.close()
    lib.bh_set_option(b"proj_form", 1)
    assert not mism, "\n".join(str(m) for m in mism)
