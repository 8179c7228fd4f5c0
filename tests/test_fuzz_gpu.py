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
    lib.bh_set_option(b"cg_fused", {5: 0, 4: 2, 3: 2}.get(seed, 1))       # one seed on the round-1 iteration, two with linear equalities fused too
    mism = []
    for case in range(40):
        n = int(rng.integers(2, 90)) if rng.random() < 0.9 else int(rng.integers(90, 700))
        d = int(rng.integers(3 * n, 5 * n + 2))
        if rng.random() < 0.15:
            d = int(rng.integers(20000, 60000))     # tall: the row-streaming grid saturates (more workgroups than vector entries)
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
    lib.bh_set_option(b"cg_fused", 1)
    assert not mism, "\n".join(str(m) for m in mism)


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_cauchy_step_and_minor_iterate(bh, seed):
    """Randomised small instances of the two device-resident callers against the NumPy oracle:
    cauchy_step (same breakpoint count, same final active set, same step) and minor_iterate (same CG exit, same step)."""
    rng = np.random.default_rng(7000 + seed)
    lib = bh._lib.lib()
    mism = []
    for case in range(25):
        n = int(rng.integers(3, 70))
        d = int(rng.integers(2 * n, 4 * n + 2))
        mA = int(rng.integers(0, min(3, n - 2) + 1)) if rng.random() < 0.5 else 0
        lib.bh_set_option(b"chol_downdate", int(rng.integers(0, 2)))
        J = rng.standard_normal((d, n)) / np.sqrt(d)
        A = rng.standard_normal((mA, n))
        L0 = R.chol_lower(A @ A.T)
        xlow, xupp = -np.ones(n), np.ones(n)
        nact = int(rng.integers(0, max(1, (n - mA) // 3)))
        x = np.clip(0.5 * rng.standard_normal(n), -0.95, 0.95)
        act = rng.choice(n, nact, replace=False)
        x[act] = np.where(rng.random(nact) < 0.5, -1.0, 1.0)
        g = rng.standard_normal(n)
        delta = float(rng.choice([0.1, 0.5, 3.0])) * 0.1 * np.linalg.norm(g)
        mu = 2.0
        Ho = R.AlHessian(J, np.zeros((0, n)), mu)
        H = bh.AlHessian(J, None, mu)

        # ---- cauchy_step
        cons_o = R.make_mixed_constraints(A, L0, l=xlow, u=xupp)
        count = [0]

        class Ops(R.NumpyOps):
            def hmul(self, Hh, v):
                count[0] += 1
                return R.hmul(Hh, v)
        try:
            s_ref = R.cauchy_step(x, g, Ho, L0, cons_o, delta, Ops())
        except Exception:
            s_ref = None              # the reference itself fails here (no breakpoint left / factor lost definiteness)
        cons = bh.MixedConstraints(A, None, l=xlow, u=xupp)
        if s_ref is None:
            with pytest.raises(bh.BenlsipHipError):
                bh.cauchy_step(x, g, H, cons, delta)
        else:
            s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
            ok = np.array_equal(cons.fixvars, cons_o.fixvars) and info["n_hmul"] == count[0] and \
                np.linalg.norm(s - s_ref) <= 1e-8 * max(np.linalg.norm(s_ref), 1e-300)
            if not ok:
                # a branch scalar within rounding of its threshold?  The oracle on a 1e-14-perturbed J tells.
                cons_p = R.make_mixed_constraints(A, L0, l=xlow, u=xupp)
                s_p = R.cauchy_step(x, g, R.AlHessian(J * (1 + 1e-14 * rng.standard_normal(J.shape)), np.zeros((0, n)), mu), L0, cons_p,
                                    delta, R.NumpyOps())
                if np.array_equal(cons_p.fixvars, cons_o.fixvars) and np.linalg.norm(s_p - s_ref) <= 1e-9 * max(np.linalg.norm(s_ref), 1e-300):
                    mism.append(("cauchy", case, n, d, mA, nact, delta, info, count[0], int(np.sum(cons.fixvars != cons_o.fixvars)),
                                 float(np.linalg.norm(s - s_ref))))

        # ---- minor_iterate from the same point (fresh constraint objects: active set = the bounds x sits on)
        fix = np.zeros(n, dtype=bool)
        fix[act] = True
        if mA + nact < n:
            mo = R.make_mixed_constraints(A, L0, fix if nact else None, l=xlow, u=xupp)
            mc = bh.MixedConstraints(A, None, fix, l=xlow, u=xupp)
            s0 = np.zeros(n)
            kappa2 = float(rng.choice([0.1, 1e-2]))
            w_ref, st_ref = R.minor_iterate(x, s0, g, Ho, mo, delta, kappa2)
            w, st, info = bh.minor_iterate(x, s0, g, H, mc, delta, kappa2, full_output=True)
            ok = int(st) == int(st_ref) and (not np.all(np.isfinite(w_ref)) or relnorm(w, w_ref) <= 1e-6)
            if not ok:
                Hp = R.AlHessian(J * (1 + 1e-14 * rng.standard_normal(J.shape)), np.zeros((0, n)), mu)
                w_p, st_p = R.minor_iterate(x, s0, g, Hp, mo, delta, kappa2)
                stable = int(st_p) == int(st_ref) and (not np.all(np.isfinite(w_ref)) or relnorm(w_p, w_ref) <= 1e-8)
                if stable:
                    mism.append(("minor", case, n, d, mA, nact, kappa2, int(st), int(st_ref), relnorm(w, w_ref)))
            mc.close()
        cons.close()
        H.close()
    lib.bh_set_option(b"chol_downdate", 0)
    assert not mism, "\n".join(str(m) for m in mism)


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_medium_shapes(bh, seed):
    """Medium shapes that cross the kernel-geometry classes (n from 100 to 5000: every row-stream configuration below the
    wide ones), factor orders on both sides of 64 (one-wave vs blocked Cholesky / triangular solves, 4 vs 16 row groups in
    the transposed multiply) and both projection forms: projection and projected_cg against the oracles."""
    rng = np.random.default_rng(9000 + seed)
    lib = bh._lib.lib()
    mism = []
    try:        # the GPU box's 128 BLAS threads make these medium-sized oracle calls an order of magnitude slower than 8 do
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=8)
    except Exception:
        limiter = None
    omp_prev = BO.set_num_threads(8)
    for case in range(10):
        n = int(rng.choice([100, 130, 257, 512, 700, 1025, 1500, 2049, 3000, 4100, 5000]))
        d = int(rng.integers(n // 2, 2 * n))
        mA = int(rng.choice([0, 1, 7, 63, 64, 65, 100, 129, 200])) if n >= 512 else int(rng.choice([0, 3, 40]))
        nfix = int(rng.integers(0, (n - mA) // 2))
        form = int(rng.integers(0, 2)) if mA + nfix <= 1500 else 1         # keep the oracle's augmented factor affordable
        lib.bh_set_option(b"proj_form", form)
        J = rng.standard_normal((d, n)) / np.sqrt(d)
        A = rng.standard_normal((mA, n))
        fix = np.zeros(n, dtype=bool)
        if nfix:
            fix[rng.choice(n, nfix, replace=False)] = True
        cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix if nfix else None, l=-np.ones(n), u=np.ones(n))
        H, Ho = bh.AlHessian(J, None, 1.0), R.AlHessian(J, np.zeros((0, n)), 1.0)
        cons = bh.MixedConstraints(A, cons_o.chol_L, fix)
        r = rng.standard_normal(n)
        v_ref = R.projection(cons_o, r)
        v = bh.projection(cons, r)
        if np.linalg.norm(v - v_ref) > 1e-9 * np.linalg.norm(r):
            mism.append(("projection", case, n, d, mA, nfix, form, np.linalg.norm(v - v_ref) / np.linalg.norm(r)))
        x_minor = np.clip(0.3 * rng.standard_normal(n), -0.9, 0.9)
        x_minor[fix] = 1.0
        g = rng.standard_normal(n)
        w_l, w_u = R.build_step_bounds(x_minor, cons_o, 0.5 * np.linalg.norm(g))
        w0, s0, it0 = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.1)
        w, st, info = bh.projected_cg(g, H, w_l, w_u, cons, 0.1, full_output=True)
        ok = int(st) == int(s0) and info["iters"] == it0 and (not np.all(np.isfinite(w0)) or relnorm(w, w0) <= 1e-6)
        if not ok:
            # rounding or bug?  the C restatement decides (reduced instances only need its chol of the augmented matrix)
            w1, s1, it1, _, _ = BO.projected_cg(g, J, np.zeros((0, n)), 1.0, w_l, w_u, A, fix, cons_o.chol_L, 0.1)
            if (int(s1), it1) == (int(s0), it0) and (not np.all(np.isfinite(w0)) or relnorm(w1, w0) <= 1e-2 * max(relnorm(w, w0), 1e-300)):
                mism.append(("pcg", case, n, d, mA, nfix, form, int(s0), int(st), it0, info["iters"], relnorm(w, w0)))
        H.close()
        cons.close()
    lib.bh_set_option(b"proj_form", 1)
    if limiter is not None:
        limiter.restore_original_limits()
    BO.set_num_threads(omp_prev)
    assert not mism, "\n".join(str(m) for m in mism)
