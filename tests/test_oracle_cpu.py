"""CPU tests: the NumPy oracle against every fixture the reference's own tests hold for the hot path
 and against the committed golden files."""
import json
import os

import numpy as np
import pytest

import benlsip_ref as R
import sphere_problem as sp

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _hs48():
    h = json.load(open(os.path.join(GOLD, "hs48_projection.json")))
    A = np.array(h["A"], dtype=np.float64)
    x = np.array(h["x"], dtype=np.float64)
    fix = np.zeros(A.shape[1], dtype=bool)
    fix[np.array(h["fixed_1based"]) - 1] = True
    return A, x, fix, np.array(h["projection"], dtype=np.float64)


def test_hs48_projection_known_answer():
    """test/structures.jl:37-58 — the one numeric golden vector of the reference."""
    A, x, fix, expected = _hs48()
    L0 = R.chol_lower(A @ A.T)
    lincons = R.make_mixed_constraints(A, L0, fix)
    B = np.vstack([A, np.eye(5)[fix]])
    y = np.random.default_rng(0).random(4)
    np.testing.assert_allclose(R.left_mul_tr(lincons, y), B.T @ y, rtol=1.5e-8)      # :51
    np.testing.assert_allclose(R.left_mul(lincons, x), B @ x, rtol=1.5e-8)           # :52
    proj = R.projection(lincons, x)
    Ap = A @ proj
    assert np.all(proj[fix] <= np.finfo(float).eps) and float(Ap @ Ap) <= np.finfo(float).eps   # :55-56
    assert np.max(np.abs(proj - expected)) <= 1e-14                                  # :57, SURVEY §8c tolerance


def test_gauss_newton_hessian_structure():
    """test/structures.jl:1-16 (random 5x5, property form)."""
    rng = np.random.default_rng(1)
    for _ in range(5):
        n = 5
        J, C, mu, v = rng.random((n, n)), rng.random((n, n)), rng.random(), rng.random(n)
        H = R.AlHessian(J, C, mu)
        H_test = J.T @ J + mu * C.T @ C
        np.testing.assert_allclose(R.hmul(H, v), H_test @ v, rtol=1.5e-8)
        assert R.vthv(H, v) == pytest.approx(float(v @ (H_test @ v)), rel=1.5e-8)


def test_mixed_constraints_structure():
    """test/structures.jl:18-35."""
    rng = np.random.default_rng(2)
    m, n = 3, 6
    A = rng.random((m, n))
    L0 = R.chol_lower(A @ A.T)
    cons = R.make_mixed_constraints(A, L0, l=-rng.random(n), u=rng.random(n) + 1)
    act = np.array([2, 4, 6]) - 1
    cons.fixvars[act] = True
    R.update_chol(cons, L0)
    B = np.vstack([A, np.eye(n)[act]])
    assert list(np.flatnonzero(cons.fixvars)) == list(act)
    np.testing.assert_allclose(cons.chol_L, np.linalg.cholesky(B @ B.T), rtol=1.5e-8, atol=1e-12)


def test_active_bounds_identification():
    """test/structures.jl:60-78."""
    rng = np.random.default_rng(3)
    m, n = 3, 7
    A = rng.random((m, n))
    L0 = R.chol_lower(A @ A.T)
    cons = R.make_mixed_constraints(A, L0, l=-10 * np.ones(n), u=10 * np.ones(n))
    x = rng.random(n)
    x[1] = -10.0
    R.active_bounds_inplace(cons, x, L0)
    assert cons.fixvars[1] and not cons.fixvars[np.setdiff1d(np.arange(n), [1])].any()
    R.add_active(cons, L0, np.array([3, 5]) - 1)
    assert cons.fixvars[[2, 4]].all()
    R.add_active(cons, L0, 7 - 1)
    active = np.array([2, 3, 5, 7]) - 1
    assert cons.fixvars[active].all() and not cons.fixvars[np.setdiff1d(np.arange(n), active)].any()
    assert cons.chol_L.shape == (m + 4, m + 4)


def test_sphere_regression_acceptance():
    """test/problems/sphere_regression.jl:63-65 — the three acceptance inequalities (config 1)."""
    xs, ys = R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u,
                          max_outer_iter=100, max_inner_iter=250)
    grad = sp.jac_r(xs).T @ sp.r(xs) + sp.jac_c(xs).T @ ys
    P = R.projection_polyhedron_small(xs - grad, sp.A, sp.b, sp.x_l, sp.x_u)
    assert np.linalg.norm(sp.c(xs)) < R.SQRT_EPS
    assert R.is_feasible(xs, sp.A, sp.x_l, sp.x_u, sp.b)
    assert np.linalg.norm(xs - P) < 1e-7
    gold = json.load(open(os.path.join(GOLD, "sphere_regression.json")))
    np.testing.assert_allclose(xs, gold["oracle_x"], rtol=1e-9)


def test_box_projection_degenerates_to_mask():
    """with A = 0 x n the augmented factor is I_p and projection = masking."""
    n = 11
    A = np.zeros((0, n))
    L0 = R.chol_lower(A @ A.T)
    fix = np.zeros(n, dtype=bool)
    fix[[0, 3, 10]] = True
    cons = R.make_mixed_constraints(A, L0, fix)
    np.testing.assert_array_equal(cons.chol_L, np.eye(3))
    r = np.arange(1.0, n + 1)
    np.testing.assert_array_equal(R.projection(cons, r), np.where(fix, 0.0, r))
    cons0 = R.make_mixed_constraints(A, L0)
    np.testing.assert_array_equal(R.projection(cons0, r), r)


def test_factor_to_boundary_inf_operands():
    """+-Inf bounds on free variables; |p| < 1e-10 entries are skipped."""
    p = np.array([-1.0, 2.0, 1e-11, -1e-11, 0.5])
    w = np.array([0.1, 0.2, 0.0, 0.0, 0.0])
    wl = np.array([-np.inf, -np.inf, -1.0, -1.0, -1.0])
    wu = np.array([np.inf, np.inf, 1.0, 1.0, 0.25])
    assert R.factor_to_boundary(p, w, wl, wu) == 0.5
    assert R.factor_to_boundary(p[:4], w[:4], wl[:4], wu[:4]) == np.inf
    assert R.factor_to_boundary(np.zeros(3), np.zeros(3), -np.ones(3), np.ones(3)) == np.inf


def test_pcg_golden_cases_reproduce():
    """The committed projected_cg fixtures are what the oracle produces today (guards oracle drift) and cover every
    reachable exit: solved, bound_hit, negative_curvature, `nothing` (max_iter = 0 and iterations exhausted)."""
    cases = json.load(open(os.path.join(GOLD, "pcg_cases.json")))["cases"]
    seen = set()
    for c in cases:
        d, n, q, mA, mpp = c["d"], c["n"], c["q"], c["mA"], c["mpp"]
        J = np.array(c["J"]).reshape((d, n), order="F")
        C = np.array(c["C"]).reshape((q, n), order="F")
        A = np.array(c["A"]).reshape((mA, n), order="F")
        L = np.array(c["L"]).reshape((mpp, mpp), order="F")
        fix = np.array(c["fixvars"], dtype=bool)
        cons = R.MixedConstraints(A, -np.ones(n), np.ones(n), fix, L)
        g = np.array([float(x) for x in c["g"]])
        wl = np.array([float(x) for x in c["w_l"]])
        wu = np.array([float(x) for x in c["w_u"]])
        w, status, iters = R.projected_cg(g, R.AlHessian(J, C, c["mu"]), wl, wu, cons, c["kappa2"])
        assert int(status) == c["status"] and iters == c["iters"], c["name"]
        w_gold = np.array([float(x) for x in c["w"]])
        # CG amplifies the 1e-16 summation-order differences of J@v (C- vs F-contiguous dgemv) by ~cond(H): up to
        # 2e-10 on these cases (measured, also vs a long-double H*p); 1e-8 normwise is the documented oracle tolerance.
        assert np.linalg.norm(w - w_gold) <= 1e-8 * max(np.linalg.norm(w_gold), 1e-300), c["name"]
        seen.add(int(status))
    assert seen == {0, 1, 2, 4}       # status 3 is unreachable in the reference


def test_status_none_when_iterations_exhausted():
    c = [x for x in json.load(open(os.path.join(GOLD, "pcg_cases.json")))["cases"] if x["name"] == "maxiter_exhaust"][0]
    max_iter = 2 * (c["n"] - c["mA"] - c["nfix"])
    assert c["status"] == int(R.CGStatus.none) and c["iters"] == max_iter + 1 and c["n_hmul"] == max_iter


def test_synthetic_generator_is_deterministic_and_sharded():
    J = R.synthetic_J(16, 8, seed=1)
    J2 = np.vstack([R.synthetic_J(8, 8, seed=1, row0=0, d_total=16), R.synthetic_J(8, 8, seed=1, row0=8, d_total=16)])
    np.testing.assert_array_equal(J, J2)
    assert np.all(np.abs(J) <= 1 / 4.0) and abs(J.mean()) < 0.1
    u = R.splitmix_uniform(7, np.arange(4))
    assert np.all((u >= -1) & (u < 1)) and len(set(u.tolist())) == 4


def test_whole_solve_divergence_is_pinned_to_a_rounding_dominated_rho():
    """The instruments the GPU tests use to say WHERE two whole solves part (driver decision log, first_decision_difference,
    assert_rounding_dominated), exercised on the oracle against itself: the same solve with H*v summed in two row blocks takes the
    same decisions for hundreds of log entries and then parts at a trust-region ratio rho = ared/pred whose numerator is a few
    ulps of mx (src/basic_tralcnlss.jl:353-354) — the reference's algorithm amplifies rounding there, whoever computes H*v."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _util import assert_rounding_dominated, first_decision_difference
    from nls_problem import NLSProblem
    P = NLSProblem(256, 48, 2, seed=1)
    kw = dict(max_outer_iter=30, max_inner_iter=60)

    class TwoBlocks(R.NumpyOps):
        def hmul(self, H, v):
            h = H.J.shape[0] // 2
            z = H.J[:h].T @ (H.J[:h] @ v)
            z = z + H.J[h:].T @ (H.J[h:] @ v)
            return z + H.C.T @ ((H.mu * H.C) @ v)

        def projected_cg(self, g, H, wl, wu, lc, k2):
            w, s, _ = R.projected_cg(g, H, wl, wu, lc, k2, hmul_fn=self.hmul)
            return w, s

    logs = []
    xs = []
    for ops in (R.NumpyOps(), TwoBlocks()):
        log = []
        x, y = R.tralcnllss(P.x0, P.r, P.jac_r, P.c, P.jac_c, P.A, P.b, P.x_l, P.x_u, ops=ops, log=log, **kw)
        logs.append(log)
        xs.append(x)
    diff = first_decision_difference(logs[0], logs[1])
    assert diff is not None and diff[0] >= 300          # a long common prefix ...
    assert_rounding_dominated(diff)                     # ... ending at a noise rho
    assert sum(e[0] == "minor" for e in logs[0]) != sum(e[0] == "minor" for e in logs[1])      # 515 vs 559 minor iterates
    assert np.linalg.norm(xs[0] - xs[1]) <= 1e-6 * np.linalg.norm(xs[0])


def test_factor_to_boundary_propagates_nan_like_julia_min():
    """Julia's min(gamma, NaN) is NaN (src/basic_tralcnlss.jl:803,805 fold with `min`); the device's OpMinNan and the C oracle do
    the same.  Python's built-in min drops a NaN that comes second — the restatement must not."""
    p = np.array([1.0, -1.0, 2.0])
    w = np.array([np.nan, 0.0, 0.0])
    assert np.isnan(R.factor_to_boundary(p, w, -np.ones(3), np.ones(3)))                       # NaN among the upper-bound ratios
    assert np.isnan(R.factor_to_boundary(-p, w, -np.ones(3), np.ones(3)))                      # ... among the lower-bound ratios
    assert np.isnan(R.factor_to_boundary(np.array([1.0]), np.array([np.inf]), np.array([-1.0]), np.array([np.inf])))   # Inf - Inf
    assert R.factor_to_boundary(np.array([1e-11, 1.0]), np.array([np.nan, 0.0]), -np.ones(2), np.ones(2)) == 1.0      # a skipped entry's NaN does not count
    import benlsip_oracle as BO
    for args in ((p, w, -np.ones(3), np.ones(3)), (-p, w, -np.ones(3), np.ones(3))):
        assert np.isnan(BO.factor_to_boundary(*args))


def test_reference_bound_on_config_1_is_decided_by_rounding_in_the_oracle_itself():
    """test/problems/sphere_regression.jl:65 asserts opt_measure < 1e-7.  The oracle — the literal restatement — meets it in the
    reference's own evaluation order of H*v, and misses it when the SAME product is summed in a mathematically equivalent order
    (long double accumulation, rows reversed): the last trust-region iterates accept or reject on rho = ared/pred with |ared|
    worth a few ulps of mx.  So the reference's third inequality cannot be guaranteed by any fp64 implementation, Julia's own
    BLAS (whose summation order is unspecified) included; the GPU tests therefore hold every device variant to the band the oracle
    spans here (tests/test_parity_gpu.py::test_sphere_regression_through_c_abi), computed in the test, not to a fitted constant."""
    from _util import sphere_oracle_band
    band = sphere_oracle_band()
    print("\n[config 1, oracle under re-association of H*v] " + ", ".join("%s: %.3e" % kv for kv in band.items()))
    assert band["reference order (mu*C)*v"] < 1e-7
    assert min(band.values()) < 1e-7 < max(band.values())
    assert max(band.values()) < 1e-6


def test_cauchy_search_is_multimodal_on_the_pinned_operands():
    """tests/golden/cauchy_events.json pins the operands of Cauchy searches (src/basic_tralcnlss.jl:574-639) late in a solve, where
    the trust region has collapsed to 1e-11 .. 1e-14 and the direction P(-g) cancels 1e7 .. 1e9 of its digits.  Claim: the outcome
    there is not determined in fp64 — the ORACLE ALONE, fed g perturbed by at most one unit in the last place, lands on several
    final active sets whose steps differ by more than 1e-2, and a second CPU restatement of the same projector (reduced form) lands
    on yet other ones.  (What the device does on these operands is checked in tests/test_parity_gpu.py.)"""
    from _util import ReducedFormOps, cauchy_outcomes, load_cauchy_events
    P, events = load_cauchy_events()
    multi = [e for e in events if e["fix_dev"] != e["fix_cpu"]]
    assert len(multi) >= 3
    for e in multi:
        aug = cauchy_outcomes(P, e, R.NumpyOps(), 32, seed=e["minor"])
        assert np.array_equal(aug[0][0], e["s_cpu"]), "the fixture's s_cpu is no longer what the oracle computes on these operands"
        sets = {}
        for s, key in aug:
            sets.setdefault(key, s)
        red_sets = {key for _, key in cauchy_outcomes(P, e, ReducedFormOps(), 32, seed=e["minor"])}
        keys = list(sets)
        far = max(np.linalg.norm(sets[a] - sets[b]) / np.linalg.norm(sets[b]) for a in keys for b in keys)
        print("\n[Cauchy event, minor iterate %d] delta = %.1e; oracle under 1-ulp perturbations of g: active-set sizes %s, steps up to %.2e apart; "
              "reduced-form restatement: sizes %s (%d sets not reached by the augmented form)"
              % (e["minor"], e["delta"], sorted({len(k) for k in keys}), far, sorted({len(k) for k in red_sets}), len(red_sets - set(keys))))
        assert len(keys) >= 2, "the oracle is unimodal on these operands: the discrepancy would be the device's"
        assert far >= 1e-2
        assert len(red_sets - set(keys)) >= 1


def test_oracles_against_answers_derived_by_hand():
    """Both CPU restatements against projected_cg instances solved by hand from the reference's text (tests/_util.py::closed_form_cases:
    dyadic data, every intermediate exact): the oracle is pinned to src/basic_tralcnlss.jl:702-761 on these, not to itself."""
    import benlsip_oracle as BO
    from _util import closed_form_cases
    n_cases = 0
    for c in closed_form_cases():
        n = c["g"].shape[0]
        L = R.chol_lower(c["A"] @ c["A"].T)
        cons = R.make_mixed_constraints(c["A"], L, c["fix"] if c["fix"].any() else None, l=-1e3 * np.ones(n), u=1e3 * np.ones(n))
        tr = R.CGTrace()
        w, st, it = R.projected_cg(c["g"], R.AlHessian(c["J"], c["C"], c["mu"]), c["wl"], c["wu"], cons, c["kappa2"], trace=tr)
        assert (int(st), it, tr.n_hmul) == (c["status"], c["iters"], c["n_hmul"]), (c["name"], int(st), it, tr.n_hmul)
        assert np.max(np.abs(w - c["w"])) <= c["rtol"] * np.max(np.abs(c["w"])) if c["rtol"] else np.array_equal(w, c["w"]), (c["name"], w, c["w"])
        w2, st2, it2, nh2, _ = BO.projected_cg(c["g"], c["J"], c["C"], c["mu"], c["wl"], c["wu"], c["A"], c["fix"], cons.chol_L, c["kappa2"])
        assert (int(st2), it2, nh2) == (c["status"], c["iters"], c["n_hmul"]), (c["name"], "C port", int(st2), it2, nh2)
        assert np.max(np.abs(w2 - c["w"])) <= c["rtol"] * np.max(np.abs(c["w"])) if c["rtol"] else np.array_equal(w2, c["w"]), (c["name"], "C port", w2)
        n_cases += 1
    assert n_cases == 9


def test_oracle_cauchy_step_against_answers_derived_by_hand():
    """The NumPy restatement of cauchy_step on the instances of tests/_util.py::closed_form_cauchy_cases (exact arithmetic, incl. an exact tie
    between the segment's minimiser and the next breakpoint): pinned to src/basic_tralcnlss.jl:574-639 by hand, not to itself."""
    from _util import closed_form_cauchy_cases
    Z = np.zeros((0, 2))
    L0 = R.chol_lower(Z @ Z.T)
    for c in closed_form_cauchy_cases():
        cons = R.make_mixed_constraints(Z, L0, l=-np.ones(2), u=np.ones(2))
        count = [0]

        class Ops(R.NumpyOps):
            def hmul(self, H, v):
                count[0] += 1
                return R.hmul(H, v)
        s = R.cauchy_step(np.zeros(2), c["g"], R.AlHessian(np.eye(2), Z, 1.0), L0, cons, c["delta"], Ops())
        assert np.array_equal(s, c["s"]) and np.array_equal(cons.fixvars, c["fix"]) and count[0] == c["n_hmul"], (c["name"], s, cons.fixvars, count[0])


def test_oracle_minor_iterate_against_an_answer_derived_by_hand():
    """minor_iterate (:649-675) with H = I: projected_cg returns w = -mask(g) (closed_form_cases `fixed_variable`), the line search has
    w'Hw = |w|^2 = -g'w, so alpha_opt = 1 exactly, no bound on a free variable (:662-665): the minor step is -mask(g) to the last bit."""
    g = np.array([1.0, -2.0, 0.5, 4.0])
    fix = np.array([False, True, False, False])
    x = np.array([0.0, 1.0, 0.0, 0.0])
    Z = np.zeros((0, 4))
    cons = R.make_mixed_constraints(Z, R.chol_lower(Z @ Z.T), fix, l=-np.ones(4), u=np.ones(4))
    w, st = R.minor_iterate(x, np.zeros(4), g, R.AlHessian(np.eye(4), Z, 1.0), cons, 100.0, 0.1)
    assert int(st) == 0 and np.array_equal(w, np.where(fix, 0.0, -g))


def test_oracle_functions_keep_the_reference_names_and_argument_order():
    """The NumPy restatement follows the reference function by function; at the interface level that is checkable mechanically: for
    the functions of the path and of the driver chain, the oracle has a function of the reference's name (`!` dropped) whose positional
    parameters are the reference's, in order (tests/golden/reference_signatures.json) — with `chol_aat` carried as its lower factor
    (`chol_aat_L`), `polyhedron` called `lincons`, and in-place output arguments returned instead."""
    import inspect
    import json
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_signatures.json")))
    by = {}
    for f in ("basic_tralcnlss.jl", "polyhedral_constraints.jl"):
        for s in ref[f]:
            by.setdefault(s["name"].split(".")[-1].rstrip("!"), []).append(s["positional"])
    names = ["projected_cg", "minor_iterate", "cauchy_step", "inner_step", "vthv", "projection", "linesearch", "factor_to_boundary",
             "next_breakpoint", "left_mul", "left_mul_tr", "update_tr", "initial_tr", "tralcnllss", "solve_subproblem", "active_bounds",
             "add_active", "cholesky_aug_aat", "update_chol", "least_squares_multipliers", "norm_reduced_gradient", "projection_nullspace",
             "projection_subspace"]
    rename = {"chol_aat": "chol_aat_L", "polyhedron": "lincons", "indx": "ind"}
    for name in names:
        fn = getattr(R, name)
        mine = [p.name for p in inspect.signature(fn).parameters.values()
                if p.default is inspect.Parameter.empty and p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        wanted = [[rename.get(a, a) for a in sig] for sig in by[name]]
        wanted += [w[:-1] for w in wanted if w and w[-1] == "v"]             # projection!(lincons, r, v) -> v = projection(lincons, r)
        assert mine in wanted, (name, mine, wanted)
