"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the golden fixtures.

Tolerances (fp64, summation order differs from any BLAS so bit parity is not meaningful):
  * one application of J v, J'u, H*v, vthv, projection:  ||y - y_ref||_2 <= 1e-12 * || |J| |v| ||_2 (resp. ||y_ref||)
  * projected_cg: identical status and iteration count; ||w - w_ref|| <= max(1e-9, 20 x oracle rounding sensitivity) ||w_ref||
  * HS48 known answer: max |v - [0,0,0,2,-2]| <= 1e-14
"""
import json
import os

import numpy as np
import pytest

import benlsip_ref as R
import sphere_problem as sp
from _util import assert_iters_in_oracle_band, assert_w_close, matvec_scale, note_tol, oracle_iteration_band, relnorm, w_tolerance

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL1 = 1e-12


def _flt(xs):
    return np.array([float(x) for x in xs], dtype=np.float64)


# ----------------------------------------------------------------------------- operators
@pytest.mark.parametrize("d,n,q", [(4, 3, 1), (1, 1, 0), (0, 5, 2), (5, 5, 5), (37, 5, 2), (64, 128, 0), (300, 130, 0), (257, 129, 1),
                                   (1000, 1024, 3), (513, 2049, 0), (256, 4096, 2), (100, 4095, 0), (130, 8000, 1),
                                   (8192, 1024, 0), (3, 600, 0)])
def test_matvec_parity(bh, d, n, q):
    """AlHessian products (src/basic_tralcnlss.jl:92-106) incl. the reference's structure test (test/structures.jl:1-16)."""
    rng = np.random.default_rng(1000 + d + n)
    J, C, mu = rng.standard_normal((d, n)), rng.standard_normal((q, n)), 0.75
    v, u = rng.standard_normal(n), rng.standard_normal(d)
    H, Ho = bh.AlHessian(J, C, mu), R.AlHessian(J, C, mu)
    assert np.linalg.norm(H.jv(v) - J @ v) <= TOL1 * matvec_scale(J, v)
    assert np.linalg.norm(H.jtv(u) - J.T @ u) <= TOL1 * matvec_scale(J.T, u)
    hv = H * v
    scale = np.linalg.norm(np.abs(J).T @ (np.abs(J) @ np.abs(v)) + mu * np.abs(C).T @ (np.abs(C) @ np.abs(v)))
    assert np.linalg.norm(hv - R.hmul(Ho, v)) <= TOL1 * scale
    # reference property test: H*v ≈ (J'J + mu C'C) v and vthv ≈ v'Hv at rtol sqrt(eps)
    if n <= 2049:
        H_test = J.T @ J + mu * C.T @ C
        np.testing.assert_allclose(hv, H_test @ v, rtol=1.5e-8, atol=1.5e-8 * np.linalg.norm(hv))
    assert bh.vthv(H, v) == pytest.approx(R.vthv(Ho, v), rel=1e-12)
    H.mu = 2.5
    Ho.mu = 2.5
    assert np.linalg.norm(H * v - R.hmul(Ho, v)) <= TOL1 * 4 * scale
    H.close()


def test_matvec_is_deterministic(bh):
    rng = np.random.default_rng(5)
    J = rng.standard_normal((3000, 1000))
    v = rng.standard_normal(1000)
    H = bh.AlHessian(J, None, 1.0)
    a, b = H * v, H * v
    assert np.array_equal(a, b)          # two-stage fixed-order reductions, no atomics
    H.close()


def test_kernel_timers_and_read_probe(bh):
    """bh_time_kernel: the three row-stream classes and the read-only stream probe (kinds 3..6) run on any shape and return
    a positive time; the probe must not disturb the handle (its output lands in the slab buffer the next product rewrites)."""
    rng = np.random.default_rng(5)
    J = rng.standard_normal((777, 130))
    H = bh.AlHessian(J, rng.standard_normal((2, 130)), 3.0)
    v = rng.standard_normal(130)
    before = H * v
    for kind in range(7):
        assert H.time_kernel(kind, 2) > 0.0
    with pytest.raises(bh.BenlsipHipError):
        H.time_kernel(7, 1)
    assert np.array_equal(H * v, before)


def test_caller_supplied_stream(bh):
    """bh_set_stream: the library enqueues on a stream the caller owns (here a torch stream: the same hipStream_t a host
    framework would hand over); results are the same bits as on the library's own stream, and NULL switches back."""
    import torch
    rng = np.random.default_rng(9)
    J = rng.standard_normal((900, 260)) / 30.0
    H = bh.AlHessian(J, None, 1.0)
    Z = np.zeros((0, 260))
    cons = bh.MixedConstraints(Z)
    g = rng.standard_normal(260)
    big = np.full(260, 10.0)
    w0, st0, info0 = bh.projected_cg(g, H, -big, big, cons, 1e-3, full_output=True)
    hv0 = H * g
    lib = bh._lib.lib()
    stream = torch.cuda.Stream()
    bh._lib.check(lib.bh_set_stream(bh._lib.C.c_void_p(stream.cuda_stream)), "bh_set_stream")
    try:
        w1, st1, info1 = bh.projected_cg(g, H, -big, big, cons, 1e-3, full_output=True)
        hv1 = H * g
    finally:
        bh._lib.check(lib.bh_set_stream(None), "bh_set_stream")
    assert int(st1) == int(st0) and info1["iters"] == info0["iters"] and info0["iters"] > 8
    assert np.array_equal(w1, w0) and np.array_equal(hv1, hv0)
    assert np.array_equal(H * g, hv0)


@pytest.mark.parametrize("bad", [np.nan, np.inf])
def test_non_finite_gradient_runs_the_loop_out_like_the_reference(bh, cg_fused, bad):
    """A NaN / Inf in g_minor makes every comparison of the loop false (src/basic_tralcnlss.jl:725, :735, :747): the reference
    iterates until `iter > max_iter` and returns `nothing` with NaN in w.  Same exit, same iteration count, same NaN pattern."""
    rng = np.random.default_rng(1)
    n, d = 40, 200
    J = rng.standard_normal((d, n))
    g = rng.standard_normal(n)
    g[3] = bad
    A = np.zeros((0, n))
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), None, l=-np.ones(n), u=np.ones(n))
    with np.errstate(all="ignore"):
        w_ref, s_ref, it_ref = R.projected_cg(g, R.AlHessian(J, np.zeros((0, n)), 1.0), -np.ones(n), np.ones(n), cons_o, 0.1)
    H = bh.AlHessian(J, None, 1.0)
    w, st, info = bh.projected_cg(g, H, -np.ones(n), np.ones(n), bh.MixedConstraints(A, None, None), 0.1, full_output=True)
    assert int(st) == int(s_ref) == int(R.CGStatus.none) and info["iters"] == it_ref == 2 * n + 1
    assert np.array_equal(np.isnan(w), np.isnan(w_ref)) and np.isnan(w).any()
    H.close()


def test_degenerate_shapes_match_the_oracle(bh, cg_fused):
    """No residual rows (d = 0, with and without C), n = 1 and 2, every variable fixed, no degrees of freedom left (mA + p = n),
    mu = 0, a single row: H*v, vthv, projection and projected_cg against the oracle (tests/edge_shapes_check.py)."""
    import edge_shapes_check
    assert edge_shapes_check.run(bh, cg_fused) == []


def test_out_of_memory_is_an_error_not_a_state(bh):
    """An image that cannot fit the 288 GB of HBM (12M x 4096 doubles = 393 GB) is refused with BH_ERR_HIP, and the next call works:
    the runtime's sticky last-error must not leak into it (it did once: the following bh_hmul failed with 'out of memory')."""
    with pytest.raises(bh.BenlsipHipError) as e:
        bh.AlHessian.synthetic(12_000_000, 4096, seed=1, mu=1.0)
    assert "memory" in str(e.value).lower()
    J = np.random.default_rng(0).standard_normal((500, 64))
    H = bh.AlHessian(J, None, 1.0)
    v = np.ones(64)
    assert relnorm(H * v, J.T @ (J @ v)) <= 1e-12
    H.close()


def test_linearity_and_symmetry_at_full_size(bh):
    """Size-independent properties at BASELINE config 3 (d=65536, n=4096; J generated in HBM):
    H(av+bw) = aHv + bHw, v'Hw = w'Hv, v'Hv = vthv(H,v) = ||Jv||^2, and J rows check against the host generator."""
    d, n = 65536, 4096
    H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
    rng = np.random.default_rng(7)
    v, w = rng.standard_normal(n), rng.standard_normal(n)
    Hv, Hw = H * v, H * w
    lin = H * (2.0 * v - 3.0 * w)
    assert relnorm(lin, 2.0 * Hv - 3.0 * Hw) <= 1e-12
    assert abs(v @ Hw - w @ Hv) <= 1e-12 * abs(v @ Hw)
    Jv = H.jv(v)
    assert abs(v @ Hv - Jv @ Jv) <= 1e-12 * (Jv @ Jv)
    assert bh.vthv(H, v) == pytest.approx(float(Jv @ Jv), rel=1e-12)
    assert relnorm(H.jtv(Jv), Hv) <= 1e-12
    rows = np.array([0, 1, 4097, 65535])
    for i in rows:
        Ji = R.synthetic_J(1, n, seed=1, row0=int(i), d_total=d)
        assert abs(Jv[i] - float(Ji[0] @ v)) <= 1e-12 * float(np.abs(Ji[0]) @ np.abs(v))
    H.close()


# ----------------------------------------------------------------------------- projection
def test_hs48_projection_known_answer(bh):
    """The reference's one golden vector: test/structures.jl:37-58."""
    h = json.load(open(os.path.join(GOLD, "hs48_projection.json")))
    A, x = np.array(h["A"]), np.array(h["x"])
    fix = np.zeros(5, dtype=bool)
    fix[np.array(h["fixed_1based"]) - 1] = True
    L_aug = R.cholesky_aug_aat(A, fix, R.chol_lower(A @ A.T))
    cons = bh.MixedConstraints(A, L_aug, fix)
    B = np.vstack([A, np.eye(5)[fix]])
    y = np.random.default_rng(0).random(4)
    np.testing.assert_allclose(bh.left_mul_tr(cons, y), B.T @ y, rtol=1.5e-8)
    np.testing.assert_allclose(bh.left_mul(cons, x), B @ x, rtol=1.5e-8)
    eps = np.finfo(float).eps
    for form in (1, 0):                 # reduced form (default) and the reference's augmented form, with the reference's own bounds
        bh.set_option("proj_form", form)
        try:
            cons_f = bh.MixedConstraints(A, L_aug, fix)
            proj = bh.projection(cons_f, x)
        finally:
            bh.set_option("proj_form", 1)
        Ap = A @ proj
        print("[HS48, proj_form=%d] projection %s; ||A proj||^2 = %.2e (reference: <= eps = %.2e), max |proj - known answer| = %.2e"
              % (form, proj, float(Ap @ Ap), eps, np.max(np.abs(proj - np.array(h["projection"])))))
        note_tol("HS48: ||A proj||^2 vs the reference's eps (test/structures.jl:55)", float(Ap @ Ap), eps, "proj_form=%d" % form)
        note_tol("HS48: max |proj - [0,0,0,2,-2]| vs 1e-14", float(np.max(np.abs(proj - np.array(h["projection"])))), 1e-14, "proj_form=%d" % form)
        assert np.all(proj[fix] <= eps) and float(Ap @ Ap) <= eps                     # test/structures.jl:55-56, as the reference asserts them
        assert np.max(np.abs(proj - np.array(h["projection"]))) <= 1e-14              # :57 / SURVEY §8c
        v = np.empty(5)
        bh.projection_(cons_f, x, v)
        assert np.array_equal(v, proj)
        cons_f.close()


@pytest.fixture(params=[1, 0], ids=["reduced_form", "augmented_form"])
def proj_form(request, bh):
    """Both projection forms of the library: 1 = reduced mA x mA factor built on the device (default), 0 = the
    reference's augmented mpp x mpp factor handed over by the caller."""
    bh._lib.lib().bh_set_option(b"proj_form", request.param)
    yield request.param
    bh._lib.lib().bh_set_option(b"proj_form", 1)


@pytest.mark.parametrize("n,mA,nfix", [(6, 3, 3), (7, 3, 0), (40, 5, 10), (300, 20, 150), (300, 20, 0), (130, 64, 66), (1000, 64, 512),
                                       (513, 1, 1), (64, 0, 9), (64, 0, 0), (4096, 64, 512), (700, 130, 200), (900, 65, 3), (1200, 257, 300),
                                       (3500, 2, 3400)])
def test_projection_parity(bh, proj_form, n, mA, nfix):
    """projection_nullspace! / projection_subspace! (src/polyhedral_constraints.jl:104-136) incl. garbage in the unread
    upper triangle of the factor.  Tolerance: a normal-equations projector is accurate to
    ~eps*cond(B B') with B = [A; I_fix]; 1e-11*||r|| floor, 200*eps*cond above it (only (130,64,66), mpp == n, needs it).
    Factor orders: <= 64 (one-wave solve), 65 / 130 / 257 (blocked solve, partial last block), 557 / 576 (augmented form, split
    trailing update) and 3402 (augmented form, too large for the split: single-slice update)."""
    rng = np.random.default_rng(n + 7 * mA + nfix)
    A = rng.standard_normal((mA, n))
    L0 = R.chol_lower(A @ A.T)
    fix = np.zeros(n, dtype=bool)
    fix[rng.choice(n, nfix, replace=False)] = True
    cons_o = R.make_mixed_constraints(A, L0, fix if nfix else None)
    L = cons_o.chol_L.copy()
    if L.shape[0] > 1:
        L[np.triu_indices(L.shape[0], 1)] = np.nan        # uninitialised upper triangle must never be read
    cons = bh.MixedConstraints(A, L, fix)
    B = np.vstack([A, np.eye(n)[fix]])
    tol = max(1e-11, 200 * np.finfo(float).eps * np.linalg.cond(B @ B.T)) if B.shape[0] else 1e-11
    for _ in range(2):
        r = rng.standard_normal(n)
        v_ref = R.projection(cons_o, r)
        v = bh.projection(cons, r)
        assert np.linalg.norm(v - v_ref) <= tol * np.linalg.norm(r), (n, mA, nfix, np.linalg.norm(v - v_ref) / np.linalg.norm(r), tol)
        if mA:
            assert np.linalg.norm(A @ v) <= 10 * tol * np.linalg.norm(A) * np.linalg.norm(r)
        if mA == 0 or proj_form == 1:
            assert np.all(v[fix] == 0.0)
        else:
            assert np.max(np.abs(v[fix]), initial=0.0) <= tol * np.linalg.norm(r)
        # idempotence of an orthogonal projector
        assert np.linalg.norm(bh.projection(cons, v) - v) <= 100 * tol * np.linalg.norm(r)


def test_reduced_form_needs_no_host_factor_and_reports_rank_deficiency(bh):
    """Default (reduced) form: the caller's factor is optional; a rank-deficient A_free is reported like the reference's
    PosDefException instead of producing garbage."""
    bh._lib.lib().bh_set_option(b"proj_form", 1)
    rng = np.random.default_rng(5)
    n, mA = 60, 4
    A = rng.standard_normal((mA, n))
    fix = np.zeros(n, dtype=bool)
    fix[:7] = True
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix)
    cons = bh.MixedConstraints(A, None, fix)                   # no factor handed over
    r = rng.standard_normal(n)
    assert relnorm(bh.projection(cons, r), R.projection(cons_o, r)) <= 1e-12
    A2 = A.copy()
    A2[3] = 0.0                                                 # rank-deficient: a zero pivot, exactly
    bad = bh.MixedConstraints(A2, None, fix)
    with pytest.raises(bh.BenlsipHipError) as e:
        bh.projection(bad, r)
    assert e.value.code == -5


def test_projection_follows_active_set_changes(bh):
    """add_active! / active_bounds! mutate fixvars + chol between calls (src/polyhedral_constraints.jl:203-261)."""
    rng = np.random.default_rng(11)
    n, mA = 50, 4
    A = rng.standard_normal((mA, n))
    L0 = R.chol_lower(A @ A.T)
    cons_o = R.make_mixed_constraints(A, L0, l=-np.ones(n), u=np.ones(n))
    cons = bh.MixedConstraints(A, L0, l=-np.ones(n), u=np.ones(n))
    r = rng.standard_normal(n)
    assert relnorm(bh.projection(cons, r), R.projection(cons_o, r)) <= 1e-12
    for ind in ([3], [10, 11], [49]):
        R.add_active(cons_o, L0, np.array(ind))
        cons.set_active(cons_o.fixvars, cons_o.chol_L)
        assert relnorm(bh.projection(cons, r), R.projection(cons_o, r)) <= 1e-11
        assert cons.nb_fix() == R.nb_fix(cons_o)


def test_preconditions_are_reported_not_asserted(bh):
    A = np.random.default_rng(0).standard_normal((2, 4))
    cons = bh.MixedConstraints(A, np.eye(2))
    cons.set_active(np.array([True, True, True, False]), np.eye(5))       # mpp = 5 > n = 4 (:43, :128)
    with pytest.raises(bh.BenlsipHipError) as e:
        bh.projection(cons, np.ones(4))
    assert e.value.code == -5
    cons.set_active(np.array([True, False, False, False]), np.eye(2))     # factor of the wrong order
    with pytest.raises(ValueError):
        bh.projection(cons, np.ones(4))
    bh._lib.lib().bh_set_option(b"proj_form", 0)                          # reference form requires the factor
    try:
        cons.set_active(np.array([True, False, False, False]), None)
        with pytest.raises(bh.BenlsipHipError) as e:
            bh.projection(cons, np.ones(4))
        assert e.value.code == -1
    finally:
        bh._lib.lib().bh_set_option(b"proj_form", 1)
    H = bh.AlHessian(np.ones((3, 5)), None, 1.0)
    cons4 = bh.MixedConstraints(np.zeros((0, 4)))
    with pytest.raises(bh.BenlsipHipError) as e:
        bh.projected_cg(np.ones(5), H, -np.ones(5), np.ones(5), cons4, 0.1)
    assert e.value.code == -6


# ----------------------------------------------------------------------------- factor_to_boundary
def test_factor_to_boundary_parity(bh):
    rng = np.random.default_rng(3)
    for n in (1, 5, 64, 1000, 4097):
        p, w = rng.standard_normal(n), 0.1 * rng.standard_normal(n)
        p[rng.random(n) < 0.3] = 1e-11
        wl = np.where(rng.random(n) < 0.5, -np.inf, -1.0)
        wu = np.where(rng.random(n) < 0.5, np.inf, 1.0)
        assert bh.factor_to_boundary(p, w, wl, wu) == R.factor_to_boundary(p, w, wl, wu)
    # NaN operands propagate like Julia's min (src/basic_tralcnlss.jl:803,805): NaN in w, Inf - Inf, NaN in p (fails both guards: skipped)
    nan = np.nan
    for p, w, wl, wu in ((np.array([1.0, -1.0, 2.0]), np.array([nan, 0.0, 0.0]), -np.ones(3), np.ones(3)),
                         (np.array([-1.0, 1.0]), np.array([nan, 0.0]), -np.ones(2), np.ones(2)),
                         (np.array([1.0]), np.array([np.inf]), np.array([-1.0]), np.array([np.inf])),
                         (np.array([nan, 1.0]), np.array([0.0, 0.0]), -np.ones(2), np.ones(2)),
                         (np.array([1e-11, 1.0]), np.array([nan, 0.0]), -np.ones(2), np.ones(2))):
        dev, ref = bh.factor_to_boundary(p, w, wl, wu), R.factor_to_boundary(p, w, wl, wu)
        assert (np.isnan(dev) and np.isnan(ref)) or dev == ref, (p, w, dev, ref)
    assert np.isnan(bh.factor_to_boundary(np.array([1.0, -1.0, 2.0]), np.array([nan, 0.0, 0.0]), -np.ones(3), np.ones(3)))
    z = np.zeros(8)
    assert bh.factor_to_boundary(z, z, -np.ones(8), np.ones(8)) == np.inf
    assert bh.factor_to_boundary(np.array([-1.0]), np.array([0.0]), np.array([-np.inf]), np.array([np.inf])) == np.inf


# ----------------------------------------------------------------------------- projected_cg
def _load_case(c):
    d, n, q, mA, mpp = c["d"], c["n"], c["q"], c["mA"], c["mpp"]
    J = _flt(c["J"]).reshape((d, n), order="F")
    C = _flt(c["C"]).reshape((q, n), order="F")
    A = _flt(c["A"]).reshape((mA, n), order="F")
    L = _flt(c["L"]).reshape((mpp, mpp), order="F")
    fix = np.array(c["fixvars"], dtype=bool)
    return J, C, A, L, fix, _flt(c["g"]), _flt(c["w_l"]), _flt(c["w_u"])


@pytest.fixture(params=[1, 0, 2], ids=["two_kernel_iteration", "round1_iteration", "fused_also_with_linear_equalities"])
def cg_fused(request, bh):
    """The shapes of the CG iteration: 1 (default) = box constraints in two kernels (H*p with the p-update folded in + one
    reduce/update kernel), general constraints in seven; 0 = H*p + slab reduction + single-workgroup step kernel(s) everywhere
    (what multi-rank runs and A/B geometries use); 2 = linear equalities too in fused form (four kernels, opt-in)."""
    bh.set_option("cg_fused", request.param)
    yield request.param
    bh.set_option("cg_fused", 1)


def test_pcg_golden_fixtures(bh, cg_fused):
    """Committed oracle fixtures: every reachable exit status, q>0, mA>0, p>0, odd n, n=3, max_iter=0, exhaustion."""
    cases = json.load(open(os.path.join(GOLD, "pcg_cases.json")))["cases"]
    for c in cases:
        J, C, A, L, fix, g, wl, wu = _load_case(c)
        n = c["n"]
        H = bh.AlHessian(J, C, c["mu"])
        cons = bh.MixedConstraints(A, L, fix)
        w, status, info = bh.projected_cg(g, H, wl, wu, cons, c["kappa2"], trace_cap=64, full_output=True)
        assert int(status) == c["status"], c["name"]
        assert info["iters"] == c["iters"] and info["n_hmul"] == c["n_hmul"], c["name"]
        w_ref = _flt(c["w"])
        cons_o = R.MixedConstraints(A, -np.ones(n), np.ones(n), fix, L)
        tol = w_tolerance(g, R.AlHessian(J, C, c["mu"]), wl, wu, cons_o, c["kappa2"], w_ref)
        if c["name"] == "maxiter_exhaust":
            tol = 1e-6      # 8 of its 14 iterations run on rounding noise (kappa2 = 0); only status/iters are pinned tightly
        if np.all(np.isfinite(w_ref)):
            assert_w_close(w, w_ref, tol, "golden fixtures: w", c["name"])
        else:
            assert np.array_equal(np.isnan(w), np.isnan(w_ref)) and np.array_equal(w[np.isfinite(w_ref)], w_ref[np.isfinite(w_ref)])
        tr_ref = np.array([[float(x) for x in row] for row in c["trace"]]).reshape(-1, 4)
        tr = info["trace"]
        assert tr.shape == tr_ref.shape
        if c["name"] != "maxiter_exhaust" and tr.size:
            m = np.isfinite(tr_ref)
            assert np.array_equal(np.isnan(tr), np.isnan(tr_ref)), c["name"]
            # SURVEY.md §8c asks 1e-9 for the scalar trace; the scalars of late iterations inherit the instance's own rounding
            # sensitivity (the same `tol` as for w: max(1e-9, 20 x what the ORACLE's w moves under long-double accumulation))
            rt = max(1e-9, tol)
            err = np.abs(tr[m] - tr_ref[m]) / (1e-10 / rt + np.abs(tr_ref[m]))
            note_tol("golden fixtures: scalar trace (rtol max(1e-9, case tolerance))", float(err.max()), rt, c["name"])
            np.testing.assert_allclose(tr[m], tr_ref[m], rtol=rt, atol=1e-10)
        H.close()
        cons.close()


@pytest.mark.parametrize("image,fused", [(1, 1), (1, 0), (0, 0)], ids=["row_space_one_kernel_per_breakpoint", "row_space_two_kernels", "sweeping"])
def test_cauchy_step_answers_derived_by_hand(bh, image, fused):
    """bh_cauchy_step on the instances solved by hand from src/basic_tralcnlss.jl:574-639 (tests/_util.py::closed_form_cauchy_cases): step to
    the last bit, active set, number of passes — also at the exact tie between a segment's minimiser and the next breakpoint — in the
    row-space forms (one kernel per breakpoint, two) and in the one that sweeps J per breakpoint."""
    from _util import closed_form_cauchy_cases
    bh.set_option("cauchy_image", image)
    bh.set_option("cauchy_fused", fused)
    try:
        H = bh.AlHessian(np.eye(2), None, 1.0)
        for c in closed_form_cauchy_cases():
            cons = bh.MixedConstraints(np.zeros((0, 2)), None, None, l=-np.ones(2), u=np.ones(2))
            s, info = bh.cauchy_step(np.zeros(2), c["g"], H, cons, c["delta"], full_output=True)
            assert np.array_equal(s, c["s"]) and np.array_equal(np.asarray(cons.fixvars, dtype=bool), c["fix"]) and info["n_hmul"] == c["n_hmul"], \
                (c["name"], image, s, cons.fixvars, info)
            cons.close()
        H.close()
    finally:
        bh.set_option("cauchy_image", 1)
        bh.set_option("cauchy_fused", 1)


def test_minor_iterate_answer_derived_by_hand(bh, cg_fused):
    """bh_minor_iterate with H = I and one variable on its bound: w = -mask(g) from the CG loop, alpha_opt = -g'w / w'Hw = 1 exactly, no
    bound on a free variable (:662-665) — the minor step is -mask(g) to the last bit (the same derivation pins the oracle on the CPU)."""
    g = np.array([1.0, -2.0, 0.5, 4.0])
    fix = np.array([False, True, False, False])
    x = np.array([0.0, 1.0, 0.0, 0.0])
    H = bh.AlHessian(np.eye(4), None, 1.0)
    cons = bh.MixedConstraints(np.zeros((0, 4)), None, fix, l=-np.ones(4), u=np.ones(4))
    w, st, info = bh.minor_iterate(x, np.zeros(4), g, H, cons, 100.0, 0.1, full_output=True)
    assert int(st) == 0 and info["alpha"] == 1.0 and np.array_equal(w, np.where(fix, 0.0, -g)), (int(st), info, w)
    H.close()
    cons.close()


def test_pcg_answers_derived_by_hand(bh, cg_fused):
    """projected_cg through the C ABI on the instances solved by hand from the reference's text (tests/_util.py::closed_form_cases: H = I,
    a bound hit, zero and negative curvature, the penalty block alone, one equality, a fixed variable, everything fixed, a two-step
    diagonal): exit status, `iter`, product count and — the data are dyadic — the step to the last bit, in every iteration shape and
    both projection forms.  The same cases pin the two CPU oracles (tests/test_oracle_cpu.py)."""
    from _util import closed_form_cases
    for form in (1, 0):
        bh.set_option("proj_form", form)
        try:
            for c in closed_form_cases():
                H = bh.AlHessian(c["J"], c["C"], c["mu"])
                L = R.chol_lower(c["A"] @ c["A"].T) if form == 0 else None
                if form == 0:
                    n = c["g"].shape[0]
                    L = R.make_mixed_constraints(c["A"], L, c["fix"] if c["fix"].any() else None, l=-np.ones(n), u=np.ones(n)).chol_L
                cons = bh.MixedConstraints(c["A"], L, c["fix"])
                w, st, info = bh.projected_cg(c["g"], H, c["wl"], c["wu"], cons, c["kappa2"], full_output=True)
                assert (int(st), info["iters"], info["n_hmul"]) == (c["status"], c["iters"], c["n_hmul"]), (c["name"], form, int(st), info)
                if c["rtol"]:
                    assert np.max(np.abs(w - c["w"])) <= c["rtol"] * np.max(np.abs(c["w"])), (c["name"], form, w, c["w"])
                else:
                    assert np.array_equal(w, c["w"]), (c["name"], form, w, c["w"])
                H.close()
                cons.close()
        finally:
            bh.set_option("proj_form", 1)


@pytest.mark.parametrize("d,n,q,mA,nfix,seed", [(50, 20, 0, 0, 4, 1), (200, 64, 1, 3, 10, 2), (300, 100, 0, 0, 0, 3), (512, 257, 0, 2, 30, 4),
                                                (1024, 512, 0, 0, 64, 5), (2000, 1000, 2, 8, 100, 6), (600, 300, 0, 16, 0, 7)])
def test_pcg_random_instances(bh, cg_fused, d, n, q, mA, nfix, seed):
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    C = rng.standard_normal((q, n))
    A = rng.standard_normal((mA, n))
    L0 = R.chol_lower(A @ A.T)
    fix = np.zeros(n, dtype=bool)
    fix[rng.choice(n, nfix, replace=False)] = True
    cons_o = R.make_mixed_constraints(A, L0, fix if nfix else None, l=-np.ones(n), u=np.ones(n))
    x_minor = np.clip(0.3 * rng.standard_normal(n), -0.9, 0.9)
    x_minor[fix] = 1.0
    g = rng.standard_normal(n)
    w_l, w_u = R.build_step_bounds(x_minor, cons_o, 0.1 * np.linalg.norm(g))
    Ho = R.AlHessian(J, C, 10.0)
    w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.1)
    H = bh.AlHessian(J, C, 10.0)
    cons = bh.MixedConstraints(A, cons_o.chol_L, fix)
    w, status, info = bh.projected_cg(g, H, w_l, w_u, cons, 0.1, full_output=True)
    assert int(status) == int(s_ref) and info["iters"] == it_ref
    assert_w_close(w, w_ref, w_tolerance(g, Ho, w_l, w_u, cons_o, 0.1, w_ref), "projected_cg: w vs oracle (tolerance max(1e-9, 20 x oracle sensitivity))",
                   "random instance d=%d n=%d q=%d mA=%d" % (d, n, q, mA))
    # w stays in the null space of the active constraints (src/basic_tralcnlss.jl:681-684)
    if mA:
        assert np.linalg.norm(A @ w) <= 1e-9 * np.linalg.norm(A) * np.linalg.norm(w)
    assert np.max(np.abs(w[fix]), initial=0.0) <= 1e-12 * np.linalg.norm(w)
    H.close()


@pytest.mark.parametrize("dep,cond_min", [(1e-4, 1e8), (1e-6, 1e12)])
def test_pcg_ill_conditioned_equalities_stay_feasible(bh, dep, cond_min):
    """Rows of A nearly dependent (cond(A A') = 1e9 / 1e13).  The step w must stay in null(A) as well as the reference's two
    triangular solves keep it there (src/polyhedral_constraints.jl:114-115): the default three-kernel iteration applies the explicit
    inverse of the factor, which alone left |A w| 100 - 4000 x larger (1.7e-10 against 1.0e-12, 2.0e-7 against 4.7e-11:
    tests/manual/illcond_probe.py), and repairs it with one step of iterative refinement (option linv_refine).  Held to 10 x the
    larger of the oracle's and the triangular-solve shape's own |A w|/|A||w|; status and iteration count as the oracle's where the
    oracle itself is stable under an LU projector."""
    rng = np.random.default_rng(3)
    d, n, mA = 600, 300, 12
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    g = rng.standard_normal(n)
    A = rng.standard_normal((mA, n))
    A[1] = A[0] + dep * rng.standard_normal(n)
    A[5] = A[4] - A[3] + dep * rng.standard_normal(n)
    M = A @ A.T
    assert np.linalg.cond(M) > cond_min
    cons_o = R.make_mixed_constraints(A, R.chol_lower(M), None, l=-np.ones(n), u=np.ones(n))
    wl, wu = -10.0 * np.ones(n), 10.0 * np.ones(n)
    Ho = R.AlHessian(J, np.zeros((0, n)), 1.0)
    w_ref, st_ref, it_ref = R.projected_cg(g, Ho, wl, wu, cons_o, 1e-3)
    w_lu, st_lu, it_lu = R.projected_cg(g, Ho, wl, wu, cons_o, 1e-3, proj_fn=lambda cons, r: r - A.T @ np.linalg.solve(M, A @ r))
    feas = lambda w: float(np.linalg.norm(A @ w) / (np.linalg.norm(A) * np.linalg.norm(w)))
    H = bh.AlHessian(J, None, 1.0)
    out = {}
    for fused in (1, 2):
        bh.set_option("cg_fused", fused)
        try:
            cons = bh.MixedConstraints(A, None, None, l=-np.ones(n), u=np.ones(n))
            w, st, info = bh.projected_cg(g, H, wl, wu, cons, 1e-3, full_output=True)
        finally:
            bh.set_option("cg_fused", 1)
        out[fused] = (w, int(st), info["iters"])
        cons.close()
    H.close()
    bound = 10.0 * max(feas(w_ref), feas(out[2][0])) + 1e-15
    note_tol("ill-conditioned equalities: |A w|/|A||w| of the explicit-inverse iteration vs 10 x (oracle, triangular solves)", feas(out[1][0]), bound,
             "cond(A A') = %.1e: device %.1e, triangular solves %.1e, oracle %.1e" % (np.linalg.cond(M), feas(out[1][0]), feas(out[2][0]), feas(w_ref)))
    assert feas(out[1][0]) <= bound, (feas(out[1][0]), feas(out[2][0]), feas(w_ref))
    assert out[1][1] == out[2][1] == int(st_ref)
    if (int(st_lu), it_lu) == (int(st_ref), it_ref):          # the oracle's own count is stable under another projector arithmetic
        assert out[1][2] == out[2][2]
        drift = max(relnorm(w_lu, w_ref), 1e-12)
        assert relnorm(out[1][0], w_ref) <= 1e3 * drift and relnorm(out[1][0], out[2][0]) <= 1e3 * drift


@pytest.mark.parametrize("d,n,nfix,kappa2", [(60, 5000, 300, 0.1), (40, 8192, 0, 0.1), (50, 9001, 700, 0.1), (30, 16384, 1000, 0.3),
                                             (33, 16385, 5, 0.3), (2000, 2048, 100, 0.01), (700, 1000, 7, 0.01)])
def test_pcg_wide_rows_every_kernel_geometry(bh, cg_fused, d, n, nfix, kappa2):
    """projected_cg with box constraints through every geometry of the row-streaming kernel that carries the CG prologue —
    <256,2,8> … <512,8,2>, the <512,16,1> variant that parks the vector in LDS (8192 < n <= 16384) — and through the column-panel
    fallback above it (n = 16385), on both iteration shapes, against the oracle (few rows: rank-deficient H, so the loop also meets
    near-zero curvature directions)."""
    rng = np.random.default_rng(n + d)
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    fix = np.zeros(n, dtype=bool)
    fix[rng.choice(n, nfix, replace=False)] = True
    A = np.zeros((0, n))
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix if nfix else None, l=-np.ones(n), u=np.ones(n))
    g = J.T @ rng.standard_normal(d) + 1e-3 * rng.standard_normal(n)
    w_l, w_u = R.build_step_bounds(np.where(fix, 1.0, 0.0), cons_o, 0.5 * np.linalg.norm(g))
    Ho = R.AlHessian(J, np.zeros((0, n)), 2.0)
    tr = R.CGTrace()
    w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, kappa2, trace=tr)
    H = bh.AlHessian(J, None, 2.0)
    cons = bh.MixedConstraints(A, None, fix)
    w, status, info = bh.projected_cg(g, H, w_l, w_u, cons, kappa2, trace_cap=32, full_output=True)
    assert int(status) == int(s_ref) and info["iters"] == it_ref and info["n_hmul"] == tr.n_hmul, (status, info["iters"], s_ref, it_ref)
    assert_w_close(w, w_ref, w_tolerance(g, Ho, w_l, w_u, cons_o, kappa2, w_ref), "projected_cg: w vs oracle (tolerance max(1e-9, 20 x oracle sensitivity))")
    k = min(len(tr.rows), 32)
    ref_rows = np.array(tr.rows[:k])
    m = np.isfinite(ref_rows)
    np.testing.assert_allclose(info["trace"][:k][m], ref_rows[m], rtol=1e-6, atol=1e-12)
    w2, status2, info2 = bh.projected_cg(g, H, w_l, w_u, cons, kappa2, full_output=True)      # second call: launch schedule from the hint
    assert np.array_equal(w, w2) and int(status2) == int(status) and info2["iters"] == info["iters"]
    H.close()


@pytest.mark.parametrize("d,n,nfix", [(40000, 128, 9), (70000, 30, 0), (30000, 500, 40)])
def test_pcg_tall_narrow_more_workgroups_than_columns(bh, cg_fused, d, n, nfix):
    """Tall, narrow J: the row-streaming grid (up to 8 workgroups per CU = 2048) has MORE workgroups than the vectors have
    entries, so every per-workgroup array of the two-kernel iteration (partial sums, minima) must be sized by the grid, not by n
    (a buffer of n_pad doubles was overrun here once).  Against the oracle, twice (second call: launch schedule from the hint),
    then a different subproblem on the same workspace to catch a stray write."""
    rng = np.random.default_rng(d + n)
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    fix = np.zeros(n, dtype=bool)
    if nfix:
        fix[rng.choice(n, nfix, replace=False)] = True
    A = np.zeros((0, n))
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix if nfix else None, l=-np.ones(n), u=np.ones(n))
    Ho = R.AlHessian(J, np.zeros((0, n)), 2.0)
    H = bh.AlHessian(J, None, 2.0)
    cons = bh.MixedConstraints(A, None, fix)
    for trial in range(2):
        g = J.T @ rng.standard_normal(d) + 1e-3 * rng.standard_normal(n)
        w_l, w_u = R.build_step_bounds(np.where(fix, 1.0, 0.0), cons_o, 0.5 * np.linalg.norm(g))
        w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, 1e-3)
        for _ in range(2):
            w, status, info = bh.projected_cg(g, H, w_l, w_u, cons, 1e-3, full_output=True)
            assert int(status) == int(s_ref) and info["iters"] == it_ref, (trial, status, info["iters"], s_ref, it_ref)
            assert_w_close(w, w_ref, w_tolerance(g, Ho, w_l, w_u, cons_o, 1e-3, w_ref), "projected_cg: w vs oracle (tolerance max(1e-9, 20 x oracle sensitivity))")
    H.close()


def test_pcg_config5_shape_linear_constraints(bh, proj_form, cg_fused):
    """BASELINE config 5 shape at oracle-sized d: n=1024, mA=16 linear equalities + p=128 active bounds (mpp=144)."""
    d, n, mA = 2048, 1024, 16
    J = R.synthetic_J(d, n, seed=1)
    inst = R.synthetic_box_vectors(d, n, fix_every=8)
    A = R.splitmix_uniform(4, np.arange(mA * n)).reshape((mA, n), order="F")
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), inst.fixvars, l=inst.x_l, u=inst.x_u)
    Ho = R.AlHessian(J, np.zeros((0, n)), 10.0)
    g = J.T @ inst.r0
    w_l, w_u = R.build_step_bounds(inst.x, cons_o, 0.1 * np.linalg.norm(g))
    w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.01)
    H = bh.AlHessian(J, None, 10.0)
    cons = bh.MixedConstraints(A, cons_o.chol_L, inst.fixvars)
    w, status, info = bh.projected_cg(g, H, w_l, w_u, cons, 0.01, full_output=True)
    assert int(status) == int(s_ref) and info["iters"] == it_ref
    assert_w_close(w, w_ref, w_tolerance(g, Ho, w_l, w_u, cons_o, 0.01, w_ref), "projected_cg: w vs oracle (tolerance max(1e-9, 20 x oracle sensitivity))")
    assert np.linalg.norm(A @ w) <= 1e-10 * np.linalg.norm(A) * np.linalg.norm(w)


def test_pcg_config2_synthetic_box(bh):
    """BASELINE config 2: d=8192, n=1024, box bounds, p = n/8, J generated in HBM vs the host generator."""
    d, n = 8192, 1024
    J = R.synthetic_J(d, n, seed=1)
    inst = R.synthetic_box_vectors(d, n, fix_every=8)
    A = np.zeros((0, n))
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), inst.fixvars, l=inst.x_l, u=inst.x_u)
    Ho = R.AlHessian(J, np.zeros((0, n)), 10.0)
    g = J.T @ inst.r0
    w_l, w_u = R.build_step_bounds(inst.x, cons_o, 0.1 * np.linalg.norm(g))
    w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.1)
    H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
    assert np.linalg.norm(H.jtv(inst.r0) - g) <= 1e-12 * matvec_scale(J.T, inst.r0)
    cons = bh.MixedConstraints(A, None, inst.fixvars, l=inst.x_l, u=inst.x_u)
    w, status, info = bh.projected_cg(g, H, w_l, w_u, cons, 0.1, full_output=True)
    assert int(status) == int(s_ref) and info["iters"] == it_ref
    assert_w_close(w, w_ref, w_tolerance(g, Ho, w_l, w_u, cons_o, 0.1, w_ref), "projected_cg: w vs oracle (tolerance max(1e-9, 20 x oracle sensitivity))")
    # Ill-conditioned variant (columns scaled by 10^(-3j/n), cond(J'J) ~ 1e6): hundreds of iterations.  Finite-precision
    # CG is chaotic here: the ORACLE itself takes 406 / 451 / 465 iterations and moves w by 13-18 % when J@v is summed
    # in F-order, C-order or 1024-row chunks (measured in this container), so iteration-level parity is not defined.
    # Checked instead: same exit status, iteration count inside that spread, the exit test holds for the returned w,
    # and the model value agrees with the oracle's to second order.
    scale = 10.0 ** (-3.0 * np.arange(n) / n)
    Jic = R.synthetic_J(d, n, seed=1, kind=1)
    Hic_o = R.AlHessian(Jic, np.zeros((0, n)), 10.0)
    gic = Jic.T @ inst.r0
    w_l, w_u = R.build_step_bounds(inst.x, cons_o, 0.1 * np.linalg.norm(gic))
    w_ref, s_ref, it_ref = R.projected_cg(gic, Hic_o, w_l, w_u, cons_o, 1e-3)
    Hic = bh.AlHessian.synthetic(d, n, seed=1, colscale=scale, mu=10.0)
    w, status, info = bh.projected_cg(gic, Hic, w_l, w_u, cons, 1e-3, full_output=True)
    assert int(status) == int(s_ref) == 0
    # the oracle under re-association of its own H*p (the BLAS-backed variants only: long double takes minutes at this size)
    band = oracle_iteration_band(gic, Hic_o, w_l, w_u, cons_o, 1e-3, variants=("reference", "C-order sums", "1024-row chunks", "rows reversed", "3 row blocks"))
    print("[config 2, ill-conditioned] device %d iterations; oracle: %s" % (info["iters"], band))
    assert all(st == 0 for st, _ in band.values())
    assert_iters_in_oracle_band(info["iters"], band, "ill-conditioned CG: iteration count vs the oracle's own band", "config 2 ic")
    res = np.where(inst.fixvars, 0.0, R.hmul(Hic_o, w) + gic)
    v0 = np.where(inst.fixvars, 0.0, gic)
    assert abs(res @ res) < 1e-3 * np.linalg.norm(v0) * 1.001
    q = lambda x: 0.5 * R.vthv(Hic_o, x) + gic @ x
    assert abs(q(w) - q(w_ref)) <= 2e-2 * abs(q(w_ref))


def test_pcg_full_size_properties(bh):
    """BASELINE config 3 (d=65536, n=4096, box, p=512) — too large for the oracle in seconds, so size-independent
    properties: the returned w satisfies the CG exit test it claims, lies in the null space, and reduces the model."""
    d, n = 65536, 4096
    H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
    inst = R.synthetic_box_vectors(d, n, fix_every=8)
    g = H.jtv(inst.r0)
    A = np.zeros((0, n))
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), inst.fixvars, l=inst.x_l, u=inst.x_u)
    w_l, w_u = R.build_step_bounds(inst.x, cons_o, 0.1 * np.linalg.norm(g))
    cons = bh.MixedConstraints(A, None, inst.fixvars, l=inst.x_l, u=inst.x_u)
    w, status, info = bh.projected_cg(g, H, w_l, w_u, cons, 0.1, trace_cap=64, full_output=True)
    assert status == bh.CGStatus.solved and 2 <= info["iters"] <= 64
    assert np.all(w[inst.fixvars] == 0.0)
    v0 = np.where(inst.fixvars, 0.0, g)
    res = np.where(inst.fixvars, 0.0, H * w + g)                 # projected residual at exit
    assert abs(res @ res) < 0.1 * np.linalg.norm(v0) * 1.0000001  # |rtv| < kappa2*||v0|| (:710,:747)
    assert abs(info["trace"][-1, 3] - res @ res) <= 1e-8 * max(abs(res @ res), 1e-300) + 1e-12 * (v0 @ v0)
    q0, q1 = 0.0, 0.5 * bh.vthv(H, w) + g @ w
    assert q1 < q0
    w2, status2, info2 = bh.projected_cg(g, H, w_l, w_u, cons, 0.1, full_output=True)
    assert np.array_equal(w, w2) and info2["iters"] == info["iters"]       # bit-reproducible
    H.close()


# ----------------------------------------------------------------------------- config 1 through the restated driver
from hip_ops import HipOps, HipOpsDeviceAll, HipOpsDeviceMinor, HipOpsResident, ShadowOps  # noqa: E402  (backends of the restated driver)


@pytest.mark.parametrize("ops_cls", [HipOps, HipOpsDeviceMinor, HipOpsDeviceAll, HipOpsResident],
                         ids=["pcg_abi", "minor_iterate_abi", "cauchy_abi", "resident_inner_step"])
def test_sphere_regression_through_c_abi(bh, capsys, ops_cls):
    """BASELINE config 1: test/problems/sphere_regression.jl with every hot-path call on the GPU; the three acceptance
    inequalities of :63-65 and agreement with the CPU oracle's solution."""
    ops = ops_cls(bh)
    log = []
    xs, ys = R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u,
                          max_outer_iter=100, max_inner_iter=250, ops=ops, log=log)
    assert ops.n_pcg > 10
    grad = sp.jac_r(xs).T @ sp.r(xs) + sp.jac_c(xs).T @ ys
    P = R.projection_polyhedron_small(xs - grad, sp.A, sp.b, sp.x_l, sp.x_u)
    opt_measure = float(np.linalg.norm(xs - P))
    with capsys.disabled():
        print("[sphere regression, %s] opt_measure = %.3e (reference asserts < 1e-7; oracle 7.16e-8), |c(x)| = %.2e, %d minor iterates"
              % (ops_cls.__name__, opt_measure, np.linalg.norm(sp.c(xs)), sum(e[0] == "minor" for e in log)))
    assert np.linalg.norm(sp.c(xs)) < R.SQRT_EPS
    assert R.is_feasible(xs, sp.A, sp.x_l, sp.x_u, sp.b)
    # The reference's third inequality (opt_measure < 1e-7, :65) is decided by rounding: the ORACLE itself meets it in the
    # reference's evaluation order of H*v and misses it under mathematically equivalent ones (tests/test_oracle_cpu.py::
    # test_reference_bound_on_config_1_is_decided_by_rounding_in_the_oracle_itself), because the last trust-region iterates
    # accept / resize on rho = ared/pred with |ared| worth 3-4 ulps of mx (printed below: first differing decision).  Which
    # side a device variant ends on has flipped with unrelated kernel edits (round 2: 6.8e-8 / 3.15e-7; round 3: 4.2e-8 ...
    # 7.8e-8 on all four mid-round, 4.2e-8 / 7.8e-8 / 2.8e-7 / 2.8e-7 after the Cauchy kernels were rewritten).  ONE rule for every variant, with the band computed here: within a factor 2 of the largest value the
    # oracle's own re-associations produce; whether the reference's 1e-7 is met is printed, not fitted.
    from _util import assert_rounding_dominated, first_decision_difference, sphere_oracle_band
    log_ref = []
    R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u, max_outer_iter=100, max_inner_iter=250, log=log_ref)
    diff = first_decision_difference(log_ref, log)
    band = sphere_oracle_band()
    with capsys.disabled():
        if diff is not None:
            print("    first driver decision that differs from the oracle's: log entry %d of %d: %s" % (diff[0], len(log_ref), diff[3]))
        print("    opt_measure %.3e: %s the reference's 1e-7; oracle band under re-association %.2e .. %.2e; bound 2 x %.2e, used %.0f %%"
              % (opt_measure, "meets" if opt_measure < 1e-7 else "MISSES", min(band.values()), max(band.values()), max(band.values()),
                 100.0 * opt_measure / (2.0 * max(band.values()))))
    if diff is not None:
        assert_rounding_dominated(diff)
    assert opt_measure < 2.0 * max(band.values())


@pytest.mark.parametrize("fused", [0, 1, 2])
@pytest.mark.parametrize("ops_cls", [HipOps, HipOpsDeviceMinor, HipOpsDeviceAll], ids=["pcg_abi", "minor_iterate_abi", "cauchy_abi"])
def test_sphere_regression_shadow_solve_every_variant(bh, capsys, ops_cls, fused):
    """config 1 (n = 3, d = 4, q = 1, mA = 1) as a SHADOW solve — device and oracle evaluated on identical operands
    at every hot-path call — for every ops variant x every CG iteration shape.  On n = 3 there is no room for a bug to hide:
      * H*v, H*s+g, vthv agree to 1e-12 of their result, projections to 1e-12 of their operand;
      * every projected_cg / minor iterate has the oracle's exit status and iteration count, every Cauchy search the oracle's
        final active set — no exception;
      * w and the Cauchy step are cancelling computations near the solution (measured here: the ORACLE's own w moves by up to
        2e-6, its Cauchy step by up to 6e-2, when g is perturbed by one unit in the last place — at n = 3; and by 12 % when the same
        projector is evaluated in its reduced form instead of the augmented one, tests/manual/sphere_cauchy_event_probe.py): they
        must lie within 8 x the spread of the oracle family's evaluations (16 perturbed ones, for the Cauchy step also in the
        reduced form of the projector; or 1e-12), the used fraction is printed."""
    bh.set_option("cg_fused", fused)
    try:
        sh = ShadowOps(ops_cls(bh), relnorm_tol=1e-12, sens_samples=16)
        xs, ys = R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u, max_outer_iter=100, max_inner_iter=250, ops=sh)
    finally:
        bh.set_option("cg_fused", 1)
    from _util import sphere_opt_measure
    worst_ratio, worst_ev = 0.0, None
    for e in sh.events:
        assert e["op"] != "projection", e
        if e["op"] in ("projected_cg", "minor_iterate"):
            assert (e["status_dev"], e["iters_dev"]) == (e["status_cpu"], e["iters_cpu"]), e
        if e["op"] == "cauchy_step":
            assert e["fix_dev"] == e["fix_cpu"], {k: v for k, v in e.items() if k != "operands"}
        ratio = e["rel"] / max(8.0 * e["oracle_sensitivity"], 1e-12)
        if ratio > worst_ratio:
            worst_ratio, worst_ev = ratio, {k: v for k, v in e.items() if k not in ("operands", "ties")}
    with capsys.disabled():
        print("[sphere shadow, %s, cg_fused=%d] %d minor iterates, opt_measure %.2e; worst deviation per operator %s; %d calls above 1e-12, "
              "worst uses %.0f %% of 8 x the oracle's own spread: %s"
              % (ops_cls.__name__, fused, sh.minor, sphere_opt_measure(xs, ys), {k: float("%.1e" % v) for k, v in sh.worst.items()},
                 len(sh.events), 100.0 * worst_ratio, worst_ev))
    for op in ("hmul", "hmul_add", "vthv", "projection"):
        assert sh.worst.get(op, 0.0) <= 1e-12, (op, sh.worst[op])
    assert worst_ratio <= 1.0, worst_ev


@pytest.mark.parametrize("fused", [0, 2])
def test_sphere_regression_with_the_other_iteration_shapes(bh, capsys, fused):
    """The same solve with the seven-kernel (cg_fused = 0: pHp = dot(p, H*p) exactly as the reference forms it) and the four-kernel
    (cg_fused = 2) iteration for linear equalities; the default three-kernel shape runs in the test above.  The shapes round pHp
    differently, so each leaves the oracle's trajectory at its own noise rho; the same rule as above applies to all of them."""
    from _util import assert_rounding_dominated, first_decision_difference, sphere_opt_measure, sphere_oracle_band
    bh.set_option("cg_fused", fused)
    try:
        ops, log, log_ref = HipOps(bh), [], []
        kw = dict(max_outer_iter=100, max_inner_iter=250)
        xs, ys = R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u, ops=ops, log=log, **kw)
        R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u, log=log_ref, **kw)
        opt_measure, band = sphere_opt_measure(xs, ys), sphere_oracle_band()
        diff = first_decision_difference(log_ref, log)
        with capsys.disabled():
            print("[sphere regression, HipOps, cg_fused=%d] opt_measure = %.3e (%s the reference's 1e-7; %.0f %% of 2 x the oracle band's %.2e); "
                  "first differing decision: %s" % (fused, opt_measure, "meets" if opt_measure < 1e-7 else "MISSES",
                                                    100.0 * opt_measure / (2.0 * max(band.values())), max(band.values()), None if diff is None else (diff[0], diff[3])))
        assert np.linalg.norm(sp.c(xs)) < R.SQRT_EPS and R.is_feasible(xs, sp.A, sp.x_l, sp.x_u, sp.b)
        if diff is not None:
            assert_rounding_dominated(diff)
        assert opt_measure < 2.0 * max(band.values())
    finally:
        bh.set_option("cg_fused", 1)


@pytest.mark.parametrize("comm", ["rccl", "ipc", "both"])
def test_one_rank_communicator_paths(comm, capsys):
    """Both transports of the all-reduce with a 1-rank communicator in a child process (BH_FORCE_COMM=1) — RCCL: dlopen,
    ncclGetUniqueId, ncclCommInitRank, ncclAllReduce on the library stream; peer buffers: inbox, hipIpc-free self exchange
    through the fused reduce+exchange kernel — results must equal the communicator-free run bit for bit; prints what one
    all-reduce of an n-vector costs on each path (bh_time_kernel kinds 7 / 8; quoted in docs/design_history_r1_r2.md §6)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r)
import benlsip_jl_amd as bh
bh.init(0)
J = np.random.default_rng(0).standard_normal((700, 300))
v = np.random.default_rng(1).standard_normal(300)
r = np.random.default_rng(2).standard_normal(700)
H0 = bh.AlHessian(J, None, 1.0)
a0, g0, s0, q0 = H0 * v, H0.jtv(J @ v), bh.vthv(H0, v), bh.resid_sqnorm(r)
assert abs(q0 - r @ r) <= 1e-13 * (r @ r)
try:
    bh.init_distributed(0, 1, lambda b: b)
    raise SystemExit("bh_comm_init must refuse while a bh_hess is alive")
except bh.BenlsipHipError as e:
    assert e.code == bh._lib.BH_ERR_PRECONDITION
H0.close()
bh.init_distributed(0, 1, lambda b: b)
rk, n = bh._lib.C.c_int32(), bh._lib.C.c_int32()
bh._lib.lib().bh_comm_info(bh._lib.C.byref(rk), bh._lib.C.byref(n))
assert (rk.value, n.value) == (0, 1)
H = bh.AlHessian(J, None, 1.0)
cons = bh.MixedConstraints(np.zeros((0, 300)))
paths = {"rccl": [0], "ipc": [1], "both": [0, 1]}[os.environ["BH_COMM"]]
for path in paths:
    bh.set_option("comm_path", path)
    a, g, s, q = H * v, H.jtv(J @ v), bh.vthv(H, v), bh.resid_sqnorm(r)
    assert np.array_equal(a, a0) and np.array_equal(g, g0) and s == s0 and q == q0, path
    n0 = H.stats()["n_allreduce"]
    w, st, info = bh.projected_cg(g, H, np.full(300, -np.inf), np.full(300, np.inf), cons, 0.1, full_output=True)
    assert H.stats()["n_allreduce"] - n0 >= info["n_hmul"]
    Hb = bh.AlHessian.synthetic(4096, 4096, seed=1, mu=10.0)       # n = 4096: the 32 KiB message of the BASELINE configs
    print("COST path=%%d allreduce_us=%%.2f reduce_plus_allreduce_us=%%.2f" %% (path, 1e3 * Hb.time_kernel(7, 200), 1e3 * Hb.time_kernel(8, 200)))
    Hb.close()
if os.environ["BH_COMM"] == "rccl":
    try:
        bh.set_option("comm_path", 1)
        raise SystemExit("comm_path = 1 must be refused without the peer-buffer communicator")
    except bh.BenlsipHipError as e:
        assert e.code == bh._lib.BH_ERR_PRECONDITION
print("OK")
assert bh._lib.lib().bh_comm_destroy() == bh._lib.BH_ERR_PRECONDITION      # refused while handles are alive
H.close()
assert bh._lib.lib().bh_comm_destroy() == 0
Hc = bh.AlHessian.synthetic(4096, 4096, seed=1, mu=10.0)
print("COST path=none reduce_only_us=%%.2f" %% (1e3 * Hc.time_kernel(8, 200)))
""" % root
    env = dict(os.environ, BH_FORCE_COMM="1", BH_COMM=comm)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2000:])
    lines = out.stdout.strip().splitlines()
    assert "OK" in lines
    with capsys.disabled():
        for ln in lines:
            if ln.startswith("COST"):
                print("[one-rank communicator, BH_COMM=%s] %s" % (comm, ln[5:]))


def test_shutdown_and_reinit_in_one_process():
    """bh_shutdown releases the library's device state (workspace, pinned arena, LDS ceilings); a second bh_init in the same
    process starts clean and reproduces the first life's results bit for bit (wide J, a blocked factor and a CG run, so that
    every lazily raised per-kernel limit is raised again)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
import benlsip_jl_amd as bh
rng = np.random.default_rng(3)
J = rng.standard_normal((40, 9000)) / 7.0
A = rng.standard_normal((130, 400))
Js = rng.standard_normal((900, 400)) / 30.0
g = rng.standard_normal(400)
def life():
    bh.init(0)
    H = bh.AlHessian(J, None, 1.0)
    hv = H * np.ones(9000)
    Hs = bh.AlHessian(Js, None, 1.0)
    cons = bh.MixedConstraints(A, None, None)
    big = np.full(400, 5.0)
    w, st, info = bh.projected_cg(g, Hs, -big, big, cons, 1e-2, full_output=True)
    pv = bh.projection(cons, g)
    H.close(); Hs.close(); cons.close()
    assert bh._lib.lib().bh_shutdown() == 0
    return hv, w, int(st), info["iters"], pv
a = life()
try:
    bh.AlHessian(Js, None, 1.0)
    raise SystemExit("a call after bh_shutdown must fail")
except bh.BenlsipHipError as e:
    assert e.code == bh._lib.BH_ERR_NOT_INIT
b = life()
assert all(np.array_equal(x, y) for x, y in zip(a, b))
print("OK", a[2], a[3])
""" % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1].startswith("OK")


def test_workspace_reuse_across_sizes_and_odd_n(bh, cg_fused):
    """The CG workspace is shared by all calls: a larger problem must not leak into the padding of a later smaller / odd-n
    one (box and general path)."""
    rng = np.random.default_rng(42)
    for n, mA in [(96, 0), (33, 0), (95, 3), (17, 0), (64, 2), (5, 1), (3, 0)]:
        d = 3 * n
        J = rng.standard_normal((d, n)) / np.sqrt(d)
        A = rng.standard_normal((mA, n))
        fix = np.zeros(n, dtype=bool)
        fix[rng.choice(n, max(n // 6, 0), replace=False)] = True
        cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix if fix.any() else None, l=-np.ones(n), u=np.ones(n))
        g = rng.standard_normal(n)
        w_l, w_u = R.build_step_bounds(np.where(fix, 1.0, 0.0), cons_o, 0.5)
        Ho = R.AlHessian(J, np.zeros((0, n)), 1.0)
        w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.1)
        H = bh.AlHessian(J, None, 1.0)
        cons = bh.MixedConstraints(A, cons_o.chol_L, fix)
        w, status, info = bh.projected_cg(g, H, w_l, w_u, cons, 0.1, full_output=True)
        assert int(status) == int(s_ref) and info["iters"] == it_ref, (n, mA)
        assert_w_close(w, w_ref, w_tolerance(g, Ho, w_l, w_u, cons_o, 0.1, w_ref), "projected_cg: w vs oracle (tolerance max(1e-9, 20 x oracle sensitivity))")


# ----------------------------------------------------------------------------- callers on the device (a9, a10, f-2)
@pytest.mark.parametrize("d,n,q,mA,nfix,seed", [(60, 24, 1, 0, 5, 1), (200, 65, 0, 3, 9, 2), (1500, 700, 2, 8, 90, 3), (300, 128, 0, 0, 0, 4)])
def test_minor_iterate_linesearch_gradient_parity(bh, d, n, q, mA, nfix, seed):
    """minor_iterate (:649-675), linesearch (:766-791), g = J'r + C'y_bar (:45), H*s+g (:412) against the oracle."""
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    C = rng.standard_normal((q, n))
    A = rng.standard_normal((mA, n))
    L0 = R.chol_lower(A @ A.T)
    fix = np.zeros(n, dtype=bool)
    fix[rng.choice(n, nfix, replace=False)] = True
    xlow, xupp = -np.ones(n), np.ones(n)
    cons_o = R.make_mixed_constraints(A, L0, fix if nfix else None, l=xlow, u=xupp)
    x = np.clip(0.4 * rng.standard_normal(n), -0.9, 0.9)
    x[fix] = np.where(rng.random(nfix) < 0.5, -1.0, 1.0)
    s = 0.01 * rng.standard_normal(n)
    s[fix] = 0.0
    Ho = R.AlHessian(J, C, 10.0)
    H = bh.AlHessian(J, C, 10.0)
    cons = bh.MixedConstraints(A, cons_o.chol_L, fix, l=xlow, u=xupp)
    rx, ybar = rng.standard_normal(d), rng.standard_normal(q)
    g = J.T @ rx + C.T @ ybar
    assert np.linalg.norm(bh.gradient(H, rx, ybar) - g) <= 1e-12 * np.linalg.norm(np.abs(J).T @ np.abs(rx) + np.abs(C).T @ np.abs(ybar))
    gm_ref = R.hmul(Ho, s) + g
    gm = bh.hmul_add(H, s, g)
    assert relnorm(gm, gm_ref) <= 1e-12
    delta = 0.1 * np.linalg.norm(g)
    # linesearch on a fixed direction
    w_l, w_u = R.build_step_bounds(x + s, cons_o, delta)
    wdir = R.projection(cons_o, -gm_ref)
    a_ref = R.linesearch(gm_ref, Ho, wdir, w_l, w_u, cons_o.fixvars)
    a = bh.linesearch(gm_ref, H, wdir, w_l, w_u, cons)
    assert a == pytest.approx(a_ref, rel=1e-12)
    # the whole minor iterate
    w_ref, st_ref = R.minor_iterate(x, s, gm_ref, Ho, cons_o, delta, 0.1)
    w, st, info = bh.minor_iterate(x, s, gm_ref, H, cons, delta, 0.1, full_output=True)
    assert int(st) == int(st_ref)
    wl2, wu2 = R.build_step_bounds(x + s, cons_o, delta)
    w_cg, s_cg, it_cg = R.projected_cg(gm_ref, Ho, wl2, wu2, cons_o, 0.1)
    assert info["iters"] == it_cg
    tol = w_tolerance(gm_ref, Ho, wl2, wu2, cons_o, 0.1, w_cg)
    assert_w_close(w, w_ref, 10 * tol, "minor_iterate: scaled w vs oracle (10 x the CG tolerance: alpha inherits w's sensitivity)")
    if int(st_ref) != int(R.CGStatus.negative_curvature):
        a_cg = R.linesearch(gm_ref, Ho, w_cg, wl2, wu2, cons_o.fixvars)
        # alpha = -g.w / w'Hw inherits the rounding sensitivity of w (tol, measured on the oracle itself)
        assert info["alpha"] == pytest.approx(a_cg, rel=max(1e-6, 1e3 * tol)), (info["alpha"], a_cg, tol, relnorm(w, w_ref))
    # default: w'Hw from the H*w the CG loop accumulated; option 0: the reference's explicit vthv(H, w) (:775).  Same value.
    bh._lib.lib().bh_set_option(b"ls_from_cg", 0)
    try:
        w2, st2, info2 = bh.minor_iterate(x, s, gm_ref, H, cons, delta, 0.1, full_output=True)
    finally:
        bh._lib.lib().bh_set_option(b"ls_from_cg", 1)
    assert int(st2) == int(st) and info2["iters"] == info["iters"]
    if int(st_ref) != int(R.CGStatus.negative_curvature):
        assert info2["alpha"] == pytest.approx(info["alpha"], rel=1e-10)
        assert relnorm(w2, w) <= 1e-10


# ----------------------------------------------------------------------------- Cauchy step on the device (f-3)
@pytest.fixture(params=[1, 0], ids=["factor_downdate", "gram_downdate_refactor"])
def chol_downdate(request, bh):
    """Per breakpoint: downdate of the Gram matrix + refactorisation (0, the default), or — for mA > 64 — rank-one downdates of
    chol(A_free A_free') with a from-scratch rebuild every 8th breakpoint (1)."""
    bh._lib.lib().bh_set_option(b"chol_downdate", request.param)
    yield request.param
    bh._lib.lib().bh_set_option(b"chol_downdate", 0)


@pytest.mark.parametrize("d,n,mA,nact,delta_scale,seed", [(80, 30, 0, 4, 0.5, 1), (300, 120, 3, 10, 1.0, 2), (500, 200, 0, 0, 5.0, 3),
                                                          (400, 96, 8, 6, 0.2, 4), (2000, 512, 4, 40, 2.0, 5), (60, 17, 1, 2, 1.0, 6),
                                                          (900, 300, 70, 12, 1.0, 7), (1500, 400, 64, 30, 3.0, 8)])
def test_cauchy_step_parity(bh, chol_downdate, d, n, mA, nact, delta_scale, seed):
    """cauchy_step (src/basic_tralcnlss.jl:574-639) incl. next_breakpoint and the active-set growth, against the oracle:
    same breakpoints, same final active set, same step."""
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    A = rng.standard_normal((mA, n))
    L0 = R.chol_lower(A @ A.T)
    xlow, xupp = -np.ones(n), np.ones(n)
    x = np.clip(0.5 * rng.standard_normal(n), -0.95, 0.95)
    act = rng.choice(n, nact, replace=False)
    x[act] = np.where(rng.random(nact) < 0.5, -1.0, 1.0)
    g = rng.standard_normal(n)
    delta = delta_scale * 0.1 * np.linalg.norm(g)
    Ho = R.AlHessian(J, np.zeros((0, n)), 10.0)
    cons_o = R.make_mixed_constraints(A, L0, l=xlow, u=xupp)
    n_hmul_ref = [0]

    class Ops(R.NumpyOps):
        def hmul(self, H, v):
            n_hmul_ref[0] += 1
            return R.hmul(H, v)
    s_ref = R.cauchy_step(x, g, Ho, L0, cons_o, delta, Ops())
    H = bh.AlHessian(J, None, 10.0)
    cons = bh.MixedConstraints(A, L0, l=xlow, u=xupp)
    s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
    assert np.array_equal(cons.fixvars, cons_o.fixvars), (np.flatnonzero(cons.fixvars), np.flatnonzero(cons_o.fixvars))
    assert info["n_hmul"] == n_hmul_ref[0]
    assert info["n_breakpoints"] == n_hmul_ref[0] - 1 or info["n_breakpoints"] == R.nb_fix(cons_o) - nact
    assert np.linalg.norm(s - s_ref) <= 1e-9 * max(np.linalg.norm(s_ref), 1e-300), relnorm(s, s_ref)
    # the step stays inside the trust region / bounds box and in null(A)
    assert np.all(x + s <= xupp + 1e-12) and np.all(x + s >= xlow - 1e-12) and np.max(np.abs(s)) <= delta * (1 + 1e-12)
    if mA:
        assert np.linalg.norm(A @ s) <= 1e-10 * np.linalg.norm(A) * max(np.linalg.norm(s), 1e-300)
    # the handle now holds the final active set: a projection agrees with the oracle's
    r = rng.standard_normal(n)
    assert np.linalg.norm(bh.projection(cons, r) - R.projection(cons_o, r)) <= 1e-10 * np.linalg.norm(r)


@pytest.mark.parametrize("d,n,q,nact,delta_scale,seed", [(90, 33, 2, 3, 1.0, 11), (700, 257, 1, 20, 3.0, 12), (3000, 1024, 0, 100, 10.0, 13), (257, 4100, 3, 50, 1.0, 14),
                                                         (5, 3, 1, 0, 5.0, 15)])
def test_cauchy_step_in_the_row_space_of_j(bh, capsys, d, n, q, nact, delta_scale, seed):
    """Box constraints: the image-space search (option cauchy_image, default) maintains J d and J s_c by one-column updates and
    forms d'Hd = ||J d||^2_W, s'Hd = (J s).(J d)_W from them — no sweep over J per breakpoint.  Against the oracle (which sweeps, as
    the reference does, src/basic_tralcnlss.jl:609,:633): same breakpoints, same final active set, same step to 1e-9 — also with
    nonlinear-constraint rows (q > 0, weight mu) — and the same against the device's own sweeping form (cauchy_image = 0); the
    handle's H*p counter shows that no sweep ran."""
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    C = rng.standard_normal((q, n))
    xlow, xupp = -np.ones(n), np.ones(n)
    x = np.clip(0.5 * rng.standard_normal(n), -0.95, 0.95)
    act = rng.choice(n, nact, replace=False)
    x[act] = np.where(rng.random(nact) < 0.5, -1.0, 1.0)
    g = rng.standard_normal(n)
    delta = delta_scale * 0.1 * np.linalg.norm(g)
    Z = np.zeros((0, n))
    L0 = R.chol_lower(Z @ Z.T)
    Ho = R.AlHessian(J, C, 2.5)
    cons_o = R.make_mixed_constraints(Z, L0, l=xlow, u=xupp)
    calls = [0]

    class Ops(R.NumpyOps):
        def hmul(self, H, v):
            calls[0] += 1
            return R.hmul(H, v)
    s_ref = R.cauchy_step(x, g, Ho, L0, cons_o, delta, Ops())
    H = bh.AlHessian(J, C, 2.5)
    out = {}
    for mode in (2, 1, 0):             # 2: row space, one kernel per breakpoint (default); 1: row space, two kernels; 0: one H*d sweep per breakpoint
        bh.set_option("cauchy_image", 1 if mode else 0)
        bh.set_option("cauchy_fused", 1 if mode == 2 else 0)
        try:
            cons = bh.MixedConstraints(Z, None, l=xlow, u=xupp)
            n0 = H.stats()["n_hmul"]
            s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
            swept = H.stats()["n_hmul"] - n0
        finally:
            bh.set_option("cauchy_image", 1)
            bh.set_option("cauchy_fused", 1)
        assert np.array_equal(cons.fixvars, cons_o.fixvars), (mode, np.flatnonzero(cons.fixvars), np.flatnonzero(cons_o.fixvars))
        assert info["n_hmul"] == calls[0], (mode, info, calls[0])           # passes = the oracle's H*d products
        assert swept == (0 if mode else calls[0]), (mode, swept, calls[0])
        rel = relnorm(s, s_ref)
        note_tol("cauchy_step (box): step vs oracle, 1e-9", rel, 1e-9, "form=%d d=%d n=%d q=%d, %d breakpoints" % (mode, d, n, q, info["n_breakpoints"]))
        assert rel <= 1e-9, (mode, rel)
        assert np.all(x + s <= xupp + 1e-12) and np.all(x + s >= xlow - 1e-12) and np.max(np.abs(s)) <= delta * (1 + 1e-12)
        out[mode] = s
        cons.close()
    assert relnorm(out[1], out[0]) <= 1e-9 and relnorm(out[2], out[0]) <= 1e-9
    H.close()


@pytest.mark.parametrize("d,n,mA,q", [(500, 200, 5, 0), (600, 257, 33, 2), (300, 270, 64, 0), (65, 70, 17, 1)])
def test_cauchy_row_space_setup_on_the_matrix_cores(bh, d, n, mA, q):
    """B = J D A' of the row-space Cauchy search with linear equalities: one sweep over J on the fp64 matrix cores
    (image_b_mfma_kernel, option cauchy_gemm) against mA J v sweeps over the masked rows of A (cauchy_gemm = 0) and against the
    oracle: same final active set, steps to 1e-9; odd sizes on every side (rows not a multiple of 16, mA not a multiple of 16,
    n just above a padding boundary, nonlinear-constraint rows)."""
    rng = np.random.default_rng(d + n + mA)
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    C = rng.standard_normal((q, n))
    A = rng.standard_normal((mA, n))
    L0 = R.chol_lower(A @ A.T)
    xlow, xupp = -np.ones(n), np.ones(n)
    x = np.clip(0.5 * rng.standard_normal(n), -0.95, 0.95)
    x[rng.choice(n, n // 20, replace=False)] = 1.0
    g = 200.0 * rng.standard_normal(n)
    delta = 0.5 * np.linalg.norm(g)
    Ho = R.AlHessian(J, C, 4.0)
    cons_o = R.make_mixed_constraints(A, L0, l=xlow, u=xupp)
    s_ref = R.cauchy_step(x, g, Ho, L0, cons_o, delta, R.NumpyOps())
    H = bh.AlHessian(J, C, 4.0)
    out = {}
    for gemm in (1, 0):
        bh.set_option("cauchy_gemm", gemm)
        try:
            cons = bh.MixedConstraints(A, None, l=xlow, u=xupp)
            s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
        finally:
            bh.set_option("cauchy_gemm", 1)
        assert np.array_equal(cons.fixvars, cons_o.fixvars), gemm
        note_tol("cauchy_step (equalities, row-space form): step vs oracle, 1e-9", relnorm(s, s_ref), 1e-9, "gemm=%d d=%d n=%d mA=%d, %d breakpoints" % (gemm, d, n, mA, info["n_breakpoints"]))
        assert relnorm(s, s_ref) <= 1e-9, (gemm, relnorm(s, s_ref))
        out[gemm] = s
        cons.close()
    assert relnorm(out[1], out[0]) <= 1e-9
    H.close()


def test_cauchy_step_row_space_form_chosen_from_history(bh, capsys):
    """Above cauchy_image_max_ma linear equalities (default 64; 16 here) and up to 64 the row-space form of the Cauchy search is used
    when the previous search on the same handle took more than 4 (1 + mA) passes: first call sweeping (one H*d per breakpoint),
    second call in the row space (no H*d) — both against the oracle: same active set, step to 1e-9."""
    rng = np.random.default_rng(77)
    d, n, mA = 900, 400, 20
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    A = rng.standard_normal((mA, n))
    L0 = R.chol_lower(A @ A.T)
    xlow, xupp = -np.ones(n), np.ones(n)
    x = np.clip(0.5 * rng.standard_normal(n), -0.95, 0.95)
    g = 1000.0 * rng.standard_normal(n)                       # steep: hundreds of breakpoints
    delta = 0.5 * np.linalg.norm(g)
    Ho = R.AlHessian(J, np.zeros((0, n)), 10.0)
    cons_o = R.make_mixed_constraints(A, L0, l=xlow, u=xupp)
    s_ref = R.cauchy_step(x, g, Ho, L0, cons_o, delta, R.NumpyOps())
    H = bh.AlHessian(J, None, 10.0)
    cons = bh.MixedConstraints(A, None, l=xlow, u=xupp)
    swept = []
    bh.set_option("cauchy_image_max_ma", 16)        # (default 64: the form would be unconditional at mA = 20)
    try:
        for call in range(2):
            n0 = H.stats()["n_hmul"]
            s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
            swept.append(H.stats()["n_hmul"] - n0)
            assert np.array_equal(cons.fixvars, cons_o.fixvars), call
            assert relnorm(s, s_ref) <= 1e-9, (call, relnorm(s, s_ref))
    finally:
        bh.set_option("cauchy_image_max_ma", 64)
    with capsys.disabled():
        print("[Cauchy search, mA = 20] %d passes; H*d sweeps: first call %d, second call %d (row-space form chosen from history)"
              % (info["n_hmul"], swept[0], swept[1]))
    assert info["n_hmul"] > 4 * (1 + mA) and swept[0] == info["n_hmul"] and swept[1] == 0
    H.close(); cons.close()


@pytest.mark.parametrize("image", [1, 0], ids=["row_space_form", "sweep_per_breakpoint"])
def test_cauchy_step_on_the_pinned_multimodal_operands(bh, capsys, image):
    """the worst Cauchy discrepancies of the device shadow solve, with their operands committed
    (tests/golden/cauchy_events.json).  tests/test_oracle_cpu.py shows that on these operands the ORACLE ALONE lands on several
    final active sets under 1-ulp perturbations of g, and a reduced-form CPU restatement on yet others — the outcome is not
    determined in fp64 (trust region 1e-11 .. 1e-14, ||g||/||P(-g)|| = 1e7 .. 1e8).  What CAN be demanded of the device there,
    and can fail:
      * the step is feasible: inside the trust region and the bounds, in null(A) as closely as the CPU restatements' own steps;
      * its model value phi(s) = g.s + s'Hs/2 — what the search minimises — lies inside the range the CPU outcomes span (5 %
        margin), and so does the size of its final active set (+-2);
      * where CPU outcomes have exactly the device's final active set, the device's step is as close to them as they are to each
        other (or 1e-6); where every CPU outcome has the SAME active set (the two unimodal events), so has the device.
    Both forms of the device search: in the row space of J (default for mA <= 16) and with one H*d sweep per breakpoint.
    (History: this test found the one-wave factor-downdate kernel for mA <= 64 walking to the corner of the trust region on event
    511 — 46 active bounds, |As|/|A||s| = 4e-6; that kernel is gone, chol_downdate = 1 now applies to mA > 64 only.)"""
    from _util import check_cauchy_against_cpu_family, load_cauchy_events
    P, events = load_cauchy_events()
    chol_downdate = 0
    for e in events:
        H = bh.AlHessian(e["J"], e["C"], e["mu"])
        cons = bh.MixedConstraints(P.A, None, None, l=P.x_l, u=P.x_u)
        bh.set_option("cauchy_image", image)
        try:
            s, info = bh.cauchy_step(e["x"], e["g"], H, cons, e["delta"], full_output=True)
        finally:
            bh.set_option("cauchy_image", 1)
        key_dev = tuple(np.flatnonzero(cons.fixvars))
        info = check_cauchy_against_cpu_family(P, e, s, key_dev, samples=32, seed=e["minor"])
        with capsys.disabled():
            print("[pinned Cauchy event, minor iterate %d, cauchy_image=%d] device: %d active bounds, phi %.4e, |As|/|A||s| %.1e; CPU outcomes: sizes %d..%d "
                  "(%d distinct sets), phi %.4e .. %.4e, |As|/|A||s| <= %.1e; same set on the CPU: %s; nearest CPU outcome at %.1e"
                  % (e["minor"], image, info["size_dev"], info["phi_dev"], info["as_rel_dev"], info["size_min"], info["size_max"], info["n_sets"],
                     info["phi_min"], info["phi_max"], info["as_rel_cpu"],
                     "no" if not info["same_set"] else "yes, steps %.1e apart" % info["same_set_dist"], info["nearest"]))
        H.close(); cons.close()


# ----------------------------------------------------------------------------- wide J (n > 8192)
@pytest.mark.parametrize("d,n,q", [(70, 8200, 1), (40, 10001, 0), (29, 16384, 1), (31, 16385, 0), (33, 20000, 2)])
def test_wide_jacobian_column_panels(bh, d, n, q):
    """8192 < n <= 16384: one row per step, the fused kernel parks its slice of v in LDS (still a single read of J);
    n > 16384: J is swept in 4096-column panels (two-pass H*p).  CG runs on the generic n-vector kernels in both."""
    rng = np.random.default_rng(n)
    J, C, mu = rng.standard_normal((d, n)), rng.standard_normal((q, n)), 0.5
    v, u = rng.standard_normal(n), rng.standard_normal(d)
    H, Ho = bh.AlHessian(J, C, mu), R.AlHessian(J, C, mu)
    assert np.linalg.norm(H.jv(v) - J @ v) <= TOL1 * matvec_scale(J, v)
    assert np.linalg.norm(H.jtv(u) - J.T @ u) <= TOL1 * matvec_scale(J.T, u)
    scale = np.linalg.norm(np.abs(J).T @ (np.abs(J) @ np.abs(v)) + mu * np.abs(C).T @ (np.abs(C) @ np.abs(v)))
    assert np.linalg.norm(H * v - R.hmul(Ho, v)) <= TOL1 * scale
    assert bh.vthv(H, v) == pytest.approx(R.vthv(Ho, v), rel=1e-12)
    # a box-constrained CG on it (generic n-vector kernels); all but ~n/1000 (< d) variables fixed so that the rank-deficient
    # H (d << n) restricted to the free variables is positive definite and the oracle needs few iterations
    fix = np.ones(n, dtype=bool)
    fix[::1000] = False
    A = np.zeros((0, n))
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix, l=-np.ones(n), u=np.ones(n))
    g = rng.standard_normal(n)
    w_l, w_u = R.build_step_bounds(np.where(fix, 1.0, 0.0), cons_o, 0.5)
    w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.1)
    cons = bh.MixedConstraints(A, None, fix)
    w, status, info = bh.projected_cg(g, H, w_l, w_u, cons, 0.1, full_output=True)
    assert int(status) == int(s_ref) and info["iters"] == it_ref
    assert_w_close(w, w_ref, w_tolerance(g, Ho, w_l, w_u, cons_o, 0.1, w_ref), "projected_cg: w vs oracle (tolerance max(1e-9, 20 x oracle sensitivity))")


def test_config4_sized_shard_on_one_gpu(bh):
    """A 16 GiB image (d = 524288, n = 4096: all of BASELINE config 4 on one GPU) — 64-bit indexing, size-independent
    properties and rows checked against the host generator."""
    d, n = 524288, 4096
    H = bh.AlHessian.synthetic(d, n, seed=3, mu=10.0)
    rng = np.random.default_rng(1)
    v, w = rng.standard_normal(n), rng.standard_normal(n)
    Jv = H.jv(v)
    for i in (0, 65535, 65536, 300000, d - 1):
        Ji = R.synthetic_J(1, n, seed=3, row0=int(i), d_total=d)
        assert abs(Jv[i] - float(Ji[0] @ v)) <= 1e-12 * float(np.abs(Ji[0]) @ np.abs(v))
    Hv = H * v
    assert relnorm(H.jtv(Jv), Hv) <= 1e-12
    assert abs(v @ Hv - Jv @ Jv) <= 1e-12 * (Jv @ Jv)
    assert relnorm(H * (v + 2 * w), Hv + 2 * (H * w)) <= 1e-12
    H.close()


@pytest.mark.parametrize("mA,n,nfix", [(1, 40, 3), (5, 33, 0), (16, 100, 20), (17, 257, 60), (64, 1000, 300), (100, 700, 150)])
def test_gram_on_matrix_cores_matches_valu_path(bh, mA, n, nfix):
    """A_free A_free' via v_mfma_f64_16x16x4_f64 (default) against the one-wave-per-entry VALU kernel and the oracle:
    the projections built from both factors agree with the reference projector."""
    rng = np.random.default_rng(mA * 1000 + n)
    A = rng.standard_normal((mA, n))
    fix = np.zeros(n, dtype=bool)
    fix[rng.choice(n, nfix, replace=False)] = True
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix if nfix else None)
    r = rng.standard_normal(n)
    v_ref = R.projection(cons_o, r)
    lib = bh._lib.lib()
    out = {}
    for flag in (2, 0):          # 2 = matrix cores for every mA, 0 = VALU kernel (default 1 switches at mA > 96)
        lib.bh_set_option(b"gram_mfma", flag)
        cons = bh.MixedConstraints(A, None, fix)
        out[1 if flag else 0] = bh.projection(cons, r)
    lib.bh_set_option(b"gram_mfma", 1)
    B = np.vstack([A, np.eye(n)[fix]])
    tol = max(1e-11, 200 * np.finfo(float).eps * np.linalg.cond(B @ B.T))
    assert np.linalg.norm(out[1] - v_ref) <= tol * np.linalg.norm(r)
    assert np.linalg.norm(out[0] - v_ref) <= tol * np.linalg.norm(r)
    assert np.linalg.norm(out[1] - out[0]) <= tol * np.linalg.norm(r)


def test_cauchy_step_vectors_longer_than_one_batch(bh):
    """n = 9001 (odd, more than the 4096 elements a workgroup of the decision code holds in registers: its batch loop and the strided
    tail of the s_c update run), box constraints, a gradient with 300 non-zero components so that the oracle's search stays short:
    every form of the device search against the oracle — breakpoints, active set, step to 1e-9."""
    rng = np.random.default_rng(16)
    d, n, q = 120, 9001, 2
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    C = rng.standard_normal((q, n))
    xlow, xupp = -np.ones(n), np.ones(n)
    x = np.clip(0.5 * rng.standard_normal(n), -0.95, 0.95)
    act = rng.choice(n, 60, replace=False)
    x[act] = np.where(rng.random(60) < 0.5, -1.0, 1.0)
    g = np.zeros(n)
    nz = rng.choice(n, 300, replace=False)
    g[nz] = rng.standard_normal(300)
    delta = 0.001 * np.linalg.norm(g)            # a tight trust region: 253 passes
    Z = np.zeros((0, n))
    L0 = R.chol_lower(Z @ Z.T)
    Ho = R.AlHessian(J, C, 2.5)
    cons_o = R.make_mixed_constraints(Z, L0, l=xlow, u=xupp)
    calls = [0]

    class Ops(R.NumpyOps):
        def hmul(self, H, v):
            calls[0] += 1
            return R.hmul(H, v)
    s_ref = R.cauchy_step(x, g, Ho, L0, cons_o, delta, Ops())
    assert calls[0] > 200
    H = bh.AlHessian(J, C, 2.5)
    for image, fused in ((1, 1), (1, 0), (0, 0)):
        bh.set_option("cauchy_image", image)
        bh.set_option("cauchy_fused", fused)
        try:
            cons = bh.MixedConstraints(Z, None, l=xlow, u=xupp)
            s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
        finally:
            bh.set_option("cauchy_image", 1)
            bh.set_option("cauchy_fused", 1)
        assert np.array_equal(cons.fixvars, cons_o.fixvars) and info["n_hmul"] == calls[0], (image, fused, info, calls[0])
        rel = relnorm(s, s_ref)
        note_tol("cauchy_step (box): step vs oracle, 1e-9", rel, 1e-9, "n=9001 image=%d fused=%d, %d breakpoints" % (image, fused, info["n_breakpoints"]))
        assert rel <= 1e-9, (image, fused, rel)
        cons.close()
    H.close()


def test_cauchy_search_forms_agree_at_config3_scale(bh, capsys):
    """The Cauchy search at BASELINE config-3 scale (d = 65536, n = 4096: 128 workgroups of the fused kernel, 512 tiles of the
    equality form — launch shapes the small parity cases never reach), where the oracle would take minutes: the device's own forms
    against each other.  Box constraints: one kernel per breakpoint / two kernels / one H*d sweep per breakpoint (the reference's
    arithmetic, src/basic_tralcnlss.jl:609,:633) must take the same breakpoints, end on the same active set and agree in the step to
    1e-9; the step is feasible and reduces the model.  With 64 equalities: row-space form against the sweeping one."""
    syn = bh.synthetic
    d, n = 65536, 4096
    H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
    x, x_l, x_u, fix = syn.box_vectors(n, fix_every=8)
    g = H.jtv(syn.residual_rows(0, d))
    delta = syn.initial_tr(g)
    Z = np.zeros((0, n))
    A64 = syn.splitmix_uniform(4, np.arange(64 * n)).reshape((64, n), order="F")
    for A, forms in ((Z, ((1, 1), (1, 0), (0, 0))), (A64, ((1, 0), (0, 0)))):
        out = []
        for image, fused in forms:
            bh.set_option("cauchy_image", image)
            bh.set_option("cauchy_fused", fused)
            try:
                cons = bh.MixedConstraints(A, None, None, l=x_l, u=x_u)
                s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
            finally:
                bh.set_option("cauchy_image", 1)
                bh.set_option("cauchy_fused", 1)
            out.append((s, np.asarray(cons.fixvars, dtype=bool).copy(), info["n_breakpoints"], info["n_hmul"]))
            cons.close()
        s0, f0, nb0, np0 = out[-1]                                    # the sweeping form
        assert nb0 > 1000
        for s, f, nb, npass in out[:-1]:
            assert (nb, npass) == (nb0, np0) and np.array_equal(f, f0), (A.shape[0], nb, nb0, int((f != f0).sum()))
            rel = relnorm(s, s0)
            note_tol("cauchy_step at config-3 scale: row-space forms vs the sweeping form, 1e-9", rel, 1e-9, "mA=%d, %d breakpoints" % (A.shape[0], nb))
            assert rel <= 1e-9, (A.shape[0], rel)
        assert np.all(x + s0 <= x_u + 1e-12) and np.all(x + s0 >= x_l - 1e-12) and np.max(np.abs(s0)) <= delta * (1 + 1e-12)
        if A.shape[0]:
            assert np.linalg.norm(A @ s0) <= 1e-9 * np.linalg.norm(np.abs(A) @ np.abs(s0))
        model = float(g @ s0 + 0.5 * (s0 @ (H * s0)))
        assert model < 0.0
        with capsys.disabled():
            print("[Cauchy search at config-3 scale, mA = %d] %d breakpoints, %d active bounds, model reduction %.6e; forms agree to %.1e"
                  % (A.shape[0], nb0, int(f0.sum()), model, max(relnorm(s, s0) for s, _, _, _ in out[:-1])))
    H.close()


def test_pcg_config3_full_size_against_oracle(bh):
    """BASELINE config 3 itself (d = 65536, n = 4096, box, p = 512; the bench.py workload): the device-generated J against the
    host generator (full 2 GiB image through J v and J' u) and projected_cg against the oracle at full size; then config 5
    (the same J with 64 linear equalities) against the oracle, both projection forms."""
    d, n = 65536, 4096
    J = np.empty((d, n), order="F")
    for r0 in range(0, d, 8192):                      # host generator in row slabs (bounds the temporaries)
        J[r0:r0 + 8192] = R.synthetic_J(8192, n, seed=1, row0=r0, d_total=d)
    inst = R.synthetic_box_vectors(d, n, fix_every=8)
    H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
    rng = np.random.default_rng(0)
    v = rng.standard_normal(n)
    assert np.linalg.norm(H.jv(v) - J @ v) <= TOL1 * matvec_scale(J, v)
    g = J.T @ inst.r0
    assert np.linalg.norm(H.jtv(inst.r0) - g) <= TOL1 * matvec_scale(J.T, inst.r0)
    A = np.zeros((0, n))
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), inst.fixvars, l=inst.x_l, u=inst.x_u)
    Ho = R.AlHessian(J, np.zeros((0, n)), 10.0)
    w_l, w_u = R.build_step_bounds(inst.x, cons_o, R.initial_tr(g))
    cons = bh.MixedConstraints(A, None, inst.fixvars, l=inst.x_l, u=inst.x_u)
    for kappa2 in (0.1, 1e-3):
        tr = R.CGTrace()
        w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, kappa2, trace=tr)
        w, status, info = bh.projected_cg(g, H, w_l, w_u, cons, kappa2, trace_cap=64, full_output=True)
        assert int(status) == int(s_ref) and info["iters"] == it_ref and info["n_hmul"] == tr.n_hmul
        assert relnorm(w, w_ref) <= 1e-9, relnorm(w, w_ref)
        np.testing.assert_allclose(info["trace"], np.array(tr.rows), rtol=1e-9)
    # BASELINE config 5 at its full size on the same J: 64 linear equalities (A = u(4, .)) + the 512 active bounds, both
    # projection forms (reduced form on the device; the reference's augmented 576 x 576 factor from the oracle)
    mA = 64
    A5 = R.splitmix_uniform(4, np.arange(mA * n)).reshape((mA, n), order="F")
    cons5_o = R.make_mixed_constraints(A5, R.chol_lower(A5 @ A5.T), inst.fixvars, l=inst.x_l, u=inst.x_u)
    w_ref, s_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons5_o, 0.01)
    lib = bh._lib.lib()
    # reduced form with the three-kernel iteration (default), augmented form (seven kernels), reduced form with the four-kernel
    # iteration (triangular solves in their own launch), reduced form with seven kernels
    for form, fused in ((1, 1), (0, 1), (1, 2), (1, 0)):
        lib.bh_set_option(b"proj_form", form)
        lib.bh_set_option(b"cg_fused", fused)
        cons5 = bh.MixedConstraints(A5, cons5_o.chol_L, inst.fixvars, l=inst.x_l, u=inst.x_u)
        w, status, info = bh.projected_cg(g, H, w_l, w_u, cons5, 0.01, full_output=True)
        lib.bh_set_option(b"proj_form", 1)
        lib.bh_set_option(b"cg_fused", 1)
        assert int(status) == int(s_ref) and info["iters"] == it_ref, (form, int(status), int(s_ref), info["iters"], it_ref)
        print("[config 5 full size] proj_form=%d cg_fused=%d: %d iterations, |w - w_oracle|/|w_oracle| = %.2e (bound 1e-8)"
              % (form, fused, info["iters"], relnorm(w, w_ref)))
        assert relnorm(w, w_ref) <= 1e-8, (form, relnorm(w, w_ref))
        assert np.linalg.norm(A5 @ w) <= 1e-10 * np.linalg.norm(A5) * np.linalg.norm(w)
        # fixed components: exact zeros in the reduced form; rounding-level in the augmented form, as in the reference
        assert np.max(np.abs(w[inst.fixvars])) <= (0.0 if form == 1 else 1e-12 * np.linalg.norm(w))
        cons5.close()
    H.close()


def test_hessian_from_device_resident_jacobian(bh):
    """bh_hess_create_dev (f-4: a device-side jac_res hands over J in HBM; only the transpose runs): the products of the
    resulting image against the ORACLE's on the same J, C — and, as a layout check, bit-equal to the host-upload handle's."""
    rng = np.random.default_rng(9)
    for d, n, q, ld_extra in ((300, 130, 2, 0), (257, 65, 0, 7), (1, 40, 1, 0)):
        J, C = rng.standard_normal((d, n)), rng.standard_normal((q, n))
        ldJ = d + ld_extra                                         # a Jacobian that is a view into a taller device buffer
        Jbuf = np.full((ldJ, n), np.nan, order="F")
        Jbuf[:d] = J
        buf = bh.DeviceVector(ldJ * n, Jbuf.ravel(order="F"))
        H_dev = bh.AlHessian.from_device(buf.ptr, d, n, ldJ=ldJ, C=C, mu=3.0)
        Ho = R.AlHessian(J, C, 3.0)
        v, u = rng.standard_normal(n), rng.standard_normal(d)
        scale = np.linalg.norm(np.abs(J).T @ (np.abs(J) @ np.abs(v)) + 3.0 * np.abs(C).T @ (np.abs(C) @ np.abs(v)))
        assert np.linalg.norm(H_dev * v - R.hmul(Ho, v)) <= TOL1 * scale
        assert np.linalg.norm(H_dev.jv(v) - J @ v) <= TOL1 * matvec_scale(J, v)
        assert np.linalg.norm(H_dev.jtv(u) - J.T @ u) <= TOL1 * matvec_scale(J.T, u)
        assert bh.vthv(H_dev, v) == pytest.approx(R.vthv(Ho, v), rel=1e-12)
        H_host = bh.AlHessian(J, C, 3.0)
        assert np.array_equal(H_dev * v, H_host * v) and np.array_equal(H_dev.jtv(u), H_host.jtv(u))
        H_dev.close(); H_host.close(); buf.close()


@pytest.mark.parametrize("d,n,q,chunk_mb", [(3000, 700, 2, 1), (513, 4100, 0, 1), (40, 33, 1, 64), (70000, 96, 0, 8)])
def test_asynchronous_jacobian_ingest(bh, d, n, q, chunk_mb):
    """bh_hess_create_async / bh_hess_wait (f-4): J travels in pipelined column chunks (copy of chunk k+1 over the transpose
    of chunk k) while the caller keeps working; the image must give the ORACLE's products, also when the first use of the
    handle is what joins the upload, and a projected_cg on it must match the oracle's."""
    rng = np.random.default_rng(d + n)
    J, C = rng.standard_normal((d, n)) / np.sqrt(d), rng.standard_normal((q, n))
    bh.set_option("upload_chunk_mb", chunk_mb)
    try:
        H = bh.AlHessian.create_async(J, C, 2.0)
        Ho = R.AlHessian(J, C, 2.0)                               # host work while the upload runs
        v, u = rng.standard_normal(n), rng.standard_normal(d)
        ref_hv, ref_jv, ref_jtu = R.hmul(Ho, v), J @ v, J.T @ u
        H.wait()
        scale = np.linalg.norm(np.abs(J).T @ (np.abs(J) @ np.abs(v)) + 2.0 * np.abs(C).T @ (np.abs(C) @ np.abs(v)))
        assert np.linalg.norm(H * v - ref_hv) <= TOL1 * scale
        assert np.linalg.norm(H.jv(v) - ref_jv) <= TOL1 * matvec_scale(J, v)
        assert np.linalg.norm(H.jtv(u) - ref_jtu) <= TOL1 * matvec_scale(J.T, u)
        H2 = bh.AlHessian.create_async(J, C, 2.0)                 # no explicit wait: the first product joins the upload
        assert np.array_equal(H2 * v, H * v)
        H_sync = bh.AlHessian(J, C, 2.0)
        assert np.array_equal(H_sync * v, H * v) and np.array_equal(H_sync.jtv(u), H.jtv(u))
        H3 = bh.AlHessian.create_async(J, C, 2.0)                 # destroyed while the upload may still be in flight
        H3.close()
        if n <= 1024:
            A = np.zeros((0, n))
            fix = np.zeros(n, dtype=bool)
            fix[::7] = True
            cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix, l=-np.ones(n), u=np.ones(n))
            g = J.T @ rng.standard_normal(d)
            w_l, w_u = R.build_step_bounds(np.where(fix, 1.0, 0.0), cons_o, 0.5)
            w_ref, st_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.1)
            w, st, info = bh.projected_cg(g, H, w_l, w_u, bh.MixedConstraints(A, None, fix), 0.1, full_output=True)
            assert int(st) == int(st_ref) and info["iters"] == it_ref
            assert_w_close(w, w_ref, w_tolerance(g, Ho, w_l, w_u, cons_o, 0.1, w_ref), "projected_cg: w vs oracle (tolerance max(1e-9, 20 x oracle sensitivity))")
    finally:
        bh.set_option("upload_chunk_mb", 64)


@pytest.fixture(params=[0, 1], ids=["mailbox", "drained"])
def final_sync(request, bh):
    """Device-pointer entry points hand their few host-visible results over through the host-mapped mailbox and return without
    draining the stream (final_sync = 0, default), or drain it the way the host-pointer entry points do (1)."""
    bh.set_option("final_sync", request.param)
    yield request.param
    bh.set_option("final_sync", 0)


@pytest.mark.parametrize("d,n,mA,q", [(700, 130, 0, 0), (900, 257, 3, 1), (4096, 4096, 0, 0)])
def test_step_accumulate_takes_hw_from_the_cg_loop(bh, d, n, mA, q):
    """bh_step_accumulate_dev right behind the bh_minor_iterate_dev that produced w (src/basic_tralcnlss.jl:434-437): with
    option step_from_cg (default) g_minor += H*w uses the H*w the CG loop accumulated — no sweep over J (asserted on the handle's
    H*p counter) — and must equal the explicit H*(s + w) + g of the ORACLE to 1e-12 of its operands' scale, as the explicit device
    product (step_from_cg = 0) does; any other calling pattern (another w, a second call) falls back to the explicit product."""
    import ctypes as ct
    rng = np.random.default_rng(d + n)
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    C = rng.standard_normal((q, n))
    A = rng.standard_normal((mA, n))
    xl, xu = -np.ones(n), np.ones(n)
    x = np.clip(0.4 * rng.standard_normal(n), -0.9, 0.9)
    fix = np.zeros(n, dtype=bool)
    fix[rng.choice(n, n // 10, replace=False)] = True
    x[fix] = 1.0
    g = rng.standard_normal(n)
    s0 = 0.01 * rng.standard_normal(n)
    s0[fix] = 0.0
    if mA:
        s0 -= A.T @ np.linalg.solve(A @ A.T, A @ s0)
        s0[fix] = 0.0
    Ho = R.AlHessian(J, C, 3.0)
    gm0 = R.hmul(Ho, s0) + g
    delta = 0.3 * np.linalg.norm(g)
    H = bh.AlHessian(J, C, 3.0)
    cons = bh.MixedConstraints(A, None, fix, l=xl, u=xu)
    lib = bh._lib.lib()
    scale = np.linalg.norm(np.abs(J).T @ (np.abs(J) @ np.abs(s0))) + np.linalg.norm(g) + 1e-300
    out = {}
    for mode in (1, 0):
        bh.set_option("step_from_cg", mode)
        dv = {k: bh.DeviceVector(n, v) for k, v in (("x", x), ("s", s0), ("g", g), ("gm", gm0), ("xl", xl), ("xu", xu))}
        dv["w"] = bh.DeviceVector(n)
        st, it, nh, al = ct.c_int32(), ct.c_int32(), ct.c_int32(), ct.c_double()
        bh._lib.check(lib.bh_minor_iterate_dev(H.handle, cons.handle, dv["x"].ptr, dv["s"].ptr, dv["gm"].ptr, dv["xl"].ptr, dv["xu"].ptr, delta, 0.1,
                                               bh.operators.SQRT_EPS, 1e-10, dv["w"].ptr, ct.byref(st), ct.byref(it), ct.byref(nh), ct.byref(al)), "minor")
        w = dv["w"].download()
        n0 = H.stats()["n_hmul"]
        bh._lib.check(lib.bh_step_accumulate_dev(H.handle, dv["s"].ptr, dv["w"].ptr, dv["g"].ptr, dv["gm"].ptr), "step")
        swept = H.stats()["n_hmul"] - n0
        assert swept == (0 if mode == 1 else 1), (mode, swept)
        s1, gm1 = dv["s"].download(), dv["gm"].download()
        assert np.array_equal(s1, s0 + w)
        ref = R.hmul(Ho, s0 + w) + g
        out[mode] = gm1
        err = np.linalg.norm(gm1 - ref) / (scale + np.linalg.norm(np.abs(J).T @ (np.abs(J) @ np.abs(w))))
        note_tol("bh_step_accumulate_dev: g_minor vs the oracle's H*(s+w)+g (1e-12 of the operands' scale)", err, 1e-12, "step_from_cg=%d n=%d" % (mode, n))
        assert err <= 1e-12, (mode, err)
        # bh_model_reduction_dev on the same H, s, g (:458): under the option s'Hs comes from the g_minor just written, no J v sweep
        mr, j0 = ct.c_double(), H.stats()["n_jv"]
        bh._lib.check(lib.bh_model_reduction_dev(H.handle, dv["g"].ptr, dv["s"].ptr, ct.byref(mr)), "model reduction")
        assert H.stats()["n_jv"] - j0 == (0 if mode == 1 else 1)
        mr_ref = float(g @ s1) + 0.5 * R.vthv(Ho, s1)
        note_tol("bh_model_reduction_dev vs oracle (1e-10 of |g||s| + s'Hs)", abs(mr.value - mr_ref), 1e-10 * (np.linalg.norm(g) * np.linalg.norm(s1) + R.vthv(Ho, s1)),
                 "step_from_cg=%d n=%d" % (mode, n))
        assert abs(mr.value - mr_ref) <= 1e-10 * (np.linalg.norm(g) * np.linalg.norm(s1) + R.vthv(Ho, s1)), (mode, mr.value, mr_ref)
        # a second call with the same w is another step (s has changed): it must sweep J again and still be right
        n0 = H.stats()["n_hmul"]
        bh._lib.check(lib.bh_step_accumulate_dev(H.handle, dv["s"].ptr, dv["w"].ptr, dv["g"].ptr, dv["gm"].ptr), "step")
        assert H.stats()["n_hmul"] - n0 == 1
        ref2 = R.hmul(Ho, s0 + 2 * w) + g
        assert np.linalg.norm(dv["gm"].download() - ref2) <= 1e-12 * (scale + 2 * np.linalg.norm(np.abs(J).T @ (np.abs(J) @ np.abs(w))))
    bh.set_option("step_from_cg", 0)
    H.close(); cons.close()


@pytest.mark.parametrize("d,n,mA", [(4096, 512, 0), (1500, 300, 4)])
def test_inner_step_device_chain_against_oracle(bh, capsys, final_sync, d, n, mA):
    """One whole `inner_step` (src/basic_tralcnlss.jl:394-460: Cauchy search, then minor iterates) against the all-CPU
    oracle, (a) with every hot-path and "next"-row call on the device but host vectors in between (bh_cauchy_step,
    bh_minor_iterate, bh_hmul_add, bh_project, bh_vthv) and (b) DEVICE-RESIDENT: the library-side chain bh.inner_step, where
    between the upload of x, g and the download of s no n-vector crosses PCIe (asserted on bh_stats' transfer counters) and the
    active set grows on the device (bh_proj_update_active_dev: Gram downdate instead of a factor rebuild).  Same CG exit
    status in every minor iterate, same final active set, same step; prints the wall times."""
    import time
    J = R.synthetic_J(d, n, seed=1)
    inst = R.synthetic_box_vectors(d, n, fix_every=8)
    A = np.random.default_rng(5).standard_normal((mA, n))
    L0 = R.chol_lower(A @ A.T)
    x = inst.x - A.T @ np.linalg.solve(A @ A.T, A @ inst.x) if mA else inst.x      # any x works: the step stays in null(A)
    x = np.clip(x, inst.x_l, inst.x_u)
    g = J.T @ inst.r0
    delta = R.initial_tr(g)

    def run(ops):
        cons = R.make_mixed_constraints(A, L0, l=inst.x_l, u=inst.x_u)
        H = ops.new_hessian(J, np.zeros((0, n)), 10.0)
        log = []
        t0 = time.perf_counter()
        if hasattr(ops, "inner_step"):
            s, pred = ops.inner_step(x, g, H, L0, cons, delta, 50, 0.1, 0.1, log)
        else:
            s, pred = R.inner_step(x, g, H, L0, cons, delta, 50, 0.1, 0.1, ops=ops, log=log)
        return s, pred, log, cons.fixvars.copy(), time.perf_counter() - t0

    s_ref, pred_ref, log_ref, fix_ref, t_cpu = run(R.NumpyOps())
    s, pred, log, fix, t_gpu = run(HipOpsDeviceAll(bh))
    res = HipOpsResident(bh)
    run(res)                                                         # warm-up (device vectors, kernels)
    res = HipOpsResident(bh)
    s_r, pred_r, log_r, fix_r, t_res = run(res)
    per_iter = res.loop_bytes / max(res.loop_minor, 1)
    with capsys.disabled():
        print("[inner_step d=%d n=%d mA=%d] oracle (CPU) %.3f s, device ops with host vectors %.4f s, device-resident chain %.4f s; "
              "%d minor iterates, %d active bounds; PCIe bytes inside the resident loop: %d (%.0f per minor iterate; one n-vector = %d)"
              % (d, n, mA, t_cpu, t_gpu, t_res, len(log_ref), int(fix_ref.sum()), res.loop_bytes, per_iter, 8 * n))
    for lg, sv, pv, fx in ((log, s, pred, fix), (log_r, s_r, pred_r, fix_r)):
        assert [e[1] for e in lg] == [e[1] for e in log_ref]         # same CG exit status in every minor iterate
        assert [e[2] for e in lg] == [e[2] for e in log_ref]         # same active-set size after every minor iterate
        assert np.array_equal(fx, fix_ref)                           # same final active set
        assert relnorm(sv, s_ref) <= 1e-6, relnorm(sv, s_ref)
        assert pv == pytest.approx(pred_ref, rel=1e-8)
    # no n-vector moved inside the loop: what crossed is the BitVector image (n/8 bytes) plus a few scalars per call
    assert res.loop_bytes <= res.loop_minor * (n // 8 + 512)
    assert per_iter < 8 * n / 4


@pytest.mark.parametrize("d,n,q,mA,nfix,seed", [(60, 24, 1, 0, 5, 1), (200, 65, 0, 3, 9, 2), (900, 512, 2, 8, 60, 3)])
def test_device_pointer_entry_points_match_the_oracle(bh, final_sync, d, n, q, mA, nfix, seed):
    """Every *_dev caller-level entry point on its own (bh_grad_dev, bh_hmul_add_dev, bh_step_accumulate_dev, bh_linesearch_dev,
    bh_minor_iterate_dev, bh_reduced_gradient_norm_dev, bh_model_reduction_dev, bh_cauchy_step_dev) against the oracle's
    function it replaces (src/basic_tralcnlss.jl:45, :412, :436-437, :766-791, :649-675, :869-875, :458, :574-639), with the transfer
    counters checked around each call: no n-vector crosses PCIe."""
    import ctypes as ct
    rng = np.random.default_rng(seed)
    lib = bh._lib.lib()
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    C = rng.standard_normal((q, n))
    A = rng.standard_normal((mA, n))
    L0 = R.chol_lower(A @ A.T)
    fix = np.zeros(n, dtype=bool)
    fix[rng.choice(n, nfix, replace=False)] = True
    xl, xu = -np.ones(n), np.ones(n)
    x = np.clip(0.3 * rng.standard_normal(n), -0.9, 0.9)
    x[fix] = np.where(rng.random(nfix) < 0.5, -1.0, 1.0)
    cons_o = R.make_mixed_constraints(A, L0, fix, l=xl, u=xu)
    Ho, H = R.AlHessian(J, C, 4.0), bh.AlHessian(J, C, 4.0)
    cons = bh.MixedConstraints(A, None, fix, l=xl, u=xu)
    P = cons.handle
    rx, ybar = rng.standard_normal(d), rng.standard_normal(q)
    s0 = 0.01 * rng.standard_normal(n)
    s0[fix] = 0.0
    dv = {k: bh.DeviceVector(n, v) for k, v in (("x", x), ("s", s0), ("xl", xl), ("xu", xu))}
    for k in ("g", "gm", "w", "t"):
        dv[k] = bh.DeviceVector(n)
    dr = bh.DeviceVector(d, rx)
    big = 8 * n            # one n-vector

    def moved(fn):
        st0 = H.stats()
        fn()
        st1 = H.stats()
        return (st1["h2d_bytes"] - st0["h2d_bytes"]) + (st1["d2h_bytes"] - st0["d2h_bytes"])

    # g = J'r + C'ybar (:45)
    yb = np.ascontiguousarray(ybar)
    assert moved(lambda: bh._lib.check(lib.bh_grad_dev(H.handle, dr.ptr, bh._lib.ptr(yb), dv["g"].ptr), "grad")) <= 8 * q + 64
    g_ref = J.T @ rx + C.T @ ybar
    g = dv["g"].download()
    assert np.linalg.norm(g - g_ref) <= TOL1 * np.linalg.norm(np.abs(J).T @ np.abs(rx) + np.abs(C).T @ np.abs(ybar))
    # g_minor = H*s + g (:412)
    assert moved(lambda: bh._lib.check(lib.bh_hmul_add_dev(H.handle, dv["s"].ptr, dv["g"].ptr, dv["gm"].ptr), "hmul_add")) < big
    gm = dv["gm"].download()
    assert relnorm(gm, R.hmul(Ho, s0) + g) <= 1e-12
    # reduced-gradient norms (:869-875) and the model value (:458)
    out = ct.c_double()
    assert moved(lambda: bh._lib.check(lib.bh_reduced_gradient_norm_dev(P, dv["gm"].ptr, ct.byref(out)), "rgn")) < big
    assert out.value == pytest.approx(R.norm_reduced_gradient(gm, cons_o), rel=1e-10)
    assert moved(lambda: bh._lib.check(lib.bh_model_reduction_dev(H.handle, dv["g"].ptr, dv["s"].ptr, ct.byref(out)), "mr")) < big
    assert out.value == pytest.approx(float(g @ s0) + 0.5 * R.vthv(Ho, s0), rel=1e-10, abs=1e-14)
    # minor_iterate (:649-675) and linesearch on its unscaled direction (:766-791)
    delta = 0.3 * np.linalg.norm(g)
    status, iters, nh, alpha = ct.c_int32(), ct.c_int32(), ct.c_int32(), ct.c_double()
    assert moved(lambda: bh._lib.check(lib.bh_minor_iterate_dev(H.handle, P, dv["x"].ptr, dv["s"].ptr, dv["gm"].ptr, dv["xl"].ptr, dv["xu"].ptr,
                                                              delta, 0.1, R.SQRT_EPS, 1e-10, dv["w"].ptr, ct.byref(status), ct.byref(iters),
                                                              ct.byref(nh), ct.byref(alpha)), "minor_iterate_dev")) < big
    w_ref, st_ref = R.minor_iterate(x, s0, gm, Ho, cons_o, delta, 0.1)
    w = dv["w"].download()
    w_l, w_u = R.build_step_bounds(x + s0, cons_o, delta)
    w_cg, st_cg, _ = R.projected_cg(gm, Ho, w_l, w_u, cons_o, 0.1)
    # (CG amplifies summation-order noise by ~cond(H): the bar is the oracle's own sensitivity, as for projected_cg itself)
    assert status.value == int(st_ref) and relnorm(w, w_ref) <= max(1e-8, w_tolerance(gm, Ho, w_l, w_u, cons_o, 0.1, w_cg))
    dwl, dwu, dwc = bh.DeviceVector(n, w_l), bh.DeviceVector(n, w_u), bh.DeviceVector(n, w_cg)
    assert moved(lambda: bh._lib.check(lib.bh_linesearch_dev(H.handle, P, dv["gm"].ptr, dwc.ptr, dwl.ptr, dwu.ptr, ct.byref(out)), "ls")) < big
    assert out.value == pytest.approx(R.linesearch(gm, Ho, w_cg, w_l, w_u, cons_o.fixvars), rel=1e-10)
    # s .+= w ; g_minor = H*s + g (:436-437)
    assert moved(lambda: bh._lib.check(lib.bh_step_accumulate_dev(H.handle, dv["s"].ptr, dv["w"].ptr, dv["g"].ptr, dv["t"].ptr), "acc")) < big
    assert np.array_equal(dv["s"].download(), s0 + w)
    assert relnorm(dv["t"].download(), R.hmul(Ho, s0 + w) + g) <= 1e-12
    # cauchy_step (:574-639) from x, leaving the active set the reference leaves
    cau_o = R.make_mixed_constraints(A, L0, l=xl, u=xu)
    s_ref = R.cauchy_step(x, g, Ho, L0, cau_o, delta, R.NumpyOps())
    cau = bh.MixedConstraints(A, None, None, l=xl, u=xu)
    chunks = np.zeros((n + 63) // 64, dtype=np.uint64)
    nbp = ct.c_int32()
    assert moved(lambda: bh._lib.check(lib.bh_cauchy_step_dev(H.handle, cau.handle, dv["x"].ptr, dv["g"].ptr, dv["xl"].ptr, dv["xu"].ptr, delta,
                                                            dv["t"].ptr, bh._lib.ptr(chunks), ct.byref(nbp), ct.byref(nh)), "cauchy_dev")) < big
    fix_dev = np.unpackbits(chunks.view(np.uint8), bitorder="little")[:n].astype(bool)
    assert np.array_equal(fix_dev, cau_o.fixvars)
    assert np.linalg.norm(dv["t"].download() - s_ref) <= 1e-9 * max(np.linalg.norm(s_ref), 1e-300)
    H.close()


@pytest.mark.parametrize("n,mA", [(24, 0), (30, 0), (32, 0), (65, 0), (112, 0), (30, 3), (48, 5), (65, 2)])
def test_device_pointer_calls_stay_inside_the_callers_buffers(bh, cg_fused, n, mA):
    """The device-pointer entry points use the caller's vectors where they lie when the kernels' 16-byte chunk accesses fit
    (n a multiple of 16) and through the padded workspace otherwise.  Every caller vector here is carved out of ONE device
    arena with NaN canaries between the vectors: a write past the end of a vector changes a canary, a read past the end that
    reaches a result turns it into NaN (n = 24, 30: even but not a multiple of 16 — the case an in-place rule gets wrong first).
    All *_dev entry points, box constraints and linear equalities, every CG iteration shape."""
    import ctypes as ct
    rng = np.random.default_rng(n + 7 * mA)
    lib = bh._lib.lib()
    d = 40 * n
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    fix = np.zeros(n, dtype=bool)
    fix[rng.choice(n, n // 5, replace=False)] = True
    xl, xu = -np.ones(n), np.ones(n)
    x = np.where(fix, 1.0, 0.2 * rng.standard_normal(n))
    A = rng.standard_normal((mA, n))
    L0 = R.chol_lower(A @ A.T)
    cons_o = R.make_mixed_constraints(A, L0, fix, l=xl, u=xu)
    Ho = R.AlHessian(J, np.zeros((0, n)), 1.5)
    H = bh.AlHessian(J, None, 1.5)
    cons = bh.MixedConstraints(A, cons_o.chol_L if mA else None, fix, l=xl, u=xu)
    P = cons.handle
    rx = rng.standard_normal(d)
    g = J.T @ rx
    s0 = np.zeros(n)
    delta = 0.4 * np.linalg.norm(g)
    w_l, w_u = R.build_step_bounds(x + s0, cons_o, delta)
    vecs = {"x": x, "s": s0, "g": g, "gm": g, "xl": xl, "xu": xu, "wl": w_l, "wu": w_u, "w": np.zeros(n), "w2": np.zeros(n), "t": np.zeros(n),
            "pv": np.zeros(n), "gr": np.zeros(n), "cs": np.zeros(n), "jt": np.zeros(n), "rx": rx, "u": np.zeros(d)}
    gap = 34                                           # canaries between the vectors (even: every vector stays 16-byte aligned)
    off, pos = {}, gap
    for k, v in vecs.items():
        off[k] = pos
        pos += v.size + (v.size % 2) + gap
    host = np.full(pos, np.nan)
    for k, v in vecs.items():
        host[off[k]:off[k] + v.size] = v
    arena = bh.DeviceVector(host.size, host)
    ptr = {k: ct.c_void_p(arena.ptr.value + 8 * off[k]) for k in vecs}
    st, it, nh, al, out = ct.c_int32(), ct.c_int32(), ct.c_int32(), ct.c_double(), ct.c_double()
    chk = bh._lib.check
    chk(lib.bh_pcg_dev(H.handle, P, ptr["g"], ptr["wl"], ptr["wu"], 0.1, R.SQRT_EPS, 1e-10, ptr["w"], ct.byref(st), ct.byref(it), None, 0, ct.byref(nh)), "pcg_dev")
    w_ref, st_ref, it_ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.1)
    chk(lib.bh_minor_iterate_dev(H.handle, P, ptr["x"], ptr["s"], ptr["gm"], ptr["xl"], ptr["xu"], delta, 0.1, R.SQRT_EPS, 1e-10, ptr["w2"],
                                 ct.byref(st), ct.byref(it), ct.byref(nh), ct.byref(al)), "minor_iterate_dev")
    w2_ref, st2_ref = R.minor_iterate(x, s0, g, Ho, cons_o, delta, 0.1)
    chk(lib.bh_hmul_dev(H.handle, ptr["g"], ptr["t"]), "hmul_dev")
    chk(lib.bh_project_dev(P, ptr["g"], ptr["pv"]), "project_dev")
    chk(lib.bh_jv_dev(H.handle, ptr["g"], ptr["u"]), "jv_dev")
    chk(lib.bh_jtv_dev(H.handle, ptr["rx"], ptr["jt"]), "jtv_dev")
    chk(lib.bh_grad_dev(H.handle, ptr["rx"], None, ptr["gr"]), "grad_dev")
    chk(lib.bh_linesearch_dev(H.handle, P, ptr["g"], ptr["w"], ptr["wl"], ptr["wu"], ct.byref(out)), "linesearch_dev")
    ls = out.value
    chk(lib.bh_reduced_gradient_norm_dev(P, ptr["g"], ct.byref(out)), "norm")
    nrm = out.value
    chk(lib.bh_model_reduction_dev(H.handle, ptr["g"], ptr["w2"], ct.byref(out)), "mr")
    mr = out.value
    chk(lib.bh_step_accumulate_dev(H.handle, ptr["s"], ptr["w2"], ptr["g"], ptr["gm"]), "acc")
    mid = arena.download()
    # the Cauchy search last: it replaces the active set of the handle
    chunks = np.zeros((n + 63) // 64, dtype=np.uint64)
    nbp = ct.c_int32()
    chk(lib.bh_cauchy_step_dev(H.handle, P, ptr["x"], ptr["g"], ptr["xl"], ptr["xu"], delta, ptr["cs"], bh._lib.ptr(chunks), ct.byref(nbp), ct.byref(nh)),
        "cauchy_step_dev")
    back = arena.download()
    mask = np.ones(host.size, dtype=bool)
    for k, v in vecs.items():
        mask[off[k]:off[k] + v.size] = False
    assert np.all(np.isnan(back[mask])), "a canary between the caller's vectors was overwritten: %s" % np.flatnonzero(~np.isnan(back[mask]))[:8]
    get = lambda k, src=mid: src[off[k]:off[k] + vecs[k].size]
    for k in ("w", "w2", "t", "gm", "pv", "u", "jt", "gr"):
        assert np.all(np.isfinite(get(k))), k
    tol = max(1e-8, w_tolerance(g, Ho, w_l, w_u, cons_o, 0.1, w_ref))
    assert_w_close(get("w"), w_ref, tol, "device-pointer calls: w vs oracle (max(1e-8, CG tolerance))")
    assert_w_close(get("w2"), w2_ref, tol, "device-pointer calls: w vs oracle (max(1e-8, CG tolerance))")
    assert relnorm(get("t"), R.hmul(Ho, g)) <= 1e-12 and relnorm(get("u"), J @ g) <= 1e-12 and relnorm(get("jt"), g) <= 1e-12 and relnorm(get("gr"), g) <= 1e-12
    assert relnorm(get("pv"), R.projection(cons_o, g)) <= 1e-9
    assert ls == pytest.approx(R.linesearch(g, Ho, get("w"), w_l, w_u, cons_o.fixvars), rel=1e-9)
    assert nrm == pytest.approx(R.norm_reduced_gradient(g, cons_o), rel=1e-9)
    assert mr == pytest.approx(float(g @ get("w2")) + 0.5 * R.vthv(Ho, get("w2")), rel=1e-9, abs=1e-12)
    assert relnorm(get("s"), s0 + get("w2")) <= 1e-15 and relnorm(get("gm"), R.hmul(Ho, get("s")) + g) <= 1e-10
    cau = R.make_mixed_constraints(A, L0, l=xl, u=xu)
    s_ref = R.cauchy_step(x, g, Ho, L0, cau, delta, R.NumpyOps())
    assert np.all(np.isfinite(get("cs", back))) and relnorm(get("cs", back), s_ref) <= 1e-8
    H.close()


@pytest.mark.parametrize("n,mA,seed", [(40, 0, 1), (200, 5, 2), (700, 64, 3), (300, 100, 4)])
def test_device_side_active_set_update_matches_oracle(bh, final_sync, n, mA, seed):
    """bh_proj_update_active_dev against the reference's active_bounds + add_active! / active_bounds!
    (src/polyhedral_constraints.jl:203-261; src/basic_tralcnlss.jl:439-453): same |active_indx|, same branch, same fixvars, and
    the projector of the updated set (factor obtained by a Gram DOWNDATE over the newly fixed columns) agrees with the
    oracle's rebuilt-from-scratch one; then a second update on top of the first (incremental twice), and the :452 branch."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((mA, n))
    L0 = R.chol_lower(A @ A.T)
    xl, xu = -np.ones(n), np.ones(n)
    x = rng.uniform(-0.9, 0.9, n)
    fix0 = np.zeros(n, dtype=bool)
    fix0[rng.choice(n, n // 10, replace=False)] = True
    x[fix0] = np.where(rng.random(int(fix0.sum())) < 0.5, -1.0, 1.0)
    delta = 0.4
    cons_o = R.make_mixed_constraints(A, L0, fix0, l=xl, u=xu)
    cons = bh.MixedConstraints(A, None, fix0, l=xl, u=xu)
    dv = {k: bh.DeviceVector(n, v) for k, v in (("x", x), ("xl", xl), ("xu", xu))}
    ds = bh.DeviceVector(n)
    lib = bh._lib.lib()
    import ctypes as ct
    chunks = np.zeros((n + 63) // 64, dtype=np.uint64)
    s = np.zeros(n)
    for rnd in range(3):
        # a step that parks some free variables on their bound / on the trust-region face (round 2: enough of them for :452)
        s = s.copy()
        free = np.flatnonzero(~cons_o.fixvars)
        if rnd < 2:
            pick = rng.choice(free, min(max(1, len(free) // 8), len(free)), replace=False)
            s[pick] = np.where(rng.random(len(pick)) < 0.5, np.maximum(xl[pick] - x[pick], -delta), np.minimum(xu[pick] - x[pick], delta))
            rest = ~np.isin(np.arange(n), pick) & ~cons_o.fixvars
            s[rest] = rng.uniform(-0.05, 0.05, int(rest.sum()))
        else:
            s[free] = -delta * np.sign(x[free])       # every free variable on the trust-region face, none on a true bound
        ds.upload(s)
        idx = R.active_bounds(cons_o, x, s, delta)                                   # :439
        if mA + idx.shape[0] <= n:
            R.add_active(cons_o, L0, idx)
            branch_ref = 0
        else:
            R.active_bounds_inplace(cons_o, x + s, L0)
            branch_ref = 1
        n_at, n_fix, br = ct.c_int32(), ct.c_int32(), ct.c_int32()
        cons._sync()
        bh._lib.check(lib.bh_proj_update_active_dev(cons._h, dv["x"].ptr, ds.ptr, dv["xl"].ptr, dv["xu"].ptr, delta, R.SQRT_EPS,
                                                    ct.byref(n_at), ct.byref(n_fix), ct.byref(br), bh._lib.ptr(chunks)), "update_active")
        fix_dev = np.unpackbits(chunks.view(np.uint8), bitorder="little")[:n].astype(bool)
        assert (n_at.value, br.value) == (idx.shape[0], branch_ref)
        assert np.array_equal(fix_dev, cons_o.fixvars) and n_fix.value == int(cons_o.fixvars.sum())
        cons._fixvars, cons._dirty = fix_dev, False
        if mA + n_fix.value <= n and (mA == 0 or n_fix.value < n - mA):
            r = rng.standard_normal(n)
            v_ref = R.projection(cons_o, r)
            assert np.linalg.norm(bh.projection(cons, r) - v_ref) <= 1e-10 * np.linalg.norm(r)


def test_interleaved_handles_share_the_workspace_safely(bh):
    """Several AlHessian / MixedConstraints objects alive at once (the reference holds the old and the new Hessian across
    an accepted step) and used alternately: the shared CG workspace and progress word must not leak state between them."""
    rng = np.random.default_rng(123)
    probs = []
    for (d, n, mA, nfix) in [(120, 40, 0, 5), (300, 96, 2, 10), (90, 33, 0, 0), (500, 128, 4, 20)]:
        J = rng.standard_normal((d, n)) / np.sqrt(d)
        A = rng.standard_normal((mA, n))
        fix = np.zeros(n, dtype=bool)
        fix[rng.choice(n, nfix, replace=False)] = True
        cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix if nfix else None, l=-np.ones(n), u=np.ones(n))
        g = rng.standard_normal(n)
        w_l, w_u = R.build_step_bounds(np.where(fix, 1.0, 0.0), cons_o, 0.5)
        Ho = R.AlHessian(J, np.zeros((0, n)), 2.0)
        ref = R.projected_cg(g, Ho, w_l, w_u, cons_o, 0.1)
        probs.append(dict(H=bh.AlHessian(J, None, 2.0), cons=bh.MixedConstraints(A, cons_o.chol_L, fix), g=g, w_l=w_l, w_u=w_u,
                          ref=ref, Ho=Ho, r=rng.standard_normal(n), cons_o=cons_o))
    first = {}
    for rnd in range(3):
        for i in (0, 1, 2, 3, 2, 0, 3, 1):
            p = probs[i]
            w, st, info = bh.projected_cg(p["g"], p["H"], p["w_l"], p["w_u"], p["cons"], 0.1, full_output=True)
            assert int(st) == int(p["ref"][1]) and info["iters"] == p["ref"][2]
            assert relnorm(w, p["ref"][0]) <= 1e-8
            if i in first:
                assert np.array_equal(w, first[i])            # bit-identical every time, whatever ran in between
            first[i] = w
            assert relnorm(bh.projection(p["cons"], p["r"]), R.projection(p["cons_o"], p["r"])) <= 1e-10
            assert relnorm(p["H"] * p["g"], R.hmul(p["Ho"], p["g"])) <= 1e-12


def test_row_sharded_products_emulated_on_one_gpu(bh):
    """SURVEY.md §8(e) on one device: G handles, one per row block (what each rank holds; C only with block 0), their
    products summed in rank order (what the all-reduce does) — against the unsharded handle and the oracle, for H*v, J'u,
    vthv and a whole projected_cg driven shard-wise."""
    rng = np.random.default_rng(2024)
    d, n, q, G = 1000, 192, 2, 4
    J = R.synthetic_J(d, n, seed=7)
    C = rng.standard_normal((q, n))
    mu = 10.0
    H_full, Ho = bh.AlHessian(J, C, mu), R.AlHessian(J, C, mu)
    shards = []
    for k in range(G):
        lo, hi = bh.row_shard(d, k, G)
        shards.append((lo, hi, bh.AlHessian(J[lo:hi], C if k == 0 else None, mu)))
    v, u = rng.standard_normal(n), rng.standard_normal(d)

    def sharded_hmul(_H, x):
        z = np.zeros(n)
        for lo, hi, Hk in shards:            # fixed rank order, like a ring/tree all-reduce result
            z += Hk * x
        return z
    scale = np.linalg.norm(np.abs(J).T @ (np.abs(J) @ np.abs(v)) + mu * np.abs(C).T @ (np.abs(C) @ np.abs(v)))
    assert np.linalg.norm(sharded_hmul(None, v) - H_full * v) <= 1e-12 * scale
    assert np.linalg.norm(sharded_hmul(None, v) - R.hmul(Ho, v)) <= 1e-12 * scale
    jtu = sum(Hk.jtv(u[lo:hi]) for lo, hi, Hk in shards)
    assert np.linalg.norm(jtu - J.T @ u) <= 1e-12 * matvec_scale(J.T, u)
    assert sum(bh.vthv(Hk, v) for _, _, Hk in shards) == pytest.approx(R.vthv(Ho, v), rel=1e-12)
    # the CG loop with shard-wise products (replicated n-vector work) reaches the same exit as the unsharded device run
    inst = R.synthetic_box_vectors(d, n, fix_every=6)
    A = np.zeros((0, n))
    cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), inst.fixvars, l=inst.x_l, u=inst.x_u)
    g = J.T @ inst.r0
    w_l, w_u = R.build_step_bounds(inst.x, cons_o, R.initial_tr(g))
    w_sh, s_sh, it_sh = R.projected_cg(g, None, w_l, w_u, cons_o, 0.01, hmul_fn=sharded_hmul)
    cons = bh.MixedConstraints(A, None, inst.fixvars)
    w, st, info = bh.projected_cg(g, H_full, w_l, w_u, cons, 0.01, full_output=True)
    assert int(st) == int(s_sh) and info["iters"] == it_sh
    assert relnorm(w, w_sh) <= 1e-9


def test_full_solve_medium_nls_through_c_abi(bh, capsys):
    """The restated outer iteration (tralcnllss -> solve_subproblem -> inner_step) on a 48-parameter constrained NLS with
    every hot-path / next-row call on the device (hundreds of minor iterates, Cauchy searches, active-set changes, mu and
    Hessian updates) against the all-CPU oracle run.

    A whole solve is a chaotic map of its rounding errors: the ORACLE itself takes 515 / 559 / 572 minor iterates when its H*v
    sums in fp64 / in two row blocks / in long double (measured in this container), because near the end of an outer
    iteration rho = ared/pred (src/basic_tralcnlss.jl:353-354) divides a difference of two objective values that agree to
    ~16 digits by an equally tiny model reduction.  So the test pins the divergence instead of tolerating it:
      (1) free-running device solve vs oracle solve: every driver decision (CG exit status and active-set size of each minor
          iterate, minor-loop exit, trust-region accept / resize, outer exit) is identical up to the FIRST differing one, and
          that one must be a rounding-dominated rho (numerator worth <= 512 ulps of mx) — the named iterate is printed;
      (2) shadow solve: device and oracle evaluated on IDENTICAL operands at every hot-path call of the device's trajectory
          — any call whose CG status / iteration count / active set differs must carry a logged tie (bh_pcg_tie_info)."""
    import time
    from _util import first_decision_difference
    from nls_problem import NLSProblem
    from _util import assert_rounding_dominated
    P = NLSProblem(256, 48, 2, seed=1)
    kw = dict(max_outer_iter=30, max_inner_iter=60)
    t0 = time.perf_counter()
    log_ref = []
    x_ref, y_ref = R.tralcnllss(P.x0, P.r, P.jac_r, P.c, P.jac_c, P.A, P.b, P.x_l, P.x_u, log=log_ref, **kw)
    t_cpu = time.perf_counter() - t0
    ops = HipOpsDeviceAll(bh)
    log = []
    t0 = time.perf_counter()
    x, y = R.tralcnllss(P.x0, P.r, P.jac_r, P.c, P.jac_c, P.A, P.b, P.x_l, P.x_u, ops=ops, log=log, **kw)
    t_gpu = time.perf_counter() - t0
    obj = lambda z: 0.5 * float(P.r(z) @ P.r(z))
    n_ref, n_dev = sum(e[0] == "minor" for e in log_ref), sum(e[0] == "minor" for e in log)
    diff = first_decision_difference(log_ref, log)
    with capsys.disabled():
        print("[full solve n=48 d=256] oracle %.2f s (%d minor iterates), device ops %.2f s (%d); |x - x_ref| = %.2e, obj %.9f vs %.9f"
              % (t_cpu, n_ref, t_gpu, n_dev, np.linalg.norm(x - x_ref), obj(x), obj(x_ref)))
        if diff is None:
            print("    every driver decision identical (%d log entries)" % len(log))
        else:
            k, a, b, why = diff
            print("    first differing decision: log entry %d = trust-region iterate after minor iterate #%d; oracle %s, device %s; %s"
                  % (k, sum(e[0] == "minor" for e in log_ref[:k]), a[:4], b[:4], why))
    if diff is None:
        assert len(log) == len(log_ref)
    else:
        assert_rounding_dominated(diff)
        assert diff[0] >= 100                      # a long common prefix: hundreds of identical decisions come first
    assert np.linalg.norm(P.c(x)) < 1e-6 and np.linalg.norm(P.A @ x - P.b) < 1e-10
    assert np.all(x >= P.x_l - 1e-12) and np.all(x <= P.x_u + 1e-12)
    assert obj(x) == pytest.approx(obj(x_ref), rel=1e-5)
    assert np.linalg.norm(x - x_ref) <= 1e-4 * np.linalg.norm(x_ref)

    sh = ShadowOps(HipOpsDeviceAll(bh))
    xs, ys = R.tralcnllss(P.x0, P.r, P.jac_r, P.c, P.jac_c, P.A, P.b, P.x_l, P.x_u, ops=sh, **kw)
    with capsys.disabled():
        print("    shadow solve: %d minor iterates compared call by call on identical operands: %d discrepancies; worst relative deviation "
              "per operator %s; closest CG branch margin of the solve %.2e (%s)"
              % (sh.minor, len(sh.events), {k: float("%.1e" % v) for k, v in sh.worst.items()}, sh.min_margin[0],
                 None if sh.min_margin[1] is None else "minor iterate %d, test %s" % (sh.min_margin[1]["minor"], bh.operators.TIE_KINDS.get(sh.min_margin[1]["min_margin_kind"]))))
        for e in sorted(sh.events, key=lambda e: -e["rel"])[:3]:
            print("        largest:", {k: v for k, v in e.items() if k != "operands"})
    check_shadow_events(sh)


def check_shadow_events(sh):
    """Acceptance rule for a shadow solve: on identical operands
      * a minor iterate whose CG status or iteration count differs from the oracle's must carry a logged tie;
      * a projection must agree to 1e-10 of its operand's norm (ShadowOps records anything above);
      * the result of a Cauchy search / minor iterate may deviate by more than 1e-6 only as far as the ORACLE's own result
        moves when its right-hand side is perturbed in the last bit (both are cancelling computations near a critical point:
        the direction P(-g) carries ||g||/||P(-g)|| = 1e5 ... 1e9 of amplification there, whoever computes it);
      * a Cauchy search that ends on another active set than the oracle's must pass the property check of the pinned-operands test
        against the family of CPU outcomes on ITS operands (tests/_util.py::check_cauchy_against_cpu_family)."""
    for e in sh.events:
        ev = {k: v for k, v in e.items() if k != "operands"}
        if e["op"] == "minor_iterate" and (e["status_dev"] != e["status_cpu"] or e["iters_dev"] != e["iters_cpu"]):
            assert e["ties"] is not None and e["ties"]["tie_flags"] != 0, "status / iteration count differ on identical operands without a logged tie: %r" % (ev,)
        elif e["op"] == "projection":
            raise AssertionError("projection deviates on identical operands: %r" % (ev,))
        elif e["op"] == "cauchy_step" and e["fix_dev"] != e["fix_cpu"]:
            # another final active set than the oracle's on identical operands: accepted only if the outcome is within what the CPU
            # restatements themselves produce on these operands under 1-ulp perturbations of g (feasibility, model value, set size,
            # agreement with outcomes that share its set) — the rule of test_cauchy_step_on_the_pinned_multimodal_operands
            from types import SimpleNamespace
            from _util import check_cauchy_against_cpu_family
            op = e["operands"]
            check_cauchy_against_cpu_family(SimpleNamespace(A=op["A"], x_l=op["x_l"], x_u=op["x_u"]), op, op["s_dev"],
                                            tuple(np.flatnonzero(op["fix_dev_set"])), samples=16, seed=e["minor"])
        else:
            assert e["rel"] <= max(1e-6, 20.0 * e["oracle_sensitivity"]), ev
