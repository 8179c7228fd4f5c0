"""CPU tests: the plain-C oracle (oracle/benlsip_oracle.c) against the reference's HS48 known answer, against the NumPy
oracle and against the committed golden fixtures — two independent restatements must agree."""
import json
import os

import numpy as np

import benlsip_oracle as BO
import benlsip_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _flt(xs):
    return np.array([float(x) for x in xs], dtype=np.float64)


def test_c_oracle_hs48_known_answer():
    h = json.load(open(os.path.join(GOLD, "hs48_projection.json")))
    A, x = np.array(h["A"]), np.array(h["x"])
    fix = np.zeros(5, dtype=bool)
    fix[np.array(h["fixed_1based"]) - 1] = True
    L = R.cholesky_aug_aat(A, fix, R.chol_lower(A @ A.T))
    proj = BO.projection(A, fix, L, x)
    assert np.max(np.abs(proj - np.array(h["projection"]))) <= 1e-14          # test/structures.jl:57


def test_c_oracle_operators_match_numpy_oracle():
    rng = np.random.default_rng(0)
    for d, n, q in [(4, 3, 1), (50, 20, 0), (300, 130, 2), (1000, 257, 1)]:
        J, Cm, mu, v = rng.standard_normal((d, n)), rng.standard_normal((q, n)), 0.7, rng.standard_normal(n)
        H = R.AlHessian(J, Cm, mu)
        scale = np.linalg.norm(np.abs(J).T @ (np.abs(J) @ np.abs(v)) + mu * np.abs(Cm).T @ (np.abs(Cm) @ np.abs(v)))
        assert np.linalg.norm(BO.hmul(J, Cm, mu, v) - R.hmul(H, v)) <= 1e-13 * scale
        assert abs(BO.vthv(J, Cm, mu, v) - R.vthv(H, v)) <= 1e-13 * R.vthv(H, v)
    p, w = rng.standard_normal(100), 0.1 * rng.standard_normal(100)
    wl = np.where(rng.random(100) < 0.5, -np.inf, -1.0)
    wu = np.where(rng.random(100) < 0.5, np.inf, 1.0)
    assert BO.factor_to_boundary(p, w, wl, wu) == R.factor_to_boundary(p, w, wl, wu)
    assert BO.num_threads() >= 1


def test_c_oracle_reproduces_golden_pcg_cases():
    cases = json.load(open(os.path.join(GOLD, "pcg_cases.json")))["cases"]
    for c in cases:
        d, n, q, mA, mpp = c["d"], c["n"], c["q"], c["mA"], c["mpp"]
        J = _flt(c["J"]).reshape((d, n), order="F")
        Cm = _flt(c["C"]).reshape((q, n), order="F")
        A = _flt(c["A"]).reshape((mA, n), order="F")
        L = _flt(c["L"]).reshape((mpp, mpp), order="F")
        fix = np.array(c["fixvars"], dtype=bool)
        w, status, iters, n_hmul, trace = BO.projected_cg(_flt(c["g"]), J, Cm, c["mu"], _flt(c["w_l"]), _flt(c["w_u"]), A, fix, L,
                                                          c["kappa2"], trace_cap=64)
        assert (status, iters, n_hmul) == (c["status"], c["iters"], c["n_hmul"]), c["name"]
        w_gold = _flt(c["w"])
        tol = 1e-6 if c["name"] == "maxiter_exhaust" else 1e-8
        if np.all(np.isfinite(w_gold)):
            assert np.linalg.norm(w - w_gold) <= tol * max(np.linalg.norm(w_gold), 1e-300), c["name"]
        else:
            assert np.array_equal(np.isnan(w), np.isnan(w_gold))
