#!/usr/bin/env python3
"""Regenerate tests/golden/*.json.

The reference is pure Julia and cannot run in this image (no Julia toolchain), so:
  * hs48_projection.json holds the DATA of the reference's one known-answer test
    (test/structures.jl:37-58: A, x, fixed set, expected projection) — a fixture, not code;
  * sphere_regression.json holds the data of test/problems/sphere_regression.jl (bounds, A, b, x0) and the
    solution the NumPy oracle reaches, with the three acceptance measures of :63-65;
  * pcg_cases.json holds seeded projected_cg inputs -> (w, status, iters, scalar trace) produced by the
    NumPy oracle (oracle/benlsip_ref.py).  projected_cg has no golden data in the reference, so these pin the
    HIP path to the oracle, not to Julia ("parity unpinned" at iteration level).
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import benlsip_ref as R  # noqa: E402


def f(a):
    return [float(x) if np.isfinite(x) else (str(x)) for x in np.asarray(a, dtype=np.float64).ravel()]


def pcg_case(name, seed, d, n, q, mA, nfix, kind="plain", delta=0.5, kappa2=0.1):
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((d, n)) / np.sqrt(max(d, 1))
    if kind == "illcond":
        J = J * (10.0 ** (-2.0 * np.arange(n) / n))[None, :]
    C = rng.standard_normal((q, n))
    mu = 10.0
    if kind == "negcurv":           # H = 0 -> pHp = 0 <= atol with |pHp| <= atol: no boundary move
        J = np.zeros((d, n)); C = np.zeros((q, n))
    A = rng.standard_normal((mA, n))
    L0 = R.chol_lower(A @ A.T)
    fix = np.zeros(n, dtype=bool)
    if nfix:
        fix[rng.choice(n, nfix, replace=False)] = True
    lo = R.make_mixed_constraints(A, L0, fix if nfix else None, l=-np.ones(n), u=np.ones(n))
    x_minor = 0.3 * rng.standard_normal(n).clip(-2, 2)
    x_minor[fix] = np.where(rng.random(nfix) < 0.5, -1.0, 1.0)
    w_l, w_u = R.build_step_bounds(x_minor, lo, delta)
    if kind == "bound":             # finite bounds on the free variables so alpha > gamma fires
        w_l = np.where(fix, w_l, -1e-3); w_u = np.where(fix, w_u, 1e-3)
    g = rng.standard_normal(n)
    if kind == "exhaust":           # kappa2 = 0 and a huge gradient: never "solved", pHp stays > atol -> runs out of
        g = g * 1e30                # iterations; the reference then returns `nothing`, never status 3
    H = R.AlHessian(J, C, mu)
    tr = R.CGTrace()
    w, status, iters = R.projected_cg(g, H, w_l, w_u, lo, kappa2, trace=tr)
    return {
        "name": name, "d": d, "n": n, "q": q, "mA": mA, "nfix": nfix, "mu": mu, "kappa2": kappa2,
        "J": f(np.asfortranarray(J).ravel(order="F")), "C": f(np.asfortranarray(C).ravel(order="F")),
        "A": f(np.asfortranarray(A).ravel(order="F")), "L": f(np.asfortranarray(lo.chol_L).ravel(order="F")),
        "mpp": int(lo.chol_L.shape[0]), "fixvars": [int(b) for b in fix], "g": f(g), "w_l": f(w_l), "w_u": f(w_u),
        "w": f(w), "status": int(status), "iters": int(iters), "n_hmul": tr.n_hmul,
        "trace": [f(row) for row in tr.rows],
    }


def main():
    hs48 = {
        "source": "test/structures.jl:37-58",
        "A": [[1.0, 1, 1, 1, 1], [0, 0, 1, -2, -2]], "b": [5.0, -3], "x": [3.0, 5, -3, 2, -2],
        "fixed_1based": [1, 2], "projection": [0.0, 0, 0, 2, -2],
    }
    json.dump(hs48, open(os.path.join(HERE, "hs48_projection.json"), "w"), indent=1)

    cases = [
        pcg_case("box_small", 11, 40, 16, 0, 0, 3),
        pcg_case("box_nofix", 12, 64, 24, 0, 0, 0),
        pcg_case("box_q", 13, 50, 20, 2, 0, 4),
        pcg_case("lin_nullspace", 14, 60, 24, 1, 3, 0),
        pcg_case("lin_subspace", 15, 80, 32, 1, 3, 6),
        pcg_case("illcond_maxiter", 16, 30, 12, 0, 2, 4, kind="illcond", kappa2=1e-14),
        pcg_case("bound_hit", 17, 48, 20, 0, 0, 3, kind="bound"),
        pcg_case("zero_hessian_negcurv", 18, 10, 8, 1, 0, 2, kind="negcurv"),
        pcg_case("all_but_fixed", 19, 20, 6, 0, 2, 4),          # max_iter = 0 -> status none
        pcg_case("tiny_n3", 20, 4, 3, 1, 1, 0),
        pcg_case("odd_n", 21, 33, 17, 0, 2, 5),
        pcg_case("maxiter_exhaust", 22, 40, 10, 0, 0, 3, kind="exhaust", kappa2=0.0),
    ]
    json.dump({"generator": "oracle/benlsip_ref.py via tests/golden/make_golden.py", "cases": cases},
              open(os.path.join(HERE, "pcg_cases.json"), "w"))
    print({c["name"]: (c["status"], c["iters"], c["n_hmul"]) for c in cases})

    # sphere regression: data of the reference test + the oracle's solution and acceptance measures
    x_l = np.array([-2., -1.5, 0]); x_u = np.array([2., 1.5, 2.]); A = np.array([[1., 2, -1]]); b = np.array([0.5])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sphere_problem import c, jac_c, jac_r, r, x0
    xs, ys = R.tralcnllss(x0, r, jac_r, c, jac_c, A, b, x_l, x_u, max_outer_iter=100, max_inner_iter=250)
    grad = jac_r(xs).T @ r(xs) + jac_c(xs).T @ ys
    P = R.projection_polyhedron_small(xs - grad, A, b, x_l, x_u)
    sph = {"source": "test/problems/sphere_regression.jl:9-48,63-65", "x_l": f(x_l), "x_u": f(x_u), "A": f(A), "b": f(b),
           "x0": f(x0), "oracle_x": f(xs), "oracle_y": f(ys), "norm_c": float(np.linalg.norm(c(xs))),
           "opt_measure": float(np.linalg.norm(xs - P))}
    json.dump(sph, open(os.path.join(HERE, "sphere_regression.json"), "w"), indent=1)
    print(sph["oracle_x"], sph["norm_c"], sph["opt_measure"])


if __name__ == "__main__":
    main()
