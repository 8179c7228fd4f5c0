"""Child process of test_rccl_two_kernel_iteration_on_a_one_rank_communicator: BH_FORCE_COMM=1 BH_COMM=rccl brings up a 1-rank RCCL
communicator, which routes box-constrained projected_cg through the RCCL form of the two-kernel iteration
(row_stream_kernel<..., CGP = 3> + reduce_partials_sq_kernel + ncclAllReduce) and projected_cg with linear equalities through
the four-kernel one (H*p, slab reduction, ncclAllReduce, update, projection).  Every case of the golden file and a few
random instances (with traces, through bh_pcg and through bh_minor_iterate) are compared with the oracle.  Test infrastructure:
imports the oracle."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, os.path.join(ROOT, "oracle"), TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

import benlsip_ref as R                                        # noqa: E402
from _util import relnorm, w_tolerance                         # noqa: E402


def flt(seq):
    return np.array([float(x) for x in seq], dtype=np.float64)


def main():
    import benlsip_jl_amd as bh
    bh.init(0)
    bh.init_distributed(0, 1, lambda b: b)
    n_checked, worst = 0, 0.0
    cases = json.load(open(os.path.join(TESTS, "golden", "pcg_cases.json")))["cases"]
    n_gen = 0
    for c in cases:
        d, n, q, mA = c["d"], c["n"], c["q"], c["mA"]
        J = flt(c["J"]).reshape((d, n), order="F")
        C = flt(c["C"]).reshape((q, n), order="F")
        A = flt(c["A"]).reshape((mA, n), order="F")
        L = flt(c["L"]).reshape((c["mpp"], c["mpp"]), order="F")
        fix = np.array(c["fixvars"], dtype=bool)
        g, wl, wu = flt(c["g"]), flt(c["w_l"]), flt(c["w_u"])
        H = bh.AlHessian(J, C, c["mu"])
        cons = bh.MixedConstraints(A, L if mA else None, fix)
        a0 = H.stats()["n_allreduce"]
        w, status, info = bh.projected_cg(g, H, wl, wu, cons, c["kappa2"], trace_cap=64, full_output=True)
        assert int(status) == c["status"] and info["iters"] == c["iters"] and info["n_hmul"] == c["n_hmul"], (c["name"], int(status), info)
        if c["n_hmul"] >= 1:
            assert H.stats()["n_allreduce"] - a0 >= c["n_hmul"], c["name"]          # the collective really ran: this IS the RCCL path
            # two kernels + collective (box) / slab reduction + collective + update + projection behind the H*p launch (equalities)
            assert H.stats()["cg_kernels"] == (4 if mA else 2), (c["name"], H.stats()["cg_kernels"])
            n_gen += 1 if mA else 0
        w_ref = flt(c["w"])
        cons_o = R.MixedConstraints(A, -np.ones(n), np.ones(n), fix, L)
        tol = 1e-6 if c["name"] == "maxiter_exhaust" else w_tolerance(g, R.AlHessian(J, C, c["mu"]), wl, wu, cons_o, c["kappa2"], w_ref)
        if np.all(np.isfinite(w_ref)):
            rel = relnorm(w, w_ref)
            worst = max(worst, rel / tol)
            assert rel <= tol, (c["name"], rel, tol)
        else:
            assert np.array_equal(np.isnan(w), np.isnan(w_ref)), c["name"]
        tr_ref = np.array([[float(x) for x in row] for row in c["trace"]]).reshape(-1, 4)
        tr = info["trace"]
        assert tr.shape == tr_ref.shape, c["name"]
        if c["name"] != "maxiter_exhaust" and tr.size:
            m = np.isfinite(tr_ref)
            assert np.array_equal(np.isnan(tr), np.isnan(tr_ref)), c["name"]
            np.testing.assert_allclose(tr[m], tr_ref[m], rtol=max(1e-9, tol), atol=1e-10, err_msg=c["name"])
        n_checked += 1
        H.close(); cons.close()
    # random box instances: odd n, wide rows, many iterations; bh_pcg and bh_minor_iterate (H*w accumulated next to w)
    for d, n, nfix, kappa2, seed in ((300, 97, 9, 1e-4, 1), (8200, 4096, 300, 0.1, 2), (2000, 512, 40, 1e-3, 3), (9000, 4500, 100, 0.1, 4), (40, 3, 0, 0.1, 5)):
        rng = np.random.default_rng(seed)
        J = rng.standard_normal((d, n)) / np.sqrt(d)
        fix = np.zeros(n, dtype=bool)
        fix[rng.choice(n, nfix, replace=False)] = True
        g = rng.standard_normal(n)
        xl, xu = -np.ones(n), np.ones(n)
        x = np.clip(0.3 * rng.standard_normal(n), -0.9, 0.9)
        x[fix] = 1.0
        Z = np.zeros((0, n))
        cons_o = R.make_mixed_constraints(Z, R.chol_lower(Z @ Z.T), fix, l=xl, u=xu)
        Ho = R.AlHessian(J, Z, 1.0)
        delta = 0.5 * np.linalg.norm(g)
        wl, wu = R.build_step_bounds(x, cons_o, delta)
        w_ref, st_ref, it_ref = R.projected_cg(g, Ho, wl, wu, cons_o, kappa2)
        H = bh.AlHessian(J, None, 1.0)
        cons = bh.MixedConstraints(Z, None, fix, l=xl, u=xu)
        w, status, info = bh.projected_cg(g, H, wl, wu, cons, kappa2, full_output=True)
        assert int(status) == int(st_ref) and info["iters"] == it_ref, (d, n, int(status), int(st_ref), info["iters"], it_ref)
        tol = w_tolerance(g, Ho, wl, wu, cons_o, kappa2, w_ref)
        worst = max(worst, relnorm(w, w_ref) / tol)
        assert relnorm(w, w_ref) <= tol, (d, n, relnorm(w, w_ref), tol)
        s0 = np.zeros(n)
        wm_ref, stm_ref = R.minor_iterate(x, s0, g, Ho, cons_o, delta, kappa2)
        wm, stm, infom = bh.minor_iterate(x, s0, g, H, cons, delta, kappa2, full_output=True)
        assert int(stm) == int(stm_ref) and relnorm(wm, wm_ref) <= 10 * tol, (d, n, relnorm(wm, wm_ref), tol)
        n_checked += 1
        H.close(); cons.close()
    # random instances with linear equalities (the four-kernel RCCL form): few and 64 rows of A, fixed variables, a repeated call
    for d, n, mA, nfix, kappa2, seed in ((400, 130, 3, 7, 1e-6, 11), (2100, 1024, 64, 50, 1e-3, 12), (700, 301, 17, 0, 1e-4, 13)):
        rng = np.random.default_rng(seed)
        J = rng.standard_normal((d, n)) / np.sqrt(d)
        A = rng.standard_normal((mA, n))
        fix = np.zeros(n, dtype=bool)
        fix[rng.choice(n, nfix, replace=False)] = True
        g = rng.standard_normal(n)
        xl, xu = -np.ones(n), np.ones(n)
        x = np.clip(0.3 * rng.standard_normal(n), -0.9, 0.9)
        x[fix] = 1.0
        cons_o = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix if nfix else None, l=xl, u=xu)
        Ho = R.AlHessian(J, np.zeros((0, n)), 1.0)
        delta = 0.5 * np.linalg.norm(g)
        wl, wu = R.build_step_bounds(x, cons_o, delta)
        w_ref, st_ref, it_ref = R.projected_cg(g, Ho, wl, wu, cons_o, kappa2)
        H = bh.AlHessian(J, None, 1.0)
        cons = bh.MixedConstraints(A, None, fix if nfix else None, l=xl, u=xu)
        tol = w_tolerance(g, Ho, wl, wu, cons_o, kappa2, w_ref)
        for rep in range(2):            # the second call sizes its first batch from the first: one collective per H*p
            a0 = H.stats()["n_allreduce"]
            w, status, info = bh.projected_cg(g, H, wl, wu, cons, kappa2, full_output=True)
            assert int(status) == int(st_ref) and info["iters"] == it_ref, (d, n, mA, int(status), int(st_ref), info["iters"], it_ref)
            assert H.stats()["cg_kernels"] == 4
            worst = max(worst, relnorm(w, w_ref) / tol)
            assert relnorm(w, w_ref) <= tol, (d, n, mA, relnorm(w, w_ref), tol)
        if info["n_hmul"] <= 32:
            assert H.stats()["n_allreduce"] - a0 == info["n_hmul"], (H.stats()["n_allreduce"] - a0, info["n_hmul"])
        s0 = np.zeros(n)
        wm_ref, stm_ref = R.minor_iterate(x, s0, g, Ho, cons_o, delta, kappa2)
        wm, stm, infom = bh.minor_iterate(x, s0, g, H, cons, delta, kappa2, full_output=True)
        assert int(stm) == int(stm_ref) and relnorm(wm, wm_ref) <= 10 * tol, (d, n, mA, relnorm(wm, wm_ref), tol)
        n_checked += 1
        n_gen += 1
        H.close(); cons.close()
    assert n_gen >= 4
    bh._lib.check(bh._lib.lib().bh_comm_destroy(), "bh_comm_destroy")
    print("OK %d cases, worst w deviation %.0f %% of its tolerance" % (n_checked, 100.0 * worst))


if __name__ == "__main__":
    main()
