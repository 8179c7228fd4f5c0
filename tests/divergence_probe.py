"""Where does a full solve on the device leave the oracle's trajectory, and why?  (manual diagnostic; uses the oracle, so
it lives under tests/.  `python tests/divergence_probe.py` on the GPU box.)

For BASELINE config 1 (sphere regression) and the 48-parameter NLS it prints
  * the reference's optimality measure of each C-ABI variant (test/problems/sphere_regression.jl:61-65 asserts < 1e-7),
  * the first DECISION of the restated driver that differs between the free-running device solve and the oracle solve,
    with the deciding scalar (tests/_util.py::first_decision_difference),
  * a SHADOW solve: device and oracle evaluated on identical operands at every hot-path call — any call whose status /
    iteration count / active set differs, with the device's tie log for it, and the closest branch margin of the solve.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import benlsip_ref as R                       # noqa: E402
import sphere_problem as sp                   # noqa: E402
from _util import first_decision_difference   # noqa: E402
from hip_ops import HipOps, HipOpsDeviceAll, HipOpsDeviceMinor, ShadowOps   # noqa: E402
from nls_problem import NLSProblem            # noqa: E402


def main():
    import benlsip_jl_amd as bh
    bh.init(0)
    out = {}
    # ---- config 1 -------------------------------------------------------------------------------------------------------
    kw = dict(max_outer_iter=100, max_inner_iter=250)
    log_ref = []
    xo, yo = R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u, log=log_ref, **kw)

    def opt(xs, ys):
        grad = sp.jac_r(xs).T @ sp.r(xs) + sp.jac_c(xs).T @ ys
        return float(np.linalg.norm(xs - R.projection_polyhedron_small(xs - grad, sp.A, sp.b, sp.x_l, sp.x_u)))
    print("[sphere] oracle: opt_measure %.3e, %d minor iterates" % (opt(xo, yo), sum(e[0] == "minor" for e in log_ref)))
    for cls in (HipOps, HipOpsDeviceMinor, HipOpsDeviceAll):
        ops, log = cls(bh), []
        xs, ys = R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u, ops=ops, log=log, **kw)
        diff = first_decision_difference(log_ref, log)
        print("[sphere] %-18s opt_measure %.3e  |c| %.2e  |x-x_oracle| %.2e  minor iterates %d  logged ties %d  first decision difference: %s"
              % (cls.__name__, opt(xs, ys), np.linalg.norm(sp.c(xs)), np.linalg.norm(xs - xo), sum(e[0] == "minor" for e in log), len(ops.ties), diff))
        out["sphere_" + cls.__name__] = dict(opt=opt(xs, ys), minor=sum(e[0] == "minor" for e in log), same_log_len=len(log) == len(log_ref))
    sh = ShadowOps(HipOpsDeviceAll(bh))
    xs, ys = R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u, ops=sh, **kw)
    print("[sphere] shadow solve: %d minor iterates, %d same-operand discrepancies; worst relative deviations %s; closest margin %s"
          % (sh.minor, len(sh.events), {k: "%.1e" % v for k, v in sh.worst.items()}, sh.min_margin))
    for e in sh.events[:30]:
        print("        ", {k: v for k, v in e.items() if k != "operands"})

    # ---- 48-parameter NLS ----------------------------------------------------------------------------------------------
    P = NLSProblem(256, 48, 2, seed=1)
    kw = dict(max_outer_iter=30, max_inner_iter=60)
    log_ref = []
    t0 = time.perf_counter()
    x_ref, y_ref = R.tralcnllss(P.x0, P.r, P.jac_r, P.c, P.jac_c, P.A, P.b, P.x_l, P.x_u, log=log_ref, **kw)
    t_cpu = time.perf_counter() - t0
    for cls in (HipOps, HipOpsDeviceAll):
        ops, log = cls(bh), []
        t0 = time.perf_counter()
        x, y = R.tralcnllss(P.x0, P.r, P.jac_r, P.c, P.jac_c, P.A, P.b, P.x_l, P.x_u, ops=ops, log=log, **kw)
        t_gpu = time.perf_counter() - t0
        diff = first_decision_difference(log_ref, log)
        print("[nls48] %-16s %d minor iterates (oracle %d), %.2f s (oracle %.2f s), |x - x_ref| %.2e, logged ties %d"
              % (cls.__name__, sum(e[0] == "minor" for e in log), sum(e[0] == "minor" for e in log_ref), t_gpu, t_cpu, np.linalg.norm(x - x_ref), len(ops.ties)))
        print("        first decision difference at log entry %s" % (None if diff is None else diff[0]))
        if diff is not None:
            k, a, b, why = diff
            print("        oracle entry %s\n        device entry %s\n        deciding: %s" % (a, b, why))
            nm = sum(e[0] == "minor" for e in log_ref[:k])
            same = all((ea[1], ea[2]) == (eb[1], eb[2]) for ea, eb in zip(log_ref[:k], log[:k]) if ea[0] == "minor")
            print("        before it: %d minor iterates with identical CG status and active-set size: %s" % (nm, same))
        for t in ops.ties[:5]:
            print("        tie:", t)
    sh = ShadowOps(HipOpsDeviceAll(bh))
    x, y = R.tralcnllss(P.x0, P.r, P.jac_r, P.c, P.jac_c, P.A, P.b, P.x_l, P.x_u, ops=sh, **kw)
    print("[nls48] shadow solve: %d minor iterates, %d same-operand discrepancies; worst relative deviations %s; closest margin %s"
          % (sh.minor, len(sh.events), {k: "%.1e" % v for k, v in sh.worst.items()}, sh.min_margin))
    for e in sh.events[:40]:
        print("        ", {k: v for k, v in e.items() if k != "operands"})
    cau = [e for e in sh.events if e["op"] == "cauchy_step"]
    if cau:
        worst = max(cau, key=lambda e: e["rel"] / max(e["oracle_sensitivity"], 1e-300))
        print("[nls48] worst Cauchy event relative to the oracle's own sensitivity:", {k: v for k, v in worst.items() if k != "operands"})
        op = worst["operands"]
        np.savez(os.path.join(ROOT, "gpurun_out", "cauchy_worst.npz"), A=P.A, x_l=P.x_l, x_u=P.x_u, **op)
        # the same operands through the device variants
        A = P.A
        L0 = R.chol_lower(A @ A.T)
        Ho = R.AlHessian(op["J"], op["C"], float(op["mu"]))
        Hd = bh.AlHessian(op["J"], op["C"], float(op["mu"]))
        for dd in (1, 0):
            bh.set_option("chol_downdate", dd)
            cons = bh.MixedConstraints(A, None, op["fix0"], l=P.x_l, u=P.x_u)
            sd, info = bh.cauchy_step(op["x"], op["g"], Hd, cons, float(op["delta"]), full_output=True)
            print("        device, chol_downdate=%d: %d breakpoints, |s - s_cpu|/|s_cpu| = %.2e, |s - s_dev(solve)|/|s| = %.2e"
                  % (dd, info["n_breakpoints"], np.linalg.norm(sd - op["s_cpu"]) / np.linalg.norm(op["s_cpu"]), np.linalg.norm(sd - op["s_dev"]) / np.linalg.norm(op["s_dev"])))
        bh.set_option("chol_downdate", 1)
        # the oracle in extended precision for the projections' inner products: where is the truth?
        lc = R.make_mixed_constraints(A, L0, l=P.x_l, u=P.x_u)
        R.active_bounds_inplace(lc, op["x"], L0)
        d0 = R.projection(lc, -op["g"])
        dd0 = bh.projection(bh.MixedConstraints(A, None, lc.fixvars, l=P.x_l, u=P.x_u), -op["g"])
        print("        initial direction d = P(-g): |g| = %.3e, |d_cpu| = %.3e, |d_dev - d_cpu| = %.3e" % (np.linalg.norm(op["g"]), np.linalg.norm(d0), np.linalg.norm(dd0 - d0)))
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "divergence_probe.json"), "w"), indent=1) if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None


if __name__ == "__main__":
    main()
