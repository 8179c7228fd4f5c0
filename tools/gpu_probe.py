#!/usr/bin/env python3
"""First-contact GPU probe: self-test, small parity checks against the NumPy oracle, kernel timings."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import benlsip_jl_amd as bh  # noqa: E402
import benlsip_ref as R  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def main():
    lib = bh.init()
    print("selftest rc", lib.bh_selftest(), lib.bh_last_error_detail())
    rng = np.random.default_rng(0)
    for (d, n, q) in [(4, 3, 1), (37, 5, 2), (300, 130, 0), (1000, 1024, 3), (513, 2049, 0), (256, 4096, 0), (200, 8000, 1)]:
        J = rng.standard_normal((d, n)); Cm = rng.standard_normal((q, n)); mu = 0.7
        v = rng.standard_normal(n); u = rng.standard_normal(d)
        H = bh.AlHessian(J, Cm, mu)
        Ho = R.AlHessian(J, Cm, mu)
        e = [rel(H.jv(v), J @ v), rel(H.jtv(u), J.T @ u), rel(H * v, R.hmul(Ho, v)), abs(bh.vthv(H, v) - R.vthv(Ho, v)) / R.vthv(Ho, v)]
        print("matvec", (d, n, q), ["%.2e" % x for x in e])
        H.close()
    # HS48
    A = np.array([[1., 1, 1, 1, 1], [0, 0, 1, -2, -2]])
    L0 = R.chol_lower(A @ A.T)
    fix = np.array([True, True, False, False, False])
    lo = R.make_mixed_constraints(A, L0, fix)
    lc = bh.MixedConstraints(A, lo.chol_L, fix)
    x = np.array([3., 5, -3, 2, -2])
    print("HS48", bh.projection(lc, x), bh.left_mul(lc, x), bh.left_mul_tr(lc, np.arange(4.0)), R.left_mul_tr(lo, np.arange(4.0)))
    # general projection, larger
    n, mA = 300, 20
    A = rng.standard_normal((mA, n)); L0 = R.chol_lower(A @ A.T)
    fix = np.zeros(n, bool); fix[rng.choice(n, 150, replace=False)] = True
    lo = R.make_mixed_constraints(A, L0, fix); lc = bh.MixedConstraints(A, lo.chol_L, fix)
    r = rng.standard_normal(n)
    print("proj general", rel(bh.projection(lc, r), R.projection(lo, r)))
    lo2 = R.make_mixed_constraints(A, L0); lc2 = bh.MixedConstraints(A, L0)
    print("proj nullspace", rel(bh.projection(lc2, r), R.projection(lo2, r)))
    # pcg small
    for (d, n, mA, nf) in [(50, 20, 0, 4), (200, 64, 3, 10), (4, 3, 1, 0), (300, 100, 0, 0)]:
        J = rng.standard_normal((d, n)) / np.sqrt(d); Cm = rng.standard_normal((1, n)); mu = 10.0
        A = rng.standard_normal((mA, n)); L0 = R.chol_lower(A @ A.T)
        fix = np.zeros(n, bool); fix[rng.choice(n, nf, replace=False)] = True
        lo = R.make_mixed_constraints(A, L0, fix if nf else None)
        lc = bh.MixedConstraints(A, lo.chol_L, fix)
        g = rng.standard_normal(n)
        wl, wu = R.build_step_bounds(np.zeros(n), R.MixedConstraints(A, -np.ones(n), np.ones(n), fix, None), 0.5)
        Ho = R.AlHessian(J, Cm, mu); H = bh.AlHessian(J, Cm, mu)
        tr = R.CGTrace()
        w0, s0, it0 = R.projected_cg(g, Ho, wl, wu, lo, 0.1, trace=tr)
        w1, s1, info = bh.projected_cg(g, H, wl, wu, lc, 0.1, trace_cap=64, full_output=True)
        print("pcg", (d, n, mA, nf), int(s0), it0, "|", int(s1), info["iters"], info["n_hmul"], "relw %.2e" % rel(w1, w0))
    # timings, config 3
    d, n = 65536, 4096
    t0 = time.time()
    H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
    print("synthetic create s", time.time() - t0)
    Jsmall = R.synthetic_J(8, n, seed=1, d_total=d)
    v = rng.standard_normal(n)
    print("synth check", rel(H.jv(v)[:8], Jsmall @ v))
    res = {}
    for variant in [0, 1, 2, 3, 4]:
        lib.bh_set_option(b"rs_variant", variant)
        for bpc in [1, 2, 3, 4]:
            lib.bh_set_option(b"blocks_per_cu", bpc)
            ms = [H.time_kernel(k, 10) for k in (0, 1, 2)]
            gb = [8.0 * d * n / (m * 1e-3) / 1e9 for m in ms]
            res["v%d_b%d" % (variant, bpc)] = gb
            print("variant", variant, "bpc", bpc, "ms", ["%.3f" % m for m in ms], "GB/s", ["%.0f" % g for g in gb], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "probe_variants.json"), "w"))


if __name__ == "__main__":
    main()
