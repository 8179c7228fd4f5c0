#!/usr/bin/env python3
"""What a projected_cg call costs beyond its CG iterations: time whole bh_pcg_dev calls on instances that
take different numbers of iterations and fit  t(call) = a + m * b  (m = H*p products per call) — b is the steady-state
iteration, a the per-call cost (first-launch enqueue, the launch that finds the loop finished, the host's poll and return,
the init kernels of the general-constraint form).  BASELINE configs 3 (box) and 5 (64 linear equalities), J = 65536 x 4096.

    python tools/percall_breakdown.py [--calls 200]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402
import bench  # noqa: E402


def time_calls(H, cons, dv, kappa2, calls):
    bench.run_steps(bh, H, cons, dv, kappa2, 5)
    bh._lib.lib().bh_synchronize()
    per = []
    for _ in range(calls):
        t0 = time.perf_counter()
        st, it, nh = bench.run_steps(bh, H, cons, dv, kappa2, 1)
        per.append(time.perf_counter() - t0)
    # back-to-back as bench.py times them (one clock around all calls)
    t0 = time.perf_counter()
    bench.run_steps(bh, H, cons, dv, kappa2, calls)
    bh._lib.lib().bh_synchronize()
    b2b = (time.perf_counter() - t0) / calls
    per = np.sort(np.asarray(per))
    return dict(status=st.name, n_hmul=nh, median_us=1e6 * float(np.median(per)), p10_us=1e6 * float(per[len(per) // 10]),
                back_to_back_us=1e6 * b2b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--configs", default="3,5")
    a = ap.parse_args()
    bh.init(0)
    for config in [int(c) for c in a.configs.split(",")]:
        rows = []
        for kind, kappas in ((0, (0.1, 1e-3)), (1, (0.1,))):
            H, cons, dv, _ = bench.setup_instance(bh, 0, 1, kind, config=config)
            for k2 in kappas:
                r = time_calls(H, cons, dv, k2, a.calls)
                r.update(variant="wc" if kind == 0 else "ic", kappa2=k2)
                rows.append(r)
                print("config %d %s kappa2=%g: %s, %d H*p per call: median %.1f us, p10 %.1f us, back-to-back %.1f us per call"
                      % (config, r["variant"], k2, r["status"], r["n_hmul"], r["median_us"], r["p10_us"], r["back_to_back_us"]), flush=True)
            H.close()
        m = np.array([r["n_hmul"] for r in rows], dtype=float)
        for key in ("back_to_back_us", "median_us"):
            t = np.array([r[key] for r in rows])
            A = np.vstack([np.ones_like(m), m]).T
            (a0, b0), *_ = np.linalg.lstsq(A, t, rcond=None)
            print("config %d fit on %s:  t(call) = %.1f us + m * %.1f us   (residuals %s)"
                  % (config, key, a0, b0, ", ".join("%.1f" % x for x in (t - A @ np.array([a0, b0])))), flush=True)


if __name__ == "__main__":
    main()
