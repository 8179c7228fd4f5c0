#!/usr/bin/env python3
"""BASELINE config 2 (d=8192, n=1024, box, p=128): J = 64 MiB fits the 256 MiB Infinity Cache, so GB/s here is cache
bandwidth ("effective"), never an HBM-roofline fraction."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402
import bench  # noqa: E402


def main():
    bh.init(0)
    for kind, name in ((0, "wc"), (1, "ic")):
        H, cons, dv, host = bench.setup_instance(bh, 0, 1, kind, d_per_gpu=8192, n=1024)
        bench.run_steps(bh, H, cons, dv, 0.1, 5)
        bh._lib.lib().bh_synchronize()
        t0 = time.perf_counter()
        steps = 200
        st, it, nh = bench.run_steps(bh, H, cons, dv, 0.1, steps)
        el = (time.perf_counter() - t0) / steps
        ms = [H.time_kernel(k, 50) for k in (0, 1, 2)]
        gb = [8.0 * 8192 * 1024 / (m * 1e-3) / 1e9 for m in ms]
        print("config2 %s: %s iters=%d n_hmul=%d  %.1f us per subproblem (%.0f/s), %.1f us per CG iteration; kernels fused/jv/jtv %s us = %s GB/s effective"
              % (name, st.name, it, nh, 1e6 * el, 1 / el, 1e6 * el / max(nh, 1), ["%.1f" % (1e3 * m) for m in ms], ["%.0f" % g for g in gb]), flush=True)


if __name__ == "__main__":
    main()
