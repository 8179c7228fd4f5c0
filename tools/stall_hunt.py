#!/usr/bin/env python3
"""Look for intermittent multi-millisecond stalls in back-to-back bh_pcg_dev calls (wc box instance)."""
import gc
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402
import bench  # noqa: E402


def main():
    bh.init(0)
    H, cons, dv, host = bench.setup_instance(bh, 0, 1, 0)
    for label, disable_gc in (("gc on", False), ("gc off", True)):
        if disable_gc:
            gc.disable()
        bench.run_steps(bh, H, cons, dv, 0.1, 5)
        per = []
        for _ in range(3000):
            t0 = time.perf_counter()
            bench.run_steps(bh, H, cons, dv, 0.1, 1)
            per.append(time.perf_counter() - t0)
        per = np.array(per) * 1e3
        out = np.flatnonzero(per > 2 * np.median(per))
        print(label, "median %.3f ms mean %.3f ms max %.2f ms; outliers (idx, ms):" % (np.median(per), per.mean(), per.max()),
              [(int(i), round(float(per[i]), 2)) for i in out[:20]], flush=True)
        gc.enable()


if __name__ == "__main__":
    main()
