#!/usr/bin/env python3
"""Per-call cost of the projection kernels for mA = 64, 128, 256, 512 at n = 4096 (reduced form) — run under rocprofv3
--kernel-trace --stats to see the per-kernel split, or plain for the end-to-end bh_project_dev time."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402


def main():
    if os.environ.get("BH_AB_LIB"):          # A/B against another build of the library
        bh._lib._build.OUT = os.path.abspath(os.environ["BH_AB_LIB"])
    bh.init(0)
    lib = bh._lib.lib()
    syn = bh.synthetic
    n = 4096
    for mA in (64, 128, 256, 512):
        A = syn.splitmix_uniform(4, np.arange(mA * n)).reshape((mA, n), order="F")
        fix = np.zeros(n, dtype=bool)
        fix[::8] = True
        cons = bh.MixedConstraints(A, None, fix)
        r = bh.DeviceVector(n, np.ones(n))
        v = bh.DeviceVector(n)
        h = cons.handle
        for _ in range(5):
            lib.bh_project_dev(h, r.ptr, v.ptr)
        ts = []
        for _ in range(200):
            t0 = time.perf_counter()
            lib.bh_project_dev(h, r.ptr, v.ptr)
            ts.append(time.perf_counter() - t0)
        out = v.download()
        # median: the box shows an occasional 30-80 ms host stall (CPU quota), which would dominate a mean over 200 calls
        print("mA=%4d: median %.1f us, max %.0f us per projection (incl. host sync), |A v| = %.2e"
              % (mA, 1e6 * sorted(ts)[len(ts) // 2], 1e6 * max(ts), np.linalg.norm(A @ out)), flush=True)


if __name__ == "__main__":
    main()
