#!/usr/bin/env python3
"""Kernel rates for wide Jacobians (2 GiB images): n = 8192 (registers), 12288 and 16384 (v parked in LDS), 20480 (column
panels, two passes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402


def main():
    bh.init(0)
    for n in (8192, 12288, 16384, 20480):
        d = (2 << 30) // (8 * n)
        H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
        if n <= 16384:
            ms = [min(H.time_kernel(k, 10) for _ in range(2)) for k in (0, 1, 2)]
            print("n=%5d d=%6d: fused %.3f ms = %.0f GB/s | J v %.3f ms = %.0f GB/s | J'u %.3f ms = %.0f GB/s"
                  % ((n, d) + tuple(x for m in ms for x in (m, 8.0 * d * n / m / 1e6))), flush=True)
        import time
        import numpy as np
        v = np.ones(n)
        H * v
        t0 = time.perf_counter()
        for _ in range(10):
            H * v
        el = (time.perf_counter() - t0) / 10
        print("n=%5d d=%6d: bh_hmul (host vectors) %.3f ms per H*v = %.0f GB/s of J per product" % (n, d, 1e3 * el, 8.0 * d * n / el / 1e9), flush=True)
        H.close()


if __name__ == "__main__":
    main()
