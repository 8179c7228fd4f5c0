#!/usr/bin/env python3
"""Time bh_hess_create (PCIe upload + device transpose) for a 2 GiB column-major J handed over from pageable host memory."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402


def main():
    bh.init(0)
    for d, n in ((8192, 1024), (65536, 4096)):
        J = np.asfortranarray(np.random.default_rng(0).standard_normal((d, n)))
        for rep in range(3):
            t0 = time.perf_counter()
            H = bh.AlHessian(J, None, 1.0)
            el = time.perf_counter() - t0
            print("d=%d n=%d: bh_hess_create %.1f ms -> %.1f GB/s" % (d, n, 1e3 * el, J.nbytes / el / 1e9), flush=True)
            v = np.ones(n)
            if rep == 0:
                print("   check", np.linalg.norm(H.jv(v)[:5] - (J[:5] @ v)))
            H.close()


if __name__ == "__main__":
    main()
