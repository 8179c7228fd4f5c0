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
        # asynchronous ingest: how long the caller is held, and whether host work hides behind the upload
        for chunk_mb in (16, 64, 256):
            bh.set_option("upload_chunk_mb", chunk_mb)
            t0 = time.perf_counter()
            H = bh.AlHessian.create_async(J, None, 1.0)
            t_ret = time.perf_counter() - t0
            busy = 0.0
            x = np.ones(1 << 18)
            while busy < 0.030:                              # ~30 ms of host work (what evaluate_al costs the caller)
                tb = time.perf_counter()
                x = np.sqrt(x * x + 1.0)
                busy += time.perf_counter() - tb
            t1 = time.perf_counter()
            H.wait()
            t_wait = time.perf_counter() - t1
            tot = time.perf_counter() - t0
            print("d=%d n=%d async (chunks of %d MiB): create returned after %.2f ms, %.1f ms of host work, wait %.1f ms more, total %.1f ms"
                  % (d, n, chunk_mb, 1e3 * t_ret, 1e3 * busy, 1e3 * t_wait, 1e3 * tot), flush=True)
            H.close()
        bh.set_option("upload_chunk_mb", 64)


if __name__ == "__main__":
    main()
