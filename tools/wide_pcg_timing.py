#!/usr/bin/env python3
"""projected_cg per-iteration cost for wide Jacobians (2 GiB images, box bounds): n = 8192 (<512,8,2> geometry) and n = 16384
(<512,16,1>, v parked in LDS) in the two-kernel iteration (cg_fused = 1: the p-update in the prologue of the H*p launch) and in the
three-kernel one (cg_fused = 0).  The fused variants of these geometries exceed the register budget and spill; this says whether that costs.

    python tools/wide_pcg_timing.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import benlsip_jl_amd as bh
    import bench
    bh.init(0)
    for n in (8192, 16384):
        d = (2 << 30) // (8 * n)
        for fused in (1, 0, 1, 0):
            bh.set_option("cg_fused", fused)
            H, cons, dv, _ = bench.setup_instance(bh, 0, 1, 1, d_per_gpu=d, n=n)
            bench.run_steps(bh, H, cons, dv, 0.1, 3)
            bh._lib.lib().bh_synchronize()
            t0 = time.perf_counter()
            st, it, nh = bench.run_steps(bh, H, cons, dv, 0.1, 5)
            bh._lib.lib().bh_synchronize()
            el = (time.perf_counter() - t0) / 5
            print("n=%5d d=%6d cg_fused=%d: %s, %d H*p, %.1f us per H*p = %.0f GB/s (kernels per iteration: %d)"
                  % (n, d, fused, st.name, nh, 1e6 * el / max(nh, 1), H.stats()["bytes_per_hmul"] / (el / max(nh, 1)) / 1e9,
                     H.stats()["cg_kernels"]), flush=True)
            H.close()
            cons.close()
    bh.set_option("cg_fused", 1)


if __name__ == "__main__":
    main()
