#!/usr/bin/env python3
"""The RCCL call site with ONE rank (BH_FORCE_COMM=1 BH_COMM=rccl: real librccl, a 1-rank ncclAllReduce per H*p on the library
stream): per-iteration time of the two-kernel form (H*p launch with the update of the previous iteration in its prologue + slab
reduction) against the three-kernel form (cg_fused = 0), config 3, 2- and 23-iteration instances.  RCCL elides a 1-rank
all-reduce, so this measures the launch structure around the collective, not the collective.

    BH_FORCE_COMM=1 BH_COMM=rccl python tools/rccl_iteration_timing.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402
import bench  # noqa: E402


def main():
    assert os.environ.get("BH_FORCE_COMM") == "1", "run with BH_FORCE_COMM=1 BH_COMM=rccl"
    bh.init(0)
    bh.init_distributed(0, 1, lambda b: b)
    for kind, name in ((0, "wc"), (1, "ic")):
        H, cons, dv, _ = bench.setup_instance(bh, 0, 1, kind)
        for rnd in range(2):
            for fused in (1, 0):
                bh.set_option("cg_fused", fused)
                bench.run_steps(bh, H, cons, dv, 0.1, 5)
                bh._lib.lib().bh_synchronize()
                a0 = H.stats()["n_allreduce"]
                t0 = time.perf_counter()
                st, it, nh = bench.run_steps(bh, H, cons, dv, 0.1, 50)
                bh._lib.lib().bh_synchronize()
                el = (time.perf_counter() - t0) / 50
                print("%s cg_fused=%d (%s): %s, %d H*p, %.1f us per subproblem, %.1f us per iteration, %.2f all-reduces per H*p"
                      % (name, fused, "two kernels + collective" if fused else "three kernels + collective", st.name, nh, 1e6 * el, 1e6 * el / nh,
                         (H.stats()["n_allreduce"] - a0) / (50.0 * nh)), flush=True)
        bh.set_option("cg_fused", 1)
        H.close()
    bh._lib.lib().bh_comm_destroy()


if __name__ == "__main__":
    main()
