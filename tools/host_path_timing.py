#!/usr/bin/env python3
"""Host-pointer API (what the Julia shim uses) vs device-pointer API on the config-3 instance."""
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
    lib = bh._lib.lib()
    H, cons, dv, host = bench.setup_instance(bh, 0, 1, 0)
    g, w_l, w_u = host["g"], host["w_l"], host["w_u"]
    x, x_l, x_u = host["x"], host["x_l"], host["x_u"]
    cons_full = bh.MixedConstraints(np.zeros((0, 4096)), None, host["fix"], l=x_l, u=x_u)
    delta = bh.synthetic.initial_tr(g)
    s0 = np.zeros(4096)

    def timeit(fn, reps=200):
        for _ in range(5):
            fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return 1e3 * (time.perf_counter() - t0) / reps

    print("bh_pcg_dev          %.4f ms" % timeit(lambda: bench.run_steps(bh, H, cons, dv, 0.1, 1)))
    print("bh_pcg (host ptrs)  %.4f ms" % timeit(lambda: bh.projected_cg(g, H, w_l, w_u, cons, 0.1)))
    print("bh_minor_iterate    %.4f ms" % timeit(lambda: bh.minor_iterate(x, s0, g, H, cons_full, delta, 0.1)))
    print("bh_hmul (host)      %.4f ms" % timeit(lambda: bh.hmul(H, g)))
    print("bh_vthv (host)      %.4f ms" % timeit(lambda: bh.vthv(H, g)))
    print("bh_project (host)   %.4f ms" % timeit(lambda: bh.projection(cons, g)))
    print("bh_hmul_add (host)  %.4f ms" % timeit(lambda: bh.hmul_add(H, g, g)))


if __name__ == "__main__":
    main()
