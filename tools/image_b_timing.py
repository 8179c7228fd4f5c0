#!/usr/bin/env python3
"""The set-up GEMM of the row-space Cauchy search (image_b_mfma_kernel: B = J (D A'), 65536 x 4096 times 4096 x 64, fp64 MFMA) in
isolation: a handful of short Cauchy searches at the config-5 shape (tiny trust region: few breakpoints), so that a profiler sees
a few dispatches of the kernel and little else.  Used by tools/image_b_counters.sh."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402


def main():
    bh.init(0)
    syn = bh.synthetic
    d, n, mA = 65536, 4096, 64
    H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
    x, x_l, x_u, fix = syn.box_vectors(n, fix_every=8)
    g = H.jtv(syn.residual_rows(0, d))
    A = syn.splitmix_uniform(4, np.arange(mA * n)).reshape((mA, n), order="F")
    delta = 1e-6 * syn.initial_tr(g)
    for gemm in (1, 0):
        bh.set_option("cauchy_gemm", gemm)
        for rep in range(4):
            cons = bh.MixedConstraints(A, None, None, l=x_l, u=x_u)
            t0 = time.perf_counter()
            s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
            el = time.perf_counter() - t0
            if rep:
                print("cauchy_gemm=%d: %d passes, %.2f ms per search (set-up included)" % (gemm, info["n_hmul"], 1e3 * el), flush=True)
    bh.set_option("cauchy_gemm", 1)


if __name__ == "__main__":
    main()
