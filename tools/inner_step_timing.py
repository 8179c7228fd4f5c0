#!/usr/bin/env python3
"""One whole device-resident inner step (src/basic_tralcnlss.jl:394-460: Cauchy search, H*s+g, minor iterates, model reduction) at
config-3 scale (J 65536 x 4096, box bounds) through bh.inner_step, with the round-3 forms on and off:
  cauchy_image = 1 / 0      Cauchy search in the row space of J / one H*d sweep per breakpoint
(the H*w reuse for g_minor and the model reduction from g_minor are part of bh.inner_step; their per-call effect is in
tools/minor_loop_breakdown.py).

    python tools/inner_step_timing.py
"""
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
    d, n = 65536, 4096
    H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
    x, x_l, x_u, fix = syn.box_vectors(n, fix_every=8)
    g = H.jtv(syn.residual_rows(0, d))
    mA = int(sys.argv[1]) if len(sys.argv) > 1 else 0          # python tools/inner_step_timing.py 64: the config-5 shape
    A = syn.splitmix_uniform(4, np.arange(mA * n)).reshape((mA, n), order="F") if mA else np.zeros((0, n))
    for dscale in (1.0, 0.1):
        delta = dscale * syn.initial_tr(g)
        for image in (1, 0, 1, 0):
            bh.set_option("cauchy_image", image)
            cons = bh.MixedConstraints(A, None, None, l=x_l, u=x_u)
            t0 = time.perf_counter()
            s, mr, info = bh.inner_step(x, g, H, cons, delta, 50, 0.1, 0.1, full_output=True)
            el = time.perf_counter() - t0
            print("mA = %d, delta = %.3g, cauchy_image = %d: %d Cauchy breakpoints, %d minor iterates, %d active bounds, model reduction %.6e: %.2f ms"
                  % (mA, delta, image, info["n_breakpoints"], len(info["minor"]), cons.nb_fix(), mr, 1e3 * el), flush=True)
    bh.set_option("cauchy_image", 1)


if __name__ == "__main__":
    main()
