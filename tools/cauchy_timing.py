#!/usr/bin/env python3
"""Time the device-resident Cauchy step at BASELINE config-3 scale (d=65536, n=4096, box) and config-5 shape (mA=64)."""
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
    for mA, image in ((0, 1), (0, 0), (8, 1), (8, 0), (16, 1), (64, 0), (64, 1)):
        # box constraints: image-space search (default: J d, J s_c maintained by one-column updates, no sweep per breakpoint) and
        # the sweeping form (cauchy_image = 0: one H*d per breakpoint, as the reference); linear equalities: sweeping form only
        bh.set_option("cauchy_image", image)
        A = syn.splitmix_uniform(4, np.arange(mA * n)).reshape((mA, n), order="F")
        for dscale in (0.1, 1.0, 10.0):
            cons = bh.MixedConstraints(A, None, None, l=x_l, u=x_u)
            delta = dscale * syn.initial_tr(g)
            bh.cauchy_step(x, g, H, cons, delta)
            # (between 17 and 64 equalities the row-space form is chosen from the previous search on the SAME handle: reuse it)
            cons2 = cons if (image and mA > 16) else bh.MixedConstraints(A, None, None, l=x_l, u=x_u)
            t0 = time.perf_counter()
            s, info = bh.cauchy_step(x, g, H, cons2, delta, full_output=True)
            el = time.perf_counter() - t0
            print("mA=%d %s delta=%.3g: %d breakpoints, %d passes, %.3f ms total, %.1f us per pass, nfix %d -> %d, |s|=%.6e"
                  % (mA, "image-space" if image else "H*d sweep per breakpoint", delta, info["n_breakpoints"], info["n_hmul"], 1e3 * el,
                     1e6 * el / max(info["n_hmul"], 1), int(fix.sum()), cons2.nb_fix(), np.linalg.norm(s)), flush=True)
    bh.set_option("cauchy_image", 1)


if __name__ == "__main__":
    main()
