"""Scratch: box-constrained Cauchy search at config-3 scale, the three forms, no profiler attached."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh
bh.init(0)
syn = bh.synthetic
d, n = 65536, 4096
H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
x, x_l, x_u, fix = syn.box_vectors(n, fix_every=8)
g = H.jtv(syn.residual_rows(0, d))
for image, fused, name in ((1, 1, "row space, one kernel per breakpoint"), (1, 0, "row space, two kernels per breakpoint")):
    bh.set_option("cauchy_image", image); bh.set_option("cauchy_fused", fused)
    for dscale in (0.1, 1.0):
        best = 1e9
        for rep in range(4):
            cons = bh.MixedConstraints(np.zeros((0, n)), None, None, l=x_l, u=x_u)
            delta = dscale * syn.initial_tr(g)
            t0 = time.perf_counter()
            s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
            best = min(best, time.perf_counter() - t0)
        print("%-40s delta x%-4g %5d passes, %8.3f ms, %6.2f us per pass, |s| = %.9e" % (name, dscale, info["n_hmul"], 1e3 * best, 1e6 * best / info["n_hmul"], np.linalg.norm(s)), flush=True)
