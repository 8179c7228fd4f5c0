"""Scratch: fused box Cauchy search at config-3 scale for several grids of cauchy_fused_kernel."""
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
for grid in (0, 16, 32, 64, 128, 256):
    bh.set_option("cauchy_fused_grid", grid)
    best = 1e9
    for rep in range(3):
        cons = bh.MixedConstraints(np.zeros((0, n)), None, None, l=x_l, u=x_u)
        delta = 1.0 * syn.initial_tr(g)
        t0 = time.perf_counter()
        s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
        best = min(best, time.perf_counter() - t0)
    print("grid %3d: %5d passes, %8.3f ms, %6.2f us per pass, |s| = %.9e" % (grid, info["n_hmul"], 1e3 * best, 1e6 * best / info["n_hmul"], np.linalg.norm(s)), flush=True)
