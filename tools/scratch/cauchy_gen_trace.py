"""Scratch: the row-space Cauchy search with linear equalities at config-3 scale (mA from argv), two searches (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh
bh.init(0)
syn = bh.synthetic
mA = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d, n = 65536, 4096
H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
x, x_l, x_u, fix = syn.box_vectors(n, fix_every=8)
g = H.jtv(syn.residual_rows(0, d))
A = syn.splitmix_uniform(4, np.arange(mA * n)).reshape((mA, n), order="F")
for rep in range(2):
    cons = bh.MixedConstraints(A, None, None, l=x_l, u=x_u)
    delta = 1.0 * syn.initial_tr(g)
    t0 = time.perf_counter()
    s, info = bh.cauchy_step(x, g, H, cons, delta, full_output=True)
    el = time.perf_counter() - t0
    print("mA=%d: %d passes, %.3f ms, %.2f us per pass" % (mA, info["n_hmul"], 1e3 * el, 1e6 * el / info["n_hmul"]), flush=True)
