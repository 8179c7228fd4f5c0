#!/usr/bin/env python3
"""Device-memory leak check: 240 create / use / destroy cycles of AlHessian (synchronous and asynchronous ingest) and
MixedConstraints handles of random sizes, with a projection, a projected_cg and a Cauchy search on each; hipMemGetInfo after every 40 (free memory must come back to where it was, give or
take the image pool's parked buffers)."""
import ctypes as C, sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import benlsip_jl_amd as bh
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 2**20
bh.init(0)
rng = np.random.default_rng(0)
base = None
for rep in range(6):
    for k in range(40):
        d, n = int(rng.integers(50, 3000)), int(rng.integers(8, 900))
        J = rng.standard_normal((d, n))
        H = bh.AlHessian(J, None, 1.0) if k % 2 else bh.AlHessian.create_async(J, None, 1.0)
        v = rng.standard_normal(n)
        _ = H * v
        mA = int(rng.integers(0, 4))
        cons = bh.MixedConstraints(rng.standard_normal((mA, n)), None, None)
        fix = np.zeros(n, dtype=bool); fix[::7] = True
        cons.set_active(fix, None)
        _ = bh.projection(cons, v)
        w, st = bh.projected_cg(v, H, -np.ones(n), np.ones(n), cons, 0.1)
        # the Cauchy search allocates its row-space vectors (and, with equalities, the rows x (1 + mA) set-up block) on the handle
        cau = bh.MixedConstraints(rng.standard_normal((mA, n)), None, None, l=-np.ones(n), u=np.ones(n))
        _ = bh.cauchy_step(np.zeros(n), v, H, cau, 0.3 * np.linalg.norm(v))
        H.close(); cons.close(); cau.close()
    bh._lib.lib().bh_synchronize()
    f = free_mb()
    if base is None: base = f
    print("after %d cycles: free %.1f MiB (delta %.1f)" % (40 * (rep + 1), f, f - base), flush=True)
