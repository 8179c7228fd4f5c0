"""Exploration (GPU): ShadowOps on sphere regression (config 1, n = 3) for every ops variant x cg_fused 0/1/2 — which calls
deviate from the oracle on identical operands, by how much, relative to what."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, os.path.join(ROOT, "oracle"), TESTS):
    sys.path.insert(0, p)
import benlsip_jl_amd as bh
import benlsip_ref as R
import sphere_problem as sp
from hip_ops import HipOps, HipOpsDeviceAll, HipOpsDeviceMinor, ShadowOps

bh.init(0)
for fused in (0, 1, 2):
    bh.set_option("cg_fused", fused)
    for cls in (HipOps, HipOpsDeviceMinor, HipOpsDeviceAll):
        sh = ShadowOps(cls(bh), relnorm_tol=1e-12)
        xs, ys = R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u, max_outer_iter=100, max_inner_iter=250, ops=sh)
        grad = sp.jac_r(xs).T @ sp.r(xs) + sp.jac_c(xs).T @ ys
        opt = float(np.linalg.norm(xs - R.projection_polyhedron_small(xs - grad, sp.A, sp.b, sp.x_l, sp.x_u)))
        print("cg_fused=%d %-18s minor=%d events=%d opt=%.3e worst=%s" % (fused, cls.__name__, sh.minor, len(sh.events), opt,
              {k: float("%.1e" % v) for k, v in sh.worst.items()}), flush=True)
        for e in sorted(sh.events, key=lambda e: -e["rel"])[:6]:
            print("     ", {k: (float("%.3g" % v) if isinstance(v, float) else v) for k, v in e.items() if k not in ("operands", "ties")})
bh.set_option("cg_fused", 1)
