"""Scratch: one case of tests/multirank/fuzz_cases.py on ONE rank, device vs oracle (python tests/manual/fuzz_case_probe.py seed k)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/manual -> repo root
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "multirank")):
    sys.path.insert(0, p)
import numpy as np
import benlsip_ref as R
import benlsip_jl_amd as bh
from fuzz_cases import cases
seed, kk = int(sys.argv[1]), int(sys.argv[2])
bh.init(0)
for c in cases(seed, kk + 1):
    if c["k"] != kk:
        continue
    n = c["n"]
    print({k: c[k] for k in ("n", "d", "q", "mA", "delta", "kappa2", "mu")}, "nfix", int(c["fix"].sum()))
    Ho = R.AlHessian(c["J"], c["C"], c["mu"])
    L0 = R.chol_lower(c["A"] @ c["A"].T)
    fixed = c["fix"] if c["fix"].any() else None
    cons_o = R.make_mixed_constraints(c["A"], L0, fixed, l=c["xl"], u=c["xu"])
    H = bh.AlHessian(c["J"], c["C"], c["mu"])
    for fused in (1, 0):
        bh.set_option("cg_fused", fused)
        cons = bh.MixedConstraints(c["A"], None, c["fix"], l=c["xl"], u=c["xu"])
        w, st, info = bh.projected_cg(c["g"], H, c["wl"], c["wu"], cons, c["kappa2"], trace_cap=64, full_output=True)
        wo, sto, ito = R.projected_cg(c["g"], Ho, c["wl"], c["wu"], cons_o, c["kappa2"])
        print("cg_fused", fused, "pcg device", int(st), info["iters"], info["n_hmul"], "oracle", int(sto), ito, "finite", np.isfinite(w).all(), np.isfinite(wo).all())
        print(" trace", info["trace"][:4])
        wm, stm, infom = bh.minor_iterate(c["x"], np.zeros(n), c["g"], H, cons, c["delta"], c["kappa2"], full_output=True)
        wmo, stmo = R.minor_iterate(c["x"], np.zeros(n), c["g"], Ho, cons_o, c["delta"], c["kappa2"])
        print(" minor device", int(stm), infom, "finite", np.isfinite(wm).all(), "oracle", int(stmo), np.isfinite(wmo).all())
        if np.isfinite(wm).all():
            print(" rel", np.linalg.norm(wm - wmo) / np.linalg.norm(wmo))
        cons.close()
