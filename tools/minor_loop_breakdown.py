#!/usr/bin/env python3
"""Per-call cost of the device-resident minor-loop body on the config-3 instance (src/basic_tralcnlss.jl:434-447)."""
import ctypes as ct
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
    n = 4096
    g, x, x_l, x_u, fix = host["g"], host["x"], host["x_l"], host["x_u"], host["fix"]
    delta = bh.synthetic.initial_tr(g)
    s0 = np.zeros(n)
    d = {k: bh.DeviceVector(n, v) for k, v in (("x", x), ("s", s0), ("g", g), ("xl", x_l), ("xu", x_u))}
    d["w"], d["gm"] = bh.DeviceVector(n), bh.DeviceVector(n)
    chunks = np.zeros(n // 64, dtype=np.uint64)
    P = bh.MixedConstraints(np.zeros((0, n)), None, fix, l=x_l, u=x_u)
    Ph = P.handle
    st, it, nh, al = ct.c_int32(), ct.c_int32(), ct.c_int32(), ct.c_double()
    na, nf, br = ct.c_int32(), ct.c_int32(), ct.c_int32()
    a = ct.c_double()
    eps = 1.4901161193847656e-08
    calls = {
        "bh_pcg_dev (2 H*p)": lambda: bench.run_steps(bh, H, cons, dv, 0.1, 1),
        "bh_minor_iterate_dev": lambda: lib.bh_minor_iterate_dev(H.handle, Ph, d["x"].ptr, d["s"].ptr, d["g"].ptr, d["xl"].ptr, d["xu"].ptr, delta, 0.1,
                                                                 eps, 1e-10, d["w"].ptr, ct.byref(st), ct.byref(it), ct.byref(nh), ct.byref(al)),
        "bh_step_accumulate_dev (explicit: 1 H*p)": lambda: lib.bh_step_accumulate_dev(H.handle, d["s"].ptr, d["w"].ptr, d["g"].ptr, d["gm"].ptr),
        # the pair as the inner step calls it (s and g_minor re-uploaded first so that every repetition is the same minor iterate),
        # and the same without the accumulate: the difference is bh_step_accumulate_dev with H*w taken from the CG loop
        "upload s, g_minor + bh_minor_iterate_dev": lambda: (
            d["s"].upload(s0), d["gm"].upload(g),
            lib.bh_minor_iterate_dev(H.handle, Ph, d["x"].ptr, d["s"].ptr, d["gm"].ptr, d["xl"].ptr, d["xu"].ptr, delta, 0.1,
                                     eps, 1e-10, d["w"].ptr, ct.byref(st), ct.byref(it), ct.byref(nh), ct.byref(al))),
        "upload s, g_minor + bh_minor_iterate_dev + bh_step_accumulate_dev (H*w from the CG loop)": lambda: (
            d["s"].upload(s0), d["gm"].upload(g),
            lib.bh_minor_iterate_dev(H.handle, Ph, d["x"].ptr, d["s"].ptr, d["gm"].ptr, d["xl"].ptr, d["xu"].ptr, delta, 0.1,
                                     eps, 1e-10, d["w"].ptr, ct.byref(st), ct.byref(it), ct.byref(nh), ct.byref(al)),
            lib.bh_step_accumulate_dev(H.handle, d["s"].ptr, d["w"].ptr, d["g"].ptr, d["gm"].ptr)),
        "bh_hmul_dev (1 H*p)": lambda: lib.bh_hmul_dev(H.handle, d["s"].ptr, d["gm"].ptr),
        "bh_proj_update_active_dev": lambda: lib.bh_proj_update_active_dev(Ph, d["x"].ptr, d["s"].ptr, d["xl"].ptr, d["xu"].ptr, delta, eps,
                                                                           ct.byref(na), ct.byref(nf), ct.byref(br), bh._lib.ptr(chunks)),
        "bh_reduced_gradient_norm_dev": lambda: lib.bh_reduced_gradient_norm_dev(Ph, d["g"].ptr, ct.byref(a)),
        "bh_model_reduction_dev (1 J*v)": lambda: lib.bh_model_reduction_dev(H.handle, d["g"].ptr, d["s"].ptr, ct.byref(a)),
        "bh_synchronize (idle stream)": lambda: lib.bh_synchronize(),
    }
    lib.bh_set_option(b"step_from_cg", 1)          # as the resident inner-step mirrors do around their minor loop
    for name, fn in calls.items():
        for _ in range(5):
            fn()
        d["s"].upload(s0)
        P.fixvars = fix
        _ = P.handle
        t0 = time.perf_counter()
        reps = 100
        for _ in range(reps):
            fn()
        lib.bh_synchronize()          # (calls that owe the host nothing return once their work is enqueued)
        print("%-92s %8.1f us" % (name, 1e6 * (time.perf_counter() - t0) / reps), flush=True)


if __name__ == "__main__":
    main()
