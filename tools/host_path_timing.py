#!/usr/bin/env python3
"""Host-pointer API (what the Julia shim uses) vs device-pointer API on the config-3 instance."""
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
    g, w_l, w_u = host["g"], host["w_l"], host["w_u"]
    x, x_l, x_u = host["x"], host["x_l"], host["x_u"]
    cons_full = bh.MixedConstraints(np.zeros((0, 4096)), None, host["fix"], l=x_l, u=x_u)
    delta = bh.synthetic.initial_tr(g)
    s0 = np.zeros(4096)

    def timeit(fn, reps=200):
        for _ in range(5):
            fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return 1e3 * (time.perf_counter() - t0) / reps

    print("bh_pcg_dev          %.4f ms" % timeit(lambda: bench.run_steps(bh, H, cons, dv, 0.1, 1)))
    print("bh_pcg (host ptrs)  %.4f ms" % timeit(lambda: bh.projected_cg(g, H, w_l, w_u, cons, 0.1)))
    print("bh_minor_iterate    %.4f ms" % timeit(lambda: bh.minor_iterate(x, s0, g, H, cons_full, delta, 0.1)))
    print("bh_hmul (host)      %.4f ms" % timeit(lambda: bh.hmul(H, g)))
    print("bh_vthv (host)      %.4f ms" % timeit(lambda: bh.vthv(H, g)))
    print("bh_project (host)   %.4f ms" % timeit(lambda: bh.projection(cons, g)))
    print("bh_hmul_add (host)  %.4f ms" % timeit(lambda: bh.hmul_add(H, g, g)))

    # One pass of the minor loop body (src/basic_tralcnlss.jl:434-447: minor_iterate; s .+= w; g_minor = H*s+g; active set
    # update; two reduced-gradient norms) — host vectors between the calls (what the per-method shim does) vs device-resident.
    import ctypes as ct
    n = 4096
    fix = host["fix"]

    def body_host():
        w, st = bh.minor_iterate(x, s0, g, H, cons_full, delta, 0.1)
        s = s0 + w
        gm = bh.hmul_add(H, s, g)
        s_l, s_u = np.maximum(x_l - x, -delta), np.minimum(x_u - x, delta)
        at = ((s - s_l) <= 1.5e-8) | ((s_u - s) <= 1.5e-8)             # active_bounds on the host
        cons_full.fixvars = fix | at                                  # add_active!: re-pushes the mask (box: no factor)
        a = np.linalg.norm(bh.projection(cons_full, -g))
        b = np.linalg.norm(bh.projection(cons_full, -gm))
        cons_full.fixvars = fix
        return a, b

    d = {k: bh.DeviceVector(n, v) for k, v in (("x", x), ("s", s0), ("g", g), ("xl", x_l), ("xu", x_u))}
    d["w"], d["gm"] = bh.DeviceVector(n), bh.DeviceVector(n)
    chunks = np.zeros(n // 64, dtype=np.uint64)
    cons_dev = bh.MixedConstraints(np.zeros((0, n)), None, fix, l=x_l, u=x_u)

    def body_dev():
        cons_dev.fixvars = fix
        P = cons_dev.handle
        st, it, nh, al = ct.c_int32(), ct.c_int32(), ct.c_int32(), ct.c_double()
        d["s"].upload(s0)                                             # (reset between repetitions; not part of the loop body)
        bh._lib.check(lib.bh_minor_iterate_dev(H.handle, P, d["x"].ptr, d["s"].ptr, d["g"].ptr, d["xl"].ptr, d["xu"].ptr, delta, 0.1,
                                               1.4901161193847656e-08, 1e-10, d["w"].ptr, ct.byref(st), ct.byref(it), ct.byref(nh), ct.byref(al)), "mi")
        bh._lib.check(lib.bh_step_accumulate_dev(H.handle, d["s"].ptr, d["w"].ptr, d["g"].ptr, d["gm"].ptr), "acc")
        na, nf, br = ct.c_int32(), ct.c_int32(), ct.c_int32()
        bh._lib.check(lib.bh_proj_update_active_dev(P, d["x"].ptr, d["s"].ptr, d["xl"].ptr, d["xu"].ptr, delta, 1.4901161193847656e-08,
                                                    ct.byref(na), ct.byref(nf), ct.byref(br), bh._lib.ptr(chunks)), "upd")
        a, b = ct.c_double(), ct.c_double()
        bh._lib.check(lib.bh_reduced_gradient_norm_dev(P, d["g"].ptr, ct.byref(a)), "n1")
        bh._lib.check(lib.bh_reduced_gradient_norm_dev(P, d["gm"].ptr, ct.byref(b)), "n2")
        return a.value, b.value

    ah, ad = body_host(), body_dev()
    print("minor-loop body, host vectors between calls   %.4f ms   (norms %.6e %.6e)" % (timeit(body_host, 100), ah[0], ah[1]))
    t_dev = timeit(body_dev, 100)
    t_up = timeit(lambda: (cons_dev.__setattr__("fixvars", fix), cons_dev.handle, d["s"].upload(s0)), 100)
    print("minor-loop body, device-resident              %.4f ms   (norms %.6e %.6e; includes %.4f ms of per-repetition reset)" % (t_dev, ad[0], ad[1], t_up))


if __name__ == "__main__":
    main()
