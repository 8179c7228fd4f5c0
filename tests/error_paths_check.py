"""Manual check (not collected by pytest): every refused call leaves the library usable — invalid arguments, shape
mismatches, a rank-deficient A_free, an allocation that cannot fit, each followed by a valid call checked against NumPy."""
import ctypes as ct
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402

bh.init(0)
lib = bh._lib.lib()
rng = np.random.default_rng(0)
n, d = 96, 700
J = rng.standard_normal((d, n))
H = bh.AlHessian(J, None, 1.0)
v = rng.standard_normal(n)
ref = J.T @ (J @ v)


def still_fine(tag):
    err = np.linalg.norm(H * v - ref) / np.linalg.norm(ref)
    A = np.zeros((0, n))
    cons = bh.MixedConstraints(A, None, None)
    w, st = bh.projected_cg(v, H, -np.ones(n), np.ones(n), cons, 0.1)
    assert err < 1e-12 and np.all(np.isfinite(w)), (tag, err)
    print("ok after:", tag, flush=True)


def expect_error(tag, fn):
    try:
        fn()
    except (bh.BenlsipHipError, ValueError, TypeError) as e:
        print("  refused (%s): %s" % (tag, str(e)[:110]))
        still_fine(tag)
        return
    raise SystemExit("NOT refused: " + tag)


out = np.zeros(n)
expect_error("NULL vector", lambda: bh._lib.check(lib.bh_hmul(H.handle, None, bh._lib.ptr(out)), "bh_hmul"))
A3 = rng.standard_normal((3, n + 5))
P_bad = bh.MixedConstraints(A3, None, None)
expect_error("H.n != lincons.n", lambda: bh.projected_cg(v, H, -np.ones(n), np.ones(n), P_bad, 0.1))
Adep = np.vstack([np.ones(n), np.ones(n)])               # rank-deficient rows
Pdep = bh.MixedConstraints(Adep, None, None)
expect_error("A_free A_free' not positive definite", lambda: bh.projection(Pdep, v))
expect_error("mpp > n", lambda: bh.MixedConstraints(rng.standard_normal((4, n)), None, np.ones(n, dtype=bool)).handle)
expect_error("image larger than HBM", lambda: bh.AlHessian.synthetic(12_000_000, 4096, seed=1, mu=1.0))
expect_error("unknown option", lambda: bh.set_option("no_such_option", 1))
expect_error("blocks_per_cu out of range", lambda: bh.set_option("blocks_per_cu", 99))
expect_error("negative trace_cap", lambda: bh._lib.check(lib.bh_pcg(H.handle, bh.MixedConstraints(np.zeros((0, n)), None, None).handle, bh._lib.ptr(v), bh._lib.ptr(-np.ones(n)), bh._lib.ptr(np.ones(n)), 0.1, 1e-8, 1e-10, bh._lib.ptr(out), None, None, bh._lib.ptr(out), -5, None), "bh_pcg"))
print("all error paths recover")
