"""ctypes wrapper of the plain-C oracle (oracle/benlsip_oracle.c).  TEST INFRASTRUCTURE ONLY — see the C file's header."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load(build=True):
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(HERE, "libbenlsip_oracle.so")
    src = os.path.join(HERE, "benlsip_oracle.c")
    if build and (not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-s", "-C", HERE])
    lib = C.CDLL(path)
    vp, L, D = C.c_void_p, C.c_long, C.c_double
    lib.bo_num_threads.restype = C.c_int
    lib.bo_set_num_threads.argtypes = [C.c_int]
    lib.bo_set_num_threads.restype = None
    lib.bo_hmul.argtypes = [vp, L, L, L, vp, L, L, D, vp, vp, vp]
    lib.bo_hmul.restype = None
    lib.bo_vthv.argtypes = [vp, L, L, L, vp, L, L, D, vp, vp]
    lib.bo_vthv.restype = D
    lib.bo_projection.argtypes = [vp, L, L, L, vp, vp, L, L, vp, vp, vp]
    lib.bo_projection.restype = None
    lib.bo_factor_to_boundary.argtypes = [vp, vp, vp, vp, L, D]
    lib.bo_factor_to_boundary.restype = D
    lib.bo_projected_cg.argtypes = [vp, L, L, L, vp, L, L, D, vp, L, L, vp, vp, L, L, vp, vp, vp, D, D, D, vp,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), vp, L]
    lib.bo_projected_cg.restype = C.c_int
    _LIB = lib
    return lib


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def num_threads():
    return load().bo_num_threads()


def set_num_threads(n):
    """Cap the OpenMP team size of the following calls; returns the previous maximum."""
    prev = load().bo_num_threads()
    load().bo_set_num_threads(int(n))
    return prev


def hmul(J, Cm, mu, v):
    J, Cm, v = _f(J), _f(Cm), np.ascontiguousarray(v, dtype=np.float64)
    d, n = J.shape
    q = Cm.shape[0]
    out, work = np.empty(n), np.empty(d + 2 * q + n + 8)
    load().bo_hmul(_p(J), d, n, max(d, 1), _p(Cm), q, max(q, 1), mu, _p(v), _p(out), _p(work))
    return out


def vthv(J, Cm, mu, v):
    J, Cm, v = _f(J), _f(Cm), np.ascontiguousarray(v, dtype=np.float64)
    d, n = J.shape
    q = Cm.shape[0]
    work = np.empty(d + q + 8)
    return load().bo_vthv(_p(J), d, n, max(d, 1), _p(Cm), q, max(q, 1), mu, _p(v), _p(work))


def projection(A, fix, L, r):
    A, L, r = _f(A), _f(L), np.ascontiguousarray(r, dtype=np.float64)
    mA, n = A.shape
    mpp = L.shape[0]
    fixb = np.ascontiguousarray(fix, dtype=np.uint8)
    v, work = np.empty(n), np.empty(2 * mpp + 8)
    load().bo_projection(_p(A), mA, n, max(mA, 1), _p(fixb), _p(L), mpp, max(mpp, 1), _p(r), _p(v), _p(work))
    return v


def factor_to_boundary(p, w, wl, wu, atol=1e-10):
    p, w, wl, wu = (np.ascontiguousarray(x, dtype=np.float64) for x in (p, w, wl, wu))
    return load().bo_factor_to_boundary(_p(p), _p(w), _p(wl), _p(wu), p.shape[0], atol)


def projected_cg(g, J, Cm, mu, w_l, w_u, A, fix, L, kappa2, atol=np.sqrt(np.finfo(float).eps), atol_f2b=1e-10, trace_cap=0):
    """Returns (w, status, iters, n_hmul, trace)."""
    J, Cm, A, L = _f(J), _f(Cm), _f(A), _f(L)
    d, n = J.shape
    q, mA, mpp = Cm.shape[0], A.shape[0], L.shape[0]
    g, w_l, w_u = (np.ascontiguousarray(x, dtype=np.float64) for x in (g, w_l, w_u))
    fixb = np.ascontiguousarray(fix, dtype=np.uint8)
    w = np.empty(n)
    st, it, nh = C.c_int(-1), C.c_int(0), C.c_int(0)
    trace = np.full((max(trace_cap, 1), 4), np.nan)
    rc = load().bo_projected_cg(_p(J), d, n, max(d, 1), _p(Cm), q, max(q, 1), mu, _p(A), mA, max(mA, 1), _p(fixb), _p(L), mpp,
                                max(mpp, 1), _p(g), _p(w_l), _p(w_u), kappa2, atol, atol_f2b, _p(w), C.byref(st), C.byref(it),
                                C.byref(nh), _p(trace), trace_cap)
    assert rc == 0
    return w, st.value, it.value, nh.value, trace[:min(trace_cap, nh.value)]
