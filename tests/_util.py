"""Shared helpers of the parity tests."""
import numpy as np

import benlsip_ref as R


def relnorm(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def matvec_scale(J, v):
    """|| |J| |v| ||_2 — the normwise scale of SURVEY.md §8c's matvec tolerance."""
    return float(np.linalg.norm(np.abs(J) @ np.abs(v)))


def hmul_longdouble(H, v):
    Jl, Cl, vl = H.J.astype(np.longdouble), H.C.astype(np.longdouble), v.astype(np.longdouble)
    return (Jl.T @ (Jl @ vl) + Cl.T @ ((np.longdouble(H.mu) * Cl) @ vl)).astype(np.float64)


def pcg_sensitivity(g, H, w_l, w_u, cons, kappa2, w_ref):
    """How far the ORACLE's own w moves when H*p is accumulated in long double instead of fp64 — the rounding
    sensitivity of this CG instance (CG amplifies summation-order noise by ~cond(H)).  Parity of w is demanded to
    max(1e-9, 20 x this): tighter than that no two fp64 implementations (Julia's BLAS included) agree."""
    w2, s2, it2 = R.projected_cg(g, H, w_l, w_u, cons, kappa2, hmul_fn=hmul_longdouble)
    return relnorm(w2, w_ref)


def w_tolerance(g, H, w_l, w_u, cons, kappa2, w_ref):
    return max(1e-9, 20.0 * pcg_sensitivity(g, H, w_l, w_u, cons, kappa2, w_ref))
