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


def driver_decisions(entry, eta1=0.25, eta2=0.75):
    """The branch decisions one log entry of the restated driver stands for (oracle/benlsip_ref.py: inner_step appends
    ("minor", CG status, active bounds, exit ratio), solve_subproblem appends ("tr", rho, pix/omega, delta))."""
    if entry[0] == "minor":
        return ("minor", entry[1], entry[2], bool(entry[3] <= 1.0))
    return ("tr", bool(entry[1] > eta1), bool(entry[1] > eta2), bool(entry[1] < eta1), bool(entry[2] < 1.0))


def first_decision_difference(log_a, log_b, eta1=0.25, eta2=0.75):
    """Index of the first entry where two driver logs take different decisions (None if one is a prefix of the other and
    all shared decisions agree), with the deciding scalars and their relative distance to the threshold they straddle."""
    for k, (a, b) in enumerate(zip(log_a, log_b)):
        da, db = driver_decisions(a, eta1, eta2), driver_decisions(b, eta1, eta2)
        if da == db:
            continue
        why = []
        if a[0] != b[0]:
            why.append(("sequence", a[0], b[0], None))
        elif a[0] == "minor":
            if a[1] != b[1]:
                why.append(("cg_status", a[1], b[1], None))
            if a[2] != b[2]:
                why.append(("active_bounds", a[2], b[2], None))
            if (a[3] <= 1.0) != (b[3] <= 1.0):
                why.append(("minor-loop exit ratio vs 1 (src/basic_tralcnlss.jl:449)", a[3], b[3], max(abs(a[3] - 1.0), abs(b[3] - 1.0))))
        else:
            for thr, name in ((eta1, "rho vs eta1 (:358,:828)"), (eta2, "rho vs eta2 (:824)")):
                if (a[1] > thr) != (b[1] > thr) or (a[1] < thr) != (b[1] < thr):
                    # rho = ared/pred with ared = mx_next - mx: how many ulps of |mx| the numerator is worth on either side
                    noise = max(abs(a[4]), abs(b[4])) / (np.finfo(float).eps * max(a[6], b[6], 1e-300))
                    why.append((name, a[1], b[1], {"ared_in_ulps_of_mx": noise, "ared": (a[4], b[4]), "pred": (a[5], b[5])}))
            if (a[2] < 1.0) != (b[2] < 1.0):
                why.append(("pix vs omega (:372)", a[2], b[2], max(abs(a[2] - 1.0), abs(b[2] - 1.0))))
        return k, a, b, why
    return None


def assert_rounding_dominated(diff):
    """The first differing decision of two driver logs must be one the reference's own arithmetic cannot decide: a
    trust-region ratio whose numerator ared = mx_next - mx is worth no more than a few hundred ulps of mx."""
    k, a, b, why = diff
    assert why, (k, a, b)
    for name, va, vb, extra in why:
        assert name.startswith("rho vs"), "driver decision %r differs (oracle %r, device %r) at log entry %d" % (name, va, vb, k)
        assert extra["ared_in_ulps_of_mx"] <= 512.0, (k, name, extra)
