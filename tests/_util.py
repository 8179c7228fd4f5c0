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


# --------------------------------------------------------------------------------------------------------------------------
# Pinned operands of Cauchy searches whose outcome is decided by rounding (tests/golden/cauchy_events.json, written by
# tests/golden/make_cauchy_events.py from a device shadow solve of the 48-parameter NLS instance)
# --------------------------------------------------------------------------------------------------------------------------
def load_cauchy_events():
    import json
    import os
    from nls_problem import NLSProblem
    data = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cauchy_events.json")))
    P = NLSProblem(256, 48, 2, seed=1)
    hx = lambda seq: np.array([float.fromhex(v) for v in seq])
    events = []
    for e in data["events"]:
        x = hx(e["x"])
        fix0 = np.zeros(P.n, dtype=bool)
        fix0[e["fix0"]] = True
        events.append(dict(minor=e["minor"], x=x, g=hx(e["g"]), delta=float.fromhex(e["delta"]), mu=float.fromhex(e["mu"]), fix0=fix0,
                           J=P.jac_r(x), C=P.jac_c(x), s_dev=hx(e["s_dev"]), s_cpu=hx(e["s_cpu"]), fix_dev=e["fix_dev"], fix_cpu=e["fix_cpu"]))
    return P, events


class ReducedFormOps(R.NumpyOps):
    """A second CPU restatement of the SAME projector (the form the device uses): zero the fixed components,
    project the free part onto null(A_free) with chol(A_free A_free').  Mathematically identical to the reference's augmented
    form; the rounding differs."""

    def projection(self, lincons, r):
        from scipy.linalg import solve_triangular
        free = ~lincons.fixvars
        Af, rf = lincons.lineq[:, free], r[free]
        v = np.zeros_like(r)
        if Af.shape[0] == 0:
            v[free] = rf
            return v
        L = np.linalg.cholesky(Af @ Af.T)
        y = solve_triangular(L.T, solve_triangular(L, Af @ rf, lower=True), lower=False)
        v[free] = rf - Af.T @ y
        return v


def cauchy_outcomes(P, e, ops, samples, seed):
    """The oracle's cauchy_step (src/basic_tralcnlss.jl:574-639) on the event's operands: nominal g first, then `samples - 1`
    copies of g perturbed by at most one unit in the last place per entry.  Returns [(s, final active set as a tuple)]."""
    L0 = R.chol_lower(P.A @ P.A.T)
    H = R.AlHessian(e["J"], e["C"], e["mu"])
    rng = np.random.default_rng(seed)
    out = []
    for k in range(samples):
        g = e["g"] if k == 0 else e["g"] * (1.0 + 2.2e-16 * rng.uniform(-1.0, 1.0, e["g"].shape[0]))
        lc = R.make_mixed_constraints(P.A, L0, e["fix0"].copy(), l=P.x_l, u=P.x_u)
        s = R.cauchy_step(e["x"], g, H, L0, lc, e["delta"], ops)
        out.append((s, tuple(np.flatnonzero(lc.fixvars))))
    return out


def check_cauchy_against_cpu_family(P, e, s, key_dev, samples=32, seed=0):
    """What can be demanded of a Cauchy step where the outcome is decided by rounding (P: anything with .A, .x_l, .x_u; e: the
    operands x, g, delta, mu, J, C, fix0): feasibility, and model value / active-set size inside the range that the two CPU
    restatements (augmented form = the oracle, reduced form) span under 1-ulp perturbations of g; where CPU outcomes share the
    device's final set, the step as close to them as they are to each other (or 1e-6).  Returns a summary dict."""
    cpu = cauchy_outcomes(P, e, R.NumpyOps(), samples, seed=seed) + cauchy_outcomes(P, e, ReducedFormOps(), samples, seed=seed)
    phis = np.array([model_value(e, sc) for sc, _ in cpu])
    sizes = [len(k) for _, k in cpu]
    as_rel = lambda v: float(np.linalg.norm(P.A @ v) / (max(np.linalg.norm(P.A), 1e-300) * max(np.linalg.norm(v), 1e-300))) if P.A.shape[0] else 0.0
    phi_dev = model_value(e, s)
    same_set = [sc for sc, k in cpu if k == key_dev]
    info = dict(phi_dev=phi_dev, phi_min=float(phis.min()), phi_max=float(phis.max()), size_dev=len(key_dev), size_min=min(sizes), size_max=max(sizes),
                n_sets=len({k for _, k in cpu}), as_rel_dev=as_rel(s), as_rel_cpu=max(as_rel(sc) for sc, _ in cpu), same_set=len(same_set),
                nearest=min(relnorm(s, sc) for sc, _ in cpu))
    assert np.max(np.abs(s)) <= e["delta"] * (1 + 1e-12), info
    assert np.all(e["x"] + s <= P.x_u + 1e-12) and np.all(e["x"] + s >= P.x_l - 1e-12), info
    assert info["as_rel_dev"] <= 10.0 * info["as_rel_cpu"] + 1e-12, info
    note_tol("Cauchy step where rounding decides: model value inside the CPU outcomes' range (+-5 %)",
             max(phi_dev - phis.max(), phis.min() - phi_dev, 0.0) + 0.0, 0.05 * max(abs(phis.min()), abs(phis.max())), "delta %.1e" % e["delta"])
    assert phis.min() - 0.05 * abs(phis.min()) <= phi_dev <= phis.max() + 0.05 * abs(phis.max()), info
    assert min(sizes) - 2 <= len(key_dev) <= max(sizes) + 2, info
    if same_set:
        spread = max(relnorm(sa, sb) for sa in same_set[:8] for sb in same_set)
        info["same_set_dist"] = min(relnorm(s, sc) for sc in same_set)
        assert info["same_set_dist"] <= max(1e-6, spread), (info, spread)
    if info["n_sets"] == 1:
        assert key_dev == cpu[0][1], info
    return info


def model_value(e, s):
    """phi(s) = g.s + 1/2 s'Hs — what the Cauchy search minimises along the projected-gradient path (:610-611)."""
    return float(e["g"] @ s + 0.5 * R.vthv(R.AlHessian(e["J"], e["C"], e["mu"]), s))


# --------------------------------------------------------------------------------------------------------------------------
# Re-association variants of the oracle's H*v (the band the reference's own arithmetic spans on config 1)
# --------------------------------------------------------------------------------------------------------------------------
def _hmul_mu_outside(H, v):
    return H.J.T @ (H.J @ v) + H.mu * (H.C.T @ (H.C @ v))


def _hmul_sequential(H, v):
    def mv(M, x):          # plain left-to-right dot products, row by row
        out = np.zeros(M.shape[0])
        for i in range(M.shape[0]):
            acc = 0.0
            for j in range(M.shape[1]):
                acc += M[i, j] * x[j]
            out[i] = acc
        return out
    return mv(H.J.T, mv(H.J, v)) + mv(H.C.T, mv(H.mu * H.C, v))


def _cached(H, key, make):
    """Derived copies of J are built once per AlHessian (the variants are called hundreds of times per CG run)."""
    cache = H.__dict__.setdefault("_variant_cache", {})
    if key not in cache:
        cache[key] = make()
    return cache[key]


def _hmul_reversed(H, v):
    Jr = _cached(H, "rev", lambda: np.asfortranarray(H.J[::-1]))
    Cr = H.C[::-1]
    return (Jr.T @ (Jr @ v)) + (Cr.T @ ((H.mu * Cr) @ v))


HMUL_VARIANTS = {"reference order (mu*C)*v": None, "mu*(C'(C v))": _hmul_mu_outside, "long double": hmul_longdouble,
                 "sequential sums": _hmul_sequential, "rows reversed": _hmul_reversed}


class OracleVariantOps(R.NumpyOps):
    """The oracle with its H*v evaluated in another, mathematically equivalent order."""

    def __init__(self, fn):
        self.fn = fn

    def hmul(self, H, v):
        return R.hmul(H, v) if self.fn is None else self.fn(H, v)

    def projected_cg(self, g_minor, H, w_l, w_u, lincons, kappa2):
        w, status, _ = R.projected_cg(g_minor, H, w_l, w_u, lincons, kappa2, **({} if self.fn is None else {"hmul_fn": self.fn}))
        return w, status


def sphere_opt_measure(xs, ys):
    """test/problems/sphere_regression.jl:58-61: || x - P(x - grad L) || with P the projection onto {Ax = b, l <= x <= u}."""
    import sphere_problem as sp
    grad = sp.jac_r(xs).T @ sp.r(xs) + sp.jac_c(xs).T @ ys
    return float(np.linalg.norm(xs - R.projection_polyhedron_small(xs - grad, sp.A, sp.b, sp.x_l, sp.x_u)))


_SPHERE_BAND = {}


def sphere_oracle_band():
    """opt_measure of the ORACLE's config-1 solve under each re-association variant of its own H*v: {variant: value}."""
    import sphere_problem as sp
    if not _SPHERE_BAND:
        for name, fn in HMUL_VARIANTS.items():
            xs, ys = R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u, max_outer_iter=100, max_inner_iter=250,
                                  ops=OracleVariantOps(fn))
            _SPHERE_BAND[name] = sphere_opt_measure(xs, ys)
    return dict(_SPHERE_BAND)


# --------------------------------------------------------------------------------------------------------------------------
# How much of each tolerance the run actually used (printed at the end of the session by tests/conftest.py)
# --------------------------------------------------------------------------------------------------------------------------
TOL_USED = {}


def note_tol(label, value, bound, detail=""):
    """Record value / bound for the end-of-session table; returns the ratio."""
    ratio = float(value) / float(bound) if bound > 0 else (0.0 if value == 0 else float("inf"))
    n, worst, wdetail, vals = TOL_USED.get(label, (0, -1.0, "", (0.0, 0.0)))
    if ratio > worst:
        worst, wdetail, vals = ratio, detail, (float(value), float(bound))
    TOL_USED[label] = (n + 1, worst, wdetail, vals)
    return ratio


def assert_w_close(w, w_ref, tol, label, detail=""):
    """||w - w_ref|| <= tol ||w_ref||, with the used fraction of the tolerance recorded under `label`."""
    rel = relnorm(w, w_ref)
    note_tol(label, rel, tol, detail)
    assert rel <= tol, (label, detail, rel, tol)


def _hmul_c_order(H, v):
    Jc = _cached(H, "c", lambda: np.ascontiguousarray(H.J))
    return Jc.T @ (Jc @ v) + H.C.T @ ((H.mu * H.C) @ v)


def _hmul_row_chunks(H, v, chunk=1024):
    out = H.C.T @ ((H.mu * H.C) @ v)
    for lo in range(0, H.J.shape[0], chunk):
        Jb = H.J[lo:lo + chunk]
        out = out + Jb.T @ (Jb @ v)
    return out


def _hmul_row_blocks(k):
    """J'(J v) as the sum over k contiguous row blocks, added in block order (what k row-sharded ranks compute)."""
    def fn(H, v):
        out = H.C.T @ ((H.mu * H.C) @ v)
        d = H.J.shape[0]
        base, extra = divmod(d, k)
        lo = 0
        for b in range(k):
            hi = lo + base + (1 if b < extra else 0)
            out = out + H.J[lo:hi].T @ (H.J[lo:hi] @ v)
            lo = hi
        return out
    return fn


CG_HMUL_VARIANTS = {"reference": None, "long double": hmul_longdouble, "C-order sums": _hmul_c_order, "1024-row chunks": _hmul_row_chunks,
                    "rows reversed": _hmul_reversed, "2 row blocks": _hmul_row_blocks(2), "3 row blocks": _hmul_row_blocks(3),
                    "4 row blocks": _hmul_row_blocks(4), "7 row blocks": _hmul_row_blocks(7)}


def oracle_iteration_band(g, H, w_l, w_u, cons, kappa2, variants=None, perturbed=0):
    """Iteration counts (and exit statuses) of the ORACLE's projected_cg when its H*p is summed in mathematically equivalent
    orders — and, with `perturbed` = k, on k copies of the instance whose J carries relative perturbations of 1e-15 (seeded):
    {variant: (status, iters)}.  On an ill-conditioned instance finite-precision CG is chaotic — this is the band inside which an
    iteration count means anything."""
    out = {}
    for name, fn in CG_HMUL_VARIANTS.items():
        if variants is not None and name not in variants:
            continue
        w, st, it = R.projected_cg(g, H, w_l, w_u, cons, kappa2, **({} if fn is None else {"hmul_fn": fn}))
        out[name] = (int(st), int(it))
    rng = np.random.default_rng(1234)
    for k in range(perturbed):
        Hk = R.AlHessian(H.J * (1.0 + 1e-15 * rng.uniform(-1.0, 1.0, H.J.shape)), H.C, H.mu)
        w, st, it = R.projected_cg(g, Hk, w_l, w_u, cons, kappa2)
        out["J perturbed 1e-15 #%d" % k] = (int(st), int(it))
    return out


def assert_iters_in_oracle_band(iters, band, label, detail=""):
    """The device's count must lie inside the oracle's own band widened on either side by the band's width (at least 5 % of the
    count: a handful of oracle runs under-samples the spread — the device rounds every reduction of the loop differently, not only
    H*p)."""
    its = [it for _, it in band.values()]
    lo, hi = min(its), max(its)
    slack = max(hi - lo, int(np.ceil(0.05 * hi)))
    note_tol(label, abs(iters - 0.5 * (lo + hi)), 0.5 * (hi - lo) + slack, "%s: device %d, oracle band %d..%d" % (detail, iters, lo, hi))
    assert lo - slack <= iters <= hi + slack, (label, detail, iters, band)


def closed_form_cases():
    """projected_cg instances whose answer follows by hand from src/basic_tralcnlss.jl:702-761 — dyadic data, so every product and sum
    on the way is exact in fp64 and the expected step holds to the last bit whatever the summation order (case `diag` excepted: its
    alpha = 2/5).  They pin the oracle and the device to the reference's text independently of each other.
    Yields dicts: J, C, mu, A, fix, g, wl, wu, kappa2 and the expected w, status (0 solved, 1 bound_hit, 2 negative_curvature, 4 none),
    iters (the reference's `iter` at exit, starts at 1) and n_hmul, plus `rtol` for w."""
    g4 = np.array([1.0, -2.0, 0.5, 4.0])
    wide = 100.0 * np.ones(4)
    Z4 = np.zeros((0, 4))
    I4 = np.eye(4)
    nofix = np.zeros(4, dtype=bool)
    # H = I: v = g, p = -g, p'Hp = r'v = 21.25, alpha = 1 < gamma = 100/4; w = -g, r = 0: solved after one product (:733-748)
    yield dict(name="identity", J=I4, C=Z4, mu=1.0, A=Z4, fix=nofix, g=g4, wl=-wide, wu=wide, kappa2=0.1,
               w=-g4, status=0, iters=2, n_hmul=1, rtol=0.0)
    # the same with |w_i| <= 1/2: gamma = min(.5/1, .5/2, .5/.5, .5/4) = 1/8 < alpha = 1: w = p/8, bound_hit, iter stays 1 (:735-737)
    yield dict(name="bound_hit", J=I4, C=Z4, mu=1.0, A=Z4, fix=nofix, g=g4, wl=-0.5 * np.ones(4), wu=0.5 * np.ones(4), kappa2=0.1,
               w=-g4 / 8.0, status=1, iters=1, n_hmul=1, rtol=0.0)
    # H = 0: p'Hp = 0 <= atol and |p'Hp| <= atol: no step at all (:725-727), negative_curvature with w = 0
    yield dict(name="zero_curvature", J=np.zeros((1, 4)), C=Z4, mu=1.0, A=Z4, fix=nofix, g=g4, wl=-wide, wu=wide, kappa2=0.1,
               w=np.zeros(4), status=2, iters=1, n_hmul=1, rtol=0.0)
    # H = mu C'C with mu = -1, C = I: p'Hp = -21.25 < -atol: the step to the boundary of :727-729, w = p/8
    yield dict(name="negative_curvature_step", J=np.zeros((1, 4)), C=I4, mu=-1.0, A=Z4, fix=nofix, g=g4, wl=-0.5 * np.ones(4),
               wu=0.5 * np.ones(4), kappa2=0.1, w=-g4 / 8.0, status=2, iters=1, n_hmul=1, rtol=0.0)
    # H = 4 I through the C block (J = 0, C = I, mu = 4): alpha = 1/4, w = -g/4
    yield dict(name="penalty_block", J=np.zeros((1, 4)), C=I4, mu=4.0, A=Z4, fix=nofix, g=g4, wl=-wide, wu=wide, kappa2=0.1,
               w=-g4 / 4.0, status=0, iters=2, n_hmul=1, rtol=0.0)
    # one equality sum(w) = 0, H = I: v = g - mean(g) = g - 7/8 (A A' = 4, all dyadic), alpha = 1, w = -v, then P(r) = P(7/8 * 1) = 0
    yield dict(name="equality_mean", J=I4, C=Z4, mu=1.0, A=np.ones((1, 4)), fix=nofix, g=g4, wl=-wide, wu=wide, kappa2=0.1,
               w=-(g4 - 0.875), status=0, iters=2, n_hmul=1, rtol=4e-16)
    # a fixed variable (w_l = w_u = 0 there): v = mask(g), w = -mask(g); max_iter = 2 (4 - 0 - 1)
    fix1 = np.array([False, True, False, False])
    yield dict(name="fixed_variable", J=I4, C=Z4, mu=1.0, A=Z4, fix=fix1, g=g4, wl=np.where(fix1, 0.0, -wide), wu=np.where(fix1, 0.0, wide),
               kappa2=0.1, w=np.where(fix1, 0.0, -g4), status=0, iters=2, n_hmul=1, rtol=0.0)
    # everything fixed: max_iter = 0, the loop is never entered, iter = 1 != max_iter: status none (:759-761), w = 0
    fixall = np.ones(2, dtype=bool)
    yield dict(name="all_fixed", J=np.eye(2), C=np.zeros((0, 2)), mu=1.0, A=np.zeros((0, 2)), fix=fixall, g=np.array([1.0, -2.0]),
               wl=np.zeros(2), wu=np.zeros(2), kappa2=0.1, w=np.zeros(2), status=4, iters=1, n_hmul=0, rtol=0.0)
    # H = diag(1, 4), g = (1, 1), tight tolerance: CG is exact after two products, w = -(1, 1/4) (alpha_1 = 2/5 is not dyadic: 4 ulps)
    yield dict(name="diag", J=np.diag([1.0, 2.0]), C=np.zeros((0, 2)), mu=1.0, A=np.zeros((0, 2)), fix=np.zeros(2, dtype=bool),
               g=np.ones(2), wl=-100.0 * np.ones(2), wu=100.0 * np.ones(2), kappa2=1e-12, w=np.array([-1.0, -0.25]), status=0, iters=3,
               n_hmul=2, rtol=1e-15)


def closed_form_cauchy_cases():
    """cauchy_step instances solved by hand from src/basic_tralcnlss.jl:574-639 (H = I in two variables, x = 0, bounds +-1, dyadic data:
    every phi', phi'' and breakpoint is exact).  Yields g, delta and the expected step, active set and number of H*d products."""
    # d = (4, 1/2): phi' = -16.25, phi'' = 16.25, dt = 1 > theta = 1/4 (variable 0 reaches its bound): s = (1, 1/8), d = (0, 1/2);
    # then phi' = -1/4 + 1/16, phi'' = 1/4, dt = 3/4 < theta = 7/4: the minimiser lies inside the second segment, s = (1, 1/2)
    yield dict(name="interior_minimum_on_second_segment", g=np.array([-4.0, -0.5]), delta=100.0, s=np.array([1.0, 0.5]), fix=np.array([True, False]), n_hmul=2)
    # trust region 1/2: theta = 1/8, s = (1/2, 1/16); then dt = 7/8 EQUALS theta = 7/8 — `dt < theta` (:622) is false, the search
    # advances to the breakpoint and fixes variable 1 too; with n - mA variables fixed the loop ends (:615) after a third product
    yield dict(name="tie_between_minimiser_and_breakpoint", g=np.array([-4.0, -0.5]), delta=0.5, s=np.array([0.5, 0.5]), fix=np.array([True, True]), n_hmul=3)
    # g = 0: phi' = 0 >= 0 at once (:620)
    yield dict(name="stationary_point", g=np.zeros(2), delta=1.0, s=np.zeros(2), fix=np.array([False, False]), n_hmul=1)
    # d = (4, 1): theta = 1/4, then dt = 3/4 = theta again: both variables end on their bounds
    yield dict(name="corner", g=np.array([-4.0, -1.0]), delta=100.0, s=np.ones(2), fix=np.array([True, True]), n_hmul=3)
