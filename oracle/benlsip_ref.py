"""CPU oracle (NumPy) for the BEnlsip.jl trust-region subproblem hot path.

TEST INFRASTRUCTURE ONLY.  This file is a literal CPU restatement of the
reference's algorithm and exists solely as the checker for the HIP path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it; the product package (``benlsip.jl_amd/``) never does.

Parity pinning.  The reference is pure Julia and no Julia
toolchain exists in this image, so the reference itself cannot be run.  The
oracle is pinned by the fixtures the reference's own tests hold:

* ``test/structures.jl:37-58``  HS48 projection known answer ``[0,0,0,2,-2]``
  plus the ``left_mul`` / ``left_mul_tr`` identities            -> ``projection``
* ``test/structures.jl:1-16``   ``H*v`` / ``vthv`` vs explicit ``J'J + mu C'C``
* ``test/structures.jl:18-35``  augmented factor ``L ~ chol(B B')``
* ``test/structures.jl:60-78``  active-set flags
* ``test/problems/sphere_regression.jl``  three acceptance inequalities
  (the only, indirect, pin on ``projected_cg`` / ``factor_to_boundary`` /
  ``linesearch`` / ``minor_iterate``).

``projected_cg`` has NO direct golden data in the reference: its iteration-level
behaviour is pinned only by this line-by-line restatement ("parity unpinned"
at iteration level; pinned end-to-end through sphere_regression's inequalities).

All citations are ``path:line`` relative to ``/root/reference``.
Symbols: d = rows of J, n = columns, q = rows of C, mA = rows of A,
p = number of fixed variables, mpp = mA + p  (SURVEY.md §0.1).
"""
from __future__ import annotations

import enum
import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Tuple

import numpy as np
from scipy.linalg import solve_triangular

SQRT_EPS = math.sqrt(np.finfo(np.float64).eps)


# --------------------------------------------------------------------------- #
# src/basic_tralcnlss.jl:12  @enum CG_status solved bound_hit negative_curvature
# max_iter_reached ; plus Julia's `nothing` as code 4.
# --------------------------------------------------------------------------- #
class CGStatus(enum.IntEnum):
    solved = 0
    bound_hit = 1
    negative_curvature = 2
    max_iter_reached = 3
    none = 4


# --------------------------------------------------------------------------- #
# AlHessian — src/basic_tralcnlss.jl:6-10
# --------------------------------------------------------------------------- #
@dataclass
class AlHessian:
    J: np.ndarray  # d x n
    C: np.ndarray  # q x n
    mu: float


def vthv(H: AlHessian, v: np.ndarray) -> float:
    """src/basic_tralcnlss.jl:92-96."""
    Jv = H.J @ v
    Cv = H.C @ v
    return float(np.dot(Jv, Jv) + H.mu * np.dot(Cv, Cv))


def hmul(H: AlHessian, v: np.ndarray) -> np.ndarray:
    """``Base.:*(H::AlHessian, v)`` — src/basic_tralcnlss.jl:102-106.

    ``H.mu*H.C*v`` is evaluated left to right in Julia, i.e. ``(mu*C)*v``.
    """
    Jv = H.J @ v
    muCv = (H.mu * H.C) @ v
    return H.J.T @ Jv + H.C.T @ muCv


# --------------------------------------------------------------------------- #
# MixedConstraints — src/polyhedral_constraints.jl:1-7
# ``chol`` is stored as the lower factor L (mpp x mpp, or mA x mA when nothing
# is fixed); only its lower triangle is meaningful.
# --------------------------------------------------------------------------- #
@dataclass
class MixedConstraints:
    lineq: np.ndarray           # mA x n
    xlow: np.ndarray            # n
    xupp: np.ndarray            # n
    fixvars: np.ndarray         # n bool   (Julia BitVector)
    chol_L: np.ndarray          # lower factor currently in use


def chol_lower(M: np.ndarray) -> np.ndarray:
    """Lower Cholesky factor; accepts the 0 x 0 matrix (box-only configs)."""
    if M.shape[0] == 0:
        return np.zeros((0, 0))
    return np.linalg.cholesky(M)


def _solve_lower(L: np.ndarray, b: np.ndarray) -> np.ndarray:
    if L.shape[0] == 0:
        return np.zeros_like(b, dtype=np.float64)
    return solve_triangular(L, b, lower=True, check_finite=False)


def _solve_upper_from_lower(L: np.ndarray, b: np.ndarray) -> np.ndarray:
    """``chol.U \\ b`` with ``U = L'`` (only the lower triangle of L is read)."""
    if L.shape[0] == 0:
        return np.zeros_like(b, dtype=np.float64)
    return solve_triangular(L, b, lower=True, trans="T", check_finite=False)


def make_mixed_constraints(A, chol_aat_L, fixed=None, l=None, u=None) -> MixedConstraints:
    """Constructors src/polyhedral_constraints.jl:9-29."""
    A = np.asarray(A, dtype=np.float64)
    n = A.shape[1]
    l = np.full(n, -np.inf) if l is None else np.asarray(l, dtype=np.float64)
    u = np.full(n, np.inf) if u is None else np.asarray(u, dtype=np.float64)
    if fixed is None:
        return MixedConstraints(A, l, u, np.zeros(n, dtype=bool), chol_aat_L)
    fixed = np.asarray(fixed, dtype=bool).copy()
    chol = cholesky_aug_aat(A, fixed, chol_aat_L) if fixed.any() else chol_aat_L
    return MixedConstraints(A, l, u, fixed, chol)


def nb_fix(lincons: MixedConstraints) -> int:
    """src/polyhedral_constraints.jl:31."""
    return int(np.count_nonzero(lincons.fixvars))


def cholesky_aug_aat(A: np.ndarray, fix_bounds: np.ndarray, chol_aat_L: np.ndarray) -> np.ndarray:
    """src/polyhedral_constraints.jl:35-59.  Returns the mpp x mpp lower factor
    ``[L_A 0; G' chol(I - G'G)]`` (upper triangle explicitly zero here; the
    reference leaves it uninitialised)."""
    m, n = A.shape
    p = int(np.count_nonzero(fix_bounds))
    mpp = m + p
    assert mpp <= n
    A_act_cols = A[:, fix_bounds]
    G = _solve_lower(chol_aat_L, A_act_cols) if m > 0 else np.zeros((0, p))
    Hm = np.eye(p) - G.T @ G
    L = np.zeros((mpp, mpp))
    L[:m, :m] = chol_aat_L
    L[m:, :m] = G.T
    L[m:, m:] = chol_lower(Hm)
    return L


def update_chol(lincons: MixedConstraints, chol_aat_L: np.ndarray) -> None:
    """src/polyhedral_constraints.jl:62-68."""
    lincons.chol_L = cholesky_aug_aat(lincons.lineq, lincons.fixvars, chol_aat_L)


def left_mul_tr(lincons: MixedConstraints, y: np.ndarray) -> np.ndarray:
    """src/polyhedral_constraints.jl:72-84."""
    m, n = lincons.lineq.shape
    if lincons.fixvars.any():
        x = lincons.lineq.T @ y[:m]
        x[lincons.fixvars] += y[m:]
    else:
        x = lincons.lineq.T @ y
    return x


def left_mul(lincons: MixedConstraints, x: np.ndarray) -> np.ndarray:
    """src/polyhedral_constraints.jl:86-98."""
    m, _ = lincons.lineq.shape
    if lincons.fixvars.any():
        y = np.empty(m + nb_fix(lincons))
        y[:m] = lincons.lineq @ x
        y[m:] = x[lincons.fixvars]
    else:
        y = lincons.lineq @ x
    return y


def projection_nullspace(lincons: MixedConstraints, r: np.ndarray) -> np.ndarray:
    """src/polyhedral_constraints.jl:104-118."""
    assert not lincons.fixvars.any()
    y = _solve_lower(lincons.chol_L, lincons.lineq @ r)
    w = _solve_upper_from_lower(lincons.chol_L, y)
    return r - lincons.lineq.T @ w


def projection_subspace(lincons: MixedConstraints, r: np.ndarray) -> np.ndarray:
    """src/polyhedral_constraints.jl:120-136."""
    m, n = lincons.lineq.shape
    mpp = m + nb_fix(lincons)
    assert m < mpp <= n
    y = _solve_lower(lincons.chol_L, left_mul(lincons, r))
    w = _solve_upper_from_lower(lincons.chol_L, y)
    return r - left_mul_tr(lincons, w)


def projection(lincons: MixedConstraints, r: np.ndarray) -> np.ndarray:
    """``projection`` / ``projection!`` — src/polyhedral_constraints.jl:150-170."""
    if lincons.fixvars.any():
        return projection_subspace(lincons, r)
    return projection_nullspace(lincons, r)


def active_bounds_inplace(lincons: MixedConstraints, x: np.ndarray, chol_aat_L: np.ndarray,
                          atol: float = SQRT_EPS) -> None:
    """``active_bounds!`` — src/polyhedral_constraints.jl:203-215."""
    lincons.fixvars = ((x - lincons.xlow) <= atol) | ((lincons.xupp - x) <= atol)
    update_chol(lincons, chol_aat_L)


def active_bounds(lincons: MixedConstraints, x: np.ndarray, s: np.ndarray, delta: float,
                  atol: float = SQRT_EPS) -> np.ndarray:
    """src/polyhedral_constraints.jl:219-237 (returns 0-based indices)."""
    s_l = np.maximum(lincons.xlow - x, -delta)
    s_u = np.minimum(lincons.xupp - x, delta)
    at_bound = ((s - s_l) <= atol) | ((s_u - s) <= atol)
    return np.flatnonzero(at_bound)


def add_active(lincons: MixedConstraints, chol_aat_L: np.ndarray, ind) -> None:
    """``add_active!`` (scalar and vector methods) — src/polyhedral_constraints.jl:240-261."""
    lincons.fixvars = lincons.fixvars.copy()
    lincons.fixvars[ind] = True
    update_chol(lincons, chol_aat_L)


# --------------------------------------------------------------------------- #
# factor_to_boundary — src/basic_tralcnlss.jl:793-809
# --------------------------------------------------------------------------- #
def _julia_min(a: float, b: float) -> float:
    """Julia's ``min`` propagates NaN (``min(Inf, NaN) === NaN``); Python's built-in drops it when it comes second."""
    if a != a or b != b:
        return math.nan
    return min(a, b)


def factor_to_boundary(p, w, w_l, w_u, atol: float = 1e-10) -> float:
    gamma = math.inf
    with np.errstate(divide="ignore", invalid="ignore"):
        neg = p <= -atol
        pos = p >= atol
        if neg.any():
            gamma = _julia_min(gamma, float(np.min((w_l[neg] - w[neg]) / p[neg])))     # np.min propagates NaN like Julia's min
        if pos.any():
            gamma = _julia_min(gamma, float(np.min((w_u[pos] - w[pos]) / p[pos])))
    return gamma


# --------------------------------------------------------------------------- #
# projected_cg — src/basic_tralcnlss.jl:690-764
# --------------------------------------------------------------------------- #
@dataclass
class CGTrace:
    """Per-iteration scalars (pHp, alpha, gamma, rtv_after) for fixtures."""
    rows: List[Tuple[float, float, float, float]] = field(default_factory=list)
    n_hmul: int = 0


def projected_cg(g_minor, H, w_l, w_u, lincons: MixedConstraints, kappa2: float,
                 atol: float = SQRT_EPS, atol_f2b: float = 1e-10,
                 hmul_fn: Callable = hmul, proj_fn: Callable = projection,
                 trace: Optional[CGTrace] = None):
    """Returns ``(w, status, iters)``; ``iters`` is the reference's ``iter``
    variable at exit (starts at 1, :713)."""
    m, n = lincons.lineq.shape

    w = np.zeros(n)                                   # :702
    r = np.array(g_minor, dtype=np.float64, copy=True)  # :703-705
    v = proj_fn(lincons, r)                           # :706
    rtv = float(np.dot(r, v))                         # :707
    p = -v                                            # :708

    tol_cg = kappa2 * float(np.linalg.norm(v))        # :710
    tol_negcurve = atol                               # :711

    it = 1                                            # :713
    max_iter = 2 * (n - m - nb_fix(lincons))          # :714
    approx_solved = False                             # :716
    neg_curvature = False
    outside_region = False

    with np.errstate(all="ignore"):
        while (not approx_solved) and (not outside_region) and (not neg_curvature) and it <= max_iter:  # :720
            Hp = hmul_fn(H, p)                        # :722
            pHp = float(np.dot(p, Hp))                # :723
            if trace is not None:
                trace.n_hmul += 1
            alpha = math.nan
            gamma = math.nan
            if pHp <= tol_negcurve:                   # :725
                neg_curvature = True
                if abs(pHp) > tol_negcurve:           # :727
                    gamma = factor_to_boundary(p, w, w_l, w_u, atol_f2b)  # :728
                    w = w + gamma * p                 # :729
            else:
                rtv = float(np.dot(r, v))             # :732
                alpha = rtv / pHp                     # :733
                gamma = factor_to_boundary(p, w, w_l, w_u, atol_f2b)      # :734
                outside_region = alpha > gamma        # :735
                if outside_region:
                    w = w + gamma * p                 # :737
                else:
                    w = w + alpha * p                 # :739
                    r = r + alpha * Hp                # :740
                    v = proj_fn(lincons, r)           # :741
                    rtv_next = float(np.dot(r, v))    # :743
                    beta = rtv_next / rtv             # :744
                    p = -v + beta * p                 # :745
                    rtv = rtv_next                    # :746
                    approx_solved = abs(rtv) < tol_cg  # :747
                    it += 1                           # :748
            if trace is not None:
                trace.rows.append((pHp, alpha, gamma, rtv))

    if approx_solved:                                 # :753-761
        status = CGStatus.solved
    elif outside_region:
        status = CGStatus.bound_hit
    elif neg_curvature:
        status = CGStatus.negative_curvature
    elif it == max_iter:
        status = CGStatus.max_iter_reached
    else:
        status = CGStatus.none
    return w, status, it


# --------------------------------------------------------------------------- #
# linesearch — src/basic_tralcnlss.jl:766-791
# --------------------------------------------------------------------------- #
def linesearch(g_model, H, w, w_l, w_u, fix_bounds, vthv_fn: Callable = vthv) -> float:
    wHw = vthv_fn(H, w)                               # :775
    with np.errstate(all="ignore"):
        alpha_opt = (-float(np.dot(g_model, w)) / wHw) if wHw > 0 else math.inf  # :776
        alpha_allowed = math.inf                      # :779
        free = ~np.asarray(fix_bounds, dtype=bool)
        neg = free & (w < 0)
        pos = free & (w > 0)
        if neg.any():
            alpha_allowed = min(alpha_allowed, float(np.min(w_l[neg] / w[neg])))  # :783
        if pos.any():
            alpha_allowed = min(alpha_allowed, float(np.min(w_u[pos] / w[pos])))  # :785
    return min(alpha_opt, alpha_allowed)              # :790


# --------------------------------------------------------------------------- #
# Backend hooks: the caller chain below goes through an ``Ops`` object so the
# tests can run the *same* restated outer iteration over the HIP C-ABI.
# --------------------------------------------------------------------------- #
class NumpyOps:
    """Default backend: the oracle's own operators."""

    def new_hessian(self, J, C, mu):
        return AlHessian(np.asarray(J, dtype=np.float64), np.asarray(C, dtype=np.float64), float(mu))

    def hmul(self, H, v):
        return hmul(H, v)

    def vthv(self, H, v):
        return vthv(H, v)

    def projection(self, lincons, r):
        return projection(lincons, r)

    def projected_cg(self, g_minor, H, w_l, w_u, lincons, kappa2):
        w, status, _ = projected_cg(g_minor, H, w_l, w_u, lincons, kappa2)
        return w, status

    # The three places where the callers touch the residual rows directly.  A row-sharded backend (one rank holds
    # rows [lo, hi) of r and J) overrides them with all-reduced forms; here they are the reference's own expressions.
    def residual_sqnorm(self, rx):
        """``dot(rx,rx)`` in ``mx`` — src/basic_tralcnlss.jl:44 (new_point), :58 (evaluate_al)."""
        return float(np.dot(rx, rx))

    def gradient(self, H, Jx, rx, Cx, y_bar):
        """``g = Jx'*rx + Cx'*y_bar`` — src/basic_tralcnlss.jl:45 (new_point), :74 (first_derivatives)."""
        return Jx.T @ rx + Cx.T @ y_bar

    def jtr(self, J, r):
        """``jac_res(x)' * residuals(x)`` — src/basic_tralcnlss.jl:893 (least_squares_multipliers)."""
        return J.T @ r


def _hook(ops, name):
    """Backends written before a hook existed fall back to the reference expression."""
    return getattr(ops, name, None) or getattr(NumpyOps(), name)


def build_step_bounds(x_minor, lincons: MixedConstraints, delta: float):
    """The ``w_l``/``w_u`` construction of ``minor_iterate`` —
    src/basic_tralcnlss.jl:660-665 (quirk: only the FIXED variables get finite
    bounds)."""
    n = x_minor.shape[0]
    w_u = np.full(n, np.inf)
    w_l = np.full(n, -np.inf)
    f = lincons.fixvars
    w_u[f] = np.minimum(lincons.xupp[f] - x_minor[f], delta)
    w_l[f] = np.maximum(lincons.xlow[f] - x_minor[f], -delta)
    return w_l, w_u


def minor_iterate(x, s, g_model, H, lincons, delta, kappa2, ops=None):
    """src/basic_tralcnlss.jl:649-675."""
    ops = ops or NumpyOps()
    x_minor = x + s
    w_l, w_u = build_step_bounds(x_minor, lincons, delta)
    w, cg_status = ops.projected_cg(g_model, H, w_l, w_u, lincons, kappa2)  # :667
    if cg_status != CGStatus.negative_curvature:       # :669
        alpha = linesearch(g_model, H, w, w_l, w_u, lincons.fixvars, vthv_fn=ops.vthv)
        with np.errstate(all="ignore"):
            w = alpha * w
    return w, cg_status


# --------------------------------------------------------------------------- #
# Callers (context, needed only to drive BASELINE config 1 without Julia).
# --------------------------------------------------------------------------- #
def next_breakpoint(d, s, d_l, d_u, fix_bounds):
    """src/basic_tralcnlss.jl:536-562 (0-based index, -1 if none)."""
    theta = math.inf
    ind = -1
    for i in range(d.shape[0]):
        if not fix_bounds[i]:
            if d[i] < 0:
                theta_try = (d_l[i] - s[i]) / d[i]
            elif d[i] > 0:
                theta_try = (d_u[i] - s[i]) / d[i]
            else:
                theta_try = math.inf
            if theta_try < theta:
                theta = theta_try
                ind = i
    return theta, ind


def cauchy_step(x, g, H, chol_aat_L, lincons, delta, ops=None):
    """6-argument method, src/basic_tralcnlss.jl:574-639."""
    ops = ops or NumpyOps()
    m, n = lincons.lineq.shape
    nmm = n - m
    s_c = np.zeros(n)
    active_bounds_inplace(lincons, x, chol_aat_L)      # :591
    d = ops.projection(lincons, -g)                    # :592
    d_u = np.minimum(lincons.xupp - x, delta)          # :602
    d_l = np.maximum(lincons.xlow - x, -delta)         # :603
    Hd = ops.hmul(H, d)                                # :609
    phi_p = float(np.dot(s_c, Hd) + np.dot(g, d))      # :610
    phi_pp = float(np.dot(d, Hd))                      # :611
    min_found = False
    while (not min_found) and (nb_fix(lincons) < nmm):  # :615
        theta, ind = next_breakpoint(d, s_c, d_l, d_u, lincons.fixvars)
        delta_t = (-phi_p / phi_pp) if phi_pp > 0 else 0.0
        if phi_p >= 0:                                 # :620
            min_found = True
        elif phi_p < 0 and phi_pp > 0 and delta_t < theta:
            delta_t = -phi_p / phi_pp
            s_c = s_c + delta_t * d                    # :625
            min_found = True
        else:
            s_c = s_c + theta * d                      # :628
            add_active(lincons, chol_aat_L, ind)       # :631
            d = ops.projection(lincons, -g)            # :632
            Hd = ops.hmul(H, d)                        # :633
            phi_p = float(np.dot(s_c, Hd) + np.dot(g, d))
            phi_pp = float(np.dot(d, Hd))
    return s_c


def norm_reduced_gradient(g, lincons, ops=None):
    """src/basic_tralcnlss.jl:869-875."""
    ops = ops or NumpyOps()
    return float(np.linalg.norm(ops.projection(lincons, -g)))


def inner_step(x, g, H, chol_aat_L, lincons, delta, nb_minor_step, kappa2, kappa3, ops=None, log=None):
    """src/basic_tralcnlss.jl:394-460."""
    ops = ops or NumpyOps()
    m, n = lincons.lineq.shape
    if hasattr(ops, "cauchy_step"):         # backend provides the whole Cauchy search (bh_cauchy_step); it must leave
        s = ops.cauchy_step(x, g, H, chol_aat_L, lincons, delta)   # lincons.fixvars / chol_L as the reference would
    else:
        s = cauchy_step(x, g, H, chol_aat_L, lincons, delta, ops)  # :410
    hmul_add = getattr(ops, "hmul_add", lambda H_, s_, g_: ops.hmul(H_, s_) + g_)
    g_minor = hmul_add(H, s, g)                                    # :412
    j = 1
    norm_reduced_g = norm_reduced_gradient(g, lincons, ops)        # :420
    norm_reduced_g_minor = norm_reduced_gradient(g_minor, lincons, ops)
    approx_solved = norm_reduced_g_minor <= kappa3 * norm_reduced_g
    allowed_minor_step = n - m - nb_fix(lincons)                   # :425
    max_minor_step = min(nb_minor_step, allowed_minor_step)
    cg_stop = False
    while j <= max_minor_step and (not approx_solved) and (not cg_stop):  # :430
        if hasattr(ops, "minor_iterate"):      # backend provides the whole minor iterate (bh_minor_iterate)
            w, cg_status = ops.minor_iterate(x, s, g_minor, H, lincons, delta, kappa2)
        else:
            w, cg_status = minor_iterate(x, s, g_minor, H, lincons, delta, kappa2, ops)
        cg_stop = cg_status == CGStatus.negative_curvature
        s = s + w                                                  # :436
        g_minor = hmul_add(H, s, g)                                # :437
        active_indx = active_bounds(lincons, x, s, delta)          # :439
        if m + active_indx.shape[0] <= n:                          # :441
            add_active(lincons, chol_aat_L, active_indx)
            norm_reduced_g = norm_reduced_gradient(g, lincons, ops)
            norm_reduced_g_minor = norm_reduced_gradient(g_minor, lincons, ops)
            approx_solved = norm_reduced_g_minor <= kappa3 * norm_reduced_g
        else:
            approx_solved = True
            active_bounds_inplace(lincons, x + s, chol_aat_L)      # :452
        if log is not None:
            # (tag, CG status, active bounds after the update, the ratio the loop's exit test compares with 1 (:449))
            log.append(("minor", int(cg_status), nb_fix(lincons),
                        norm_reduced_g_minor / (kappa3 * norm_reduced_g) if norm_reduced_g > 0 else math.inf))
        j += 1
    model_reduction = float(np.dot(g, s)) + 0.5 * ops.vthv(H, s)   # :458
    return s, model_reduction


def initial_tr(g, tr_factor: float = 0.1) -> float:
    """src/basic_tralcnlss.jl:817-819."""
    return tr_factor * float(np.linalg.norm(g))


def update_tr(delta, rho, eta1, eta2, gamma1, gamma2):
    """src/basic_tralcnlss.jl:821-837."""
    if rho > eta2:
        return gamma2 * delta
    if rho < eta1:
        return gamma1 * delta
    return delta


def solve_subproblem(x0, y, mu, residuals, nlconstraints, jac_res, jac_nlcons, chol_aat_L, lincons,
                     nb_minor_step, k_max, omega_tol, eta1, eta2, gamma1, gamma2, kappa2, kappa3,
                     ops=None, log=None):
    """src/basic_tralcnlss.jl:303-378 (+ new_point :32-49, evaluate_al :51-61,
    first_derivatives :63-77, second_derivatives :79-85)."""
    ops = ops or NumpyOps()
    x = np.array(x0, dtype=np.float64, copy=True)
    sqnorm, gradient = _hook(ops, "residual_sqnorm"), _hook(ops, "gradient")
    rx, cx = residuals(x), nlconstraints(x)
    Jx, Cx = jac_res(x), jac_nlcons(x)
    y_bar = y + mu * cx
    mx = 0.5 * sqnorm(rx) + np.dot(y, cx) + 0.5 * mu * np.dot(cx, cx)    # :44
    H = ops.new_hessian(Jx, Cx, mu)                                      # :46 (built first so a device backend can use it for g)
    g = gradient(H, Jx, rx, Cx, y_bar)                                   # :45
    pix = math.inf
    delta = initial_tr(g)
    k = 1
    solved = False
    while (not solved) and k <= k_max:
        if hasattr(ops, "inner_step"):       # backend owns the whole inner step (device-resident minor loop)
            s, pred = ops.inner_step(x, g, H, chol_aat_L, lincons, delta, nb_minor_step, kappa2, kappa3, log)
        else:
            s, pred = inner_step(x, g, H, chol_aat_L, lincons, delta, nb_minor_step, kappa2, kappa3, ops, log)
        x_next = x + s
        rx_next, cx_next = residuals(x_next), nlconstraints(x_next)
        mx_next = 0.5 * sqnorm(rx_next) + np.dot(y, cx_next) + 0.5 * mu * np.dot(cx_next, cx_next)   # :58
        ared = mx_next - mx
        with np.errstate(all="ignore"):
            rho = ared / pred
        if rho > eta1:                                             # :358
            x = x_next
            rx, cx, mx = rx_next, cx_next, mx_next
            Jx, Cx = jac_res(x), jac_nlcons(x)
            y_bar = y + mu * cx
            H = ops.new_hessian(Jx, Cx, mu)                              # :84
            g = gradient(H, Jx, rx, Cx, y_bar)                           # :74
        delta = update_tr(delta, rho, eta1, eta2, gamma1, gamma2)
        pix = norm_reduced_gradient(g, lincons, ops)               # :369
        solved = pix < omega_tol
        if log is not None:
            # (tag, rho, pix/omega, new delta, ared, pred, |mx|): rho = ared/pred (:353-354) is the ratio of a difference of two
            # nearly equal objective values to a tiny model reduction once the steps get small — the log keeps the operands
            # so that a reader can tell a rounding-dominated rho (|ared| ~ eps*|mx|) from a real disagreement
            log.append(("tr", float(rho), float(pix / omega_tol) if omega_tol > 0 else math.inf, float(delta),
                        float(ared), float(pred), float(max(abs(mx), abs(mx_next)))))
        k += 1
    return x, cx, pix


def least_squares_multipliers(x, residuals, jac_res, jac_nlcons, ops=None):
    """src/basic_tralcnlss.jl:887-903."""
    g = _hook(ops or NumpyOps(), "jtr")(jac_res(x), residuals(x))       # :893
    C = jac_nlcons(x)
    L = chol_lower(C @ C.T)
    b = -C @ g
    v = _solve_lower(L, b)
    return _solve_upper_from_lower(L, v)


def tralcnllss(x0, residuals, jac_res, nlconstraints, jac_nlcons, A, b, x_l, x_u, *,
               mu0=10.0, tau=100.0, omega0=1.0, eta0=1.0, feas_tol=SQRT_EPS, crit_tol=SQRT_EPS,
               k_crit=1.0, k_feas=0.1, beta_crit=1.0, beta_feas=0.9, eta1=0.25, eta2=0.75,
               gamma1=0.0625, gamma2=2.0, gamma_c=10.0, kappa1=1e-2, kappa2=0.1, kappa3=0.1,
               max_outer_iter=500, max_inner_iter=500, max_minor_iter=50, ops=None, log=None):
    """src/basic_tralcnlss.jl:167-298 (logging of src/misc.jl omitted; ``b`` is
    accepted and never used, as in the reference)."""
    ops = ops or NumpyOps()
    assert (0 < eta1 <= eta2 < 1) and (0 < gamma1 < 1 < gamma2)
    A = np.asarray(A, dtype=np.float64)
    chol_aat_L = chol_lower(A @ A.T)                               # :206
    x = np.array(x0, dtype=np.float64, copy=True)
    cx = nlconstraints(x)
    mu = mu0
    omega = omega0 / (mu0 ** k_crit)                               # :153-163
    eta = eta0 / (mu0 ** k_feas)
    y = least_squares_multipliers(x, residuals, jac_res, jac_nlcons, ops)
    polyhedron = make_mixed_constraints(A, chol_aat_L, l=x_l, u=x_u)
    first_order_critical = False
    outer_iter = 1
    while (not first_order_critical) and outer_iter <= max_outer_iter:   # :246
        x_next, cx_next, pix = solve_subproblem(
            x, y, mu, residuals, nlconstraints, jac_res, jac_nlcons, chol_aat_L, polyhedron,
            max_minor_iter, max_inner_iter, omega, eta1, eta2, gamma1, gamma2, kappa2, kappa3,
            ops=ops, log=log)
        feas_measure = float(np.linalg.norm(cx_next))
        if feas_measure <= eta:                                    # :273
            x = x_next
            cx = cx_next
            first_order_critical = (pix <= crit_tol) and (feas_measure <= feas_tol)
            if not first_order_critical:
                y = y + mu * cx                                    # :905-911
                omega /= mu ** beta_crit
                eta /= mu ** beta_feas
        else:
            mu *= tau
            omega = omega0 / (mu ** k_crit)
            eta = eta0 / (mu ** k_feas)
        outer_iter += 1
    return x, y


def is_feasible(x, A, x_l, x_u, b):
    """src/basic_tralcnlss.jl:142-150 (``isapprox`` = rtol sqrt(eps) on the 2-norm)."""
    Ax = A @ x
    ok = np.linalg.norm(Ax - b) <= SQRT_EPS * max(np.linalg.norm(Ax), np.linalg.norm(b))
    return bool(ok and np.all(x_l <= x) and np.all(x <= x_u))


# --------------------------------------------------------------------------- #
# Stand-in for ``projection_polyhedron`` (src/polyhedral_constraints.jl:179-198,
# a JuMP/Ipopt QP; only used by the reference's *test* for its optimality
# measure).  Exact enumeration of bound patterns — small n only.
# --------------------------------------------------------------------------- #
def projection_polyhedron_small(x, A, b, l, u):
    import itertools
    n = x.shape[0]
    assert n <= 8
    best = None
    for pattern in itertools.product((0, 1, 2), repeat=n):      # 0 free, 1 at lower, 2 at upper
        pattern = np.array(pattern)
        free = pattern == 0
        v = np.where(pattern == 1, l, np.where(pattern == 2, u, 0.0))
        if not np.all(np.isfinite(v[~free])):
            continue
        Af = A[:, free]
        rhs = b - A[:, ~free] @ v[~free]
        xf = x[free]
        if Af.shape[1] == 0:
            if np.linalg.norm(rhs) > 1e-10:
                continue
            lam = np.zeros(A.shape[0])
        else:
            M = Af @ Af.T
            if np.linalg.matrix_rank(M) < M.shape[0]:
                if np.linalg.norm(rhs - Af @ xf) > 1e-10 and Af.shape[1] < A.shape[0]:
                    continue
                lam = np.linalg.lstsq(M, Af @ xf - rhs, rcond=None)[0]
            else:
                lam = np.linalg.solve(M, Af @ xf - rhs)
            v[free] = xf - Af.T @ lam
        if np.linalg.norm(A @ v - b) > 1e-9:
            continue
        if np.any(v < l - 1e-12) or np.any(v > u + 1e-12):
            continue
        dist = float(np.dot(v - x, v - x))
        if best is None or dist < best[0]:
            best = (dist, v.copy())
    assert best is not None
    return best[1]


# --------------------------------------------------------------------------- #
# Synthetic instances of SURVEY.md §8(d): splitmix64 counter generator shared by
# host (here), the C oracle and the device generator, so nobody ships gigabytes.
# --------------------------------------------------------------------------- #
_M64 = (1 << 64) - 1


def splitmix_uniform(seed: int, k: np.ndarray) -> np.ndarray:
    """u(seed,k): splitmix64 of ``seed + (k+1)*0x9E3779B97F4A7C15`` (the state
    after k+1 increments), top 53 bits -> uniform in [-1, 1)."""
    k = np.asarray(k, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _M64) + (k + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def synthetic_J(d: int, n: int, seed: int = 1, kind: int = 0, row0: int = 0, d_total: Optional[int] = None):
    """Rows ``[row0, row0+d)`` of the d_total x n synthetic Jacobian, element
    (i,j) = u(seed, i + j*d_total) / sqrt(d_total); ``kind=1`` scales column j
    by 10^(-3 j / n) (the ill-conditioned "ic" variant)."""
    d_total = d if d_total is None else d_total
    i = np.arange(row0, row0 + d, dtype=np.uint64)[:, None]
    j = np.arange(n, dtype=np.uint64)[None, :]
    J = splitmix_uniform(seed, i + j * np.uint64(d_total)) / math.sqrt(d_total)
    if kind == 1:
        J = J * (10.0 ** (-3.0 * np.arange(n) / n))[None, :]
    return np.asfortranarray(J)


@dataclass
class SyntheticBoxInstance:
    d: int
    n: int
    x: np.ndarray
    x_l: np.ndarray
    x_u: np.ndarray
    fixvars: np.ndarray
    r0: np.ndarray
    mu: float = 10.0
    kappa2: float = 0.1


def synthetic_box_vectors(d: int, n: int, fix_every: int = 8) -> SyntheticBoxInstance:
    """Config 2/3 recipe of SURVEY.md §8(d): bounds +-1, x = 0.5*u(2,.), every
    ``fix_every``-th variable on a bound (alternating lower/upper), r0 = u(3,.)."""
    x = 0.5 * splitmix_uniform(2, np.arange(n))
    x_l = -np.ones(n)
    x_u = np.ones(n)
    fix = np.zeros(n, dtype=bool)
    if fix_every > 0:
        idx = np.arange(0, n, fix_every)
        fix[idx] = True
        x[idx[0::2]] = -1.0
        x[idx[1::2]] = 1.0
    r0 = splitmix_uniform(3, np.arange(d))
    return SyntheticBoxInstance(d, n, x, x_l, x_u, fix, r0)
