"""Synthetic benchmark instances of SURVEY.md §8(d) (product side; the oracle has its own copy and the tests
cross-check the two).  Counter-based generator so host and device produce identical values without shipping
gigabytes: u(seed,k) = splitmix64(seed + (k+1)*0x9E3779B97F4A7C15) -> top 53 bits -> uniform in [-1, 1)."""
import math

import numpy as np


def splitmix_uniform(seed, k):
    k = np.asarray(k, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & ((1 << 64) - 1)) + (k + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def column_scale(n, kind):
    """kind 0: none (well conditioned, "wc"); kind 1: 10^(-3j/n) ("ic")."""
    if kind == 0:
        return None
    return 10.0 ** (-3.0 * np.arange(n) / n)


def box_vectors(n, fix_every=8):
    """x = 0.5*u(2,.), bounds +-1, every `fix_every`-th variable on a bound (alternating lower/upper)."""
    x = 0.5 * splitmix_uniform(2, np.arange(n))
    x_l, x_u = -np.ones(n), np.ones(n)
    fix = np.zeros(n, dtype=bool)
    if fix_every > 0:
        idx = np.arange(0, n, fix_every)
        fix[idx] = True
        x[idx[0::2]] = -1.0
        x[idx[1::2]] = 1.0
    return x, x_l, x_u, fix


def residual_rows(row_lo, row_hi):
    """r0_i = u(3, i) for this rank's rows."""
    return splitmix_uniform(3, np.arange(row_lo, row_hi))


def step_bounds(x_minor, xlow, xupp, fixvars, delta):
    """The w_l / w_u that ``minor_iterate`` hands to ``projected_cg`` — src/basic_tralcnlss.jl:662-665: +-Inf on the
    free variables, min(xupp - x, delta) / max(xlow - x, -delta) on the fixed ones."""
    n = x_minor.shape[0]
    w_u, w_l = np.full(n, np.inf), np.full(n, -np.inf)
    f = np.asarray(fixvars, dtype=bool)
    w_u[f] = np.minimum(xupp[f] - x_minor[f], delta)
    w_l[f] = np.maximum(xlow[f] - x_minor[f], -delta)
    return w_l, w_u


def initial_tr(g, tr_factor=0.1):
    """``initial_tr`` — src/basic_tralcnlss.jl:817-819."""
    return tr_factor * float(np.linalg.norm(g))
