"""Seeded instance shared by the multi-rank worker processes and the parent test (tests/test_multirank_gpu.py)."""
import numpy as np


def make_problem(seed=11, d=3001, n=384, q=5, mA=3, nfix=40):
    rng = np.random.default_rng(seed)
    P = {}
    P["J"] = rng.standard_normal((d, n)) / np.sqrt(d) * np.logspace(0, -1.5, n)
    P["C"] = rng.standard_normal((q, n))
    P["A"] = rng.standard_normal((mA, n))
    P["mu"] = 7.0
    fix = np.zeros(n, dtype=bool)
    fix[rng.choice(n, nfix, replace=False)] = True
    P["fix"] = fix
    P["xlow"], P["xupp"] = -np.ones(n), np.ones(n)
    x = np.clip(0.4 * rng.standard_normal(n), -0.9, 0.9)
    x[fix] = np.where(rng.random(nfix) < 0.5, -1.0, 1.0)
    P["x"] = x
    s = 0.01 * rng.standard_normal(n)
    s[fix] = 0.0
    P["s"] = s
    P["v"] = rng.standard_normal(n)
    P["u"] = rng.standard_normal(d)
    P["rx"] = rng.standard_normal(d)
    P["ybar"] = rng.standard_normal(q)
    P["g_cauchy"] = rng.standard_normal(n)
    return P
