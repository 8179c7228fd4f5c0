"""Seeded random instances for the multi-rank fuzz (tests/multirank/fuzz_worker.py generates them on every rank, the parent test
regenerates them for the oracle): small and odd n, fewer rows than ranks, q > 0, linear equalities, fixed variables, tight and
wide bounds.  Plain NumPy: imported by the workers, so nothing from oracle/ here."""
import numpy as np


def step_bounds(x, xl, xu, fix, delta):
    """w_l, w_u of src/basic_tralcnlss.jl:660-666 (fixed variables keep a zero-width interval)."""
    wl = np.where(fix, 0.0, np.maximum(xl - x, -delta))
    wu = np.where(fix, 0.0, np.minimum(xu - x, delta))
    return wl, wu


def cases(seed, count):
    rng = np.random.default_rng(seed)
    for k in range(count):
        n = int(rng.integers(2, 90)) if rng.random() < 0.85 else int(rng.integers(90, 400))
        r = rng.random()
        if r < 0.12:
            d = int(rng.integers(1, 4))                   # fewer rows than ranks: some ranks own nothing
        elif r < 0.3:
            d = int(rng.integers(4, max(n + 2, 5)))       # rank-deficient H: zero / negative-curvature exits
        else:
            d = int(rng.integers(2 * n, 5 * n + 2))
        q = int(rng.integers(0, 3))
        mA = int(rng.integers(1, min(20, n - 1) + 1)) if (rng.random() < 0.5 and n > 2) else 0
        nact = int(rng.integers(0, max(1, (n - mA) // 3)))
        J = rng.standard_normal((d, n)) / np.sqrt(max(d, n))
        C = 0.3 * rng.standard_normal((q, n))
        A = rng.standard_normal((mA, n))
        box = float(rng.choice([1.0, 1.0, 1e3]))           # wide boxes: the CG loop ends by its own tests, not at a bound
        xl, xu = -box * np.ones(n), box * np.ones(n)
        x = np.clip(0.5 * rng.standard_normal(n), -0.95, 0.95)
        act = rng.choice(n, nact, replace=False)
        x[act] = np.where(rng.random(nact) < 0.5, -box, box)
        fix = np.zeros(n, dtype=bool)
        fix[act] = True
        g = rng.standard_normal(n) * float(rng.choice([1.0, 0.02]))
        delta = float(rng.choice([0.05, 0.5, 5.0])) * np.linalg.norm(g)
        wl, wu = step_bounds(x, xl, xu, fix, delta)
        if rng.random() < 0.25:
            wl = np.where(fix, wl, -0.05)
            wu = np.where(fix, wu, 0.05)
        yield dict(k=k, n=n, d=d, q=q, mA=mA, J=J, C=C, A=A, xl=xl, xu=xu, x=x, fix=fix, g=g, delta=delta, wl=wl, wu=wu,
                   kappa2=float(rng.choice([0.1, 1e-2])), mu=float(rng.choice([1.0, 10.0])),
                   g_cauchy=g * float(rng.choice([1.0, 30.0])), feasible_rows=(mA + nact < n))
