"""A seeded, medium-sized constrained nonlinear least-squares instance (n parameters, d residuals, one nonlinear equality,
mA linear equalities, box bounds) for end-to-end runs of the restated outer iteration — the reference bundles only the
3-variable sphere regression."""
import numpy as np


class NLSProblem:
    def __init__(self, d=256, n=48, mA=2, seed=0):
        rng = np.random.default_rng(seed)
        self.d, self.n, self.mA = d, n, mA
        self.M = rng.standard_normal((d, n)) / np.sqrt(n)
        x_true = rng.uniform(-0.8, 0.8, n)
        self.curv = 0.1
        lin = self.M @ x_true
        self.y = lin + self.curv * lin ** 2 + 0.01 * rng.standard_normal(d)
        self.radius2 = float(x_true @ x_true)
        self.A = rng.standard_normal((mA, n))
        self.b = self.A @ x_true
        self.x_l = -np.ones(n)
        self.x_u = np.ones(n)
        # feasible start for the linear equalities and the bounds: shrink x_true towards a null-space perturbation
        Z = np.eye(n) - self.A.T @ np.linalg.solve(self.A @ self.A.T, self.A)
        pert = Z @ rng.standard_normal(n)
        pert *= 0.1 / max(np.max(np.abs(pert)), 1e-12)
        self.x0 = np.clip(x_true + pert, -0.95, 0.95)
        self.x0 = self.x0 - self.A.T @ np.linalg.solve(self.A @ self.A.T, self.A @ self.x0 - self.b)
        self.x_true = x_true

    def r(self, x):
        lin = self.M @ x
        return lin + self.curv * lin ** 2 - self.y

    def jac_r(self, x):
        lin = self.M @ x
        return (1.0 + 2.0 * self.curv * lin)[:, None] * self.M

    def c(self, x):
        return np.array([x @ x - self.radius2])

    def jac_c(self, x):
        return (2.0 * x)[None, :]
