"""Backends ("ops") that plug the C ABI into the oracle's restated outer iteration (tralcnllss -> solve_subproblem ->
inner_step -> minor_iterate), the executable stand-in for the unchanged Julia driver: every hot-path call goes through the
library exactly as julia/BEnlsipHIP.jl routes it.  Test infrastructure (imports the oracle for its CGStatus enum and for the
shadow comparison)."""
import numpy as np

import benlsip_ref as R


class HipOps:
    """`*`, vthv, projection!, projected_cg on the device (the minimal shim: SURVEY.md §8 a2-a8)."""

    def __init__(self, bh):
        self.bh = bh
        self.n_pcg = 0
        self.ties = []                 # (minor-iterate index, tie info) of every projected_cg whose tie log is not empty

    def new_hessian(self, J, C, mu):
        return self.bh.AlHessian(J, C, mu)

    def hmul(self, H, v):
        return self.bh.hmul(H, v)

    def vthv(self, H, v):
        return self.bh.vthv(H, v)

    def _dev(self, lincons):
        dev = getattr(lincons, "_dev", None)
        if dev is None:
            dev = self.bh.MixedConstraints(lincons.lineq, None, None, l=lincons.xlow, u=lincons.xupp)
            lincons._dev = dev
        dev.set_active(lincons.fixvars, lincons.chol_L)
        return dev

    def projection(self, lincons, r):
        return self.bh.projection(self._dev(lincons), r)

    def _note(self, info):
        t = info.get("ties")
        if t is not None:
            self.last_ties = t
            if t["tie_flags"]:
                self.ties.append((self.n_pcg, t))

    def projected_cg(self, g_minor, H, w_l, w_u, lincons, kappa2):
        self.n_pcg += 1
        w, status, info = self.bh.projected_cg(g_minor, H, w_l, w_u, self._dev(lincons), kappa2, full_output=True)
        self.last_iters = info["iters"]
        self._note(info)
        return w, R.CGStatus(int(status))


class HipOpsDeviceMinor(HipOps):
    """Same, with the whole minor iterate (step bounds + projected_cg + linesearch + scaling) and H*s+g on the device
    (bh_minor_iterate, bh_hmul_add: SURVEY.md §8 a10 / f-2)."""

    def minor_iterate(self, x, s, g_model, H, lincons, delta, kappa2):
        self.n_pcg += 1
        w, status, info = self.bh.minor_iterate(x, s, g_model, H, self._dev(lincons), delta, kappa2, full_output=True)
        self.last_iters = info["iters"]
        self._note(info)
        return w, R.CGStatus(int(status))

    def hmul_add(self, H, s, g):
        return self.bh.hmul_add(H, s, g)


class HipOpsDeviceAll(HipOpsDeviceMinor):
    """Same, plus the Cauchy search on the device (bh_cauchy_step); the oracle-side lincons is brought
    to the state the reference's cauchy_step leaves behind (fixvars + refreshed factor)."""

    def cauchy_step(self, x, g, H, chol_aat_L, lincons, delta):
        dev = self._dev(lincons)
        s = self.bh.cauchy_step(x, g, H, dev, delta)
        lincons.fixvars = dev.fixvars.copy()
        R.update_chol(lincons, chol_aat_L)
        return s


class HipOpsResident(HipOpsDeviceAll):
    """Same, with the WHOLE inner step owned by the library's device-resident chain (bh.inner_step: bh_cauchy_step_dev,
    bh_minor_iterate_dev, bh_step_accumulate_dev, bh_proj_update_active_dev, bh_reduced_gradient_norm_dev,
    bh_model_reduction_dev — SURVEY.md §8 f-1/f-2): between the upload of x, g and the download of s no n-vector crosses PCIe."""

    def __init__(self, bh):
        super().__init__(bh)
        self.loop_bytes = 0
        self.loop_minor = 0

    def inner_step(self, x, g, H, chol_aat_L, lincons, delta, nb_minor_step, kappa2, kappa3, log):
        dev = self._dev(lincons)
        s, pred, info = self.bh.inner_step(x, g, H, dev, delta, nb_minor_step, kappa2, kappa3, full_output=True)
        lincons.fixvars = dev.fixvars.copy()
        R.update_chol(lincons, chol_aat_L)          # the oracle-side object is brought to the state the reference leaves behind
        self.n_pcg += len(info["minor"])
        self.loop_bytes += info["pcie_bytes_in_loop"]
        self.loop_minor += len(info["minor"]) + 1
        self.last_info = info
        if log is not None:
            for st, iters, nfix, ratio in info["minor"]:
                log.append(("minor", int(st), int(nfix), float(ratio)))
        return s, pred


class ShardedHipOps(HipOpsDeviceAll):
    """Row-sharded run: this rank's `residuals` / `jac_res` callbacks return rows [lo, hi) only, the
    library's communicator is up, and the three places where the driver touches residual rows directly — mx (:44,:58),
    g (:45,:74) and the least-squares multipliers' J'r (:893) — are routed to their all-reduced entry points, as the
    multi-rank methods of julia/BEnlsipHIP.jl do.  Everything else is replicated and must stay bit-identical."""

    def residual_sqnorm(self, rx):
        return self.bh.resid_sqnorm(rx)

    def gradient(self, H, Jx, rx, Cx, y_bar):
        return self.bh.gradient(H, rx, y_bar)

    def jtr(self, J, r):
        H = self.bh.AlHessian(J, None, 0.0)
        try:
            return H.jtv(r)
        finally:
            H.close()


class ShardedResidentOps(ShardedHipOps, HipOpsResident):
    """Row-sharded AND device-resident: the library's inner_step chain (HipOpsResident) with the communicator up — its Cauchy
    search, minor iterates, H*s+g and model reduction all-reduce inside the device-pointer entry points, whose scalars come back
    through the mailbox on every rank."""


class ShadowOps:
    """Runs the device backend AND the oracle on identical inputs at every hot-path call and carries the DEVICE result
    forward.  A divergence of two free-running solves says nothing about where it started; here every call is compared on
    the same operands, so the first discrepancy in status / iteration count / active set is pinned to one call, together
    with the device's tie log for it (such a discrepancy must be a logged tie)."""

    def __init__(self, dev_ops, relnorm_tol=1e-6, sens_samples=None):
        self.dev = dev_ops
        self.cpu = R.NumpyOps()
        self.events = []          # discrepancies: dicts
        self.calls = 0
        self.minor = 0
        self.worst = {}           # op -> largest relative deviation seen
        self.min_margin = (np.inf, None)
        self.tol = relnorm_tol
        self.n_cg_samples = sens_samples or 3          # perturbed oracle evaluations behind "oracle_sensitivity"
        self.n_cauchy_samples = sens_samples or 16
        # the driver asks the backend for whole-step methods with hasattr(): offer exactly what the device backend offers
        if hasattr(dev_ops, "minor_iterate"):
            self.minor_iterate = self._minor_iterate
        if hasattr(dev_ops, "cauchy_step"):
            self.cauchy_step = self._cauchy_step
        if hasattr(dev_ops, "hmul_add"):
            self.hmul_add = self._hmul_add

    def _rel(self, op, a, b):
        a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
        with np.errstate(all="ignore"):
            d = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
        if not np.isfinite(d):
            d = 0.0 if np.array_equal(a, b, equal_nan=True) else np.inf
        self.worst[op] = max(self.worst.get(op, 0.0), d)
        return d

    def new_hessian(self, J, C, mu):
        return (self.dev.new_hessian(J, C, mu), self.cpu.new_hessian(J, C, mu))

    def hmul(self, H, v):
        out = self.dev.hmul(H[0], v)
        self._rel("hmul", out, self.cpu.hmul(H[1], v))
        return out

    def _hmul_add(self, H, s, g):
        out = self.dev.hmul_add(H[0], s, g)
        self._rel("hmul_add", out, self.cpu.hmul(H[1], s) + g)
        return out

    def vthv(self, H, v):
        out = self.dev.vthv(H[0], v)
        self._rel("vthv", [out], [self.cpu.vthv(H[1], v)])
        return out

    def projection(self, lincons, r):
        out = self.dev.projection(lincons, r)
        ref = self.cpu.projection(lincons, r)
        # a projection is a difference r - A~'(...): its error scales with ||r||, not with the (possibly tiny) result
        with np.errstate(all="ignore"):
            d = float(np.linalg.norm(out - ref) / max(np.linalg.norm(r), 1e-300))
        self.worst["projection"] = max(self.worst.get("projection", 0.0), d)
        if d > 1e-10:
            self.events.append(dict(op="projection", minor=self.minor, rel=d, nfix=int(lincons.fixvars.sum()),
                                    shrink=float(np.linalg.norm(ref) / max(np.linalg.norm(r), 1e-300))))
        return out

    def projected_cg(self, g_minor, H, w_l, w_u, lincons, kappa2):
        """(backends without a whole-step minor_iterate: the oracle's minor_iterate calls this, then linesearch -> vthv)"""
        self.minor += 1
        w, st = self.dev.projected_cg(g_minor, H[0], w_l, w_u, lincons, kappa2)
        it_dev, ties = self.dev.last_iters, getattr(self.dev, "last_ties", None)
        w_o, st_o, it_o = R.projected_cg(g_minor, H[1], w_l, w_u, lincons, kappa2)
        d = self._rel("projected_cg", w, w_o)
        if ties is not None and ties["min_margin"] < self.min_margin[0]:
            self.min_margin = (ties["min_margin"], dict(ties, minor=self.minor))
        if int(st) != int(st_o) or it_dev != it_o or d > self.tol:
            rng = np.random.default_rng(self.minor)
            sens = 0.0
            for _ in range(self.n_cg_samples):
                g2 = g_minor * (1.0 + 2.2e-16 * rng.uniform(-1.0, 1.0, g_minor.shape[0]))
                w2, _, _ = R.projected_cg(g2, H[1], w_l, w_u, lincons, kappa2)
                sens = max(sens, float(np.linalg.norm(w2 - w_o) / max(np.linalg.norm(w_o), 1e-300)))
            self.events.append(dict(op="projected_cg", minor=self.minor, status_dev=int(st), status_cpu=int(st_o), iters_dev=it_dev,
                                    iters_cpu=it_o, rel=d, oracle_sensitivity=sens, ties=ties))
        return w, st

    def _minor_iterate(self, x, s, g_model, H, lincons, delta, kappa2):
        self.minor += 1
        w, st = self.dev.minor_iterate(x, s, g_model, H[0], lincons, delta, kappa2)
        it_dev, ties = self.dev.last_iters, getattr(self.dev, "last_ties", None)
        # the oracle on the same operands (its own projected_cg reports the iteration count)
        w_l, w_u = R.build_step_bounds(x + s, lincons, delta)
        w_o, st_o, it_o = R.projected_cg(g_model, H[1], w_l, w_u, lincons, kappa2)
        if st_o != R.CGStatus.negative_curvature:
            with np.errstate(all="ignore"):
                w_o = R.linesearch(g_model, H[1], w_o, w_l, w_u, lincons.fixvars) * w_o
        d = self._rel("minor_iterate", w, w_o)
        if ties is not None and ties["min_margin"] < self.min_margin[0]:
            self.min_margin = (ties["min_margin"], dict(ties, minor=self.minor))
        if int(st) != int(st_o) or it_dev != it_o or d > self.tol:
            rng = np.random.default_rng(self.minor)
            sens = 0.0
            for _ in range(self.n_cg_samples):           # the oracle's own w under last-bit perturbations of its right-hand side
                g2 = g_model * (1.0 + 2.2e-16 * rng.uniform(-1.0, 1.0, g_model.shape[0]))
                w2, st2, _ = R.projected_cg(g2, H[1], w_l, w_u, lincons, kappa2)
                if st2 != R.CGStatus.negative_curvature:
                    with np.errstate(all="ignore"):
                        w2 = R.linesearch(g2, H[1], w2, w_l, w_u, lincons.fixvars) * w2
                sens = max(sens, float(np.linalg.norm(w2 - w_o) / max(np.linalg.norm(w_o), 1e-300)))
            self.events.append(dict(op="minor_iterate", minor=self.minor, status_dev=int(st), status_cpu=int(st_o), iters_dev=it_dev,
                                    iters_cpu=it_o, rel=d, oracle_sensitivity=sens, ties=ties))
        return w, st

    def _cauchy_step(self, x, g, H, chol_aat_L, lincons, delta):
        import copy
        shadow = copy.copy(lincons)
        shadow._dev = None
        shadow.fixvars = lincons.fixvars.copy()
        fix0, chol0 = lincons.fixvars.copy(), lincons.chol_L
        s = self.dev.cauchy_step(x, g, H[0], chol_aat_L, lincons, delta)
        s_o = R.cauchy_step(x, g, H[1], chol_aat_L, shadow, delta, self.cpu)
        d = self._rel("cauchy_step", s, s_o)
        if not np.array_equal(shadow.fixvars, lincons.fixvars) or d > self.tol:
            # how far does the ORACLE's own Cauchy step move when g is perturbed in its last bits?  (the search direction is
            # P(-g), a cancelling difference once g is nearly orthogonal to the null space: near a critical point its
            # relative accuracy is eps*||g||/||P(-g)||, whoever computes it)
            rng = np.random.default_rng(self.minor)
            sens = 0.0
            for _ in range(self.n_cauchy_samples):         # (the outcome can be bimodal — e.g. 37 or 41 breakpoints — so a handful of samples is not enough)
                sh2 = copy.copy(lincons)
                sh2._dev = None
                sh2.fixvars = fix0.copy()
                sh2.chol_L = chol0
                g2 = g * (1.0 + 2.2e-16 * rng.uniform(-1.0, 1.0, g.shape[0]))
                s2 = R.cauchy_step(x, g2, H[1], chol_aat_L, sh2, delta, self.cpu)
                sens = max(sens, float(np.linalg.norm(s2 - s_o) / max(np.linalg.norm(s_o), 1e-300)))
            # ... and when the SAME projector is evaluated in its reduced form (mask, then chol(A_free A_free'): mathematically the
            # reference's augmented form, rounded differently — the form the device uses)?  Near a critical point phi' = g'd is
            # -|d|^2 + g'(error of d), so two exact-arithmetic-identical projectors disagree on the second segment's minimiser by
            # eps |g|^2 / |d|^2 (measured on config 1: 12 % at |g|/|P(-g)| = 8e6, while 1-ulp perturbations of g move the augmented
            # form by 1e-9: tests/manual/sphere_cauchy_event_probe.py).  The oracle FAMILY's spread is what a device can be held to.
            from _util import ReducedFormOps
            for k in range(max(2, self.n_cauchy_samples // 4)):
                sh2 = copy.copy(lincons)
                sh2._dev = None
                sh2.fixvars = fix0.copy()
                sh2.chol_L = chol0
                g2 = g if k == 0 else g * (1.0 + 2.2e-16 * rng.uniform(-1.0, 1.0, g.shape[0]))
                s2 = R.cauchy_step(x, g2, H[1], chol_aat_L, sh2, delta, ReducedFormOps())
                sens = max(sens, float(np.linalg.norm(s2 - s_o) / max(np.linalg.norm(s_o), 1e-300)))
            red = self.cpu.projection(shadow, -g)
            self.events.append(dict(op="cauchy_step", minor=self.minor, rel=d, oracle_sensitivity=sens,
                                    fix_dev=int(lincons.fixvars.sum()), fix_cpu=int(shadow.fixvars.sum()),
                                    g_over_reduced_g=float(np.linalg.norm(g) / max(np.linalg.norm(red), 1e-300)),
                                    operands=dict(x=x.copy(), g=g.copy(), delta=delta, fix0=fix0.copy(), J=H[1].J.copy(), C=H[1].C.copy(), mu=H[1].mu,
                                                  s_dev=s.copy(), s_cpu=s_o.copy(), fix_dev_set=lincons.fixvars.copy(),
                                                  A=lincons.lineq.copy(), x_l=lincons.xlow.copy(), x_u=lincons.xupp.copy())))
        return s
