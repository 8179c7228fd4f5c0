"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI.

The reference is Julia; its seam is dispatch on ``AlHessian`` / ``MixedConstraints`` and the function
``projected_cg``.  With no Julia toolchain in this image the host side above the C ABI
is written in Python with the same names, argument meaning and error behaviour, so the parity tests
read like the reference's own tests (``test/structures.jl``).  ``julia/BEnlsipHIP.jl`` is the
``ccall`` shim for the real package.

    reference                                                   here
    AlHessian(J, C, mu)          src/basic_tralcnlss.jl:6-10    AlHessian(J, C, mu)
    H * v                        :102-106                       H * v   /  hmul(H, v)
    vthv(H, v)                   :92-96                         vthv(H, v)
    MixedConstraints(A, chol…)   src/polyhedral_constraints.jl:1-29   MixedConstraints(A, chol_L, fixed, l, u)
    left_mul / left_mul_tr       :72-98                         left_mul / left_mul_tr
    projection / projection!     :150-170                       projection / projection_
    projected_cg(...)            src/basic_tralcnlss.jl:690-764 projected_cg(...)
    factor_to_boundary(...)      :793-809                       factor_to_boundary(...)
    CG_status                    :12                            CGStatus (+ ``none`` for Julia's `nothing`)

Everything numerical runs on the GPU; there is no CPU fallback.
"""
import ctypes as ct
import enum
import math

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr

SQRT_EPS = math.sqrt(np.finfo(np.float64).eps)


class CGStatus(enum.IntEnum):
    """``@enum CG_status`` (src/basic_tralcnlss.jl:12); ``none`` = the reference returning `nothing`."""
    solved = 0
    bound_hit = 1
    negative_curvature = 2
    max_iter_reached = 3
    none = 4


class AlHessian:
    """Implicit Gauss-Newton Hessian ``J'J + mu C'C`` resident in HBM (src/basic_tralcnlss.jl:6-10).

    ``J`` is this rank's row block (d x n); ``C`` (q x n) is replicated.  The host arrays are only read
    during construction (the reference builds a new AlHessian whenever J changes, :46,:84).
    """

    def __init__(self, J, C=None, mu=0.0):
        lib = _lib.lib()
        J = np.asarray(J, dtype=np.float64)
        if J.ndim != 2:
            raise ValueError("J must be a matrix")
        d, n = J.shape
        Jf = np.asfortranarray(J)
        if C is None:
            C = np.zeros((0, n))
        C = np.asarray(C, dtype=np.float64)
        if C.ndim != 2 or C.shape[1] != n:
            raise ValueError("C must be q x n")
        q = C.shape[0]
        Cf = np.asfortranarray(C)
        self._h = _null_handle()
        check(lib.bh_hess_create(_byref(self._h), ptr(Jf) if d > 0 else None, d, n, max(d, 1),
                                 ptr(Cf) if q > 0 else None, q, max(q, 1), float(mu)), "bh_hess_create")
        self.d, self.n, self.q = d, n, q
        self._mu = float(mu)

    @classmethod
    def create_async(cls, J, C=None, mu=0.0):
        """``bh_hess_create_async``: returns while J is still on its way to HBM (chunked copy overlapped with the device
        transpose); ``wait()`` — or the first product — joins the upload.  The object keeps ``J`` alive until then."""
        lib = _lib.lib()
        self = cls.__new__(cls)
        Jf = np.asfortranarray(np.asarray(J, dtype=np.float64))
        d, n = Jf.shape
        Cm = np.zeros((0, n)) if C is None else np.asarray(C, dtype=np.float64)
        q = Cm.shape[0]
        Cf = np.asfortranarray(Cm)
        self._h = _null_handle()
        self._pending_J = Jf
        check(lib.bh_hess_create_async(_byref(self._h), ptr(Jf) if d > 0 else None, d, n, max(d, 1), ptr(Cf) if q > 0 else None, q,
                                       max(q, 1), float(mu)), "bh_hess_create_async")
        self.d, self.n, self.q = d, n, q
        self._mu = float(mu)
        return self

    def wait(self):
        check(_lib.lib().bh_hess_wait(self._h), "bh_hess_wait")
        self._pending_J = None

    @classmethod
    def from_device(cls, J_dev_ptr, d, n, ldJ=None, C=None, mu=0.0):
        """J already in HBM (column-major d x n at device address ``J_dev_ptr``, e.g. ``tensor.data_ptr()`` or a
        ``DeviceVector.ptr``): only the device transpose runs (``bh_hess_create_dev``)."""
        lib = _lib.lib()
        self = cls.__new__(cls)
        self._h = _null_handle()
        Cm = np.zeros((0, n)) if C is None else np.asarray(C, dtype=np.float64)
        q = Cm.shape[0]
        Cf = np.asfortranarray(Cm)
        check(lib.bh_hess_create_dev(_byref(self._h), J_dev_ptr, d, n, d if ldJ is None else ldJ, ptr(Cf) if q > 0 else None, q,
                                     max(q, 1), float(mu)), "bh_hess_create_dev")
        self.d, self.n, self.q = d, n, q
        self._mu = float(mu)
        return self

    @classmethod
    def synthetic(cls, d, n, row0=0, d_total=None, seed=1, colscale=None, mu=10.0):
        """Benchmark instance generated in HBM; see ``bh_hess_create_synthetic``."""
        lib = _lib.lib()
        self = cls.__new__(cls)
        self._h = _null_handle()
        d_total = d if d_total is None else d_total
        cs = None if colscale is None else as_f64(colscale, n)
        check(lib.bh_hess_create_synthetic(_byref(self._h), d, n, row0, d_total, seed, ptr(cs), float(mu)),
              "bh_hess_create_synthetic")
        self.d, self.n, self.q = d, n, 0
        self._mu = float(mu)
        return self

    @property
    def mu(self):
        return self._mu

    @mu.setter
    def mu(self, value):
        check(_lib.lib().bh_hess_set_mu(self._h, float(value)), "bh_hess_set_mu")
        self._mu = float(value)

    @property
    def handle(self):
        return self._h

    def __mul__(self, v):
        return hmul(self, v)

    def __matmul__(self, v):
        return hmul(self, v)

    def jv(self, v):
        """``H.J*v`` (src/basic_tralcnlss.jl:93,103)."""
        v = as_f64(v, self.n)
        out = np.empty(self.d)
        check(_lib.lib().bh_jv(self._h, ptr(v), ptr(out)), "bh_jv")
        return out

    def jtv(self, u):
        """``H.J'*u`` (src/basic_tralcnlss.jl:105; g = Jx'*rx at :45)."""
        u = as_f64(u, self.d)
        out = np.empty(self.n)
        check(_lib.lib().bh_jtv(self._h, ptr(u), ptr(out)), "bh_jtv")
        return out

    def stats(self):
        st = _lib.bh_stats_t()
        check(_lib.lib().bh_stats(self._h, ct.byref(st)), "bh_stats")
        return {name: getattr(st, name) for name, _ in st._fields_}

    def reset_stats(self):
        check(_lib.lib().bh_stats_reset(self._h), "bh_stats_reset")

    def time_kernel(self, kind, reps=20):
        """Average hipEvent milliseconds of one launch: kind 0 = fused J'(Jp), 1 = J v, 2 = J'u, 3..6 = read-only stream probe with 1/2/4/8 workgroups per CU."""
        ms = ct.c_double(0.0)
        check(_lib.lib().bh_time_kernel(self._h, kind, reps, ct.byref(ms)), "bh_time_kernel")
        return ms.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().bh_hess_destroy(self._h)
            self._h = _null_handle()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _null_handle():
    return ct.c_void_p()


def _byref(h):
    return ct.byref(h)


def hmul(H, v):
    """``Base.:*(H::AlHessian, v)`` — src/basic_tralcnlss.jl:102-106."""
    v = as_f64(v, H.n)
    out = np.empty(H.n)
    check(_lib.lib().bh_hmul(H.handle, ptr(v), ptr(out)), "bh_hmul")
    return out


def vthv(H, v):
    """``vthv(H, v)`` — src/basic_tralcnlss.jl:92-96."""
    v = as_f64(v, H.n)
    out = ct.c_double(0.0)
    check(_lib.lib().bh_vthv(H.handle, ptr(v), ct.byref(out)), "bh_vthv")
    return out.value


def pack_bitvector(fixvars):
    """Julia ``BitVector.chunks`` layout: bit i%64 of word i//64 <=> element i."""
    f = np.asarray(fixvars, dtype=bool)
    b = np.packbits(f, bitorder="little")
    pad = (-b.shape[0]) % 8
    if pad:
        b = np.concatenate([b, np.zeros(pad, dtype=np.uint8)])
    if b.shape[0] == 0:
        b = np.zeros(8, dtype=np.uint8)
    return np.ascontiguousarray(b).view(np.uint64)


class MixedConstraints:
    """``MixedConstraints`` (src/polyhedral_constraints.jl:1-29): ``lineq`` (mA x n), ``xlow``, ``xupp``,
    ``fixvars`` and the Cholesky factor ``chol`` (lower factor L of Ã Ãᵀ; only its lower triangle is read).

    The factor is maintained by the caller exactly as in the reference (``cholesky_aug_aat``, ``add_active!`` …
    stay on the host); assign ``fixvars`` / ``chol`` (or call ``set_active``) after every
    change and the device image is refreshed before the next projection.
    """

    def __init__(self, A, chol_L=None, fixed=None, l=None, u=None):
        lib = _lib.lib()
        A = np.asarray(A, dtype=np.float64)
        if A.ndim != 2:
            raise ValueError("A must be mA x n")
        mA, n = A.shape
        self.lineq = A
        self.xlow = np.full(n, -np.inf) if l is None else as_f64(l, n)
        self.xupp = np.full(n, np.inf) if u is None else as_f64(u, n)
        Af = np.asfortranarray(A)
        self._h = _null_handle()
        check(lib.bh_proj_create(_byref(self._h), ptr(Af) if mA > 0 else None, mA, n, max(mA, 1)), "bh_proj_create")
        self.mA, self.n = mA, n
        self._fixvars = np.zeros(n, dtype=bool) if fixed is None else np.asarray(fixed, dtype=bool).copy()
        self._chol = None if chol_L is None else np.asarray(chol_L, dtype=np.float64)
        self._dirty = True

    # -- state the reference mutates between CG calls -------------------------------------
    @property
    def fixvars(self):
        return self._fixvars

    @fixvars.setter
    def fixvars(self, value):
        self._fixvars = np.asarray(value, dtype=bool)
        self._dirty = True

    @property
    def chol(self):
        return self._chol

    @chol.setter
    def chol(self, value):
        self._chol = None if value is None else np.asarray(value, dtype=np.float64)
        self._dirty = True

    def set_active(self, fixvars, chol_L):
        self._fixvars = np.asarray(fixvars, dtype=bool)
        self._chol = None if chol_L is None else np.asarray(chol_L, dtype=np.float64)
        self._dirty = True

    def mark_dirty(self):
        self._dirty = True

    def nb_fix(self):
        """``nb_fix`` — src/polyhedral_constraints.jl:31."""
        return int(np.count_nonzero(self._fixvars))

    @property
    def handle(self):
        self._sync()
        return self._h

    def _sync(self):
        if not self._dirty:
            return
        chunks = pack_bitvector(self._fixvars)
        mpp = self.mA + self.nb_fix()
        L = None
        ldL = max(mpp, 1)
        if self.mA > 0 and self._chol is not None:
            # the reference's augmented factor; only needed by the library's proj_form = 0 (reference form) —
            # the default reduced form factors A_free A_free' on the device and ignores it
            L = np.asfortranarray(self._chol)
            if L.shape != (mpp, mpp):
                raise ValueError("chol factor is %r, expected (%d, %d) = mA + count(fixvars)" % (L.shape, mpp, mpp))
        check(_lib.lib().bh_proj_set_active(self._h, ptr(chunks), self.n, ptr(L), mpp, ldL), "bh_proj_set_active")
        self._dirty = False

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().bh_proj_destroy(self._h)
            self._h = _null_handle()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def left_mul(lincons, x):
    """``left_mul`` — src/polyhedral_constraints.jl:86-98."""
    x = as_f64(x, lincons.n)
    out = np.empty(lincons.mA + lincons.nb_fix())
    check(_lib.lib().bh_left_mul(lincons.handle, ptr(x), ptr(out)), "bh_left_mul")
    return out


def left_mul_tr(lincons, y):
    """``left_mul_tr`` — src/polyhedral_constraints.jl:72-84."""
    y = as_f64(y, lincons.mA + lincons.nb_fix())
    out = np.empty(lincons.n)
    check(_lib.lib().bh_left_mul_tr(lincons.handle, ptr(y), ptr(out)), "bh_left_mul_tr")
    return out


def projection_(lincons, r, v):
    """``projection!(lincons, r, v)`` — src/polyhedral_constraints.jl:158-170 (writes into ``v``)."""
    r = as_f64(r, lincons.n)
    if not (isinstance(v, np.ndarray) and v.dtype == np.float64 and v.flags.c_contiguous and v.shape == (lincons.n,)):
        raise ValueError("v must be a contiguous float64 vector of length n")
    check(_lib.lib().bh_project(lincons.handle, ptr(r), ptr(v)), "bh_project")


def projection(lincons, r):
    """``projection(lincons, r)`` — src/polyhedral_constraints.jl:150-155."""
    v = np.empty(lincons.n)
    projection_(lincons, r, v)
    return v


def factor_to_boundary(p, w, w_l, w_u, atol=1e-10):
    """``factor_to_boundary`` — src/basic_tralcnlss.jl:793-809."""
    p = as_f64(p)
    n = p.shape[0]
    w, w_l, w_u = as_f64(w, n), as_f64(w_l, n), as_f64(w_u, n)
    out = ct.c_double(0.0)
    check(_lib.lib().bh_factor_to_boundary(ptr(p), ptr(w), ptr(w_l), ptr(w_u), n, float(atol), ct.byref(out)),
          "bh_factor_to_boundary")
    return out.value


TIE_KINDS = {1: "pHp<=tol_negcurve (:725)", 2: "|pHp|>tol (:727)", 4: "alpha>gamma (:735)", 8: "|rtv|<tol_cg (:747)"}


def tie_info(H):
    """Tie log of the last ``projected_cg`` on ``H`` (``bh_pcg_tie_info``): which branch tests came within
    1e-10 (relative) of their threshold, and the closest any test came in that call."""
    flags, first, kind, at = ct.c_int32(0), ct.c_int32(0), ct.c_int32(0), ct.c_int32(0)
    margin = ct.c_double(0.0)
    check(_lib.lib().bh_pcg_tie_info(H.handle, ct.byref(flags), ct.byref(first), ct.byref(margin), ct.byref(kind), ct.byref(at)),
          "bh_pcg_tie_info")
    return {"tie_flags": flags.value, "first_tie_hmul": first.value, "min_margin": margin.value,
            "min_margin_kind": kind.value, "min_margin_hmul": at.value}


def projected_cg(g_minor, H, w_l, w_u, lincons, kappa2, atol=SQRT_EPS, atol_f2b=1e-10, trace_cap=0, full_output=False):
    """``projected_cg(g_minor, H, w_l, w_u, lincons, kappa2; atol)`` — src/basic_tralcnlss.jl:690-764.

    Returns ``(w, status)`` like the reference; with ``full_output`` also a dict with ``iters`` (the
    reference's ``iter`` at exit), ``n_hmul`` and ``trace`` (rows ``pHp, alpha, gamma, rtv``).
    """
    n = H.n
    g = as_f64(g_minor, n)
    wl, wu = as_f64(w_l, n), as_f64(w_u, n)
    w = np.empty(n)
    status, iters, n_hmul = ct.c_int32(-1), ct.c_int32(0), ct.c_int32(0)
    trace = np.full((trace_cap, 4), np.nan) if trace_cap > 0 else None
    check(_lib.lib().bh_pcg(H.handle, lincons.handle, ptr(g), ptr(wl), ptr(wu), float(kappa2), float(atol), float(atol_f2b),
                            ptr(w), ct.byref(status), ct.byref(iters), ptr(trace), trace_cap, ct.byref(n_hmul)), "bh_pcg")
    st = CGStatus(status.value)
    if full_output:
        info = {"iters": iters.value, "n_hmul": n_hmul.value,
                "trace": None if trace is None else trace[:min(trace_cap, n_hmul.value)], "ties": tie_info(H)}
        return w, st, info
    return w, st


def linesearch(g_model, H, w, w_l, w_u, lincons):
    """``linesearch(g_model, H, w, w_l, w_u, fix_bounds)`` — src/basic_tralcnlss.jl:766-791 (``fix_bounds`` = lincons.fixvars)."""
    n = H.n
    out = ct.c_double(0.0)
    g, w, wl, wu = as_f64(g_model, n), as_f64(w, n), as_f64(w_l, n), as_f64(w_u, n)
    check(_lib.lib().bh_linesearch(H.handle, lincons.handle, ptr(g), ptr(w), ptr(wl), ptr(wu), ct.byref(out)), "bh_linesearch")
    return out.value


def minor_iterate(x, s, g_model, H, lincons, delta, kappa2, atol=SQRT_EPS, atol_f2b=1e-10, full_output=False):
    """``minor_iterate(x, s, g_model, H, lincons, delta, kappa2)`` — src/basic_tralcnlss.jl:649-675, one device-resident
    call (step bounds, projected_cg, linesearch, scaling).  Returns ``(w, cg_status)`` like the reference."""
    n = H.n
    x, s, g = as_f64(x, n), as_f64(s, n), as_f64(g_model, n)
    w = np.empty(n)
    status, iters, n_hmul, alpha = ct.c_int32(-1), ct.c_int32(0), ct.c_int32(0), ct.c_double(0.0)
    check(_lib.lib().bh_minor_iterate(H.handle, lincons.handle, ptr(x), ptr(s), ptr(g), ptr(lincons.xlow), ptr(lincons.xupp),
                                      float(delta), float(kappa2), float(atol), float(atol_f2b), ptr(w), ct.byref(status),
                                      ct.byref(iters), ct.byref(n_hmul), ct.byref(alpha)), "bh_minor_iterate")
    st = CGStatus(status.value)
    if full_output:
        return w, st, {"iters": iters.value, "n_hmul": n_hmul.value, "alpha": alpha.value, "ties": tie_info(H)}
    return w, st


def cauchy_step(x, g, H, lincons, delta, full_output=False):
    """``cauchy_step(x, g, H, chol_aat, lincons, delta)`` — src/basic_tralcnlss.jl:574-639, device-resident.  Returns the
    Cauchy step ``s_c``; like the reference it leaves ``lincons.fixvars`` at the active set found along the projected
    gradient path (``lincons.chol`` is NOT rebuilt: the device keeps its own reduced factor)."""
    n = H.n
    x, g = as_f64(x, n), as_f64(g, n)
    s = np.empty(n)
    nwords = (n + 63) // 64
    chunks = np.zeros(nwords, dtype=np.uint64)
    nbp, nh = ct.c_int32(0), ct.c_int32(0)
    lincons._sync()
    check(_lib.lib().bh_cauchy_step(H.handle, lincons._h, ptr(x), ptr(g), ptr(lincons.xlow), ptr(lincons.xupp), float(delta), ptr(s),
                                    ptr(chunks), ct.byref(nbp), ct.byref(nh)), "bh_cauchy_step")
    bits = np.unpackbits(chunks.view(np.uint8), bitorder="little")[:n].astype(bool)
    lincons._fixvars = bits
    lincons._chol = None
    lincons._dirty = False          # the device already holds this active set
    if full_output:
        return s, {"n_breakpoints": nbp.value, "n_hmul": nh.value}
    return s


def gradient(H, rx, y_bar=None):
    """``g = Jx'*rx + Cx'*y_bar`` — src/basic_tralcnlss.jl:45,:74."""
    r = as_f64(rx, H.d)
    yb = np.zeros(H.q) if y_bar is None else as_f64(y_bar, H.q)
    out = np.empty(H.n)
    check(_lib.lib().bh_grad(H.handle, ptr(r), ptr(yb), ptr(out)), "bh_grad")
    return out


def resid_sqnorm(rx):
    """``dot(rx, rx)`` of ``mx = 0.5*dot(rx,rx) + ...`` — src/basic_tralcnlss.jl:44,:58; ``rx`` = this rank's rows, the
    result is the global value (all-reduced), identical on every rank."""
    r = as_f64(rx)
    out = ct.c_double(0.0)
    check(_lib.lib().bh_resid_sqnorm(ptr(r), r.shape[0], ct.byref(out)), "bh_resid_sqnorm")
    return out.value


def set_option(key, value):
    check(_lib.lib().bh_set_option(key.encode(), int(value)), "bh_set_option(%s)" % key)


def hmul_add(H, s, g):
    """``H*s + g`` — src/basic_tralcnlss.jl:412,:437."""
    s, g = as_f64(s, H.n), as_f64(g, H.n)
    out = np.empty(H.n)
    check(_lib.lib().bh_hmul_add(H.handle, ptr(s), ptr(g), ptr(out)), "bh_hmul_add")
    return out


def transfer_counters():
    """Library-wide host<->device traffic since ``bh_init`` (``bh_stats``): (h2d_bytes, d2h_bytes, h2d_calls, d2h_calls)."""
    st = _lib.bh_stats_t()
    h = _null_handle()
    # any live handle reports the library-wide counters; a throw-away 1x1 image serves when the caller has none at hand
    H = AlHessian(np.zeros((1, 1)), None, 0.0)
    try:
        check(_lib.lib().bh_stats(H.handle, ct.byref(st)), "bh_stats")
    finally:
        H.close()
    return st.h2d_bytes, st.d2h_bytes, st.h2d_calls, st.d2h_calls


def _xfer(H):
    st = _lib.bh_stats_t()
    check(_lib.lib().bh_stats(H.handle, ct.byref(st)), "bh_stats")
    return st.h2d_bytes + st.d2h_bytes


def inner_step(x, g, H, lincons, delta, nb_minor_step, kappa2, kappa3, atol=SQRT_EPS, atol_f2b=1e-10, full_output=False):
    """``inner_step(x, g, H, chol_aat, lincons, delta, nb_minor_step, kappa2, kappa3)`` — src/basic_tralcnlss.jl:394-460 — with
    every n-vector of the minor loop resident in HBM: x, g and the bounds go up once, the Cauchy search,
    the minor iterates, ``s .+= w``, ``H*s+g``, the active-set growth (device-side ``active_bounds`` + Gram downdate instead
    of the reference's factor rebuild) and the reduced-gradient norms all work on device vectors, and ``s`` comes down once.
    Returns ``(s, model_reduction)`` and leaves ``lincons.fixvars`` as the reference's method does.  This is the executable
    mirror of the ``BEnlsip.inner_step`` method in julia/BEnlsipHIP.jl."""
    lib = _lib.lib()
    n, mA = H.n, lincons.mA
    x, g = as_f64(x, n), as_f64(g, n)
    dv = getattr(lincons, "_resident", None)
    if dv is None:
        dv = {k: DeviceVector(n) for k in ("x", "g", "s", "w", "gm")}
        dv["xlow"], dv["xupp"] = DeviceVector(n, lincons.xlow), DeviceVector(n, lincons.xupp)
        lincons._resident = dv
    dv["x"].upload(x)
    dv["g"].upload(g)
    lincons._sync()
    P = lincons._h
    nwords = (n + 63) // 64
    chunks = np.zeros(nwords, dtype=np.uint64)
    nbp, nh = ct.c_int32(0), ct.c_int32(0)
    t0 = _xfer(H)
    check(lib.bh_cauchy_step_dev(H.handle, P, dv["x"].ptr, dv["g"].ptr, dv["xlow"].ptr, dv["xupp"].ptr, float(delta), dv["s"].ptr,
                                 ptr(chunks), ct.byref(nbp), ct.byref(nh)), "bh_cauchy_step_dev")            # :410
    check(lib.bh_hmul_add_dev(H.handle, dv["s"].ptr, dv["g"].ptr, dv["gm"].ptr), "bh_hmul_add_dev")          # :412
    nfix = int(np.unpackbits(chunks.view(np.uint8), bitorder="little")[:n].sum())

    def red_norm(vec):
        out = ct.c_double(0.0)
        check(lib.bh_reduced_gradient_norm_dev(P, vec.ptr, ct.byref(out)), "bh_reduced_gradient_norm_dev")
        return out.value
    nrg, nrgm = red_norm(dv["g"]), red_norm(dv["gm"])                                                        # :420-421
    approx_solved = nrgm <= kappa3 * nrg
    max_minor_step = min(nb_minor_step, n - mA - nfix)                                                       # :425-426
    j, cg_stop, statuses = 1, False, []
    # from here on dv["gm"] holds H*s + g for the current s by construction: bh_step_accumulate_dev may add the H*w the CG loop of
    # the preceding bh_minor_iterate_dev accumulated instead of sweeping J again, and bh_model_reduction_dev may take s'Hs from
    # g_minor (option step_from_cg, switched off again below)
    check(lib.bh_set_option(b"step_from_cg", 1), "bh_set_option")
    try:
        while j <= max_minor_step and not approx_solved and not cg_stop:                                         # :430
            status, iters, n_hmul, alpha = ct.c_int32(-1), ct.c_int32(0), ct.c_int32(0), ct.c_double(0.0)
            check(lib.bh_minor_iterate_dev(H.handle, P, dv["x"].ptr, dv["s"].ptr, dv["gm"].ptr, dv["xlow"].ptr, dv["xupp"].ptr, float(delta),
                                           float(kappa2), float(atol), float(atol_f2b), dv["w"].ptr, ct.byref(status), ct.byref(iters),
                                           ct.byref(n_hmul), ct.byref(alpha)), "bh_minor_iterate_dev")           # :434
            cg_stop = status.value == int(CGStatus.negative_curvature)
            check(lib.bh_step_accumulate_dev(H.handle, dv["s"].ptr, dv["w"].ptr, dv["g"].ptr, dv["gm"].ptr), "bh_step_accumulate_dev")   # :436-437
            n_at, n_fixed, branch = ct.c_int32(0), ct.c_int32(0), ct.c_int32(0)
            check(lib.bh_proj_update_active_dev(P, dv["x"].ptr, dv["s"].ptr, dv["xlow"].ptr, dv["xupp"].ptr, float(delta), float(atol),
                                                ct.byref(n_at), ct.byref(n_fixed), ct.byref(branch), ptr(chunks)), "bh_proj_update_active_dev")   # :439-453
            if branch.value == 0:
                nrg, nrgm = red_norm(dv["g"]), red_norm(dv["gm"])                                                # :446-447
                approx_solved = nrgm <= kappa3 * nrg
            else:
                approx_solved = True
            statuses.append((CGStatus(status.value), iters.value, n_fixed.value, nrgm / (kappa3 * nrg) if nrg > 0 else math.inf))
            j += 1
        # :458 — still under the invariant: s'Hs = s.(g_minor - g) from the resident g_minor instead of a J v sweep
        mr = ct.c_double(0.0)
        check(lib.bh_model_reduction_dev(H.handle, dv["g"].ptr, dv["s"].ptr, ct.byref(mr)), "bh_model_reduction_dev")
    finally:
        lib.bh_set_option(b"step_from_cg", 0)
    loop_bytes = _xfer(H) - t0
    s = dv["s"].download()
    lincons._fixvars = np.unpackbits(chunks.view(np.uint8), bitorder="little")[:n].astype(bool)
    lincons._chol = None
    lincons._dirty = False              # the device already holds this active set
    if full_output:
        return s, mr.value, {"minor": statuses, "n_breakpoints": nbp.value, "pcie_bytes_in_loop": int(loop_bytes)}
    return s, mr.value


class DeviceVector:
    """A float64 vector resident in HBM (plumbing for the ``*_dev`` entry points)."""

    def __init__(self, n, host=None):
        self.n = int(n)
        self._p = _null_handle()
        check(_lib.lib().bh_dev_alloc(_byref(self._p), 8 * max(self.n, 1)), "bh_dev_alloc")
        if host is not None:
            self.upload(host)

    @property
    def ptr(self):
        return self._p

    def upload(self, host):
        a = as_f64(host, self.n)
        check(_lib.lib().bh_dev_upload(self._p, ptr(a), 8 * self.n), "bh_dev_upload")

    def download(self):
        out = np.empty(self.n)
        check(_lib.lib().bh_dev_download(ptr(out), self._p, 8 * self.n), "bh_dev_download")
        return out

    def close(self):
        if self._p.value:
            _lib.lib().bh_dev_free(self._p)
            self._p = _null_handle()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_PCG_OUT = (ct.c_int32(-1), ct.c_int32(0), ct.c_int32(0))       # reused output cells: one ctypes allocation less per call
_PCG_REF = tuple(ct.byref(x) for x in _PCG_OUT)


def projected_cg_dev(g_dev, H, wl_dev, wu_dev, lincons, kappa2, w_out_dev, atol=SQRT_EPS, atol_f2b=1e-10):
    """``bh_pcg_dev``: all vectors already in HBM (what bench.py times).  Returns (status, iters, n_hmul)."""
    rc = _lib.lib().bh_pcg_dev(H._h, lincons.handle, g_dev._p, wl_dev._p, wu_dev._p, kappa2, atol, atol_f2b, w_out_dev._p,
                               _PCG_REF[0], _PCG_REF[1], None, 0, _PCG_REF[2])
    if rc != 0:
        check(rc, "bh_pcg_dev")
    return CGStatus(_PCG_OUT[0].value), _PCG_OUT[1].value, _PCG_OUT[2].value
