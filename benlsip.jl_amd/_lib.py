"""ctypes binding of the C ABI declared in include/benlsip_hip.h.

The product path has NO CPU fallback: if the HIP library is missing or no GPU is
visible, calls raise (BenlsipHipError / OSError).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

BH_OK = 0
BH_ERR_INVALID_ARG, BH_ERR_NOT_INIT, BH_ERR_HIP, BH_ERR_RCCL = -1, -2, -3, -4
BH_ERR_PRECONDITION, BH_ERR_SHAPE, BH_ERR_NO_DEVICE, BH_ERR_UNSUPPORTED = -5, -6, -7, -8
BH_FLAG_PROFILE = 1
BH_UNIQUE_ID_BYTES = 128

EXPORTS = [
    "bh_init", "bh_shutdown", "bh_set_stream", "bh_synchronize", "bh_strerror", "bh_last_error_detail", "bh_device_info",
    "bh_comm_unique_id", "bh_comm_init", "bh_comm_destroy", "bh_comm_info",
    "bh_hess_create", "bh_hess_create_async", "bh_hess_wait", "bh_hess_create_dev", "bh_hess_create_synthetic", "bh_hess_set_mu", "bh_hess_destroy", "bh_hess_shape",
    "bh_hmul", "bh_vthv", "bh_jv", "bh_jtv", "bh_hmul_dev", "bh_jv_dev", "bh_jtv_dev",
    "bh_proj_create", "bh_proj_set_active", "bh_proj_destroy", "bh_proj_shape", "bh_project", "bh_project_dev",
    "bh_left_mul", "bh_left_mul_tr",
    "bh_pcg", "bh_pcg_dev", "bh_pcg_tie_info", "bh_resid_sqnorm", "bh_factor_to_boundary", "bh_minor_iterate", "bh_linesearch", "bh_grad", "bh_hmul_add", "bh_cauchy_step",
    "bh_cauchy_step_dev", "bh_minor_iterate_dev", "bh_linesearch_dev", "bh_grad_dev", "bh_hmul_add_dev", "bh_step_accumulate_dev",
    "bh_proj_update_active_dev", "bh_reduced_gradient_norm_dev", "bh_model_reduction_dev",
    "bh_dev_alloc", "bh_dev_free", "bh_dev_upload", "bh_dev_download", "bh_stats", "bh_stats_reset",
    "bh_set_option", "bh_time_kernel", "bh_selftest",
]


class BenlsipHipError(RuntimeError):
    def __init__(self, code, what, detail):
        super().__init__("%s (code %d): %s" % (what, code, detail))
        self.code = code


class bh_stats_t(C.Structure):
    _fields_ = [("n_hmul", C.c_int64), ("n_jv", C.c_int64), ("n_jtv", C.c_int64), ("n_proj", C.c_int64),
                ("n_pcg", C.c_int64), ("n_cg_iter", C.c_int64), ("n_allreduce", C.c_int64), ("hmul_ms", C.c_double),
                ("hmul_timed", C.c_int64), ("bytes_per_hmul", C.c_double),
                ("h2d_bytes", C.c_int64), ("d2h_bytes", C.c_int64), ("h2d_calls", C.c_int64), ("d2h_calls", C.c_int64),
                ("cg_kernels", C.c_int64)]


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_i32, _i64, _f64 = C.c_int32, C.c_int64, C.c_double

_PROTOS = {
    "bh_init": ([_i32, _i32], _i32),
    "bh_shutdown": ([], _i32),
    "bh_set_stream": ([_vp], _i32),
    "bh_synchronize": ([], _i32),
    "bh_strerror": ([_i32], C.c_char_p),
    "bh_last_error_detail": ([], C.c_char_p),
    "bh_device_info": ([C.c_char_p, _i64, C.POINTER(_i32), C.c_char_p, _i64], _i32),
    "bh_comm_unique_id": ([_vp], _i32),
    "bh_comm_init": ([_i32, _i32, _vp], _i32),
    "bh_comm_destroy": ([], _i32),
    "bh_comm_info": ([C.POINTER(_i32), C.POINTER(_i32)], _i32),
    "bh_hess_create": ([C.POINTER(_vp), _vp, _i64, _i64, _i64, _vp, _i64, _i64, _f64], _i32),
    "bh_hess_create_async": ([C.POINTER(_vp), _vp, _i64, _i64, _i64, _vp, _i64, _i64, _f64], _i32),
    "bh_hess_wait": ([_vp], _i32),
    "bh_hess_create_dev": ([C.POINTER(_vp), _vp, _i64, _i64, _i64, _vp, _i64, _i64, _f64], _i32),
    "bh_hess_create_synthetic": ([C.POINTER(_vp), _i64, _i64, _i64, _i64, C.c_uint64, _vp, _f64], _i32),
    "bh_hess_set_mu": ([_vp, _f64], _i32),
    "bh_hess_destroy": ([_vp], _i32),
    "bh_hess_shape": ([_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)], _i32),
    "bh_hmul": ([_vp, _vp, _vp], _i32),
    "bh_vthv": ([_vp, _vp, _dp], _i32),
    "bh_jv": ([_vp, _vp, _vp], _i32),
    "bh_jtv": ([_vp, _vp, _vp], _i32),
    "bh_hmul_dev": ([_vp, _vp, _vp], _i32),
    "bh_jv_dev": ([_vp, _vp, _vp], _i32),
    "bh_jtv_dev": ([_vp, _vp, _vp], _i32),
    "bh_proj_create": ([C.POINTER(_vp), _vp, _i64, _i64, _i64], _i32),
    "bh_proj_set_active": ([_vp, _vp, _i64, _vp, _i64, _i64], _i32),
    "bh_proj_destroy": ([_vp], _i32),
    "bh_proj_shape": ([_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)], _i32),
    "bh_project": ([_vp, _vp, _vp], _i32),
    "bh_project_dev": ([_vp, _vp, _vp], _i32),
    "bh_left_mul": ([_vp, _vp, _vp], _i32),
    "bh_left_mul_tr": ([_vp, _vp, _vp], _i32),
    "bh_pcg": ([_vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _vp, C.POINTER(_i32), C.POINTER(_i32), _vp, _i64,
                C.POINTER(_i32)], _i32),
    "bh_pcg_dev": ([_vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _vp, C.POINTER(_i32), C.POINTER(_i32), _vp, _i64,
                    C.POINTER(_i32)], _i32),
    "bh_pcg_tie_info": ([_vp, C.POINTER(_i32), C.POINTER(_i32), _dp, C.POINTER(_i32), C.POINTER(_i32)], _i32),
    "bh_resid_sqnorm": ([_vp, _i64, _dp], _i32),
    "bh_factor_to_boundary": ([_vp, _vp, _vp, _vp, _i64, _f64, _dp], _i32),
    "bh_minor_iterate": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _f64, _vp, C.POINTER(_i32), C.POINTER(_i32),
                         C.POINTER(_i32), _dp], _i32),
    "bh_linesearch": ([_vp, _vp, _vp, _vp, _vp, _vp, _dp], _i32),
    "bh_cauchy_step": ([_vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _vp, C.POINTER(_i32), C.POINTER(_i32)], _i32),
    "bh_grad": ([_vp, _vp, _vp, _vp], _i32),
    "bh_cauchy_step_dev": ([_vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _vp, C.POINTER(_i32), C.POINTER(_i32)], _i32),
    "bh_minor_iterate_dev": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _f64, _vp, C.POINTER(_i32), C.POINTER(_i32),
                             C.POINTER(_i32), _dp], _i32),
    "bh_linesearch_dev": ([_vp, _vp, _vp, _vp, _vp, _vp, _dp], _i32),
    "bh_grad_dev": ([_vp, _vp, _vp, _vp], _i32),
    "bh_hmul_add_dev": ([_vp, _vp, _vp, _vp], _i32),
    "bh_step_accumulate_dev": ([_vp, _vp, _vp, _vp, _vp], _i32),
    "bh_proj_update_active_dev": ([_vp, _vp, _vp, _vp, _vp, _f64, _f64, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), _vp], _i32),
    "bh_reduced_gradient_norm_dev": ([_vp, _vp, _dp], _i32),
    "bh_model_reduction_dev": ([_vp, _vp, _vp, _dp], _i32),
    "bh_hmul_add": ([_vp, _vp, _vp, _vp], _i32),
    "bh_dev_alloc": ([C.POINTER(_vp), _i64], _i32),
    "bh_dev_free": ([_vp], _i32),
    "bh_dev_upload": ([_vp, _vp, _i64], _i32),
    "bh_dev_download": ([_vp, _vp, _i64], _i32),
    "bh_stats": ([_vp, C.POINTER(bh_stats_t)], _i32),
    "bh_stats_reset": ([_vp], _i32),
    "bh_set_option": ([C.c_char_p, _i64], _i32),
    "bh_time_kernel": ([_vp, _i32, _i32, _dp], _i32),
    "bh_selftest": ([], _i32),
}

_lib = None
_initialised = False


def library_path():
    return _build.lib_path()


def load(build_if_missing=True):
    """dlopen the in-tree shared library (building it first if it is absent and hipcc exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise OSError("HIP extension %s is missing: run `python benlsip.jl_amd/build.py`" % path)
        _build.build()
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (argtypes, restype) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def check(rc, what):
    if rc != BH_OK:
        lib = load()
        raise BenlsipHipError(rc, "%s: %s" % (what, lib.bh_strerror(rc).decode()), lib.bh_last_error_detail().decode())


def init(device=None, flags=0):
    """bh_init on `device` (default: LOCAL_RANK or 0).  Raises when no GPU is visible — there is no CPU path."""
    global _initialised
    lib = load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    check(lib.bh_init(int(device), int(flags)), "bh_init")
    _initialised = True
    return lib


def lib():
    if not _initialised:
        init()
    return _lib


def as_f64(x, n=None):
    a = np.ascontiguousarray(x, dtype=np.float64)
    if n is not None and a.shape != (n,):
        raise ValueError("expected a vector of length %d, got shape %r" % (n, a.shape))
    return a


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None
