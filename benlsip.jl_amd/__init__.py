"""benlsip.jl_amd — MI355X (gfx950) backend for BEnlsip.jl's trust-region subproblem hot path.

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C ABI of ``include/benlsip_hip.h``),
the ctypes binding and the host-side mirror of the reference's operator interface.
"""
from . import _lib, build  # noqa: F401
from ._lib import BenlsipHipError, init, library_path, load  # noqa: F401
from . import synthetic  # noqa: F401
from .distributed import init_distributed, row_shard, torch_broadcast_bytes  # noqa: F401
from .operators import (AlHessian, CGStatus, DeviceVector, MixedConstraints, cauchy_step, factor_to_boundary, gradient, hmul, hmul_add,  # noqa: F401
                        inner_step, transfer_counters,
                        left_mul, left_mul_tr, linesearch, minor_iterate, pack_bitvector, projected_cg, projected_cg_dev,
                        projection, projection_, resid_sqnorm, set_option, tie_info, vthv)
