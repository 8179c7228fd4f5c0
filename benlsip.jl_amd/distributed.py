"""Row sharding of J over ranks (one process per GPU) and RCCL communicator bring-up.

SURVEY.md §8(e): GPU k holds rows [k*d/G, (k+1)*d/G) of J; every n-vector, the mask, A, L and C are
replicated; the only exchange is ONE all-reduce of n doubles per J'·t, issued by the library on its own
stream through RCCL.  ``torch.distributed`` is used only to hand the RCCL unique id to the other ranks.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


def row_shard(d_total, rank, nranks):
    """Half-open row range of ``rank``: contiguous blocks, the first ``d_total % nranks`` ranks get one extra row."""
    if not (0 <= rank < nranks):
        raise ValueError("rank %d outside 0..%d" % (rank, nranks - 1))
    base, extra = divmod(int(d_total), int(nranks))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def init_distributed(rank, nranks, broadcast_bytes):
    """Create the library's RCCL communicator.

    ``broadcast_bytes(buf: bytes | None) -> bytes`` must return rank 0's buffer on every rank (e.g. built on
    ``torch.distributed.broadcast`` or an MPI bcast).
    """
    import os
    lib = _lib.lib()
    if nranks == 1 and os.environ.get("BH_FORCE_COMM", "0") in ("", "0"):
        check(lib.bh_comm_init(0, 1, None), "bh_comm_init")
        return
    buf, rc0 = None, 0
    if rank == 0:
        raw = (C.c_ubyte * _lib.BH_UNIQUE_ID_BYTES)()
        rc0 = lib.bh_comm_unique_id(raw)
        # a failure here must not leave the other ranks waiting in the broadcast: an all-zero id tells them (no transport
        # produces one: RCCL ids carry a socket address, peer-buffer ids 64 random bytes)
        buf = bytes(raw) if rc0 == 0 else bytes(_lib.BH_UNIQUE_ID_BYTES)
    buf = broadcast_bytes(buf)
    if len(buf) != _lib.BH_UNIQUE_ID_BYTES:
        raise ValueError("unique id must be %d bytes" % _lib.BH_UNIQUE_ID_BYTES)
    if rank == 0 and rc0 != 0:
        check(rc0, "bh_comm_unique_id")
    if not any(buf):
        raise _lib.BenlsipHipError(_lib.BH_ERR_RCCL, "bh_comm_unique_id", "rank 0 could not create the communicator id")
    raw = (C.c_ubyte * _lib.BH_UNIQUE_ID_BYTES).from_buffer_copy(buf)
    check(lib.bh_comm_init(rank, nranks, raw), "bh_comm_init")


def torch_broadcast_bytes(device=None):
    """A ``broadcast_bytes`` callable on top of an initialised ``torch.distributed`` process group."""
    import torch
    import torch.distributed as dist

    def bcast(buf):
        t = torch.zeros(_lib.BH_UNIQUE_ID_BYTES, dtype=torch.uint8, device=device)
        if dist.get_rank() == 0:
            t.copy_(torch.from_numpy(np.frombuffer(buf, dtype=np.uint8).copy()))
        dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())

    return bcast
