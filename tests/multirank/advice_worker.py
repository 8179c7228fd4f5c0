"""One rank of two regression checks from the round-2 advisor (never imports the oracle):

 (a) the CG iteration shape must not depend on a rank's POINTERS: n = 40 (even, below ld = 48), bh_pcg_dev; rank 0 hands over
     16-byte aligned device vectors, the other ranks views that start 8 bytes into their allocations.  Before the fix rank 0
     ran the two-kernel iteration (in place) and the others the three-kernel one (staged): different exchange payloads on the same
     sequence counter.
 (b) bh_hess_create_async on a rank that owns no rows used to run the d_total all-reduce inside the create call, its peers at
     bh_hess_wait: a collective issued in between (bh_resid_sqnorm) was matched against it.  d_total = 2 over 3 ranks.
"""
import ctypes as ct
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


class View:
    """A device vector of n doubles that starts `offset_bytes` into its allocation (what a caller's sub-array looks like)."""

    def __init__(self, bh, n, host, offset_bytes):
        self.n, self.off = n, offset_bytes
        self.base = bh.DeviceVector(n + 2)
        self._p = ct.c_void_p(self.base.ptr.value + offset_bytes)
        if host is not None:
            bh._lib.check(bh._lib.lib().bh_dev_upload(self._p, host.ctypes.data_as(ct.c_void_p), 8 * n), "upload")

    def download(self, bh):
        out = np.empty(self.n)
        bh._lib.check(bh._lib.lib().bh_dev_download(out.ctypes.data_as(ct.c_void_p), self._p, 8 * self.n), "download")
        return out


def main():
    rank, world, workdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import benlsip_jl_amd as bh
    bh.init(0)
    idfile = os.path.join(workdir, "unique_id_advice.bin")

    def bcast(buf):
        if rank == 0:
            with open(idfile + ".tmp", "wb") as f:
                f.write(buf)
            os.rename(idfile + ".tmp", idfile)
            return buf
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120:
                raise RuntimeError("no unique id from rank 0")
            time.sleep(0.01)
        return open(idfile, "rb").read()

    bh.init_distributed(rank, world, bcast)
    out = {}

    # ---- (a) ------------------------------------------------------------------------------------------------------------------
    rng = np.random.default_rng(21)
    d, n = 500, 40
    J = rng.standard_normal((d, n)) / np.sqrt(d) * np.logspace(0, -1, n)
    g = rng.standard_normal(n)
    fix = np.zeros(n, dtype=bool)
    fix[[3, 17, 30]] = True
    wl = np.where(fix, 0.0, -1e3)
    wu = np.where(fix, 0.0, 1e3)
    lo, hi = bh.row_shard(d, rank, world)
    H = bh.AlHessian(J[lo:hi], None, 1.0)
    cons = bh.MixedConstraints(np.zeros((0, n)), None, fix, l=-np.ones(n), u=np.ones(n))
    off = 0 if rank == 0 else 8
    gv, lv, uv, wv = View(bh, n, g, off), View(bh, n, wl, off), View(bh, n, wu, off), View(bh, n, None, off)
    st, it, nh = bh.projected_cg_dev(gv, H, lv, uv, cons, 1e-6, wv)
    out["a_w"], out["a_st"], out["a_it"], out["a_nh"] = wv.download(bh), int(st), it, nh
    st2, it2, nh2 = bh.projected_cg_dev(gv, H, lv, uv, cons, 1e-6, wv)            # again: first batch sized by the previous call
    out["a_same"] = bool(np.array_equal(out["a_w"], wv.download(bh)) and (int(st2), it2, nh2) == (int(st), it, nh))
    H.close()

    # ---- (b) ------------------------------------------------------------------------------------------------------------------
    rng = np.random.default_rng(5)
    d_total, n = 2, 24
    J = rng.standard_normal((d_total, n))
    C = rng.standard_normal((1, n))
    gvec = rng.standard_normal(n)
    r = rng.standard_normal(d_total)
    lo, hi = bh.row_shard(d_total, rank, world)
    H = bh.AlHessian.create_async(np.asfortranarray(J[lo:hi]), C, 2.0)
    out["b_sq"] = bh.resid_sqnorm(r[lo:hi])          # a collective between create and wait, at the same point on every rank
    H.wait()
    out["b_hv"] = H * gvec
    out["b_rows"] = hi - lo
    H.close()
    np.savez(os.path.join(workdir, "advice_rank%d.npz" % rank), **out)
    bh._lib.check(bh._lib.lib().bh_comm_destroy(), "bh_comm_destroy")
    print("rank %d done" % rank, flush=True)


if __name__ == "__main__":
    main()
