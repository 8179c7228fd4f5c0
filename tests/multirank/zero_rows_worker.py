"""One rank of a run in which some ranks own NO rows of J (d_total < number of ranks): the exchange still has to happen, the
launch schedule still has to agree."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world, workdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import benlsip_jl_amd as bh
    bh.init(0)
    idfile = os.path.join(workdir, "unique_id_zero.bin")

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
    rng = np.random.default_rng(5)
    d_total, n = 2, 24
    J = rng.standard_normal((d_total, n))
    C = rng.standard_normal((1, n))
    g = rng.standard_normal(n)
    lo, hi = bh.row_shard(d_total, rank, world)
    H = bh.AlHessian(J[lo:hi], C, 2.0)
    hv = H * g
    vt = bh.vthv(H, g)
    cons = bh.MixedConstraints(np.zeros((0, n)), None, None)
    w, st, info = bh.projected_cg(g, H, -np.ones(n), np.ones(n), cons, 1e-3, full_output=True)
    np.savez(os.path.join(workdir, "zero_rank%d.npz" % rank), hv=hv, vt=vt, w=w, st=int(st), it=info["iters"], lo=lo, hi=hi)
    H.close()
    bh._lib.check(bh._lib.lib().bh_comm_destroy(), "bh_comm_destroy")
    print("rank %d rows [%d, %d) done" % (rank, lo, hi), flush=True)


if __name__ == "__main__":
    main()
