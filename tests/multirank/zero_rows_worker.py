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
    # linear equalities: projected_cg, a minor iterate, and the Cauchy search (row-space form: the rank without rows contributes
    # nothing to the two sums and launches no GEMM for its empty block of B); box-constrained Cauchy search as well
    A = rng.standard_normal((2, n))
    xl, xu = -np.ones(n), np.ones(n)
    x = np.clip(0.3 * rng.standard_normal(n), -0.9, 0.9)
    gen = bh.MixedConstraints(A, None, None, l=xl, u=xu)
    wg, stg, infog = bh.projected_cg(g, H, -0.5 * np.ones(n), 0.5 * np.ones(n), gen, 1e-6, full_output=True)
    gen_kernels = H.stats()["cg_kernels"]
    wm, stm, infom = bh.minor_iterate(x, np.zeros(n), g, H, gen, 0.5 * np.linalg.norm(g), 0.1, full_output=True)
    cau = bh.MixedConstraints(A, None, None, l=xl, u=xu)
    sc, infoc = bh.cauchy_step(x, 50.0 * g, H, cau, 25.0 * np.linalg.norm(g), full_output=True)
    caub = bh.MixedConstraints(np.zeros((0, n)), None, None, l=xl, u=xu)
    sb, infob = bh.cauchy_step(x, 50.0 * g, H, caub, 25.0 * np.linalg.norm(g), full_output=True)
    np.savez(os.path.join(workdir, "zero_rank%d.npz" % rank), hv=hv, vt=vt, w=w, st=int(st), it=info["iters"], lo=lo, hi=hi,
             A=A, x=x, wg=wg, stg=int(stg), itg=infog["iters"], gen_kernels=gen_kernels, wm=wm, stm=int(stm),
             sc=sc, sc_fix=np.asarray(cau.fixvars, dtype=bool), sc_passes=infoc["n_hmul"],
             sb=sb, sb_fix=np.asarray(caub.fixvars, dtype=bool), sb_passes=infob["n_hmul"])
    H.close()
    bh._lib.check(bh._lib.lib().bh_comm_destroy(), "bh_comm_destroy")
    print("rank %d rows [%d, %d) done" % (rank, lo, hi), flush=True)


if __name__ == "__main__":
    main()
