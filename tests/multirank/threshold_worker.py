"""One rank of the launch-schedule regression test: shards whose own streaming-time estimates sit on either
side of a launch-ahead threshold.  The batch size must come from rank-independent data, or the ranks enqueue different
numbers of iterations — and of collectives.  Never imports the oracle."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    rank, world, workdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import benlsip_jl_amd as bh
    bh.init(0)
    idfile = os.path.join(workdir, "unique_id_thr.bin")

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
    syn = bh.synthetic
    n, d_total = 4096, 64087                      # 21363 / 21362 / 21362 rows: 100.004 us vs 99.999 us at 7 TB/s
    lo, hi = bh.row_shard(d_total, rank, world)
    H = bh.AlHessian.synthetic(hi - lo, n, row0=lo, d_total=d_total, seed=1, colscale=syn.column_scale(n, 1), mu=10.0)
    x, x_l, x_u, fix = syn.box_vectors(n, fix_every=8)
    cons = bh.MixedConstraints(np.zeros((0, n)), None, fix, l=x_l, u=x_u)
    g = H.jtv(syn.residual_rows(lo, hi))
    w_l, w_u = syn.step_bounds(x, x_l, x_u, fix, syn.initial_tr(g))
    H.reset_stats()
    res = []
    for _ in range(3):                            # first call: default first batch; later ones: sized by the previous call
        w, st, info = bh.projected_cg(g, H, w_l, w_u, cons, 0.1, full_output=True)
        res.append((w, int(st), info["iters"], info["n_hmul"]))
    st = H.stats()
    np.savez(os.path.join(workdir, "thr_rank%d.npz" % rank), w=res[-1][0], status=res[-1][1], iters=res[-1][2], n_hmul=res[-1][3],
             same=all(np.array_equal(r[0], res[0][0]) and r[1:] == res[0][1:] for r in res), n_allreduce=st["n_allreduce"], lo=lo, hi=hi)
    H.close()
    bh._lib.check(bh._lib.lib().bh_comm_destroy(), "bh_comm_destroy")
    print("rank %d done: %d H*p, %d all-reduces" % (rank, res[-1][3], st["n_allreduce"]), flush=True)


if __name__ == "__main__":
    main()
