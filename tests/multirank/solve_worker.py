"""One rank of a WHOLE row-sharded solve: the restated outer iteration (tralcnllss -> solve_subproblem -> inner_step, the
stand-in for the unchanged Julia driver) runs replicated on every rank, `residuals` / `jac_res` return this rank's rows only,
and every hot-path call plus the three residual-row seams (mx :44/:58, g :45/:74, least-squares multipliers :893) go through
the library (ShardedHipOps = what julia/BEnlsipHIP.jl's multi-rank methods do).  Writes x, y and the driver's decision log;
the parent compares the ranks bit for bit and against the UNSHARDED oracle solve."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, os.path.join(ROOT, "oracle"), TESTS, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def problem(name):
    if name == "sphere":
        import sphere_problem as sp
        return dict(r=sp.r, jac_r=sp.jac_r, c=sp.c, jac_c=sp.jac_c, A=sp.A, b=sp.b, x_l=sp.x_l, x_u=sp.x_u, x0=sp.x0, d=4,
                    kw=dict(max_outer_iter=100, max_inner_iter=250))
    from nls_problem import NLSProblem
    P = NLSProblem(256, 48, 2, seed=1)
    return dict(r=P.r, jac_r=P.jac_r, c=P.c, jac_c=P.jac_c, A=P.A, b=P.b, x_l=P.x_l, x_u=P.x_u, x0=P.x0, d=256,
                kw=dict(max_outer_iter=30, max_inner_iter=60))


def main():
    rank, world, workdir, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import benlsip_jl_amd as bh
    import benlsip_ref as R                      # the restated DRIVER only; every numerical kernel call goes to the library
    from hip_ops import ShardedHipOps, ShardedResidentOps
    resident = len(sys.argv) > 5 and sys.argv[5] == "resident"
    bh.init(0)
    idfile = os.path.join(workdir, "unique_id_%s.bin" % name)

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
    P = problem(name)
    lo, hi = bh.row_shard(P["d"], rank, world)
    ops = ShardedResidentOps(bh) if resident else ShardedHipOps(bh)
    log = []
    t0 = time.perf_counter()
    x, y = R.tralcnllss(P["x0"], lambda z: P["r"](z)[lo:hi], lambda z: P["jac_r"](z)[lo:hi], P["c"], P["jac_c"], P["A"], P["b"],
                        P["x_l"], P["x_u"], ops=ops, log=log, **P["kw"])
    el = time.perf_counter() - t0
    np.savez(os.path.join(workdir, "solve_%s_rank%d.npz" % (name, rank)), x=x, y=y, lo=lo, hi=hi)
    with open(os.path.join(workdir, "solve_%s_rank%d.json" % (name, rank)), "w") as f:
        json.dump(dict(log=[[e[0]] + [float(v) for v in e[1:]] for e in log], n_pcg=ops.n_pcg, seconds=el,
                       ties=[[k, t] for k, t in ops.ties]), f)
    import gc
    gc.collect()                                 # AlHessian handles of the solve: released before the communicator goes
    bh._lib.check(bh._lib.lib().bh_comm_destroy(), "bh_comm_destroy")
    print("rank %d done: %s solve, rows [%d, %d), %d minor iterates, %.2f s" % (rank, name, lo, hi, ops.n_pcg, el), flush=True)


if __name__ == "__main__":
    main()
