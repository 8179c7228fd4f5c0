"""One rank of the multi-rank fuzz: `count` seeded instances (fuzz_cases.py), J row-sharded, every caller-level entry point of the
path — projected_cg, minor_iterate, cauchy_step — results written per case.  Never imports the oracle."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    rank, world, workdir, seed, count = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    import benlsip_jl_amd as bh
    from fuzz_cases import cases
    bh.init(0)
    idfile = os.path.join(workdir, "unique_id_fuzz.bin")

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
    for c in cases(seed, count):
        k, n = c["k"], c["n"]
        lo, hi = bh.row_shard(c["d"], rank, world)
        H = bh.AlHessian(c["J"][lo:hi], c["C"], c["mu"])
        err = {}
        if c["feasible_rows"]:
            cons = bh.MixedConstraints(c["A"], None, c["fix"], l=c["xl"], u=c["xu"])
            w, st, info = bh.projected_cg(c["g"], H, c["wl"], c["wu"], cons, c["kappa2"], full_output=True)
            out["pcg_w_%d" % k], out["pcg_st_%d" % k] = w, np.array([int(st), info["iters"], info["n_hmul"], H.stats()["cg_kernels"]])
            wm, stm, infom = bh.minor_iterate(c["x"], np.zeros(n), c["g"], H, cons, c["delta"], c["kappa2"], full_output=True)
            out["mi_w_%d" % k], out["mi_st_%d" % k] = wm, np.array([int(stm)])
            cons.close()
        cau = bh.MixedConstraints(c["A"], None, None, l=c["xl"], u=c["xu"])
        try:
            s, info = bh.cauchy_step(c["x"], c["g_cauchy"], H, cau, c["delta"], full_output=True)
            out["cs_s_%d" % k], out["cs_fix_%d" % k], out["cs_nh_%d" % k] = s, np.asarray(cau.fixvars, dtype=bool), np.array([info["n_hmul"]])
        except bh.BenlsipHipError as e:
            out["cs_err_%d" % k] = np.array([e.code])
        cau.close()
        H.close()
    np.savez(os.path.join(workdir, "fuzz_rank%d.npz" % rank), **out)
    bh._lib.check(bh._lib.lib().bh_comm_destroy(), "bh_comm_destroy")
    print("rank %d: %d cases done" % (rank, count), flush=True)


if __name__ == "__main__":
    main()
