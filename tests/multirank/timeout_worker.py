"""One rank of the "a peer never arrives" test: rank 1 skips one all-reduce; rank 0's exchange must time out (BH_PEER_TIMEOUT_S)
and surface as an error instead of hanging the device.  Never imports the oracle."""
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
    idfile = os.path.join(workdir, "unique_id_to.bin")

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
    rng = np.random.default_rng(3)
    J = rng.standard_normal((200, 64))
    lo, hi = bh.row_shard(200, rank, world)
    H = bh.AlHessian(J[lo:hi], None, 1.0)
    v = rng.standard_normal(64)
    a = H * v                                   # both ranks: a matched exchange
    outcome = "no error"
    t0 = time.time()
    if rank == 0:
        try:
            H * v                               # rank 1 never joins this one
        except bh.BenlsipHipError as e:
            outcome = "error %d after %.1f s: %s" % (e.code, time.time() - t0, str(e)[:80])
    else:
        time.sleep(6.0)                         # longer than rank 0's timeout
    np.savez(os.path.join(workdir, "to_rank%d.npz" % rank), a=a, outcome=outcome, code_rccl=bh._lib.BH_ERR_RCCL)
    H.close()
    bh._lib.lib().bh_comm_destroy()
    print("rank %d: %s" % (rank, outcome), flush=True)


if __name__ == "__main__":
    main()
