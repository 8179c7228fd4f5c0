"""One rank of the "peer inboxes map but are not reachable" test: BH_PEER_ECHO_SKIP_RANK keeps one rank away from the
reachability echo of bh_comm_init; EVERY rank's bh_comm_init must then fail (collectively, within the echo's own timeout)
and leave the library usable without a communicator.  Never imports the oracle."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    rank, world, workdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import ctypes as C
    import benlsip_jl_amd as bh
    bh.init(0)
    idfile = os.path.join(workdir, "unique_id_echo.bin")

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

    t0 = time.time()
    outcome = "no error"
    try:
        bh.init_distributed(rank, world, bcast)
    except bh.BenlsipHipError as e:
        outcome = "error %d after %.1f s: %s" % (e.code, time.time() - t0, str(e))
    r, n = C.c_int32(-1), C.c_int32(-1)
    bh._lib.lib().bh_comm_info(C.byref(r), C.byref(n))
    rng = np.random.default_rng(3)
    J = rng.standard_normal((50, 16))
    H = bh.AlHessian(J, None, 1.0)          # no communicator: a plain one-rank product must work
    v = rng.standard_normal(16)
    ok = bool(np.allclose(H * v, J.T @ (J @ v), rtol=1e-12, atol=1e-12))
    H.close()
    left = [f for f in os.listdir("/dev/shm") if f.startswith("bh_ipc_")]
    np.savez(os.path.join(workdir, "echo_rank%d.npz" % rank), outcome=outcome, comm=np.array([r.value, n.value]), ok=ok, left=len(left))
    print("rank %d: %s" % (rank, outcome), flush=True)


if __name__ == "__main__":
    main()
