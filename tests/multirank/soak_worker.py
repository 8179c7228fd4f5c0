"""One rank of the peer-exchange soak: thousands of all-reduces (n-vectors and scalars, via bh_jtv / H*v / bh_vthv /
bh_resid_sqnorm) with deliberately skewed arrival times; every result is checked on the spot against the product on the FULL
matrix (plain NumPy arithmetic on the test's own data — no oracle involved) and compared across ranks afterwards."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    rank, world, workdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 400
    import benlsip_jl_amd as bh
    bh.init(0)
    idfile = os.path.join(workdir, "unique_id_soak.bin")

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
    rng = np.random.default_rng(77)                      # the same stream on every rank: replicated inputs
    skew = np.random.default_rng(1000 + rank)            # ... but private arrival jitter
    shapes = [(601, 512), (300, 4096), (97, 33)]
    Js = [rng.standard_normal(s) / np.sqrt(s[0]) for s in shapes]
    Hs, los = [], []
    for J in Js:
        lo, hi = bh.row_shard(J.shape[0], rank, world)
        Hs.append(bh.AlHessian(J[lo:hi], None, 1.0))
        los.append((lo, hi))
    worst = 0.0
    digest = 0
    for it in range(rounds):
        k = it % len(Js)
        J, H, (lo, hi) = Js[k], Hs[k], los[k]
        d, n = J.shape
        u, v = rng.standard_normal(d), rng.standard_normal(n)
        if skew.random() < 0.3:
            time.sleep(float(skew.random()) * 2e-3)      # uneven load: this rank shows up late
        z = H.jtv(u[lo:hi])
        worst = max(worst, np.linalg.norm(z - J.T @ u) / np.linalg.norm(np.abs(J).T @ np.abs(u)))
        hv = H * v
        worst = max(worst, np.linalg.norm(hv - J.T @ (J @ v)) / np.linalg.norm(np.abs(J).T @ (np.abs(J) @ np.abs(v))))
        q = bh.vthv(H, v)
        worst = max(worst, abs(q - np.dot(J @ v, J @ v)) / np.dot(J @ v, J @ v))
        r2 = bh.resid_sqnorm(u[lo:hi])
        worst = max(worst, abs(r2 - u @ u) / (u @ u))
        digest ^= int(np.bitwise_xor.reduce(np.concatenate([z, hv, [q, r2]]).view(np.int64)))
    st = Hs[0].stats()
    np.savez(os.path.join(workdir, "soak_rank%d.npz" % rank), worst=worst, digest=np.int64(digest), n_allreduce=st["n_allreduce"])
    for H in Hs:
        H.close()
    bh._lib.check(bh._lib.lib().bh_comm_destroy(), "bh_comm_destroy")
    print("rank %d: %d rounds, worst relative error %.2e" % (rank, rounds, worst), flush=True)


if __name__ == "__main__":
    main()
