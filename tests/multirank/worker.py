"""One rank of the two-process multi-rank test: real library, real kernels, row shard [lo, hi) of J, the all-reduce staged
through tests/multirank/libstaged_rccl.so (BH_RCCL_LIB).  Writes every result to an .npz the parent compares with the
oracle on the UNSHARDED problem and, bit for bit, with the other rank's file.  Never imports the oracle."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    rank, world, workdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import benlsip_jl_amd as bh
    from problem import make_problem
    bh.init(0)
    idfile = os.path.join(workdir, "unique_id.bin")

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
    P = make_problem()
    J, C, A, mu, fix = P["J"], P["C"], P["A"], P["mu"], P["fix"]
    d, n = J.shape
    lo, hi = bh.row_shard(d, rank, world)
    out = {"lo": lo, "hi": hi}
    H = bh.AlHessian(J[lo:hi], C, mu)                       # C is replicated; the library lets only rank 0 apply it
    out["hv"] = H * P["v"]
    out["jtu"] = H.jtv(P["u"][lo:hi])
    out["vthv"] = bh.vthv(H, P["v"])
    g = bh.gradient(H, P["rx"][lo:hi], P["ybar"])
    out["g"] = g
    gm = bh.hmul_add(H, P["s"], g)
    out["gm"] = gm
    delta = 0.1 * np.linalg.norm(g)

    # box constraints: loose and tight CG tolerance (few and many iterations -> several launch batches)
    Z = np.zeros((0, n))
    box = bh.MixedConstraints(Z, None, fix, l=P["xlow"], u=P["xupp"])
    wl = np.where(fix, 0.0, np.maximum(P["xlow"] - P["x"], -delta))
    wu = np.where(fix, 0.0, np.minimum(P["xupp"] - P["x"], delta))
    out["wl"], out["wu"] = wl, wu
    w, st, info = bh.projected_cg(gm, H, wl, wu, box, 0.1, full_output=True)           # leaves through a bound after 3 iterations
    out["box_loose_w"], out["box_loose_st"], out["box_loose_it"] = w, int(st), info["iters"]
    # wide bounds, tight tolerance: ~80 iterations, i.e. many launch-ahead batches and host decisions per rank
    big = np.full(n, 1e3)
    w, st, info = bh.projected_cg(gm, H, np.where(fix, 0.0, -big), np.where(fix, 0.0, big), box, 1e-3, full_output=True)
    out["box_tight_w"], out["box_tight_st"], out["box_tight_it"], out["box_tight_nh"] = w, int(st), info["iters"], info["n_hmul"]
    w, st, info = bh.projected_cg(gm, H, np.where(fix, 0.0, -big), np.where(fix, 0.0, big), box, 3e-2, full_output=True)
    out["box_mid_w"], out["box_mid_st"], out["box_mid_it"] = w, int(st), info["iters"]
    out["box_form"] = H.stats()["cg_kernels"]
    w, st, info = bh.projected_cg(gm, H, np.where(fix, 0.0, -big), np.where(fix, 0.0, big), box, 1e-3, full_output=True)
    # the same call again: the first batch is now sized by the previous call on this handle
    ar0 = H.stats()["n_allreduce"]
    w2, st2, info2 = bh.projected_cg(gm, H, np.where(fix, 0.0, -big), np.where(fix, 0.0, big), box, 1e-3, full_output=True)
    out["box_again_same"] = bool(np.array_equal(w, w2) and int(st) == int(st2) and info["iters"] == info2["iters"])
    # collectives of the repeated call: with the previous call's count as the first batch none is enqueued past the exit
    # (capped at 32 iterations per batch, so a longer loop over RCCL still over-launches by < one batch)
    out["box_again_allreduce"], out["box_again_nh"] = H.stats()["n_allreduce"] - ar0, info2["n_hmul"]
    w3, st3, info3 = bh.projected_cg(gm, H, np.where(fix, 0.0, -big), np.where(fix, 0.0, big), box, 3e-2, full_output=True)
    ar1 = H.stats()["n_allreduce"]
    w3, st3, info3 = bh.projected_cg(gm, H, np.where(fix, 0.0, -big), np.where(fix, 0.0, big), box, 3e-2, full_output=True)
    out["box_mid_again_allreduce"], out["box_mid_again_nh"] = H.stats()["n_allreduce"] - ar1, info3["n_hmul"]

    # linear equalities + fixed variables (reduced-form projection on the device)
    gen = bh.MixedConstraints(A, None, fix, l=P["xlow"], u=P["xupp"])
    w, st, info = bh.projected_cg(gm, H, wl, wu, gen, 1e-6, full_output=True)
    out["gen_w"], out["gen_st"], out["gen_it"] = w, int(st), info["iters"]
    out["gen_form"] = H.stats()["cg_kernels"]
    w, st, info = bh.minor_iterate(P["x"], P["s"], gm, H, gen, delta, 0.1, full_output=True)
    out["mi_w"], out["mi_st"], out["mi_alpha"] = w, int(st), info["alpha"]

    # Cauchy step from a point with a few active bounds (the active set grows on the device)
    cau = bh.MixedConstraints(A, None, None, l=P["xlow"], u=P["xupp"])
    s_c, info = bh.cauchy_step(P["x"], P["g_cauchy"], H, cau, 0.5 * np.linalg.norm(P["g_cauchy"]), full_output=True)
    out["cauchy_s"], out["cauchy_fix"], out["cauchy_nh"] = s_c, np.asarray(cau.fixvars, dtype=bool), info["n_hmul"]

    # the same with box constraints only: the image-space search (every rank keeps J d, J s_c for its rows; two scalars all-reduced
    # per breakpoint)
    cau_box = bh.MixedConstraints(Z, None, None, l=P["xlow"], u=P["xupp"])
    g_big = 1000.0 * P["g_cauchy"]            # a steep gradient: the breakpoints come 1000 x closer, the search passes many of them
    s_b, info_b = bh.cauchy_step(P["x"], g_big, H, cau_box, 0.5 * np.linalg.norm(g_big), full_output=True)
    out["cauchy_box_s"], out["cauchy_box_fix"], out["cauchy_box_passes"] = s_b, np.asarray(cau_box.fixvars, dtype=bool), info_b["n_hmul"]

    st = H.stats()
    out["n_allreduce"] = st["n_allreduce"]
    np.savez(os.path.join(workdir, "rank%d.npz" % rank), **out)
    H.close()
    bh._lib.check(bh._lib.lib().bh_comm_destroy(), "bh_comm_destroy")
    print("rank %d done: %d all-reduces" % (rank, st["n_allreduce"]), flush=True)


if __name__ == "__main__":
    main()
