"""world_size-2 gloo test of the multi-GPU scheme on CPU.

The product's N>1 path = row shards + ONE all-reduce of n doubles per J'(J p) + replicated, deterministic n-vector
work.  There is no GPU here, so the shard-local products are the oracle's; what is exercised is exactly what the
scheme adds: `row_shard`, the all-reduce in place of the local product, and lock-step control flow (every rank must
reach the same status / iteration count and the same w as the unsharded solve).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import benlsip_jl_amd as bh
    import benlsip_ref as R
    d, n = 96, 24
    lo, hi = bh.row_shard(d, rank, world)
    J_full = R.synthetic_J(d, n, seed=5)
    J_loc = R.synthetic_J(hi - lo, n, seed=5, row0=lo, d_total=d)
    assert np.array_equal(J_loc, J_full[lo:hi])
    C = np.random.default_rng(9).standard_normal((1, n))
    mu = 10.0
    n_allreduce = [0]

    def sharded_hmul(H, v):
        # rank-local J_k'(J_k v); C'(mu C v) only on rank 0 (C is replicated); then ONE all-reduce of n doubles
        z = J_loc.T @ (J_loc @ v)
        if rank == 0:
            z = z + C.T @ ((mu * C) @ v)
        t = torch.from_numpy(z.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n_allreduce[0] += 1
        return t.numpy()

    inst = R.synthetic_box_vectors(d, n, fix_every=4)
    A = np.zeros((0, n))
    cons = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), inst.fixvars, l=inst.x_l, u=inst.x_u)
    g_loc = J_loc.T @ inst.r0[lo:hi]
    gt = torch.from_numpy(g_loc.copy())
    dist.all_reduce(gt)                      # g = J' r0, r0 sharded like the rows
    g = gt.numpy()
    delta = 0.1 * np.linalg.norm(g)
    w_l, w_u = R.build_step_bounds(inst.x, cons, delta)
    tr = R.CGTrace()
    w, status, iters = R.projected_cg(g, None, w_l, w_u, cons, 0.1, hmul_fn=sharded_hmul, trace=tr)
    assert n_allreduce[0] == tr.n_hmul
    # unsharded solve on every rank
    H = R.AlHessian(J_full, C, mu)
    g_ref = J_full.T @ inst.r0
    w_ref, s_ref, it_ref = R.projected_cg(g_ref, H, w_l, w_u, cons, 0.1)
    ok = (int(status) == int(s_ref)) and (iters == it_ref) and np.allclose(w, w_ref, rtol=1e-9, atol=1e-14) \
        and np.allclose(g, g_ref, rtol=1e-12)
    # lock-step: all ranks hold bit-identical w
    wt = torch.from_numpy(w.copy())
    gathered = [torch.zeros_like(wt) for _ in range(world)]
    dist.all_gather(gathered, wt)
    ok = ok and all(torch.equal(gathered[0], t) for t in gathered)
    out[rank] = 1 if ok else 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_row_sharded_pcg_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0] * world)
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    for p in procs:
        assert p.exitcode == 0
    assert list(out) == [1] * world


def _solve_worker(rank, world, port, out):
    """A whole row-sharded solve over gloo: the restated driver runs replicated, `residuals` / `jac_res` return this rank's rows,
    and the three row-dependent seams (mx :44/:58, g :45/:74, least-squares multipliers :893) are all-reduced — the CPU rehearsal of
    what tests/test_multirank_gpu.py::test_whole_row_sharded_solve_matches_unsharded_oracle runs through the library."""
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import benlsip_jl_amd as bh
    import benlsip_ref as R
    import sphere_problem as sp
    from _util import first_decision_difference

    def allsum(a):
        t = torch.from_numpy(np.atleast_1d(np.asarray(a, dtype=np.float64)).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    class ShardedNumpyOps(R.NumpyOps):
        """The oracle's operators on this rank's rows + one all-reduce where the library has one."""

        def new_hessian(self, J, C, mu):
            return R.AlHessian(np.asarray(J, dtype=np.float64), np.asarray(C, dtype=np.float64), float(mu))

        def hmul(self, H, v):
            z = H.J.T @ (H.J @ v)
            if rank == 0:
                z = z + H.C.T @ ((H.mu * H.C) @ v)
            return allsum(z)

        def vthv(self, H, v):
            Jv, Cv = H.J @ v, H.C @ v
            return float(allsum(np.dot(Jv, Jv) + (H.mu * np.dot(Cv, Cv) if rank == 0 else 0.0))[0])

        def projected_cg(self, g_minor, H, w_l, w_u, lincons, kappa2):
            w, status, _ = R.projected_cg(g_minor, H, w_l, w_u, lincons, kappa2, hmul_fn=self.hmul)
            return w, status

        def residual_sqnorm(self, rx):
            return float(allsum(np.dot(rx, rx))[0])

        def gradient(self, H, Jx, rx, Cx, y_bar):
            z = Jx.T @ rx
            if rank == 0:
                z = z + Cx.T @ y_bar
            return allsum(z)

        def jtr(self, J, r):
            return allsum(J.T @ r)

    d = 4
    lo, hi = bh.row_shard(d, rank, world)
    kw = dict(max_outer_iter=100, max_inner_iter=250)
    log = []
    x, y = R.tralcnllss(sp.x0, lambda z: sp.r(z)[lo:hi], lambda z: sp.jac_r(z)[lo:hi], sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u,
                        ops=ShardedNumpyOps(), log=log, **kw)
    log_ref = []
    x_ref, y_ref = R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u, log=log_ref, **kw)
    ok = np.linalg.norm(x - x_ref) <= 1e-6 * np.linalg.norm(x_ref) and np.linalg.norm(sp.c(x)) < R.SQRT_EPS
    diff = first_decision_difference(log_ref, log)
    if diff is not None:            # only a rounding-dominated rho may differ (see tests/test_multirank_gpu.py)
        ok = ok and all(name.startswith("rho vs") and extra["ared_in_ulps_of_mx"] <= 512.0 for name, _, _, extra in diff[3])
    xt = torch.from_numpy(np.concatenate([x, y]).copy())
    gathered = [torch.zeros_like(xt) for _ in range(world)]
    dist.all_gather(gathered, xt)
    ok = ok and all(torch.equal(gathered[0], t) for t in gathered)       # replicated control flow: identical bits
    out[rank] = 1 if ok else 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_whole_row_sharded_solve_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0] * world)
    procs = [ctx.Process(target=_solve_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(200)
    for p in procs:
        assert p.exitcode == 0
    assert list(out) == [1] * world
