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
