#!/usr/bin/env python3
"""bench.py — PCG subproblems/s + achieved HBM GB/s of the dominant kernel, BASELINE config 3 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one complete projected_cg subproblem (bh_pcg_dev: initial projection -> CG loop -> status) on the synthetic
dense NLS instance of SURVEY.md §8(d): J is (N*65536) x 4096 fp64, rows sharded 65536 per GPU (J generated in HBM), box
bounds with p = 512 active, mu = 10, kappa2 = 0.1; all vectors are resident in HBM when the timed region starts.
Weak scaling: each rank always owns a 65536 x 4096 shard (2 GiB); with N > 1 every H*p ends in one RCCL all-reduce of
n doubles.  `value` counts one unit per rank-shard per subproblem (N units per step), so it is the whole-job aggregate.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D_PER_GPU = 65536
N_COLS = 4096
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_COPY_CEILING_GBS = 6290.0   # the same guide's measured float4-copy figure (read + write streams; this path is read-only)


def cpu_share():
    """Host CPUs this process may actually use: the cgroup quota when there is one (the GPU boxes expose 128 hardware
    threads but a 16-CPU share; BLAS teams larger than the share get throttled), else the affinity mask / CPU count."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, int(round(int(quota) / int(period))))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, int(round(q / p)))
    except Exception:
        pass
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def setup_instance(bh, rank, world, kind, d_per_gpu=D_PER_GPU, n=N_COLS):
    syn = bh.synthetic
    d_total = d_per_gpu * world
    lo, hi = bh.row_shard(d_total, rank, world)
    H = bh.AlHessian.synthetic(hi - lo, n, row0=lo, d_total=d_total, seed=1, colscale=syn.column_scale(n, kind), mu=10.0)
    x, x_l, x_u, fix = syn.box_vectors(n, fix_every=8)
    cons = bh.MixedConstraints(np.zeros((0, n)), None, fix, l=x_l, u=x_u)
    g = H.jtv(syn.residual_rows(lo, hi))            # g = J' r0 (row-sharded J' t + all-reduce)
    w_l, w_u = syn.step_bounds(x, x_l, x_u, fix, syn.initial_tr(g))
    dv = {k: bh.DeviceVector(n, v) for k, v in (("g", g), ("wl", w_l), ("wu", w_u))}
    dv["w"] = bh.DeviceVector(n)
    return H, cons, dv, dict(g=g, w_l=w_l, w_u=w_u, x=x, x_l=x_l, x_u=x_u, fix=fix, lo=lo, hi=hi, d_total=d_total)


def pmc_traffic():
    """HBM bytes per launch of the fused kernel from the committed rocprofv3 PMC summary (profiles/rNN_pmc_traffic.json,
    produced by tools/profile_round.sh + tools/summarize_profile.py; gfx950 FETCH_SIZE half-count already corrected)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    data = json.load(open(files[-1]))
    for name, e in data.get("kernels", {}).items():
        if "row_stream_kernel<256, 8, 4, 2" in name and "hbm_bytes_per_launch" in e:
            return e["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    return None, None


def run_steps(bh, H, cons, dv, kappa2, steps):
    out = None
    for _ in range(steps):
        out = bh.projected_cg_dev(dv["g"], H, dv["wl"], dv["wu"], cons, kappa2, dv["w"])
    return out


def cpu_baseline(kind, kappa2, n_hmul_gpu, n=N_COLS, d_full=D_PER_GPU, d_sample=16384):
    """The oracle timed on a bounded sample of the same workload: rows [0, d_sample) of the same J (same vectors recipe),
    full projected_cg calls, scaled to the full row count and to the GPU run's H*p count.  Two ports are timed: the
    plain-C/OpenMP restatement (oracle/benlsip_oracle.c, all host cores) and the NumPy/OpenBLAS one (the dgemv family
    Julia's LinearAlgebra dispatches to); `value` is the faster of the two."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import benlsip_oracle as BO
    import benlsip_ref as R
    J = R.synthetic_J(d_sample, n, seed=1, kind=kind, d_total=d_full)
    inst = R.synthetic_box_vectors(d_sample, n, fix_every=8)
    A = np.zeros((0, n))
    Z = np.zeros((0, n))
    cons = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), inst.fixvars, l=inst.x_l, u=inst.x_u)
    g = J.T @ inst.r0
    w_l, w_u = R.build_step_bounds(inst.x, cons, R.initial_tr(g))

    def timed(fn, budget):
        n_h = fn()                                   # warm-up, returns the H*p count on the sample
        reps, t0 = 0, time.perf_counter()
        while True:
            fn()
            reps += 1
            el = time.perf_counter() - t0
            if el > budget or reps >= 50:
                break
        return el / reps / max(n_h, 1), reps

    # the C/OpenMP port at several team sizes (a 128-thread team is not the fastest for a memory-bound dgemv), best kept
    omp_max = BO.num_threads()
    t_c, reps_c, omp_best, by_team = None, 0, omp_max, {}
    share = cpu_share()
    for team in sorted({t for t in (8, 16, 32, 64, share, omp_max) if t <= omp_max}):
        BO.set_num_threads(team)
        t_k, reps_k = timed(lambda: BO.projected_cg(g, J, Z, 10.0, w_l, w_u, A, inst.fixvars, cons.chol_L, kappa2)[3], 2.5)
        by_team[team] = 1e3 * t_k * (d_full / d_sample)
        if t_c is None or t_k < t_c:
            t_c, reps_c, omp_best = t_k, reps_k, team
    BO.set_num_threads(omp_max)
    H = R.AlHessian(J, Z, 10.0)

    def np_run():
        tr = R.CGTrace()
        R.projected_cg(g, H, w_l, w_u, cons, kappa2, trace=tr)
        return tr.n_hmul
    t_np, reps_np = timed(np_run, 3.0)
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        np_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
        # the box exposes more hardware threads than its CPU share: OpenBLAS at the C port's best team size as well
        with threadpool_limits(limits=int(omp_best)):
            t_np2, reps_np2 = timed(np_run, 3.0)
        if t_np2 < t_np:
            t_np, reps_np, np_threads = t_np2, reps_np2, int(omp_best)
    except Exception:
        np_threads = os.cpu_count() or 1
    scale = d_full / d_sample
    best, cores, which = (t_c, omp_best, "C/OpenMP") if t_c <= t_np else (t_np, np_threads, "NumPy/OpenBLAS")
    t_full = best * scale * max(n_hmul_gpu, 1)
    return {
        "value": 1.0 / t_full, "unit": "PCG subproblems/s", "cores": int(cores), "kind": "port",
        "sample": "oracle projected_cg on rows [0,%d) of the same %dx%d J (1/%d of the rows; %s port, %d repeats); time per H*p "
                  "scaled x%d in rows and to the GPU run's %d H*p per subproblem"
                  % (d_sample, d_full, n, d_full // d_sample, which, reps_c if which == "C/OpenMP" else reps_np, d_full // d_sample, n_hmul_gpu),
        "ms_per_hmul_full_size": {"c_openmp": 1e3 * t_c * scale, "numpy_openblas": 1e3 * t_np * scale},
        "threads": {"c_openmp": int(omp_best), "numpy_openblas": int(np_threads)},
        "c_openmp_ms_per_hmul_by_team": by_team, "cpu_share": share, "hardware_threads": os.cpu_count(),
        "host_gbs": 2 * 8.0 * d_sample * n / best / 1e9,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--variant", choices=["wc", "ic"], default="wc",
                    help="wc: well-conditioned J (a handful of CG iterations); ic: columns scaled 10^(-3j/n) (hundreds)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-ic-extra", action="store_true",
                    help="skip the ill-conditioned extra run (its over-launched no-op kernels would pull down rocprofv3's per-kernel average)")
    args = ap.parse_args()

    # keep idle BLAS worker teams within the CPU share: a 128-thread team spinning on a 16-CPU quota gets the whole
    # process throttled for tens of milliseconds at a time — also while it only drives the GPU
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=cpu_share())
    except Exception:
        pass
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    dist = None
    # BH_BENCH_REHEARSAL=1: run the N > 1 code path on a ONE-GPU box — every rank on device 0, torch over gloo, the
    # library's all-reduce through the host-staged stand-in (BH_RCCL_LIB, tests/multirank/) — to exercise this script's
    # multi-rank flow (shards, replica check, teardown).  Its numbers mean nothing and are labelled as such.
    rehearsal = os.environ.get("BH_BENCH_REHEARSAL", "0") not in ("", "0")
    if rehearsal:
        local_rank = 0
    else:
        # a launcher that pins one visible device per rank (HIP_VISIBLE_DEVICES) leaves every rank with ordinal 0
        local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import benlsip_jl_amd as bh
    bh.init(local_rank, flags=bh._lib.BH_FLAG_PROFILE)
    if dist is not None:
        bh.init_distributed(rank, world, bh.torch_broadcast_bytes(torch.device("cuda", local_rank)))

    def barrier():
        bh._lib.lib().bh_synchronize()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    kind = 0 if args.variant == "wc" else 1
    kappa2 = 0.1
    H, cons, dv, host = setup_instance(bh, rank, world, kind)

    run_steps(bh, H, cons, dv, kappa2, args.warmup)
    H.reset_stats()
    barrier()
    t0 = time.perf_counter()
    status, iters, n_hmul = run_steps(bh, H, cons, dv, kappa2, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = H.stats()
    replicas_identical = None
    if dist is not None:
        # Lock-step check (outside the timed region): w is replicated state, every rank must hold the SAME BITS — the
        # launch-ahead schedule relies on it.  Compare a checksum of the bit patterns across ranks.
        bits = dv["w"].download().view(np.int64)
        chk = int(np.bitwise_xor.reduce(bits)) ^ (int(iters) << 20) ^ int(n_hmul)
        lo_hi = torch.tensor([chk, -chk], dtype=torch.int64, device="cuda")
        dist.all_reduce(lo_hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(int(lo_hi[0].item()) == chk and int(lo_hi[1].item()) == -chk)
        flag = torch.tensor([1 if replicas_identical else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        replicas_identical = bool(flag.item())

    traffic, traffic_src = pmc_traffic() if world == 1 else (None, None)
    ms_per_step = 1e3 * elapsed / args.steps
    hmul_ms = st["hmul_ms"] / max(st["hmul_timed"], 1)
    achieved = st["bytes_per_hmul"] / (hmul_ms * 1e-3) / 1e9 if hmul_ms > 0 else 0.0
    line = {
        "metric": "PCG subproblems/sec + achieved HBM GB/s, dense m=65536 n=4096 fp64",     # BASELINE.json's metric, verbatim
        "value": world * args.steps / elapsed,
        "unit": "PCG subproblems/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL on one GPU (all ranks share device 0, host-staged all-reduce): not a measurement",
        "config": {
            "workload": "BASELINE config 3 per GPU: synthetic dense NLS, J %d x %d fp64 (%d rows per GPU, row-sharded), box bounds, "
                        "p=512 active, mu=10, kappa2=0.1, variant=%s; one step = one projected_cg subproblem (bh_pcg_dev), "
                        "vectors resident in HBM" % (host["d_total"], N_COLS, D_PER_GPU, args.variant),
            "unit_definition": "one unit = one 65536x4096 row shard of one projected_cg call: at N GPUs a step solves ONE subproblem "
                               "on N*65536 rows and counts N units (weak scaling); subproblems_per_s_global counts it once",
            "d_total": host["d_total"], "n": N_COLS, "rows_per_gpu": D_PER_GPU, "parallelism": "row-shard x%d + 1 all-reduce(n) per H*p" % world,
            "cg_status": status.name, "cg_iters_per_subproblem": iters - 1, "hmul_per_subproblem": n_hmul,
        },
        "subproblems_per_s_global": args.steps / elapsed,
        "replicas_bitwise_identical": replicas_identical,
        "cg_iters_per_s": (iters - 1) * args.steps / elapsed,
        "ms_per_cg_iteration": ms_per_step / max(n_hmul, 1),
        "roofline": {
            "bound": "hbm", "kernel": "row_stream_kernel<256,8,4,MODE_FUSED,NT> (single-read J'(Jp), non-temporal loads)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "frac_of_guide_measured_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": st["bytes_per_hmul"], "avg_launch_ms": hmul_ms,
            "launches_timed": st["hmul_timed"],
        },
    }
    if not args.no_extras:
        d_loc = host["hi"] - host["lo"]
        mv_bytes = 8.0 * d_loc * N_COLS + 8.0 * N_COLS + 8.0 * d_loc
        ms_f, ms_jv, ms_jtv = (H.time_kernel(k, 20) for k in (0, 1, 2))
        line["matvec"] = {
            "jv_ms": ms_jv, "jv_gbs": mv_bytes / ms_jv / 1e6, "jv_frac": mv_bytes / ms_jv / 1e6 / HBM_PEAK_GBS,
            "jtv_ms": ms_jtv, "jtv_gbs": mv_bytes / ms_jtv / 1e6, "jtv_frac": mv_bytes / ms_jtv / 1e6 / HBM_PEAK_GBS,
            "fused_ms": ms_f, "fused_gbs": st["bytes_per_hmul"] / ms_f / 1e6,
            "note": "back-to-back launches timed with hipEvents on the launch stream (bh_time_kernel), this rank's shard",
        }
        # practical ceiling: a kernel that does nothing but read the same 2 GiB once (best of 1/2/4/8 workgroups per CU)
        probe_ms = {1 << (k - 3): H.time_kernel(k, 10) for k in (3, 4, 5, 6)}
        best_wg = min(probe_ms, key=probe_ms.get)
        probe_gbs = 8.0 * d_loc * N_COLS / probe_ms[best_wg] / 1e6
        line["read_probe"] = {
            "gbs": probe_gbs, "ms": probe_ms[best_wg], "workgroups_per_cu": best_wg, "frac_of_peak": probe_gbs / HBM_PEAK_GBS,
            "fused_kernel_vs_probe": (st["bytes_per_hmul"] / ms_f / 1e6) / probe_gbs,
            "ms_by_workgroups_per_cu": probe_ms,
            "note": "read_probe_kernel: 16-byte non-temporal loads + adds only, same J image; what a single-read kernel can reach here",
        }
    if world == 1 and not args.no_extras and not args.no_ic_extra and args.variant == "wc":
        # steady-state CG iteration cost on the ill-conditioned variant (23 iterations per subproblem): outside the timed region
        # (the first handle stays allocated: freeing 2 GiB here makes the driver scrub it in the background, which took
        # ~5 % of the HBM bandwidth from the next ~60 ms of kernels — tools/ic_transient.py)
        H2, cons2, dv2, _ = setup_instance(bh, rank, world, 1)
        run_steps(bh, H2, cons2, dv2, kappa2, 3)
        barrier()
        t1 = time.perf_counter()
        st2, it2, nh2 = run_steps(bh, H2, cons2, dv2, kappa2, 10)
        barrier()
        el2 = (time.perf_counter() - t1) / 10
        line["ic_variant"] = {"workload": "same instance with columns of J scaled by 10^(-3j/n)", "cg_status": st2.name,
                              "hmul_per_subproblem": nh2, "ms_per_subproblem": 1e3 * el2, "ms_per_cg_iteration": 1e3 * el2 / max(nh2, 1),
                              "subproblems_per_s": 1.0 / el2,
                              "cg_iteration_gbs": st["bytes_per_hmul"] / (el2 / max(nh2, 1)) / 1e9}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(kind, kappa2, n_hmul)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line), flush=True)
    barrier()
    if dist is not None:
        H.close()                                   # handles first: the communicator cannot change under a live handle
        bh._lib.check(bh._lib.lib().bh_comm_destroy(), "bh_comm_destroy")
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
