#!/usr/bin/env python3
"""bench.py — PCG subproblems/s + achieved HBM GB/s of the dominant kernel, BASELINE config 3 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one complete projected_cg subproblem (bh_pcg_dev: initial projection -> CG loop -> status) on the synthetic
dense NLS instance of SURVEY.md §8(d): box bounds with p = 512 active, mu = 10, kappa2 = 0.1; all vectors are resident in
HBM when the timed region starts.
  weak scaling (default): J is (N*65536) x 4096 fp64, every rank owns a 65536 x 4096 shard (2 GiB, generated in HBM);
  strong scaling: J is 65536 x 4096 whatever N is, rows split over the ranks (SURVEY.md §8e's "visible all-reduce" case).
With N > 1 every H*p ends in one all-reduce of n doubles.  `value` = subproblems per second of the WHOLE JOB (all ranks work
on the same subproblem: it is counted once); `shard_units_per_s` = N x that (one unit per 65536-row shard, weak scaling only).
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


MA_CONFIG5 = 64            # BASELINE config 5: 64 linear equalities A = u(4, .) next to the 512 active bounds


def setup_instance(bh, rank, world, kind, d_per_gpu=D_PER_GPU, n=N_COLS, strong=False, config=3):
    syn = bh.synthetic
    d_total = d_per_gpu if strong else d_per_gpu * world
    lo, hi = bh.row_shard(d_total, rank, world)
    H = bh.AlHessian.synthetic(hi - lo, n, row0=lo, d_total=d_total, seed=1, colscale=syn.column_scale(n, kind), mu=10.0)
    x, x_l, x_u, fix = syn.box_vectors(n, fix_every=8)
    mA = MA_CONFIG5 if config == 5 else 0
    A = syn.splitmix_uniform(4, np.arange(mA * n)).reshape((mA, n), order="F") if mA else np.zeros((0, n))
    cons = bh.MixedConstraints(A, None, fix, l=x_l, u=x_u)      # reduced projection form: the library factors A_free A_free' itself
    g = H.jtv(syn.residual_rows(lo, hi))            # g = J' r0 (row-sharded J' t + all-reduce)
    w_l, w_u = syn.step_bounds(x, x_l, x_u, fix, syn.initial_tr(g))
    dv = {k: bh.DeviceVector(n, v) for k, v in (("g", g), ("wl", w_l), ("wu", w_u))}
    dv["w"] = bh.DeviceVector(n)
    return H, cons, dv, dict(g=g, w_l=w_l, w_u=w_u, x=x, x_l=x_l, x_u=x_u, fix=fix, lo=lo, hi=hi, d_total=d_total)


def pmc_traffic(config=3):
    """HBM bytes per launch of the fused kernel from the committed rocprofv3 PMC summary (profiles/rNN_pmc_traffic.json, or
    rNN_config5_pmc_traffic.json for --config 5; produced by tools/profile_round.sh + tools/summarize_profile.py; gfx950
    FETCH_SIZE half-count already corrected)."""
    import glob
    import re
    pat = r"r\d+_pmc_traffic\.json$" if config == 3 else r"r\d+_config%d_pmc_traffic\.json$" % config
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")) if re.search(pat, os.path.basename(f)))
    if not files:
        return None, None
    data = json.load(open(files[-1]))
    # the dominant kernel: the fused single-read J'(Jp) launch of the two-kernel CG iteration (CGP = 1), or its plain form
    for key in ("row_stream_kernel<256, 8, 4, 2, 1, 1, 0, 1>", "row_stream_kernel<256, 8, 4, 2, 1, 1, 0>", "row_stream_kernel<256, 8, 4, 2"):
        for name, e in data.get("kernels", {}).items():
            if key in name and "hbm_bytes_per_launch" in e:
                return e["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    return None, None


def run_steps(bh, H, cons, dv, kappa2, steps):
    """`steps` back-to-back bh_pcg_dev calls (each one a complete, synchronous subproblem).  The argument objects are built once:
    what is timed is the library call, not Python's marshalling of fourteen arguments."""
    import ctypes as ct
    lib = bh._lib.lib()
    status, iters, n_hmul = ct.c_int32(-1), ct.c_int32(0), ct.c_int32(0)
    args = (H.handle, cons.handle, dv["g"].ptr, dv["wl"].ptr, dv["wu"].ptr, ct.c_double(kappa2), ct.c_double(bh.operators.SQRT_EPS),
            ct.c_double(1e-10), dv["w"].ptr, ct.byref(status), ct.byref(iters), None, ct.c_int64(0), ct.byref(n_hmul))
    fn = lib.bh_pcg_dev
    for _ in range(steps):
        rc = fn(*args)
        if rc != 0:
            bh._lib.check(rc, "bh_pcg_dev")
    return bh.CGStatus(status.value), iters.value, n_hmul.value


def host_synthetic_J(R, d, n, kind, chunk=8192):
    """The same J the device generates (counter-based), built on the host in row blocks (Fortran order)."""
    J = np.empty((d, n), order="F")
    for lo in range(0, d, chunk):
        hi = min(d, lo + chunk)
        J[lo:hi] = R.synthetic_J(hi - lo, n, seed=1, kind=kind, row0=lo, d_total=d)
    return J


def cpu_baseline(kind, kappa2, n_hmul_gpu, n=N_COLS, d_full=D_PER_GPU, repeats=5, config=3):
    """The oracle timed on the host cores, on the FULL bench workload (the same 65536 x 4096 J, 2 GiB, same vectors recipe):
    whole projected_cg calls, >= `repeats` of them per port, median.  Two ports are timed: the plain-C/OpenMP restatement
    (oracle/benlsip_oracle.c) and the NumPy/OpenBLAS one (the dgemv family Julia's LinearAlgebra dispatches to); `value` is
    the faster of the two.  Bounded: ~2 H*p of ~12-25 ms per call, a handful of calls per port and team size."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import benlsip_oracle as BO
    import benlsip_ref as R
    t_gen = time.perf_counter()
    J = host_synthetic_J(R, d_full, n, kind)
    t_gen = time.perf_counter() - t_gen
    inst = R.synthetic_box_vectors(d_full, n, fix_every=8)
    mA = MA_CONFIG5 if config == 5 else 0
    A = R.splitmix_uniform(4, np.arange(mA * n)).reshape((mA, n), order="F") if mA else np.zeros((0, n))
    Z = np.zeros((0, n))
    cons = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), inst.fixvars, l=inst.x_l, u=inst.x_u)   # the reference's augmented factor
    g = J.T @ inst.r0
    w_l, w_u = R.build_step_bounds(inst.x, cons, R.initial_tr(g))

    def timed(fn):
        n_h = fn()                                   # warm-up, returns the H*p count of one subproblem
        ts = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), n_h

    # the C/OpenMP port at a few team sizes (a 128-thread team is not the fastest for a memory-bound dgemv), best kept
    omp_max = BO.num_threads()
    share = cpu_share()
    t_c, omp_best, by_team, nh_c = None, omp_max, {}, 0
    for team in sorted({t for t in (8, 32, share, omp_max) if t <= omp_max}):
        BO.set_num_threads(team)
        t_k, nh_c = timed(lambda: BO.projected_cg(g, J, Z, 10.0, w_l, w_u, A, inst.fixvars, cons.chol_L, kappa2)[3])
        by_team[team] = 1e3 * t_k / max(nh_c, 1)
        if t_c is None or t_k < t_c:
            t_c, omp_best = t_k, team
    BO.set_num_threads(omp_max)
    H = R.AlHessian(J, Z, 10.0)

    def np_run():
        tr = R.CGTrace()
        R.projected_cg(g, H, w_l, w_u, cons, kappa2, trace=tr)
        return tr.n_hmul
    t_np, nh_np = timed(np_run)
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        np_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
        if int(omp_best) != np_threads:
            # the box exposes more hardware threads than its CPU share: OpenBLAS at the C port's best team size as well
            with threadpool_limits(limits=int(omp_best)):
                t_np2, _ = timed(np_run)
            if t_np2 < t_np:
                t_np, np_threads = t_np2, int(omp_best)
    except Exception:
        np_threads = os.cpu_count() or 1
    best, cores, which, nh = (t_c, omp_best, "C/OpenMP", nh_c) if t_c <= t_np else (t_np, np_threads, "NumPy/OpenBLAS", nh_np)
    return {
        "value": 1.0 / best, "unit": "PCG subproblems/s", "cores": int(cores), "kind": "port",
        "sample": "oracle projected_cg (%s port) on the FULL workload: the same %d x %d J (2 GiB, host copy of the device generator), "
                  "%d H*p per subproblem (GPU run: %d), median of %d whole subproblems after one warm-up"
                  % (which, d_full, n, nh, n_hmul_gpu, repeats),
        "ms_per_subproblem": {"c_openmp": 1e3 * t_c, "numpy_openblas": 1e3 * t_np},
        "ms_per_hmul": {"c_openmp": 1e3 * t_c / max(nh_c, 1), "numpy_openblas": 1e3 * t_np / max(nh_np, 1)},
        "threads": {"c_openmp": int(omp_best), "numpy_openblas": int(np_threads)},
        "c_openmp_ms_per_hmul_by_team": by_team, "cpu_share": share, "hardware_threads": os.cpu_count(),
        "host_gbs": 2 * 8.0 * d_full * n * max(nh, 1) / best / 1e9, "host_J_generation_s": t_gen,
    }


def self_launch(n_ranks, argv, script=None):
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as CHILD processes (one per GPU, the same
    environment torch.distributed.run would give them: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relay
    rank 0's JSON line on stdout and everything else on stderr, and return the worst child exit code.  This process never
    imports torch or the library and never touches HIP (tests/test_host_cpu.py checks that libamdhip64 is not mapped here),
    so nothing that has initialised a GPU is ever replaced or forked.  A rank that dies takes the job with it: the
    others are given a grace period (they may be waiting in a collective for it) and are then killed by PID."""
    import socket
    import subprocess
    import threading
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=env.get("MASTER_PORT", str(port)), BH_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, cpu_share() // n_ranks)))
    procs = []
    for r in range(n_ranks):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1))

    def relay(r, stream, is_out):
        for ln in stream:
            if is_out and r == 0 and ln.lstrip().startswith("{"):      # the JSON line; library chatter on stdout ("[Gloo] Rank 0 is connected ...") goes to stderr
                sys.stdout.write(ln)
                sys.stdout.flush()
            else:
                sys.stderr.write("[rank %d] %s" % (r, ln))
                sys.stderr.flush()

    threads = [threading.Thread(target=relay, args=(r, s, o), daemon=True)
               for r, p in enumerate(procs) for s, o in ((p.stdout, True), (p.stderr, False))]
    for t in threads:
        t.start()
    grace = float(os.environ.get("BH_BENCH_PEER_GRACE_S", "60"))
    rcs, first_fail = [None] * n_ranks, None
    try:
        while any(rc is None for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = p.poll()
                    if rcs[r] not in (None, 0) and first_fail is None:
                        first_fail = time.monotonic()
                        sys.stderr.write("bench: rank %d exited with code %d\n" % (r, rcs[r]))
            if first_fail is not None and time.monotonic() - first_fail > grace:
                for r, p in enumerate(procs):
                    if rcs[r] is None:
                        sys.stderr.write("bench: killing rank %d (pid %d), %g s after a peer failed\n" % (r, p.pid, grace))
                        p.kill()
                first_fail = float("inf")
            time.sleep(0.05)
    finally:
        for p in procs:                       # the launcher is going away (interrupt, error): no rank is left behind (exact PIDs)
            if p.poll() is None:
                p.kill()
    for t in threads:
        t.join(timeout=5)
    bad = [rc for rc in rcs if rc != 0]
    return 0 if not bad else (max(bad) if max(bad) > 0 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: 65536 rows per GPU (default); strong: the 65536 rows of config 3 split over the GPUs")
    ap.add_argument("--variant", choices=["wc", "ic"], default="wc",
                    help="wc: well-conditioned J (a handful of CG iterations); ic: columns scaled 10^(-3j/n) (hundreds)")
    ap.add_argument("--config", type=int, choices=[3, 5], default=3,
                    help="BASELINE config: 3 = box bounds (the headline), 5 = the same J with 64 linear equalities (projection kernel path)")
    ap.add_argument("--preheat", type=int, default=40,
                    help="untimed subproblems run before the W warm-up steps to bring the device back to its steady clocks after the host-side set-up")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-ic-extra", action="store_true",
                    help="skip the ill-conditioned extra run (its over-launched no-op kernels would pull down rocprofv3's per-kernel average)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # typed as `python bench.py --gpus N`: be the launcher (before anything here imports torch or touches HIP)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

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
    strong = args.scaling == "strong"

    import torch
    dist = None
    # BH_BENCH_REHEARSAL=1: run the N > 1 code path on a ONE-GPU box — every rank on device 0, torch over gloo, the
    # library's all-reduce over its peer-buffer transport (hipIpc works between processes on one device; RCCL refuses
    # duplicate devices) — to exercise this script's multi-rank flow (shards, replica check, teardown).  Its numbers mean
    # nothing and are labelled as such.
    rehearsal = os.environ.get("BH_BENCH_REHEARSAL", "0") not in ("", "0")
    if rehearsal:
        local_rank = 0
        os.environ.setdefault("BH_COMM", "ipc")
    else:
        # a launcher that pins one visible device per rank (HIP_VISIBLE_DEVICES) leaves every rank with ordinal 0
        local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # torch.distributed only carries the rendezvous (unique id), the barriers and a few scalars: BH_BENCH_PG=gloo keeps
        # torch from creating an RCCL communicator of its own next to the library's (experiment switch)
        if rehearsal or os.environ.get("BH_BENCH_PG", "nccl") == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import benlsip_jl_amd as bh
    bh.init(local_rank, flags=bh._lib.BH_FLAG_PROFILE)
    comm_note = None
    if dist is not None and world > 1:
        # Both transports are brought up when possible: RCCL (BASELINE's north_star names it) and the library's one-shot
        # peer-buffer exchange.  The K timed steps run on each; which run is the headline is decided below (valid and faster).
        bcast = bh.torch_broadcast_bytes(None if (rehearsal or os.environ.get("BH_BENCH_PG", "nccl") == "gloo") else torch.device("cuda", local_rank))
        user_choice = os.environ.get("BH_COMM", "auto") != "auto"      # BH_COMM=auto (or unset): walk the list below
        os.environ.setdefault("BH_PEER_TIMEOUT_S", "5")      # a peer exchange that does not work must not eat the run's time budget
        pg_dev = "cpu" if (rehearsal or os.environ.get("BH_BENCH_PG", "nccl") == "gloo") else "cuda"

        def bring_up(mode):
            """bh_comm_init with BH_COMM = mode on every rank; the verdict is taken TOGETHER (a rank whose own init worked tears it
            down again when a peer's did not), so all ranks walk down the same list of fall-backs."""
            os.environ["BH_COMM"] = mode
            err = None
            try:
                bh.init_distributed(rank, world, bcast)
            except Exception as e:                       # whatever went wrong here: the other ranks are waiting in the all-reduce below
                err = "%s: %s" % (type(e).__name__, e)
            flag = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=pg_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.SUM)
            if flag.item() > 0:
                if err is None:
                    bh._lib.lib().bh_comm_destroy()
                return err or "bh_comm_init failed on %d other rank(s)" % int(flag.item())
            return None

        # default: both transports; if that cannot be had, RCCL alone (what the north_star names), then the peer buffers alone
        notes = []
        for mode in ([os.environ["BH_COMM"]] if user_choice else ["both", "rccl", "ipc"]):
            err = bring_up(mode)
            if err is None:
                break
            notes.append("BH_COMM=%s: %s" % (mode, err))
        else:
            raise SystemExit("bench: no communicator could be brought up: " + " | ".join(notes))
        if notes:
            comm_note = "fell back to BH_COMM=%s (%s)" % (os.environ["BH_COMM"], " | ".join(notes))
    elif dist is not None:
        bh.init_distributed(rank, world, bh.torch_broadcast_bytes(None if (rehearsal or os.environ.get("BH_BENCH_PG", "nccl") == "gloo") else torch.device("cuda", local_rank)))
    comm_mode = os.environ.get("BH_COMM", "rccl") if world > 1 else None

    def barrier():
        bh._lib.check(bh._lib.lib().bh_synchronize(), "bh_synchronize")
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def barrier_noexcept():
        bh._lib.lib().bh_synchronize()              # its error code (a failed peer exchange) is picked up by the caller's next call
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    tdev = "cpu" if (rehearsal or os.environ.get("BH_BENCH_PG", "nccl") == "gloo") else "cuda"

    def gather(values):
        """Every rank's list of floats, as an (N, len) array on every rank."""
        if dist is None or world == 1:
            return np.asarray([values], dtype=np.float64)
        t = torch.zeros(world, len(values), dtype=torch.float64, device=tdev)
        t[rank] = torch.tensor(values, dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()

    kind = 0 if args.variant == "wc" else 1
    kappa2 = 0.1
    H, cons, dv, host = setup_instance(bh, rank, world, kind, strong=strong, config=args.config)

    def timed_run(steps, warmup):
        """(elapsed max over ranks, (status, iters, n_hmul), stats, error).  A library error on ANY rank (e.g. a peer exchange
        that timed out) is caught so that this rank still takes part in the barriers and the gather — otherwise the other
        ranks would wait for it forever — and is reported to every rank alike."""
        err, out, t0 = None, None, time.perf_counter()
        try:
            run_steps(bh, H, cons, dv, kappa2, warmup)
            H.reset_stats()
        except bh.BenlsipHipError as e:
            err = str(e)
        barrier_noexcept()
        t0 = time.perf_counter()
        if err is None:
            try:
                out = run_steps(bh, H, cons, dv, kappa2, steps)
            except bh.BenlsipHipError as e:
                err = str(e)
        barrier_noexcept()
        el = time.perf_counter() - t0
        g = gather([el, 0.0 if err is None else 1.0])
        if g[:, 1].any():
            err = err or "a peer rank reported a library error"
        return float(g[:, 0].max()), out, (H.stats() if err is None else None), err

    # hipEvents around H*p launches INSIDE the timed region: an event pair costs ~10 us of stream time, so bracketing every launch
    # slows the loop it measures (measured at K = 20: 690 instead of 630 us per subproblem, -9 %).  Long runs sample every 8th
    # launch, short ones every 4th (10 samples at K = 20, < 1 % cost); a DENSE pass outside the timed region (every launch of
    # 16 subproblems) is reported next to it as roofline.dense_sample.
    st_probe = run_steps(bh, H, cons, dv, kappa2, 1)
    timed_stride = 4 if args.steps * st_probe[2] <= 64 else 8
    timed_stride = int(os.environ.get("BH_BENCH_EV_STRIDE", timed_stride))
    bh.set_option("profile_stride", timed_stride)
    # Preheat (untimed, before the W warm-up steps, all ranks alike): the device idles for seconds while the instance is set up
    # on the host and then needs ~15-20 ms of work to be back at its steady clocks — the first ~25 subproblems after the set-up
    # run 3-6 % slower (662, 657, 656, 648, ... -> 624 us: tools/scratch/bracket_cost.py).  With W = 5 the K timed steps would sit
    # inside that ramp; the preheat is reported in the line ("preheat_steps") and changes nothing inside the timed region.
    if args.preheat > 0:
        run_steps(bh, H, cons, dv, kappa2, args.preheat)
    elapsed, out, st, run_err = timed_run(args.steps, args.warmup)
    if run_err is not None:                         # every rank sees the same verdict (gathered), so all of them stop here
        raise SystemExit("bench: the timed run failed: %s" % run_err)
    status, iters, n_hmul = out
    def replica_check(it_, nh_):
        """Lock-step check (outside the timed region): w is replicated state, every rank must hold the SAME BITS — the
        launch-ahead schedule relies on it.  Compares a checksum of the bit patterns across ranks; returns
        (identical, this rank's w)."""
        w_host = dv["w"].download()
        bits = w_host.view(np.uint64)
        x = int(np.bitwise_xor.reduce(bits))
        s = int(np.add.reduce(bits * (np.arange(bits.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1))))   # position-weighted, wraps mod 2^64
        # every field below is an integer < 2^32, exactly representable in the float64 the gather carries
        allchk = gather([float(x >> 32), float(x & 0xFFFFFFFF), float(s >> 32), float(s & 0xFFFFFFFF), float(it_), float(nh_)])
        return bool(np.all(allchk == allchk[0])), w_host

    replicas_identical, w_headline = None, None
    headline_transport, other_run = None, None
    if dist is not None and world > 1:
        replicas_identical, w_headline = replica_check(iters, n_hmul)
        headline_transport = "rccl" if comm_mode in ("rccl", "both") else "peer_buffers"
    if dist is not None and world > 1 and comm_mode == "both" and os.environ.get("BH_BENCH_HEADLINE", "auto") != "rccl":
        # Both transports are up: the same K steps once more with the all-reduce on the library's peer-buffer exchange (two kernels
        # per CG iteration, the exchange inside the update kernel).  It becomes the headline only if it is VALID — no library
        # error on any rank, replicas bit-identical, and the same w as the RCCL run up to the rounding of another summation
        # order — and faster; the other run is reported next to it.  Every decision is taken from gathered values.
        try:
            bh.set_option("comm_path", 1)
            switched = 1.0
        except bh.BenlsipHipError:
            switched = 0.0
        if gather([switched])[:, 0].min() > 0:
            el_p, out_p, st_p, err_p = timed_run(args.steps, args.warmup)
            ident_p, rel_p = False, float("inf")
            if err_p is None:
                ident_p, w_peer = replica_check(out_p[1], out_p[2])
                rel_p = float(gather([float(np.linalg.norm(w_peer - w_headline) / max(np.linalg.norm(w_headline), 1e-300))]).max())
            valid = err_p is None and ident_p and rel_p <= 1e-8 and out_p[0] == status and out_p[2] == n_hmul
            peer_run = {"transport": "peer_buffers", "steps": args.steps, "valid": bool(valid), "error": err_p,
                        "ms_per_step": 1e3 * el_p / args.steps, "value": args.steps / el_p,
                        "replicas_bitwise_identical": ident_p, "w_rel_diff_vs_rccl_run": rel_p,
                        "cg_status": None if out_p is None else out_p[0].name, "hmul_per_subproblem": None if out_p is None else out_p[2]}
            rccl_run = {"transport": "rccl", "steps": args.steps, "valid": True, "ms_per_step": 1e3 * elapsed / args.steps,
                        "value": args.steps / elapsed, "replicas_bitwise_identical": replicas_identical}
            if valid and el_p < elapsed:
                elapsed, st, (status, iters, n_hmul) = el_p, st_p, out_p
                replicas_identical, headline_transport, other_run = ident_p, "peer_buffers", rccl_run
            else:
                other_run = peer_run
        try:
            bh.set_option("comm_path", 0)           # the measurements below select their path themselves
        except bh.BenlsipHipError:
            pass

    # dense sample, outside the timed region: every H*p launch of 16 subproblems bracketed by events (all ranks do the same calls)
    dense = None
    try:
        bh.set_option("profile_stride", 1)
        H.reset_stats()
        run_steps(bh, H, cons, dv, kappa2, 16)
        barrier_noexcept()
        sd = H.stats()
        if sd["hmul_timed"] > 0:
            d_ms = sd["hmul_ms"] / sd["hmul_timed"]
            dense = {"launches_timed": sd["hmul_timed"], "avg_launch_ms": d_ms, "achieved": sd["bytes_per_hmul"] / (d_ms * 1e-3) / 1e9,
                     "note": "every H*p launch of 16 subproblems after the timed region, hipEvents on the launch stream"}
    except bh.BenlsipHipError:
        dense = None
    bh.set_option("profile_stride", timed_stride)

    traffic, traffic_src = pmc_traffic(args.config) if world == 1 else (None, None)
    ms_per_step = 1e3 * elapsed / args.steps
    hmul_ms = st["hmul_ms"] / max(st["hmul_timed"], 1)
    achieved = st["bytes_per_hmul"] / (hmul_ms * 1e-3) / 1e9 if hmul_ms > 0 else 0.0
    per_rank = gather([achieved, hmul_ms, st["bytes_per_hmul"], float(host["hi"] - host["lo"])])
    rows_per_gpu = host["d_total"] // world
    line = {
        "metric": "PCG subproblems/sec + achieved HBM GB/s, dense m=65536 n=4096 fp64",     # BASELINE.json's metric, verbatim
        "value": args.steps / elapsed,
        "unit": "PCG subproblems/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preheat_steps": args.preheat, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
        "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL on one GPU (all ranks share device 0): not a measurement",
        "config": {
            "workload": "BASELINE config %d%s: synthetic dense NLS, J %d x %d fp64 (%d rows per GPU, row-sharded), %s, "
                        "p=512 active, mu=10, kappa2=0.1, variant=%s; one step = one projected_cg subproblem (bh_pcg_dev), "
                        "vectors resident in HBM" % (args.config, " (rows split over the GPUs)" if strong else " per GPU", host["d_total"], N_COLS, rows_per_gpu,
                                                     "box bounds" if args.config == 3 else "box bounds + %d linear equalities A = u(4,.) (reduced projection form)" % MA_CONFIG5,
                                                     args.variant),
            "unit_definition": "value counts one unit per projected_cg subproblem of the whole job (all ranks work on the same subproblem); "
                               "under weak scaling the subproblem grows with N (N*65536 rows): shard_units_per_s = N * value",
            "d_total": host["d_total"], "n": N_COLS, "rows_per_gpu": rows_per_gpu,
            "parallelism": "row-shard x%d + 1 all-reduce(n) per H*p%s" % (world, "" if headline_transport is None else " over " + headline_transport),
            "cg_status": status.name, "cg_iters_per_subproblem": iters - 1, "hmul_per_subproblem": n_hmul,
        },
        "shard_units_per_s": (world * args.steps / elapsed) if not strong else None,
        "replicas_bitwise_identical": replicas_identical,
        "headline_transport": headline_transport, "other_transport_run": other_run,
        "cg_iters_per_s": (iters - 1) * args.steps / elapsed,
        "ms_per_cg_iteration": ms_per_step / max(n_hmul, 1),
        "roofline": {
            "bound": "hbm",
            "kernel": ("row_stream_kernel<256,8,4,MODE_FUSED,NT,PF,VL=0,CGP=1> (single-read J'(Jp), non-temporal loads; "
                       "rocprofv3 name: row_stream_kernel<256, 8, 4, 2, 1, 1, 0, 1>)") if headline_transport != "rccl" else
                      ("row_stream_kernel<256,8,4,MODE_FUSED,NT,PF,VL=0,CGP=0> (single-read J'(Jp), non-temporal loads; three-kernel CG "
                       "iteration around the RCCL all-reduce; rocprofv3 name: row_stream_kernel<256, 8, 4, 2, 1, 1, 0, 0>)"),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "frac_of_guide_measured_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": st["bytes_per_hmul"], "avg_launch_ms": hmul_ms,
            "launches_timed": st["hmul_timed"], "timed_region_event_stride": timed_stride, "dense_sample": dense, "rank": rank,
            "achieved_min_over_ranks": float(per_rank[:, 0].min()), "achieved_max_over_ranks": float(per_rank[:, 0].max()),
            "per_rank": [{"rank": r, "achieved": float(per_rank[r, 0]), "avg_launch_ms": float(per_rank[r, 1]),
                          "algorithmic_bytes_per_launch": float(per_rank[r, 2]), "rows": int(per_rank[r, 3])} for r in range(world)],
        },
    }
    if world > 1:
        # what the exchange costs: back-to-back all-reduces of one n-vector on the library stream (all ranks together)
        comm = {"mode": comm_mode, "timed_region_path": headline_transport, "note": comm_note}
        # rehearsal only: one rank skips the peer-path timing, the others run into the exchange's timeout — the failure this
        # section has to survive without leaving a rank behind in a collective
        fault_rank = int(os.environ.get("BH_BENCH_FAULT_RANK", "-1")) if rehearsal else -1

        def timed_exchange(kind, use_peer=False):
            """Max over ranks of bh_time_kernel(kind) in us, or None (on every rank alike) when any rank's call failed."""
            try:
                if fault_rank == rank and use_peer:
                    raise bh.BenlsipHipError(-1, "bench rehearsal", "this rank stays away from the peer exchange (BH_BENCH_FAULT_RANK)")
                v = 1e3 * H.time_kernel(kind, 200)
            except bh.BenlsipHipError as e:
                comm.setdefault("error", str(e))
                v = float("nan")
            g = gather([v])
            if np.isnan(g).any():
                comm.setdefault("error", "bh_time_kernel(%d) failed on rank(s) %s" % (kind, np.flatnonzero(np.isnan(g[:, 0])).tolist()))
                return None
            return float(g.max())

        if comm_mode in ("rccl", "both"):
            bh.set_option("comm_path", 0)
            comm["rccl_allreduce_us"] = timed_exchange(7)
            comm["rccl_reduce_plus_allreduce_us"] = timed_exchange(8)
        if comm_mode in ("ipc", "both"):
            # every decision below is taken from gathered values, so all ranks walk through the same collectives even when the
            # peer exchange fails on some of them (it then times out, raises there and is reported as comm["error"])
            try:
                bh.set_option("comm_path", 1)
                ok = 1.0
            except bh.BenlsipHipError as e:          # (a peer path that timed out earlier may not be selected again)
                comm.setdefault("error", str(e))
                ok = 0.0
            if gather([ok])[:, 0].min() > 0:
                comm["peer_allreduce_us"] = timed_exchange(7, True)
                comm["peer_reduce_plus_allreduce_us"] = timed_exchange(8, True) if comm["peer_allreduce_us"] is not None else None
            if comm_mode == "both":
                try:
                    bh.set_option("comm_path", 0)   # back to RCCL for the rest of the run (also clears a timed-out peer path)
                except bh.BenlsipHipError as e:
                    comm.setdefault("error", str(e))
        order = ("peer_allreduce_us", "rccl_allreduce_us") if headline_transport == "peer_buffers" else ("rccl_allreduce_us", "peer_allreduce_us")
        comm["allreduce_us"] = next((comm[k] for k in order if comm.get(k) is not None), None)
        line["comm"] = comm
        line["allreduce_us"] = comm["allreduce_us"]
    # (a peer-buffer-only run whose exchange failed has no working all-reduce left: the line is printed without the extras)
    if not args.no_extras and not (world > 1 and comm_mode == "ipc" and line["comm"].get("error")):
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
        line["post_stream_us"] = 1e3 * H.time_kernel(8, 200) if world == 1 else None    # slab reduction after the streaming kernel
    if world == 1 and not args.no_extras and not args.no_ic_extra and args.variant == "wc":
        # steady-state CG iteration cost on the ill-conditioned variant (23 iterations per subproblem): outside the timed region
        # (the first handle stays allocated: freeing 2 GiB here makes the driver scrub it in the background, which took
        # ~5 % of the HBM bandwidth from the next ~60 ms of kernels — tools/ic_transient.py)
        H2, cons2, dv2, _ = setup_instance(bh, rank, world, 1, config=args.config)
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
        line["cpu_baseline"] = cpu_baseline(kind, kappa2, n_hmul, config=args.config)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line), flush=True)
    barrier_noexcept()
    if dist is not None:
        H.close()                                   # handles first: the communicator cannot change under a live handle
        bh._lib.check(bh._lib.lib().bh_comm_destroy(), "bh_comm_destroy")
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
