"""Experiment: where do the ~10 us per step between tools/percall_breakdown.py (200 calls) and bench.py (K = 20) go?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh
import bench
use_torch = len(sys.argv) > 1 and sys.argv[1] == "torch"
if use_torch:
    import torch
bh.init(0, flags=bh._lib.BH_FLAG_PROFILE if "profile" in sys.argv else 0)
H, cons, dv, _ = bench.setup_instance(bh, 0, 1, 0)
lib = bh._lib.lib()
def sync():
    lib.bh_synchronize()
    if use_torch:
        torch.cuda.synchronize()
bench.run_steps(bh, H, cons, dv, 0.1, 5)
for K in (20, 20, 20, 100, 20):
    sync()
    t0 = time.perf_counter()
    per = []
    for _ in range(K):
        t1 = time.perf_counter()
        bench.run_steps(bh, H, cons, dv, 0.1, 1)
        per.append(1e6 * (time.perf_counter() - t1))
    t_loop = time.perf_counter() - t0
    sync()
    el = time.perf_counter() - t0
    print("torch=%s K=%d: %.1f us per step incl. bracket, %.1f us loop only; end sync %.1f us; first steps %s" %
          (use_torch, K, 1e6 * el / K, 1e6 * t_loop / K, 1e6 * (el - t_loop), ["%.0f" % x for x in per[:6]]), flush=True)
for _ in range(3):
    t0 = time.perf_counter(); sync(); print("idle sync: %.1f us" % (1e6 * (time.perf_counter() - t0)))
