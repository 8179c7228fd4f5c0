"""Manual soak run (not collected by pytest): `python tests/soak_gpu.py [pcg seeds] [caller seeds]` on the GPU box — 40
projected_cg fuzz cases per seed against the C oracle, 25 cauchy_step + minor_iterate cases per caller seed against the
NumPy oracle, then 6000 repeated config-3 subproblems that must return bit-identical results."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import benlsip_jl_amd as bh
import test_fuzz_gpu as F
bh.init(0)
t0 = time.time(); bad = 0
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 80          # python tests/soak_gpu.py [pcg seeds] [caller seeds]
n_caller = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for seed in range(100, 100 + n_seeds):
    try:
        F.test_fuzz_projected_cg_against_c_oracle(bh, seed)
    except AssertionError as e:
        bad += 1; print("seed", seed, "FAILED", str(e)[:400], flush=True)
    if seed % 50 == 0: print("... pcg fuzz seed", seed, flush=True)
print("fuzz soak: %d seeds x 40 projected_cg cases, failures:" % n_seeds, bad, "in %.1f s" % (time.time() - t0), flush=True)
t0 = time.time(); bad2 = 0
for seed in range(100, 100 + n_caller):
    try:
        F.test_fuzz_cauchy_step_and_minor_iterate(bh, seed)
    except AssertionError as e:
        bad2 += 1; print("caller seed", seed, "FAILED", str(e)[:400], flush=True)
    if seed % 20 == 0: print("... caller fuzz seed", seed, flush=True)
print("fuzz soak: %d seeds x 25 cauchy_step + minor_iterate cases, failures:" % n_caller, bad2, "in %.1f s" % (time.time() - t0), flush=True)
# repeated full-size calls: every call must return the same bits
import bench
H, cons, dv, host = bench.setup_instance(bh, 0, 1, 0)
bench.run_steps(bh, H, cons, dv, 0.1, 3)
ref = dv["w"].download()
n_diff = 0
t0 = time.time()
for i in range(6000):
    st, it, nh = bench.run_steps(bh, H, cons, dv, 0.1, 1)
    if i % 500 == 0 and not np.array_equal(dv["w"].download(), ref): n_diff += 1
    if (st.value, it, nh) != (0, 3, 2): n_diff += 1
print("6000 config-3 subproblems: mismatches", n_diff, "in %.1f s (%.1f/s)" % (time.time() - t0, 6000 / (time.time() - t0)), flush=True)
