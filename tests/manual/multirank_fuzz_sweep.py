"""Manual sweep (not collected by pytest): the multi-rank fuzz of tests/test_multirank_gpu.py over many seeds, both transports, 2 and 3
processes.  python tests/manual/multirank_fuzz_sweep.py [first_seed] [count]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_multirank_gpu as T      # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for seed in range(first, first + count):
    for comm, world in (("ipc", 2), ("staged", 2), ("ipc", 3), ("staged", 3)):
        if (seed + world) % 2 and world == 3:
            continue                      # half of the seeds also with three processes
        with tempfile.TemporaryDirectory() as tmp:
            try:
                T.test_multirank_fuzz_of_the_caller_level_entry_points(tmp, comm, world, seed)
            except AssertionError as e:
                bad += 1
                print("seed %d %s x%d FAILED: %s" % (seed, comm, world, str(e)[:600]), flush=True)
print("multi-rank fuzz sweep: seeds %d..%d, failures: %d" % (first, first + count - 1, bad), flush=True)
sys.exit(1 if bad else 0)
