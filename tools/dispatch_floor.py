#!/usr/bin/env python3
"""What does one gated (no-op) kernel launch cost on the stream, without a profiler attached?

The ill-conditioned instance exits after 23 H*p; with pcg_batch = B the host has enqueued up to 2B further iterations
(3 kernels each) that see state->done and return.  Time per subproblem vs B gives the per-kernel dispatch floor."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402
import bench  # noqa: E402


def main():
    bh.init(0)
    lib = bh._lib.lib()
    lib.bh_set_option(b"profile", 0)
    H, cons, dv, host = bench.setup_instance(bh, 0, 1, 1)
    rows = []
    for rnd in range(3):
        for batch in (1, 2, 4, 16, 64):
            lib.bh_set_option(b"pcg_batch", batch)
            bench.run_steps(bh, H, cons, dv, 0.1, 2)
            lib.bh_synchronize()
            t0 = time.perf_counter()
            st, it, nh = bench.run_steps(bh, H, cons, dv, 0.1, 6)
            lib.bh_synchronize()
            rows.append((batch, (time.perf_counter() - t0) / 6, nh))
    best = {}
    for b, t, nh in rows:
        best[b] = min(best.get(b, 1e9), t)
    for b in sorted(best):
        print("pcg_batch %3d: %.4f ms per subproblem (%d H*p), %.1f us per CG iteration" % (b, 1e3 * best[b], nh, 1e6 * best[b] / nh), flush=True)
    if 64 in best and 4 in best:
        # launched iterations: first batch 8, then batches of B with one batch of launch-ahead
        def launched(b):
            n = 8
            n += b
            while True:
                target = n
                n += b
                if target >= 23:
                    break
            return n
        extra = (launched(64) - launched(4)) * 3
        print("no-op kernels: %d more at batch 64 than at batch 4 -> %.2f us per gated kernel" % (extra, 1e6 * (best[64] - best[4]) / extra))


if __name__ == "__main__":
    main()
