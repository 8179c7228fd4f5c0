#!/usr/bin/env python3
"""Per-subproblem times right after a handle is created (is there a warm-up transient, and what does it follow?)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402
import bench  # noqa: E402


def series(lib, H, cons, dv, count, group):
    out = []
    for i in range(count):
        lib.bh_synchronize()
        t0 = time.perf_counter()
        st, it, nh = bench.run_steps(bh, H, cons, dv, 0.1, group)
        lib.bh_synchronize()
        out.append(1e3 * (time.perf_counter() - t0) / group)
    return out, nh


def main():
    prof = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    bh.init(0, flags=bh._lib.BH_FLAG_PROFILE if prof else 0)
    lib = bh._lib.lib()
    H, cons, dv, host = bench.setup_instance(bh, 0, 1, 0)
    ts, nh = series(lib, H, cons, dv, 16, 20)
    print("profile=%d wc fresh handle, ms per subproblem in groups of 20:" % prof, " ".join("%.4f" % t for t in ts), flush=True)
    time.sleep(0.5)
    ts, nh = series(lib, H, cons, dv, 8, 20)
    print("wc after 0.5 s idle:", " ".join("%.4f" % t for t in ts), flush=True)
    H.close()
    H2, cons2, dv2, _ = bench.setup_instance(bh, 0, 1, 1)
    ts, nh = series(lib, H2, cons2, dv2, 14, 1)
    print("ic fresh handle:", " ".join("%.3f" % t for t in ts), "n_hmul", nh, flush=True)
    time.sleep(0.5)
    ts, nh = series(lib, H2, cons2, dv2, 14, 1)
    print("ic after 0.5 s idle:", " ".join("%.3f" % t for t in ts), flush=True)
    lib.bh_set_option(b"pcg_batch", 4)
    H2.reset_stats()
    ts, nh = series(lib, H2, cons2, dv2, 14, 1)
    print("ic after stats reset:", " ".join("%.3f" % t for t in ts), flush=True)


if __name__ == "__main__":
    main()
