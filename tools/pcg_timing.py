#!/usr/bin/env python3
"""A/B timing of bh_pcg_dev under library options (interleaved rounds in one process)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402
import bench  # noqa: E402


def main():
    bh.init(0)
    lib = bh._lib.lib()
    for variant, kind, steps in (("wc", 0, 40), ("ic", 1, 6)):
        H, cons, dv, host = bench.setup_instance(bh, 0, 1, kind)
        res = {}
        for rnd in range(3):
            for prof in (0, 1):
                for pp in (0, 1):
                    lib.bh_set_option(b"profile", prof)
                    lib.bh_set_option(b"pingpong", pp)
                    bench.run_steps(bh, H, cons, dv, 0.1, 2)
                    lib.bh_synchronize()
                    t0 = time.perf_counter()
                    st, it, nh = bench.run_steps(bh, H, cons, dv, 0.1, steps)
                    lib.bh_synchronize()
                    el = (time.perf_counter() - t0) / steps
                    res.setdefault((prof, pp), []).append(el)
        for key, v in sorted(res.items()):
            print(variant, "profile=%d pingpong=%d" % key, "ms/subproblem min %.4f med %.4f" % (1e3 * min(v), 1e3 * sorted(v)[1]),
                  "per-iter us %.1f" % (1e6 * min(v) / nh), "n_hmul", nh, flush=True)
        H.close()


if __name__ == "__main__":
    main()
