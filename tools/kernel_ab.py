#!/usr/bin/env python3
"""Interleaved A/B timing of row-stream kernel variants at BASELINE config 3 (one process, several rounds)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402


def main():
    variants = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,5").split(",")]
    bh.init(0)
    lib = bh._lib.lib()
    d, n = 65536, 4096
    H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
    res = {}
    for rnd in range(5):
        for v in variants:
            lib.bh_set_option(b"rs_variant", v)
            for kind in (0, 1, 2):
                res.setdefault((v, kind), []).append(H.time_kernel(kind, 20))
    for (v, kind), ms in sorted(res.items()):
        ms = sorted(ms)
        print("variant %d kind %d  ms min %.4f med %.4f  -> %.0f GB/s (med)" % (v, kind, ms[0], ms[len(ms) // 2], 8.0 * d * n / ms[len(ms) // 2] / 1e6))


if __name__ == "__main__":
    main()
