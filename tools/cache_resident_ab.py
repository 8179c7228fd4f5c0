#!/usr/bin/env python3
"""Non-temporal vs default-policy loads of J when J fits the 256 MiB Infinity Cache (n = 4096, small d): does keeping J
cache-resident between launches beat streaming it from HBM every time?  (rs_variant 0 = nt, 5 = default policy.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402


def main():
    bh.init(0)
    lib = bh._lib.lib()
    for d in (512, 1024, 2048, 4096, 8192, 16384, 65536):
        H = bh.AlHessian.synthetic(d, 4096, seed=1, mu=10.0)
        row = []
        for variant in (0, 5):
            lib.bh_set_option(b"rs_variant", variant)
            ms = min(H.time_kernel(0, 30) for _ in range(3))
            row.append((variant, ms))
        lib.bh_set_option(b"rs_variant", 0)
        mib = 8.0 * d * 4096 / 2**20
        print("d=%6d (J = %6.0f MiB): " % (d, mib) + "  ".join("variant %d: %7.2f us = %6.0f GB/s" % (v, 1e3 * ms, 8.0 * d * 4096 / ms / 1e6) for v, ms in row), flush=True)
        H.close()


if __name__ == "__main__":
    main()
