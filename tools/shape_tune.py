#!/usr/bin/env python3
"""blocks-per-CU sweep for the smaller row widths."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402


def main():
    bh.init(0)
    lib = bh._lib.lib()
    for d, n in [(2097152, 128), (524288, 512), (262144, 1024), (131072, 2048), (65536, 4096)]:
        H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
        for bpc in (1, 2, 3, 4, 6, 8):
            lib.bh_set_option(b"blocks_per_cu", bpc)
            ms = [min(H.time_kernel(k, 10) for _ in range(2)) for k in (0, 1, 2)]
            gb = [8.0 * d * n / (m * 1e-3) / 1e9 for m in ms]
            print("n=%5d bpc=%d | fused %5.0f  Jv %5.0f  J'u %5.0f GB/s" % (n, bpc, gb[0], gb[1], gb[2]), flush=True)
        H.close()
    lib.bh_set_option(b"blocks_per_cu", 0)


if __name__ == "__main__":
    main()
