#!/usr/bin/env python3
"""Achieved GB/s of the three row-stream kernels over a sweep of shapes (J = 2 GiB unless noted)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402


def main():
    bh.init(0)
    shapes = [(2097152, 128), (524288, 512), (262144, 1024), (131072, 2048), (87381, 3000), (65536, 4096), (43690, 6000), (32768, 8192)]
    print("%9s %6s | %8s %8s %8s   (GB/s, algorithmic 8*d*n bytes / hipEvent time)" % ("d", "n", "fused", "J v", "J'u"))
    for d, n in shapes:
        H = bh.AlHessian.synthetic(d, n, seed=1, mu=10.0)
        ms = [H.time_kernel(k, 10) for k in (0, 1, 2)]
        gb = [8.0 * d * n / (m * 1e-3) / 1e9 for m in ms]
        print("%9d %6d | %8.0f %8.0f %8.0f   ms %s" % (d, n, gb[0], gb[1], gb[2], ["%.3f" % m for m in ms]), flush=True)
        H.close()


if __name__ == "__main__":
    main()
