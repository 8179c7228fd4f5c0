#!/usr/bin/env python3
"""Time bh_proj_set_active (Gram + Cholesky on the device) with the MFMA and the VALU Gram kernels (run under rocprofv3)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh  # noqa: E402


def main():
    bh.init(0)
    lib = bh._lib.lib()
    syn = bh.synthetic
    for mA, n in ((64, 4096), (256, 4096)):
        A = syn.splitmix_uniform(4, np.arange(mA * n)).reshape((mA, n), order="F")
        for flag in (1, 0):
            lib.bh_set_option(b"gram_mfma", flag)
            cons = bh.MixedConstraints(A, None, None)
            for k in range(20):
                fix = np.zeros(n, dtype=bool)
                fix[k::8] = True
                cons.set_active(fix, None)
                bh.projection(cons, np.ones(n))
    lib.bh_set_option(b"gram_mfma", 1)


if __name__ == "__main__":
    main()
