#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (one line per kernel)."""
import re
import subprocess
import sys


def main(path):
    log = open(path).read()
    blocks = re.split(r"remark: Function Name: ", log)[1:]
    for b in blocks:
        name = b.split()[0]

        def g(key):
            m = re.search(key + r": (\d+)", b)
            return m.group(1) if m else "?"

        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = dn.replace("bh::", "").split("(")[0][:72]
        vg, ag, sp = g("VGPRs"), g("AGPRs"), g("VGPRs Spill")
        sc, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print("%-74s VGPR %4s AGPR %3s spill %3s scratch %4s occ %s LDS %s" % (dn, vg, ag, sp, sc, occ, lds))


if __name__ == "__main__":
    main(sys.argv[1])
