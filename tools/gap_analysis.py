#!/usr/bin/env python3
"""Per-kernel durations and the idle gaps BETWEEN consecutive kernels from a rocprofv3 kernel_trace.csv.

    python tools/gap_analysis.py <kernel_trace.csv>

Prints, for the steady part of the trace (the last 60 % of the dispatches), the mean duration per kernel name, the mean
gap that FOLLOWS each kernel name, and the busy / idle split of the span."""
import csv
import sys
from collections import defaultdict


def short(name):
    for key in ("row_stream_kernel", "reduce_partials_kernel", "cg_reduce_update_kernel", "reduce_exchange_kernel", "cg_step_reg_kernel", "cg_step_kernel", "cg_init", "proj_", "trsv", "chol",
                "gram", "linesearch", "step_bounds", "cauchy", "synth_fill", "reduce_scalar", "weighted_sqsum"):
        if key in name:
            if key == "row_stream_kernel":
                mode = name.split("row_stream_kernel<")[1].split(">")[0].replace(" ", "")
                return "row_stream<%s>" % mode
            return name.split("(")[0].replace("void ", "").replace("bh::", "")[:60]
    return name[:60]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[int(0.4 * len(rows)):]
    dur, gap_after, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    busy = idle = 0
    for i, (s, e, name) in enumerate(rows):
        k = short(name)
        dur[k] += e - s
        cnt[k] += 1
        busy += e - s
        if i + 1 < len(rows):
            g = rows[i + 1][0] - e
            if g < 200000:              # ignore host-side pauses between subproblems (> 0.2 ms)
                gap_after[k] += g
                idle += max(g, 0)
    print("%-64s %8s %12s %14s" % ("kernel", "count", "mean us", "mean gap after us"))
    for k in sorted(dur, key=lambda k: -dur[k]):
        print("%-64s %8d %12.2f %14.2f" % (k, cnt[k], dur[k] / cnt[k] / 1e3, gap_after[k] / cnt[k] / 1e3))
    print("busy %.3f ms, idle (gaps < 0.2 ms) %.3f ms, idle share %.2f %%" % (busy / 1e6, idle / 1e6, 100.0 * idle / (busy + idle)))


if __name__ == "__main__":
    main()
