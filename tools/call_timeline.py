#!/usr/bin/env python3
"""Timeline of the LAST subproblems in a rocprofv3 kernel_trace.csv: every dispatch with its start relative to the first
kernel of its call, its duration and the idle gap in front of it.  A "call" starts after a gap > --split us (the host is
between two library calls).

    python tools/call_timeline.py <kernel_trace.csv> [--calls 2] [--split 30]
"""
import argparse
import csv


def short(name):
    name = name.replace("void ", "").replace("bh::", "")
    base = name.split("(")[0]
    return base[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--calls", type=int, default=2)
    ap.add_argument("--split", type=float, default=30.0)
    ap.add_argument("--skip-last", type=int, default=0, help="ignore this many calls at the end of the trace")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    calls, cur = [], []
    for i, r in enumerate(rows):
        if cur and (r[0] - cur[-1][1]) > a.split * 1e3:
            calls.append(cur)
            cur = []
        cur.append(r)
    if cur:
        calls.append(cur)
    if a.skip_last:
        calls = calls[:-a.skip_last]
    for call in calls[-a.calls:]:
        t0 = call[0][0]
        print("call of %d dispatches, first start to last end %.2f us" % (len(call), (call[-1][1] - t0) / 1e3))
        prev_end = None
        for s, e, name in call:
            gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
            print("  +%9.2f us  dur %8.2f  gap %6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, short(name)))
            prev_end = e


if __name__ == "__main__":
    main()
