#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 output of tools/profile_round.sh) into profiles/<tag>_*.
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts exactly half of the bytes of a 16-B/lane streaming read
(/opt/skills/guides/MI355X_MICROARCH.md §HBM), so HBM read bytes = 2 * FETCH_SIZE * 1024."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
    line = os.path.join(src, "bench_line_under_trace.json")
    if os.path.exists(line):
        shutil.copy(line, os.path.join(dst, tag + "_bench_line_under_trace.json"))
    pmc = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = sorted(glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
        if not files:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for name, vals in acc.items():
            # a launch enqueued past the CG exit sees state->done and returns at once (a gated no-op, ~0 bytes): it is a
            # dispatch of the same kernel symbol but not a launch of the operation, so it must not dilute the average
            # (the two-kernel CG iteration adds a second kind: a launch whose prologue finds the loop finished after it has
            # prefetched one row group — 32 MiB instead of 2 GiB; rare, on a fresh handle without an iteration-count hint)
            live = [v for v in vals if v >= 0.5 * max(vals)] if max(vals) > 0 else vals
            pmc.setdefault(name, {})[counter + "_KiB_avg"] = sum(live) / len(live)
            pmc[name][counter + "_dispatches"] = len(live)
            pmc[name][counter + "_gated_dispatches_excluded"] = len(vals) - len(live)
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) + --kernel-trace, bench.py --steps 5",
           "correction": "HBM bytes per launch = 2*FETCH_SIZE*1024 (gfx950 half-count of 16-B/lane reads) + WRITE_SIZE*1024",
           "kernels": {}}
    for name, c in pmc.items():
        f, w = c.get("FETCH_SIZE_KiB_avg"), c.get("WRITE_SIZE_KiB_avg")
        entry = dict(c)
        if f is not None and w is not None:
            entry["hbm_bytes_per_launch"] = 2.0 * f * 1024.0 + w * 1024.0
        out["kernels"][name] = entry
    sq_files = sorted(glob.glob(os.path.join(src, "sq", "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
    if sq_files:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(sq_files[0])):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        sq = {}
        for name, cs in acc.items():
            e = {k + "_avg": sum(v) / len(v) for k, v in cs.items()}
            e["avg_duration_ns_under_pmc"] = sum(dur[name]) / len(dur[name])
            if "GRBM_GUI_ACTIVE_avg" in e and e["avg_duration_ns_under_pmc"] > 0:
                # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
                e["effective_clock_GHz"] = e["GRBM_GUI_ACTIVE_avg"] / 8.0 / e["avg_duration_ns_under_pmc"]
            sq[name] = e
        out["sq_counters"] = sq
    json.dump(out, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
    for name, e in out["kernels"].items():
        if "hbm_bytes_per_launch" in e:
            print("%-80s %.0f bytes/launch" % (name[:80], e["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
