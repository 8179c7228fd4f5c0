#!/bin/bash
# GPU box: per-kernel split of the projection at mA = 64..512 (tools/proj_timing.py under rocprofv3 --kernel-trace).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/projtrace
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/proj_timing.py > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/*/*_kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void bh::","").replace("bh::","")) for r in csv.DictReader(open(f))))
# split the trace into the four mA phases by the chol/gram launches that precede each phase
phase, acc = -1, collections.defaultdict(lambda: collections.defaultdict(list))
for s, e, k in rows:
    if k.startswith("gram_free"): phase += 1
    if phase >= 0 and (k.startswith("proj_") or k.startswith("trsv")):
        acc[phase][k].append((e - s) / 1e3)
for ph in sorted(acc):
    print("phase", ph, {k: round(sorted(v)[len(v)//2], 2) for k, v in acc[ph].items()}, "launches", {k: len(v) for k, v in acc[ph].items()})
PY
