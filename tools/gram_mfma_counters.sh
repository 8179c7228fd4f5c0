#!/bin/bash
# GPU box: matrix-core evidence for the one GEMM-shaped kernel on the path (A_free A_free' on fp64 MFMA, mA > 96):
# kernel stats and MFMA instruction / busy counters of tools/gram_timing.py (PMC pass separate from --stats).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/gram
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/gram_timing.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -- python3 $R/tools/gram_timing.py > $OUT/pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "gram" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: sum(v) / len(v) for n, v in c.items()}, "dispatches", len(next(iter(c.values()))))
s = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(s)):
    if "gram" in r["Name"] or "chol" in r["Name"]:
        print(r["Name"].split("(")[0], r["Calls"], r["AverageNs"])
PY
