#!/bin/bash
# GPU box: matrix-core evidence for image_b_mfma_kernel (B = J (D A') of the row-space Cauchy search: the one tile of J that meets
# a dense GEMM): kernel stats and MFMA / busy counters of tools/image_b_timing.py (PMC pass separate from --stats).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/image_b
rm -rf $OUT && mkdir -p $OUT
python3 $R/tools/image_b_timing.py > $OUT/plain.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/image_b_timing.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -- python3 $R/tools/image_b_timing.py > $OUT/pmc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/tools/image_b_timing.py > $OUT/fetch.log 2>&1
cat $OUT/plain.log
python3 - <<PY
import csv, glob, collections
for kind in ("pmc", "fetch"):
    f = glob.glob("$OUT/%s/*/*_counter_collection.csv" % kind)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "image_b" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(k, {n: sum(v) / len(v) for n, v in c.items()}, "dispatches", len(next(iter(c.values()))))
s = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(s)):
    if "image_b" in r["Name"] or "row_stream_kernel<256, 8, 4, 0" in r["Name"]:
        print(r["Name"].split("(")[0], r["Calls"], r["AverageNs"])
PY
rm -rf $OUT/trace $OUT/pmc $OUT/fetch
