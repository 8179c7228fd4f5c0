#!/bin/bash
# GPU box: PMC evidence for the n <= 16384 single-read variant (v parked in LDS): HBM traffic vs algorithmic bytes and LDS bank
# conflicts of row_stream_kernel<512,16,1,...> (tools/wide_timing.py; separate --pmc passes).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/widepmc
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/tools/wide_timing.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/lds -- python3 $R/tools/wide_timing.py > $OUT/lds.log 2>&1
python3 - <<PY
import csv, glob, collections
for kind in ("fetch", "lds"):
    f = glob.glob("$OUT/%s/*/*_counter_collection.csv" % kind)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "row_stream_kernel<512, 16, 1" in k:
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        for n, v in c.items():
            vals = sorted(set(round(x) for x in v))
            print(k, n, "dispatches", len(v), "max", max(v), "distinct", vals[:6])
PY
