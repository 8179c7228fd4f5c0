#!/bin/bash
# GPU box: rocprofv3 kernel stats of the two secondary workloads — BASELINE config 5 (tools/config5_timing.py: 64 linear
# equalities + 512 active bounds, default seven-kernel iteration) and the device-resident Cauchy search (tools/cauchy_timing.py).
# Usage: tools/secondary_stats.sh <tag>   ->  gpurun_out/sec_<tag>/{config5,cauchy}_kernel_stats.csv
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sec_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in config5 cauchy; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -- python3 $R/tools/${w}_timing.py > $OUT/$w.log 2>&1
    f=$(find $OUT/$w -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $OUT/${w}_kernel_stats.csv
    tail -3 $OUT/$w.log
done
ls $OUT
