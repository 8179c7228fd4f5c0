#!/bin/bash
# scratch: kernel stats of the box-constrained row-space Cauchy search (config-3 scale)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cb -o cb -- python3 $R/tools/scratch/cauchy_box_trace.py > $R/gpurun_out/cb_trace.log 2>&1 || exit 1
f=$(find /tmp/prof_cb -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no stats file"; ls -R /tmp/prof_cb | head -20; exit 1; }
cp "$f" $R/gpurun_out/cb_kernel_stats.csv
cut -c1-220 "$f" | sed -n 1,12p
