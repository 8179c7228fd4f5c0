#!/bin/bash
# scratch: kernel stats of the row-space Cauchy search with equalities (config-3 scale), mA = 64 and 8
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in 64 8; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cg$m -o cg -- python3 $R/tools/scratch/cauchy_gen_trace.py $m > $R/gpurun_out/cg_trace_$m.log 2>&1 || exit 1
  f=$(find /tmp/prof_cg$m -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || { echo "no stats file"; exit 1; }
  cp "$f" $R/gpurun_out/cg${m}_kernel_stats.csv
  grep passes $R/gpurun_out/cg_trace_$m.log
  cut -c1-200 "$f" | sed -n 1,14p
done
