#!/bin/bash
# GPU box: kernel trace of a few ill-conditioned subproblems (23 CG iterations each) for tools/gap_analysis.py.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/gap
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --variant ic --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $OUT/run.log 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/gap_analysis.py $f > $OUT/gaps.txt
cat $OUT/gaps.txt
