#!/bin/bash
# GPU box: kernel timeline of the config-5 CG iteration (64 linear equalities + 512 active bounds), the three iteration shapes
# (cg_fused = 1: three kernels per iteration, 2: four, 0: seven).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in ${C5_SHAPES:-1 2 0}; do
    OUT=$R/gpurun_out/c5trace_$f
    rm -rf $OUT && mkdir -p $OUT
    BH_CG_FUSED=$f rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/config5_timing.py > $OUT/run.log 2>&1
    t=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
    echo "== cg_fused=$f"
    python3 $R/tools/gap_analysis.py $t
    python3 $R/tools/call_timeline.py $t --calls 1 > $OUT/timeline.txt
done
