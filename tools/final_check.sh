#!/bin/bash
# GPU box, end of round: smoke(), refreshed rocprofv3 evidence, default bench line (with cpu_baseline), ic bench line.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
bash tools/profile_round.sh ${1:-r02} > gpurun_out/profile_round.log 2>&1 || { tail -5 gpurun_out/profile_round.log; exit 1; }
cd $R
timeout -k 10 300 python bench.py > gpurun_out/bench_wc.log 2>&1 || { tail -5 gpurun_out/bench_wc.log; exit 1; }
timeout -k 10 200 python bench.py --variant ic --no-cpu-baseline --no-extras > gpurun_out/bench_ic.log 2>&1 || exit 1
tail -1 gpurun_out/bench_wc.log | cut -c1-300
tail -1 gpurun_out/bench_ic.log | cut -c1-200
