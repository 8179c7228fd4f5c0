#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the default bench, then two PMC passes
# (FETCH_SIZE and WRITE_SIZE need separate passes on gfx950: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2).
# Usage: tools/profile_round.sh <tag> [extra bench.py arguments, e.g. --config 5]      -> gpurun_out/prof_<tag>/{trace,fetch,write,sq}
set -e
TAG=${1:-r01}
shift || true
EXTRA="$@"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-ic-extra $EXTRA > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log > $OUT/bench_line_under_trace.json || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 --preheat 0 $EXTRA > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 --preheat 0 $EXTRA > $OUT/write.log 2>&1
# clock / stall picture of the same run (SQ and GRBM slots are independent of TCC)
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 --preheat 0 $EXTRA > $OUT/sq.log 2>&1 || true
find $OUT -name "*.csv" | head -20
