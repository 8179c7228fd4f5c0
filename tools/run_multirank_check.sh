#!/bin/bash
# GPU box: the two-process multi-rank test, then bench.py under torch.distributed.run with one rank (N>1 code path).
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_multirank_gpu.py -x -q > gpurun_out/mr.log 2>&1
rc=$?
tail -30 gpurun_out/mr.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 50 --no-cpu-baseline > gpurun_out/bench_tr.log 2>&1
rc=$?
tail -2 gpurun_out/bench_tr.log
exit $rc
