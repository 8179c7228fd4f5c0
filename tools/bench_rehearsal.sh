#!/bin/bash
# GPU box (one GPU): rehearse bench.py's N > 1 flow — all ranks on device 0, gloo for torch, the library's all-reduce over its
# peer-buffer transport (hipIpc between processes on one device).  Exercises the script's multi-rank flow; not a measurement.
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
export BH_BENCH_REHEARSAL=1 BH_COMM=ipc
rc=0
for cfg in "2 weak" "3 strong"; do
    set -- $cfg
    timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 2953$1 \
        bench.py --gpus $1 --steps 20 --warmup 2 --scaling $2 > gpurun_out/bench_rehearsal_$1_$2.log 2>&1 || rc=$?
    tail -1 gpurun_out/bench_rehearsal_$1_$2.log | cut -c1-1800
    [ $rc -eq 0 ] || { tail -30 gpurun_out/bench_rehearsal_$1_$2.log; exit $rc; }
done
exit $rc
