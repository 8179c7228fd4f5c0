#!/bin/bash
# GPU box (one GPU): rehearse bench.py's N = 2 flow — two ranks on device 0, gloo for torch, host-staged all-reduce.
R=$GRAFT_REPO_ROOT
cd $R
g++ -O2 -fPIC -shared -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/multirank/staged_rccl.cpp \
    -o tests/multirank/libstaged_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib || exit 1
export BH_BENCH_REHEARSAL=1 BH_RCCL_LIB=$R/tests/multirank/libstaged_rccl.so BH_STAGED_RCCL_SHM=/bh_rehearsal_$$
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 20 --warmup 2 > gpurun_out/bench_rehearsal.log 2>&1
rc=$?
rm -f /dev/shm$BH_STAGED_RCCL_SHM
tail -3 gpurun_out/bench_rehearsal.log | cut -c1-1500
exit $rc
