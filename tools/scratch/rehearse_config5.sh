#!/bin/bash
# scratch: bench.py --config 5 with 2 ranks on ONE GPU (rehearsal): peer buffers, then both transports (staged RCCL stand-in)
R=$GRAFT_REPO_ROOT
cd $R
export BH_BENCH_REHEARSAL=1
export BH_COMM=ipc
timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --config 5 > gpurun_out/rehearse_c5_ipc.log 2>&1 || { tail -20 gpurun_out/rehearse_c5_ipc.log; exit 1; }
tail -1 gpurun_out/rehearse_c5_ipc.log | cut -c1-300
g++ -O2 -fPIC -shared -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/multirank/staged_rccl.cpp -o tests/multirank/libstaged_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib || exit 1
export BH_COMM=both BH_RCCL_LIB=$R/tests/multirank/libstaged_rccl.so BH_STAGED_RCCL_SHM=/bh_rehearsal_c5_$$
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 2 --config 5 > gpurun_out/rehearse_c5_both.log 2>&1 || { tail -20 gpurun_out/rehearse_c5_both.log; rm -f /dev/shm/bh_rehearsal_c5_$$; exit 1; }
rm -f /dev/shm/bh_rehearsal_c5_$$
grep '^{' gpurun_out/rehearse_c5_both.log | tail -1 | cut -c1-300
grep -o '"replicas_bitwise_identical": [a-z]*' gpurun_out/rehearse_c5_ipc.log gpurun_out/rehearse_c5_both.log
grep -o '"comm": {[^}]*}' gpurun_out/rehearse_c5_both.log | cut -c1-500
