#!/bin/bash
# GPU box (one GPU): rehearse bench.py's N > 1 flow — all ranks on device 0, gloo for torch.  Not a measurement.
#   1. N = 2 weak and N = 3 strong scaling over the library's peer-buffer transport (hipIpc between processes on one device);
#   2. N = 2 with BH_COMM=both — the configuration bench.py picks on a real multi-GPU node: the RCCL call site is bound to the
#      host-staged stand-in (RCCL itself refuses duplicate devices), the peer-buffer transport comes up next to it, the K timed
#      steps run on each, the valid and faster run becomes the headline, and the comm section times both all-reduces.
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
export BH_BENCH_REHEARSAL=1
rc=0
#   3. the same with rank 1 staying away from the peer exchange (BH_BENCH_FAULT_RANK): rank 0 runs into the exchange's timeout;
#      the line must still be printed, with comm.error set and no rank left behind in a collective.
#   4. BH_COMM=auto with the real RCCL: it refuses two ranks on one device, on every rank, so the script must walk
#      both -> rccl -> ipc together and finish on the peer buffers with the fall-backs recorded in comm.note.
for cfg in "2 weak ipc" "3 strong ipc" "2 weak both" "2 weak both fault" "2 weak auto"; do
    set -- $cfg
    export BH_COMM=$3
    unset BH_BENCH_FAULT_RANK BH_PEER_TIMEOUT_S BH_RCCL_LIB BH_STAGED_RCCL_SHM
    if [ "$4" = "fault" ]; then export BH_BENCH_FAULT_RANK=1 BH_PEER_TIMEOUT_S=3; fi
    if [ "$3" = "both" ]; then
        g++ -O2 -fPIC -shared -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/multirank/staged_rccl.cpp \
            -o tests/multirank/libstaged_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib || exit 1
        export BH_RCCL_LIB=$R/tests/multirank/libstaged_rccl.so BH_STAGED_RCCL_SHM=/bh_rehearsal_$$
    fi
    if [ "$3" = "ipc" ]; then
        # as typed, without a launcher: bench.py starts its ranks itself (child processes; the launcher never touches HIP)
        timeout -k 10 400 python bench.py --gpus $1 --steps 20 --warmup 2 --scaling $2 > gpurun_out/bench_rehearsal_$1_$2_$3$4.log 2>&1 || rc=$?
    else
        timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 2953$1 \
            bench.py --gpus $1 --steps 20 --warmup 2 --scaling $2 > gpurun_out/bench_rehearsal_$1_$2_$3$4.log 2>&1 || rc=$?
    fi
    rm -f /dev/shm/bh_rehearsal_$$
    tail -1 gpurun_out/bench_rehearsal_$1_$2_$3$4.log | cut -c1-1200
    if [ "$3" = "auto" ]; then grep -q 'fell back to BH_COMM=ipc' gpurun_out/bench_rehearsal_$1_$2_$3$4.log || { echo "auto: no fall-back note"; rc=1; }; fi
    grep -o '"comm": {[^}]*}[^}]*}' gpurun_out/bench_rehearsal_$1_$2_$3$4.log | cut -c1-900
    [ $rc -eq 0 ] || { tail -30 gpurun_out/bench_rehearsal_$1_$2_$3$4.log; exit $rc; }
done
exit $rc
