#!/bin/bash
# GPU box (scratch copy of the repo): interleaved A/B of the library before / after the latency work on the two-kernel
# iteration (tools/ab/libbenlsip_hip_prev.so = commit bfca8e9 built by benlsip.jl_amd/build.py; not tracked).
R=$GRAFT_REPO_ROOT
cd $R
cp benlsip.jl_amd/lib/libbenlsip_hip.so /tmp/cur.so
for i in 1 2 3; do
  for which in prev cur; do
    if [ $which = prev ]; then cp tools/ab/libbenlsip_hip_prev.so benlsip.jl_amd/lib/libbenlsip_hip.so; else cp /tmp/cur.so benlsip.jl_amd/lib/libbenlsip_hip.so; fi
    python bench.py --steps 200 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$which wc  %.1f us per subproblem (%.1f/s)' % (1e3*l['ms_per_step'], l['value']))"
    python bench.py --variant ic --steps 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$which ic  %.2f us per CG iteration' % (1e3*l['ms_per_cg_iteration']))"
  done
done
cp /tmp/cur.so benlsip.jl_amd/lib/libbenlsip_hip.so
