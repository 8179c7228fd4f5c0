"""Scratch: config 2 (8192 x 1024) per-iteration time against the workgroups-per-CU of the row-stream kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import benlsip_jl_amd as bh
import bench
bh.init(0)
for bpc in (0, 1, 2, 4, 8):
    bh.set_option("blocks_per_cu", bpc)
    H, cons, dv, _ = bench.setup_instance(bh, 0, 1, 1, d_per_gpu=8192, n=1024)
    bench.run_steps(bh, H, cons, dv, 1e-3, 5)
    bh._lib.lib().bh_synchronize()
    t0 = time.perf_counter()
    st, it, nh = bench.run_steps(bh, H, cons, dv, 1e-3, 20)
    bh._lib.lib().bh_synchronize()
    el = (time.perf_counter() - t0) / 20
    print("blocks_per_cu %d: %d H*p, %.2f us per H*p; fused kernel alone %.2f us" % (bpc, nh, 1e6 * el / nh, 1e3 * H.time_kernel(0, 200)), flush=True)
    H.close(); cons.close()
bh.set_option("blocks_per_cu", 0)
