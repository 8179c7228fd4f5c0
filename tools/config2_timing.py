"""BASELINE config 2 (m = 8192, n = 1024 fp64, box bounds) on one GPU: what a projected_cg subproblem and one CG iteration cost
at a size where J (64 MiB) is re-read from the last-level cache and the iteration is launch- and latency-bound rather than
HBM-bound.  Prints per-subproblem and per-iteration times for the well- and ill-conditioned variants and the kernel times of the
stand-alone products.  Usage: python tools/config2_timing.py [d n]   (measurement tool; never imports the oracle)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    import benlsip_jl_amd as bh
    import bench
    bh.init(0, flags=bh._lib.BH_FLAG_PROFILE)
    for kind, name in ((0, "wc"), (1, "ic")):
        for kappa2 in (0.1, 1e-3):
            H, cons, dv, _ = bench.setup_instance(bh, 0, 1, kind, d_per_gpu=d, n=n)
            bench.run_steps(bh, H, cons, dv, kappa2, 50)
            bh._lib.lib().bh_synchronize()
            reps = 200
            t0 = time.perf_counter()
            st, it, nh = bench.run_steps(bh, H, cons, dv, kappa2, reps)
            bh._lib.lib().bh_synchronize()
            el = (time.perf_counter() - t0) / reps
            byt = H.stats()["bytes_per_hmul"]
            print("%s kappa2 = %-6g %-22s %3d H*p: %8.1f us per subproblem, %6.2f us per H*p, %6.0f GB/s per iteration"
                  % (name, kappa2, st.name, nh, 1e6 * el, 1e6 * el / max(nh, 1), byt / (el / max(nh, 1)) / 1e9), flush=True)
            if kind == 0 and kappa2 == 0.1:
                for k, label in ((0, "fused J'(W.(Jp))"), (1, "J v"), (2, "J'u"), (8, "slab reduction")):
                    print("    kernel %-18s %7.2f us" % (label, 1e3 * H.time_kernel(k, 200)))
            H.close()
            cons.close()


if __name__ == "__main__":
    main()
