"""Build the gfx950 shared library (C ABI of include/benlsip_hip.h) in-tree with hipcc.

    python benlsip.jl_amd/build.py [--force]

Output: benlsip.jl_amd/lib/libbenlsip_hip.so (git-ignored; it travels to the GPU box with the tree).
hipcc cross-compiles for gfx950 without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "bh_api.hip")
DEPS = [SRC, os.path.join(ROOT, "include", "benlsip_hip.h")] + [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc"))) if f.endswith(".h")]
OUT = os.path.join(HERE, "lib", "libbenlsip_hip.so")


def lib_path():
    return OUT


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in DEPS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-Wno-pass-failed", "-I/opt/rocm/include", SRC, "-o", OUT + ".tmp", "-ldl"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
