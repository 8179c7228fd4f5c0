"""The C ABI from plain C (examples/pcg_demo.c — what Julia's ccall does): the header compiles as C and the program
links against the shared library (CPU check); on a GPU it runs a small subproblem and is checked against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import benlsip_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    import benlsip_jl_amd as bh
    bh.load()
    libdir = os.path.dirname(bh.library_path())
    exe = str(tmp_path / "pcg_demo")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "pcg_demo.c"), "-o", exe, "-L" + libdir, "-lbenlsip_hip",
                           "-Wl,-rpath," + libdir, "-lm"])
    return exe


def test_header_is_valid_c_and_demo_links(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 1 and "usage" in out.stderr


@pytest.mark.gpu
def test_c_demo_matches_oracle(tmp_path):
    exe = _build(tmp_path)
    rng = np.random.default_rng(77)
    d, n, q, mA, nfix = 300, 96, 1, 3, 10
    J = rng.standard_normal((d, n)) / np.sqrt(d)
    C = rng.standard_normal((q, n))
    A = rng.standard_normal((mA, n))
    fix = np.zeros(n, dtype=bool)
    fix[rng.choice(n, nfix, replace=False)] = True
    cons = R.make_mixed_constraints(A, R.chol_lower(A @ A.T), fix, l=-np.ones(n), u=np.ones(n))
    L = cons.chol_L
    g = rng.standard_normal(n)
    w_l, w_u = R.build_step_bounds(np.where(fix, 1.0, 0.0), cons, 0.5)
    mu, kappa2 = 10.0, 0.1
    prob = tmp_path / "problem.bin"
    with open(prob, "wb") as f:
        f.write(struct.pack("<5q2d", d, n, q, mA, L.shape[0], mu, kappa2))
        for arr in (J, C, A, L):
            f.write(np.asfortranarray(arr).tobytes(order="F"))
        for vec in (fix.astype(np.float64), g, w_l, w_u):
            f.write(np.ascontiguousarray(vec).tobytes())
    res = tmp_path / "result.bin"
    out = subprocess.run([exe, str(prob), str(res)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    raw = open(res, "rb").read()
    status, iters, n_hmul = struct.unpack("<3q", raw[:24])
    vals = np.frombuffer(raw[24:], dtype=np.float64)
    w, Hg, v, gHg = vals[:n], vals[n:2 * n], vals[2 * n:3 * n], vals[3 * n]
    H = R.AlHessian(J, C, mu)
    w_ref, s_ref, it_ref = R.projected_cg(g, H, w_l, w_u, cons, kappa2)
    assert (status, iters) == (int(s_ref), it_ref)
    assert np.linalg.norm(w - w_ref) <= 1e-8 * np.linalg.norm(w_ref)
    assert np.linalg.norm(Hg - R.hmul(H, g)) <= 1e-12 * np.linalg.norm(R.hmul(H, g))
    assert np.linalg.norm(v - R.projection(cons, g)) <= 1e-11 * np.linalg.norm(g)
    assert gHg == pytest.approx(R.vthv(H, g), rel=1e-12)
