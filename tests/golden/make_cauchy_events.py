#!/usr/bin/env python3
"""Regenerate tests/golden/cauchy_events.json (needs the GPU: run through gpurun from the repo root).

    python tests/golden/make_cauchy_events.py

A shadow solve of the 48-parameter NLS instance (tests/nls_problem.py, seed 1; device = every hot-path and next-row call on
the GPU, oracle evaluated on the SAME operands at every call) records the Cauchy searches (src/basic_tralcnlss.jl:574-639)
whose device result differs from the oracle's: different final active set, or a step more than 1e-3 apart.  Late in the solve
the search direction P(-g) cancels ||g|| / ||P(-g)|| = 1e7 ... 1e9 of its digits, so the breakpoint sequence is decided by
rounding noise — in the ORACLE as well (tests/test_oracle_cpu.py::test_cauchy_search_is_multimodal_on_the_pinned_operands shows
the oracle alone landing on several active sets under 1-ulp perturbations of g).  The operands of the worst events are committed
so that this statement can be checked without re-running the solve: x, g, delta, the incoming active set and mu as exact hex
floats; J = jac_r(x) and C = jac_c(x) are regenerated from the seeded problem (verified here bit for bit against the matrices
the solve used).  Data only: inputs and the two results; no code of the reference.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, os.path.join(ROOT, "oracle"), TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

import benlsip_ref as R                      # noqa: E402
from hip_ops import HipOpsDeviceAll, ShadowOps   # noqa: E402
from nls_problem import NLSProblem           # noqa: E402


def hexvec(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def main():
    import benlsip_jl_amd as bh
    bh.init(0)
    P = NLSProblem(256, 48, 2, seed=1)
    sh = ShadowOps(HipOpsDeviceAll(bh))
    R.tralcnllss(P.x0, P.r, P.jac_r, P.c, P.jac_c, P.A, P.b, P.x_l, P.x_u, ops=sh, max_outer_iter=30, max_inner_iter=60)
    cau = [e for e in sh.events if e["op"] == "cauchy_step"]
    picked = sorted([e for e in cau if e["fix_dev"] != e["fix_cpu"]], key=lambda e: -e["rel"])[:4]
    picked += sorted([e for e in cau if e["fix_dev"] == e["fix_cpu"]], key=lambda e: -e["rel"])[:2]
    out = {"problem": "tests/nls_problem.py::NLSProblem(256, 48, 2, seed=1)", "floats": "float.hex()",
           "note": "operands of Cauchy searches of a device shadow solve whose result differs from the oracle's on identical operands",
           "events": []}
    for e in picked:
        op = e["operands"]
        assert np.array_equal(op["J"], P.jac_r(op["x"])) and np.array_equal(op["C"], P.jac_c(op["x"])), "J, C are not jac_r(x), jac_c(x)"
        out["events"].append(dict(minor=e["minor"], mu=float(op["mu"]).hex(), delta=float(op["delta"]).hex(), x=hexvec(op["x"]), g=hexvec(op["g"]),
                                  fix0=[int(i) for i in np.flatnonzero(op["fix0"])],
                                  s_dev=hexvec(op["s_dev"]), s_cpu=hexvec(op["s_cpu"]),
                                  fix_dev=int(e["fix_dev"]), fix_cpu=int(e["fix_cpu"]), rel=e["rel"],
                                  oracle_sensitivity_at_capture=e["oracle_sensitivity"], g_over_reduced_g=e["g_over_reduced_g"]))
        print("minor %d: rel %.3e, active bounds device %d / oracle %d, ||g||/||P(-g)|| = %.2e"
              % (e["minor"], e["rel"], e["fix_dev"], e["fix_cpu"], e["g_over_reduced_g"]))
    with open(os.path.join(HERE, "cauchy_events.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("%d Cauchy events of %d discrepancies written" % (len(out["events"]), len(cau)))


if __name__ == "__main__":
    main()
