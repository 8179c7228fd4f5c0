"""Manual probe: the worst Cauchy event of the sphere-regression shadow solve (device Cauchy search, cg_fused = 1) re-evaluated on its
recorded operands by the oracle and by both forms of the device search."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, os.path.join(ROOT, "oracle"), TESTS):
    sys.path.insert(0, p)
import benlsip_jl_amd as bh
import benlsip_ref as R
import sphere_problem as sp
from hip_ops import HipOpsDeviceAll, ShadowOps
np.set_printoptions(precision=17, linewidth=200)
bh.init(0)
sh = ShadowOps(HipOpsDeviceAll(bh), relnorm_tol=1e-12, sens_samples=16)
R.tralcnllss(sp.x0, sp.r, sp.jac_r, sp.c, sp.jac_c, sp.A, sp.b, sp.x_l, sp.x_u, max_outer_iter=100, max_inner_iter=250, ops=sh)
ev = [e for e in sh.events if e["op"] == "cauchy_step"]
e = max(ev, key=lambda e: e["rel"] / max(8 * e["oracle_sensitivity"], 1e-12))
o = e["operands"]
print({k: v for k, v in e.items() if k not in ("operands", "ties")})
print("x", o["x"], "\ng", o["g"], "\ndelta", o["delta"], "fix0", o["fix0"], "\nJ", o["J"], "\nC", o["C"], "mu", o["mu"], "\nA", o["A"], "\nbounds", o["x_l"], o["x_u"])
print("s_dev", o["s_dev"], "\ns_cpu", o["s_cpu"])
Ho = R.AlHessian(o["J"], o["C"], o["mu"])
L0 = R.chol_lower(o["A"] @ o["A"].T)
def oracle(g):
    cons = R.make_mixed_constraints(o["A"], L0, l=o["x_l"], u=o["x_u"])
    cons.fixvars = o["fix0"].copy(); R.update_chol(cons, L0)
    s = R.cauchy_step(o["x"], g, Ho, L0, cons, o["delta"], R.NumpyOps())
    return s, cons.fixvars.copy()
s_o, f_o = oracle(o["g"])
print("oracle again", s_o, f_o, "model", float(o["g"] @ s_o + 0.5 * R.vthv(Ho, s_o)))
H = bh.AlHessian(o["J"], o["C"], o["mu"])
for image in (1, 0):
    bh.set_option("cauchy_image", image)
    cons = bh.MixedConstraints(o["A"], None, None, l=o["x_l"], u=o["x_u"])
    s, info = bh.cauchy_step(o["x"], o["g"], H, cons, o["delta"], full_output=True)
    print("device image=%d" % image, s, np.asarray(cons.fixvars), info, "rel to oracle %.3e" % (np.linalg.norm(s - s_o) / np.linalg.norm(s_o)),
          "model", float(o["g"] @ s + 0.5 * R.vthv(Ho, s)), "|A s| %.2e" % float(np.linalg.norm(o["A"] @ s)))
bh.set_option("cauchy_image", 1)
rng = np.random.default_rng(1)
for k in range(6):
    g2 = o["g"] * (1.0 + 2.2e-16 * rng.uniform(-1, 1, 3))
    s2, f2 = oracle(g2)
    print("oracle, g perturbed by 1 ulp:", s2, f2, "rel %.3e" % (np.linalg.norm(s2 - s_o) / np.linalg.norm(s_o)))
red = R.NumpyOps().projection(R.make_mixed_constraints(o["A"], L0, l=o["x_l"], u=o["x_u"]), -o["g"])
print("P(-g) with the initial active set:", red, "|g|/|P(-g)| = %.3e" % (np.linalg.norm(o["g"]) / np.linalg.norm(red)))
