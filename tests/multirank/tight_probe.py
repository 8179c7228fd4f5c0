"""Debug aid: iteration count / model value of worker.py's kappa2=1e-3 box solve per (world, BH_CG_FUSED)."""
import os, subprocess, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
from problem import make_problem
from _util import R
P = make_problem()
Ho = R.AlHessian(P["J"], P["C"], P["mu"])
for world in (2, 3, 4):
    for fused in (0, 1):
        d = tempfile.mkdtemp()
        env = dict(os.environ, BH_COMM="ipc", BH_CG_FUSED=str(fused))
        ps = [subprocess.Popen([sys.executable, os.path.join(HERE, "worker.py"), str(r), str(world), d], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = [p.communicate(timeout=300)[0] for p in ps]
        if any(p.returncode for p in ps):
            print(world, fused, "FAILED", outs[0][-800:]); continue
        z = np.load(os.path.join(d, "rank0.npz"))
        gm = z["gm"]
        model = lambda y: float(gm @ y + 0.5 * R.vthv(Ho, y))
        print("world", world, "fused", fused, "tight it", int(z["box_tight_it"]), "st", int(z["box_tight_st"]), "model %.10e" % model(z["box_tight_w"]),
              "mid it", int(z["box_mid_it"]), "loose it", int(z["box_loose_it"]), flush=True)
