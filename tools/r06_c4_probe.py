"""C4 (the 14 scenes as one disconnected problem, bench.py's construction) under component_rest = 1 / 0 and, verbose, which components are live per LM iteration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from test_gpu_fullsize import _madrid_component
sizes = [577, 227, 450, 553, 332, 328, 2152, 1084, 572, 789, 836, 437, 5288]
scenes = [synth.make_graph(n, 12 * n, seed=400 + k, outlier_frac=0.1) for k, n in enumerate(sizes)]
scenes.insert(2, _madrid_component(os.path.join(ROOT, "tests", "golden")))
offs = np.cumsum([0] + [g["n_cams"] for g in scenes])
ei = np.concatenate([g["edge_i"] + o for o, g in zip(offs, scenes)]).astype(np.uint32)
ej = np.concatenate([g["edge_j"] + o for o, g in zip(offs, scenes)]).astype(np.uint32)
rel = np.concatenate([g["rel_aa"] for g in scenes]); cov = np.concatenate([g["cov6"] for g in scenes]); init = np.concatenate([g["init_aa"] for g in scenes])
p = RotationProblem(int(offs[-1]), ei, ej, rel, _abi.ANGLE_AXIS_COVTRACE, cov6=cov)
p.set_loss(LF.HuberLoss(0.1))
for kw in (dict(), dict(component_rest=0), dict(lm_device_control=0)):
    p.solve(init, **kw)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); r, s = p.solve(init, **kw); ts.append(time.perf_counter() - t)
    print("%s: %.2f ms, %d LM, %d PCG, %d dense steps, cost %.12e" % (kw or "default", 1e3 * min(ts), s["num_iterations"], s["num_cg_iterations"], s["num_dense_solves"], s["final_cost"]), flush=True)
if len(sys.argv) > 1:
    p.solve(init, verbose=1)
