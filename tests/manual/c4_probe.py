"""Dev probe: C4 batch (14 components incl. real Madrid), Madrid component's distance to the oracle vs the PCG tolerance."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from test_gpu_fullsize import _madrid_component
sizes = [577, 227, 450, 553, 332, 328, 2152, 1084, 572, 789, 836, 437, 5288]
scenes = [synth.make_graph(n, 12 * n, seed=400 + k, outlier_frac=0.1) for k, n in enumerate(sizes)]
scenes.insert(2, _madrid_component(os.path.join(ROOT, "tests", "golden")))
parts, off = [], 0
for g in scenes:
    parts.append((off, g)); off += g["n_cams"]
N = off
ei = np.concatenate([g["edge_i"] + o for o, g in parts]).astype(np.uint32); ej = np.concatenate([g["edge_j"] + o for o, g in parts]).astype(np.uint32)
rel = np.concatenate([g["rel_aa"] for _, g in parts]); cov = np.concatenate([g["cov6"] for _, g in parts]); init = np.concatenate([g["init_aa"] for _, g in parts])
loss = LF.HuberLoss(0.1)
ora = pyoracle.OracleProblem(N, ei, ej, rel, _abi.ANGLE_AXIS_COVTRACE, cov6=cov); ora.set_loss(loss)
ro, so = ora.solve(init)
dev = RotationProblem(N, ei, ej, rel, _abi.ANGLE_AXIS_COVTRACE, cov6=cov); dev.set_loss(loss)
sl = slice(parts[2][0], parts[2][0] + parts[2][1]["n_cams"])
for tol in (1e-12, 1e-13, 1e-14, 1e-15):
    for sr in (0, 1):
        dev.solve(init, cg_relative_tolerance=tol, pcg_single_reduction=sr)
        t = time.perf_counter(); rd, sd = dev.solve(init, cg_relative_tolerance=tol, pcg_single_reduction=sr); t = time.perf_counter() - t
        d = synth.angular_distance(synth.align_rotations(rd[sl], ro[sl]), ro[sl])
        print("tol %.0e sr=%d: %d it (oracle %d), cg %d, %.1f ms, cost rel %.1e, Madrid mean |dR| %.2e max %.2e" % (tol, sr, sd["num_iterations"], so["num_iterations"], sd["num_cg_iterations"], t * 1e3,
              abs(sd["final_cost"] - so["final_cost"]) / so["final_cost"], d.mean(), d.max()), flush=True)
