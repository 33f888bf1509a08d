"""C4 (the 14 1DSfM scenes as one disconnected problem, tests/test_gpu_fullsize.py's construction without the Madrid component's real graph
replaced: 13 synthetic scenes + a 394-camera one), a few solves; run under rocprofv3 --kernel-trace, then tools/r04b_solve_gaps.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
sizes = [577, 227, 394, 450, 553, 332, 328, 2152, 1084, 572, 789, 836, 437, 5288]
scenes = [synth.make_graph(n, 12 * n, seed=400 + k, outlier_frac=0.1) for k, n in enumerate(sizes)]
offs = np.cumsum([0] + [g["n_cams"] for g in scenes])
ei = np.concatenate([g["edge_i"] + o for o, g in zip(offs, scenes)]).astype(np.uint32)
ej = np.concatenate([g["edge_j"] + o for o, g in zip(offs, scenes)]).astype(np.uint32)
rel = np.concatenate([g["rel_aa"] for g in scenes]); cov = np.concatenate([g["cov6"] for g in scenes]); init = np.concatenate([g["init_aa"] for g in scenes])
p = RotationProblem(int(offs[-1]), ei, ej, rel, _abi.ANGLE_AXIS_COVTRACE, cov6=cov)
p.set_loss(LF.HuberLoss(0.1))
p.solve(init)
for _ in range(3):
    t = time.perf_counter(); r, s = p.solve(init); dt = time.perf_counter() - t
    print("C4-like: %.2f ms, %d LM, %d PCG iterations, %d graph launches" % (1e3 * dt, s["num_iterations"], s["num_cg_iterations"], s["num_graph_launches"]), flush=True)
