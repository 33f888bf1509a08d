"""The real Madrid graph (394 cameras, covariance + MAGSAC, the reference pipeline's default) from its spanning-tree start: three solves for a
kernel trace (rocprofv3 --kernel-trace, then tools/r04b_solve_gaps.py on the database: the last solve's kernel time by name and its gaps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "globalsfmpy_amd"))
import numpy as np
import GlobalSfMpy as sfm
from globalsfmpy_amd import _abi
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
m = np.load(os.path.join(ROOT, "tests/golden/madrid_graph.npz"))
ids = np.sort(m["view_ids"]); idx = {int(v): k for k, v in enumerate(ids)}
ei = np.array([idx[int(a)] for a in m["edge_a"]], dtype=np.uint32); ej = np.array([idx[int(b)] for b in m["edge_b"]], dtype=np.uint32)
rng = np.random.default_rng(7)
A = rng.standard_normal((len(ei), 3, 3)); C = (A @ np.transpose(A, (0, 2, 1)) + 0.5 * np.eye(3)) * 3e-8
c6 = np.stack([C[:, 0, 0], C[:, 1, 1], C[:, 2, 2], C[:, 0, 1], C[:, 0, 2], C[:, 1, 2]], axis=1)
vg = sfm.ViewGraph()
for a, b, r in zip(m["edge_a"], m["edge_b"], m["rel_aa"]):
    info = sfm.TwoViewInfo(); info.rotation_2 = r; vg.AddEdge(int(a), int(b), info)
init = sfm.MapViewIdVector3d(); sfm.OrientationsFromMaximumSpanningTree(vg, init)
x0 = np.array([init[int(v)] for v in ids])
which = sys.argv[1] if len(sys.argv) > 1 else "magsac"
et, loss = (_abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02)) if which == "magsac" else (_abi.ANGLE_AXIS, LF.SoftLOneLoss(0.1))
p = RotationProblem(len(ids), ei, ej, m["rel_aa"], et, cov6=c6); p.set_loss(loss)
p.solve(x0)
for _ in range(3):
    t = time.perf_counter(); r, s = p.solve(x0); dt = time.perf_counter() - t
    print("Madrid %s: %.2f ms, %d LM, %d dense, final cost %.17g" % (which, 1e3 * dt, s["num_iterations"], s["num_dense_solves"], s["final_cost"]), flush=True)
    time.sleep(0.01)
