"""One configuration of the real Madrid graph (covariance + MAGSAC, the pipeline's defaults), five solves: wall time per solve next to the kernel
time rocprofv3 reports for the same process (profiles/r03_madrid_kernel_stats.txt)."""
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
p = RotationProblem(len(ids), ei, ej, m["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=c6); p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
p.solve(x0)
N = 5
t = time.perf_counter()
for _ in range(N):
    r, s = p.solve(x0)
dt = (time.perf_counter() - t) / N
print("Madrid cov+MAGSAC: %.2f ms per solve (wall), %d LM iterations, %d dense solves; device timers: linear solve %.2f ms, linearise %.2f, sweeps %.2f" % (
    1e3 * dt, s["num_iterations"], s["num_dense_solves"], s["t_cg_ms"], s["t_linearize_ms"], s["t_sweep_ms"]))
