import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "globalsfmpy_amd"))
import GlobalSfMpy as sfm
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
m = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/madrid_graph.npz"))
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
for et, loss, name in ((_abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), "cov+MAGSAC"), (_abi.ANGLE_AXIS, LF.SoftLOneLoss(0.1), "SoftL1 (EstimateRotations)"), (_abi.QUATERNION_COSINE, LF.HuberLoss(0.1), "quat Huber")):
    p = RotationProblem(len(ids), ei, ej, m["rel_aa"], et, cov6=c6); p.set_loss(loss)
    p.solve(x0)
    t = time.perf_counter(); r, s = p.solve(x0); dt = time.perf_counter() - t
    tr = p.trace()
    print("Madrid %-28s %8.1f ms  %3d LM it  %5d cg it (max/it %d)  term %s  gpu ms lin %.1f sweep %.1f pcg %.1f" % (name, dt * 1e3, s["num_iterations"], s["num_cg_iterations"], int(tr[:, 7].max()), s["termination_name"], s["t_linearize_ms"], s["t_sweep_ms"], s["t_cg_ms"]))
for et, loss, name in ((_abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), "cov+MAGSAC"), (_abi.ANGLE_AXIS, LF.SoftLOneLoss(0.1), "SoftL1 (EstimateRotations)"), (_abi.QUATERNION_COSINE, LF.HuberLoss(0.1), "quat Huber")):
    p = RotationProblem(len(ids), ei, ej, m["rel_aa"], et, cov6=c6); p.set_loss(loss)
    p.solve(x0, dense_cholesky_max_cams=1000)
    t = time.perf_counter(); r, s = p.solve(x0, dense_cholesky_max_cams=1000); dt = time.perf_counter() - t
    print("Madrid %-28s dense Cholesky steps: %8.1f ms  %3d LM it  %d dense solves  term %s  gpu ms linear solve %.1f" % (name, dt * 1e3, s["num_iterations"], s["num_dense_solves"], s["termination_name"], s["t_cg_ms"]))
print("--- PCG iteration cap experiment (cov+MAGSAC) ---")
p = RotationProblem(len(ids), ei, ej, m["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=c6); p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
ref, sref = p.solve(x0)
for cap in (1000, 300, 150, 80, 40):
    t = time.perf_counter(); r, s = p.solve(x0, max_cg_iterations=cap); dt = time.perf_counter() - t
    d = synth.angular_distance(synth.align_rotations(r, ref), ref)
    print("cap %4d: %7.1f ms %3d LM it %6d cg  cost %.9e  mean dR vs uncapped %.2e" % (cap, dt * 1e3, s["num_iterations"], s["num_cg_iterations"], s["final_cost"], d.mean()))
tr = p.trace()
