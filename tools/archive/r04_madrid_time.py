"""Madrid (exact steps) and C2 with the LM control on the device against the host loop: wall time per solve."""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/globalsfmpy_amd")
import numpy as np
import GlobalSfMpy as sfm
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
m = np.load("/root/repo/tests/golden/madrid_graph.npz")
ids = np.sort(m["view_ids"]); idx = {int(v): k for k, v in enumerate(ids)}
vg = sfm.ViewGraph()
for a_, b_, r_ in zip(m["edge_a"], m["edge_b"], m["rel_aa"]):
    info = sfm.TwoViewInfo(); info.rotation_2 = r_; info.num_verified_matches = 1
    vg.AddEdge(int(a_), int(b_), info)
init = sfm.MapViewIdVector3d(); sfm.OrientationsFromMaximumSpanningTree(vg, init)
x0 = np.array([init[int(v)] for v in ids])
ei = np.array([idx[int(v)] for v in m["edge_a"]], dtype=np.uint32); ej = np.array([idx[int(v)] for v in m["edge_b"]], dtype=np.uint32)
rng = np.random.default_rng(7); A = rng.standard_normal((len(ei), 3, 3)); S = (A @ np.transpose(A, (0, 2, 1)) + 0.5 * np.eye(3)) * 3e-8
c6 = np.stack([S[:, 0, 0], S[:, 1, 1], S[:, 2, 2], S[:, 0, 1], S[:, 0, 2], S[:, 1, 2]], axis=1)
for name, et, loss, kw in (("Madrid MAGSAC cov", _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), dict(cov6=c6)), ("Madrid SoftL1", _abi.ANGLE_AXIS, LF.SoftLOneLoss(0.1), {}),
                           ("Madrid Huber quaternion", _abi.QUATERNION_COSINE, LF.HuberLoss(0.1), {})):
    p = RotationProblem(len(ids), ei, ej, m["rel_aa"], et, **kw); p.set_loss(loss)
    for dc in (0, 1):
        p.solve(x0, lm_device_control=dc); ts = []
        for _ in range(5):
            t = time.perf_counter(); r, s = p.solve(x0, lm_device_control=dc); ts.append(time.perf_counter() - t)
        print("%-26s device control %d: %.2f ms (%d LM iterations, %d exact steps, cost %.9e)" % (name, dc, 1e3 * min(ts), s["num_iterations"], s["num_dense_solves"], s["final_cost"]))
