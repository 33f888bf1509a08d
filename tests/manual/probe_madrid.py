"""Round-2 probe (dev tool): Madrid (C1) on the device, exact dense-Cholesky step vs PCG, against the oracle's Cholesky."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
import globalsfmpy_amd.GlobalSfMpy as sfm

m = np.load(os.path.join(ROOT, "tests/golden/madrid_graph.npz"))
a, b, rel = m["edge_a"], m["edge_b"], m["rel_aa"]
ids = np.sort(m["view_ids"]); idx = {int(v): k for k, v in enumerate(ids)}
vg = sfm.ViewGraph()
for i, j, r in zip(a, b, rel):
    info = sfm.TwoViewInfo(); info.rotation_2 = r; info.num_verified_matches = 1
    vg.AddEdge(int(i), int(j), info)
init = sfm.MapViewIdVector3d()
sfm.OrientationsFromMaximumSpanningTree(vg, init)
x0 = np.array([init[int(v)] for v in ids])
ei = np.array([idx[int(x)] for x in a], dtype=np.uint32); ej = np.array([idx[int(x)] for x in b], dtype=np.uint32)
rng = np.random.default_rng(7)
c6 = []
for r in rel:
    A = rng.standard_normal((3, 3)); S = (A @ A.T + 0.5 * np.eye(3)) * 3e-8
    c6.append([S[0, 0], S[1, 1], S[2, 2], S[0, 1], S[0, 2], S[1, 2]])
c6 = np.array(c6)
N = len(ids)
for name, loss, et in [("magsac", LF.MAGSACWeightBasedLoss(0.02), _abi.ANGLE_AXIS_COVARIANCE), ("softl1", LF.SoftLOneLoss(0.1), _abi.ANGLE_AXIS),
                       ("huber-quat", LF.HuberLoss(0.1), _abi.QUATERNION_COSINE)]:
    cov = c6 if et == _abi.ANGLE_AXIS_COVARIANCE else None
    o = pyoracle.OracleProblem(N, ei, ej, rel, et, cov6=cov); o.set_loss(loss); o.set_linear_solver("dense")
    d = RotationProblem(N, ei, ej, rel, et, cov6=cov); d.set_loss(loss)
    for K in (15, 30, 200):
        ro, so = o.solve(x0, max_num_iterations=K)
        for mode, kw in (("dense", dict(dense_cholesky_max_cams=4096)), ("pcg", dict(dense_cholesky_max_cams=0))):
            d.solve(x0, max_num_iterations=K, **kw)
            t = time.perf_counter(); rd, sd = d.solve(x0, max_num_iterations=K, **kw); t = time.perf_counter() - t
            diff = synth.angular_distance(synth.align_rotations(rd, ro), ro)
            print("%-10s K=%3d dev-%-5s it %3d (oracle %3d) cost rel %.2e  mean %.3e max %.3e rad  %.1f ms (dense solves %d, cg %d)" % (
                name, K, mode, sd["num_iterations"], so["num_iterations"], abs(sd["final_cost"] - so["final_cost"]) / so["final_cost"], diff.mean(), diff.max(),
                t * 1e3, sd["num_dense_solves"], sd["num_cg_iterations"]), flush=True)
