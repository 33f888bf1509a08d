"""The exact step's factorisation as one launch per block column (dense_kernels.hpp, k_chol_look) against the fused step (GSFM_CHOL_FUSED=1)
inside the product: Madrid from its spanning-tree start (covariance + MAGSAC, SoftL1) and C4 as bench.py builds it -- time per solve, LM
iterations, final cost (must agree to the last bit)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "globalsfmpy_amd"))
import numpy as np
import GlobalSfMpy as sfm
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from test_gpu_fullsize import _madrid_component

def best(p, init, reps=5, **kw):
    p.solve(init, **kw)
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r, s = p.solve(init, **kw); ts.append(1e3 * (time.perf_counter() - t))
    return min(ts), r, s

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
sizes = [577, 227, 450, 553, 332, 328, 2152, 1084, 572, 789, 836, 437, 5288]
scenes = [synth.make_graph(n, 12 * n, seed=400 + k, outlier_frac=0.1) for k, n in enumerate(sizes)]
scenes.insert(2, _madrid_component(os.path.join(ROOT, "tests", "golden")))
offs = np.cumsum([0] + [g["n_cams"] for g in scenes])
c4 = dict(n=int(offs[-1]), ei=np.concatenate([g["edge_i"] + o for o, g in zip(offs, scenes)]).astype(np.uint32), ej=np.concatenate([g["edge_j"] + o for o, g in zip(offs, scenes)]).astype(np.uint32),
          rel=np.concatenate([g["rel_aa"] for g in scenes]), cov=np.concatenate([g["cov6"] for g in scenes]), init=np.concatenate([g["init_aa"] for g in scenes]))
res = {}
for fused in ("1", "0"):
    os.environ["GSFM_CHOL_FUSED"] = fused
    name = "fused step (until round 6)" if fused == "1" else "two block columns per launch"
    for et, loss, what in ((_abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), "Madrid cov + MAGSAC"), (_abi.ANGLE_AXIS, LF.SoftLOneLoss(0.1), "Madrid SoftL1 (EstimateRotations)")):
        p = RotationProblem(len(ids), ei, ej, m["rel_aa"], et, cov6=c6); p.set_loss(loss)
        t, r, s = best(p, x0)
        print("%-30s %-34s %7.2f ms  %3d LM  %3d exact steps  cost %.17g" % (name, what, t, s["num_iterations"], s["num_dense_solves"], s["final_cost"]), flush=True)
        res[(what, fused)] = (r, s["final_cost"], s["num_iterations"])
    p = RotationProblem(c4["n"], c4["ei"], c4["ej"], c4["rel"], _abi.ANGLE_AXIS_COVTRACE, cov6=c4["cov"]); p.set_loss(LF.HuberLoss(0.1))
    t, r, s = best(p, c4["init"])
    print("%-30s %-34s %7.2f ms  %3d LM  %3d exact steps  %d PCG  cost %.17g" % (name, "C4 (14 scenes, one problem)", t, s["num_iterations"], s["num_dense_solves"], s["num_cg_iterations"], s["final_cost"]), flush=True)
    res[("C4", fused)] = (r, s["final_cost"], s["num_iterations"])
for what in sorted(set(k[0] for k in res)):
    a, b = res[(what, "1")], res[(what, "0")]
    print("%-34s rotations bit-identical: %s   cost identical: %s   LM iterations %d / %d" % (what, np.array_equal(a[0], b[0]), a[1] == b[1], a[2], b[2]))
