"""K2c (k_lin_col + finish) timing on the benchmark graph, and a check of its outputs against the row-major K2 of the same problem."""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(MAGSACWeightBasedLoss(0.02))
for _ in range(3):
    kt = p.time_kernels(g["init_aa"], reps=10)
    print({k: round(1e3 * v, 1) for k, v in kt.items()}, "us; K2c bytes %.3f GB -> %.2f TB/s" % (p.linearize_bytes() * 1e-9, p.linearize_bytes() / kt["k_lin"] * 1e-9))
lin = p.linearize(g["init_aa"])
os.environ["GSFM_K3_COLSORT"] = "0"
q = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
q.set_loss(MAGSACWeightBasedLoss(0.02))
ref = q.linearize(g["init_aa"])
for k in ("gradient", "diag_blocks"):
    print(k, "max rel diff vs row-major K2: %.2e" % (np.abs(lin[k] - ref[k]).max() / np.abs(ref[k]).max()))
v = np.random.default_rng(0).standard_normal((g["n_cams"], 3))
a, b = p.normal_matvec(v), q.normal_matvec(v)
print("mat-vec max rel diff: %.2e" % (np.abs(a - b).max() / np.abs(b).max()))
