import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
init, m = synth.spanning_tree_init(g, 2023)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(MAGSACWeightBasedLoss(0.02))
t = time.perf_counter(); r, s = p.solve(init); print("solve", time.perf_counter() - t, s["num_iterations"], s["num_cg_iterations"])
np.set_printoptions(linewidth=200, precision=4, suppress=False)
print(p.trace())
out = p.residuals(init)
print("fraction of edges in the MAGSAC tail at the tree init:", float((out["rho"][:, 2] == 0).mean()), " inliers only:", float((out["rho"][~g["is_outlier"], 2] == 0).mean()))
