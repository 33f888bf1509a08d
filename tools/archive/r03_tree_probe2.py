"""Tree-initialised C5: the PCG convergence curve of the first two LM steps (iterations needed per tolerance)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
init, m = synth.spanning_tree_init(g, 2023)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(MAGSACWeightBasedLoss(0.02))
x = init
for step in range(3):
    row = []
    for tol in (1e-1, 1e-2, 1e-3, 1e-4, 1e-6, 1e-8, 1e-10, 1e-12):
        r, s = p.solve(x, max_num_iterations=1, cg_relative_tolerance=tol, max_cg_iterations=3000)
        row.append((tol, s["num_cg_iterations"]))
    print("LM step", step + 1, "PCG iterations to tolerance:", row, flush=True)
    x = r
lin = p.linearize(init)
D = lin["diag_blocks"].reshape(-1, 6) if lin["diag_blocks"].ndim == 2 else lin["diag_blocks"]
tr = D[:, 0] + D[:, 3] + D[:, 5] if D.shape[1] == 6 else np.trace(D.reshape(-1, 3, 3), axis1=1, axis2=2)
q = np.quantile(tr, [0, 0.01, 0.1, 0.5, 0.9, 0.99, 1])
print("trace of the diagonal blocks at the tree start, quantiles 0/1/10/50/90/99/100 %:", q)
