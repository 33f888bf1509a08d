"""One solve of the coherent 100k-camera / 2M-edge graph of bench.py (cov + MAGSAC) with the automatic two-level preconditioner, for a kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 2000000, 7, outlier_frac=0.1, local_window=1000)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
for _ in range(3):
    r, s = p.solve(g["init_aa"])
print(s["num_iterations"], s["num_cg_iterations"])
