import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from globalsfmpy_amd import _abi
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from test_gpu_fullsize import _madrid_component
g = _madrid_component(os.path.join(ROOT, "tests", "golden"))
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
for _ in range(3):
    r, s = p.solve(g["init_aa"])
print(s["t_total_ms"], s["num_iterations"], s["num_dense_solves"])
