import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 2000000, 7, outlier_frac=0.1, local_window=1000)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS); p.set_loss(LF.HuberLoss(0.1))
r, s = p.solve(g["init_aa"]); print(s["num_cg_iterations"])
