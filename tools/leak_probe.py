import sys, numpy as np
sys.path.insert(0, '.')
import torch
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(2000, 60000, 3, outlier_frac=0.2)
def once(dense):
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
    p.solve(g["init_aa"], dense_cholesky_max_cams=5000 if dense else 0)
    p.close()
once(True); torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for k in range(60): once(k % 2 == 0)
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("free before %.1f MB, after 60 create/solve/destroy cycles %.1f MB, delta %.2f MB" % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
