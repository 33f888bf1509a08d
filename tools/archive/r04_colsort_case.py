import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from oracle import pyoracle as oracle
np.set_printoptions(linewidth=220, precision=6)
g = synth.make_graph(n_cams=2500, n_edges=60000, seed=9, outlier_frac=0.3)
os.environ["GSFM_K3_COLSORT"] = "1"
dev = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
dev.set_loss(LF.MAGSACWeightBasedLoss(0.02))
ora = oracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
ora.set_loss(LF.MAGSACWeightBasedLoss(0.02))
ro, so = ora.solve(g["init_aa"])
print("oracle", so["num_iterations"]); print(ora.trace())
for kw in (dict(pcg_forcing=0), dict(), dict(pcg_forcing_tolerance=1e-8), dict(pcg_forcing_tolerance=1e-9)):
    rd, sd = dev.solve(g["init_aa"], pcg_single_reduction=0, **kw)
    d = synth.angular_distance(synth.align_rotations(rd, ro), ro)
    print(kw, sd["num_iterations"], sd["num_cg_iterations"], sd["num_inexact_steps"], sd["num_forcing_refinements"], "dR mean %.2e" % d.mean()); print(dev.trace())
