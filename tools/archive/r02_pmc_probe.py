#!/usr/bin/env python3
"""Driver for the rocprofv3 PMC passes (profiles/r02_*): launches every hot kernel of the C5 problem a few times -- the three of
gsfm_rot_time_kernels (k_cost trial-cost sweep, k_lin, k_matvec) and the K1 variants of gsfm_rot_time_sweep_variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(MAGSACWeightBasedLoss(0.02))
print(p.time_kernels(g["init_aa"], reps=6))
print(p.time_sweep_variants(g["init_aa"], reps=6))
