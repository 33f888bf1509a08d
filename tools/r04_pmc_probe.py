#!/usr/bin/env python3
"""Probe for the round-4 PMC passes: the C5 problem, a few launches of each hot kernel (k_cost variants, k_lin_col, k_mv_col) and of the
scalar-weight / unit-weight / sigma-consensus specialisations of K1 and K2 on an ANGLE_AXIS problem of the same graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss, TrivialLoss, SoftLOneLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(MAGSACWeightBasedLoss(0.02))
print(p.time_kernels(g["init_aa"], reps=4))
print(p.time_sweep_variants(g["init_aa"], reps=3))
p.close()
p6 = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
p6.set_loss(TrivialLoss())
p6.set_edge_weights(np.ones(len(g["edge_i"])))
print(p6.time_sweep_variants(g["init_aa"], reps=3))
p6.close()
p1 = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)   # EstimateRotations (a1): unit weights, SoftL1(0.1)
p1.set_loss(SoftLOneLoss(0.1))
print(p1.time_kernels(g["init_aa"], reps=4))
