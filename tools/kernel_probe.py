#!/usr/bin/env python3
"""Dev probe: times the three hot kernels on the C5 graph for a few tuning settings."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
n, e = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
g = synth.make_graph(n, e, 2023, outlier_frac=0.3)
for G in (os.environ.get("PROBE_G", "64,32,16").split(",")):
    os.environ["GSFM_ROW_LANES"] = G
    p = RotationProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    p.set_loss(MAGSACWeightBasedLoss(0.02))
    t = p.time_kernels(g["init_aa"], reps=10)
    print("G=%s" % G, {k: round(v * 1e3, 1) for k, v in t.items()}, "us", flush=True)
    p.close()
