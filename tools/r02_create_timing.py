"""Phase times of gsfm_rot_problem_create on the C5 graph (GSFM_CREATE_TIMING=1) and on Madrid-size problems (dev tool)."""
import os, sys, time
os.environ["GSFM_CREATE_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
for k in range(2):
    t = time.perf_counter()
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    print("create total %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
    p.close()
