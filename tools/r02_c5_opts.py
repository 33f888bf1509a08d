"""C5 solve time by solver option (dev tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
ref = None
for opts in (dict(pcg_single_reduction=0), dict(pcg_single_reduction=1), dict(pcg_single_reduction=1, cg_check_interval=16), dict(pcg_single_reduction=0, cg_check_interval=16)):
    p.solve(g["init_aa"], **opts)
    ts = []
    for _ in range(4):
        t = time.perf_counter(); r, s = p.solve(g["init_aa"], **opts); ts.append(time.perf_counter() - t)
    if ref is None: ref = r
    print("%-60s %7.2f ms  %d LM it %4d cg it  gpu: lin %.2f sweep %.2f pcg %.2f ms; max |dR| vs first %.1e" % (opts, min(ts) * 1e3, s["num_iterations"], s["num_cg_iterations"],
          s["t_linearize_ms"], s["t_sweep_ms"], s["t_cg_ms"], synth.angular_distance(r, ref).max()), flush=True)
