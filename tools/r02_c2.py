"""C2 (10k cameras / 200k edges) timing by solver options (dev tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(10000, 200000, 11, outlier_frac=0.1)
for name, et, loss, kw in (("GM", _abi.ANGLE_AXIS, LF.GemanMcClureLoss(0.1, 1.0), {}), ("cov+MAGSAC", _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), dict(cov6=g["cov6"]))):
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, **kw); p.set_loss(loss)
    for opts in (dict(pcg_single_reduction=0), dict(pcg_single_reduction=1), dict(pcg_single_reduction=1, cg_check_interval=16)):
        p.solve(g["init_aa"], **opts)
        ts = []
        for _ in range(5):
            t = time.perf_counter(); r, s = p.solve(g["init_aa"], **opts); ts.append(time.perf_counter() - t)
        print("C2 %-11s lanes=%s %-55s %7.2f ms  %2d LM it %4d cg it  gpu: lin %.2f sweep %.2f pcg %.2f ms" % (name, os.environ.get("GSFM_ROW_LANES", "auto"), opts, min(ts) * 1e3,
              s["num_iterations"], s["num_cg_iterations"], s["t_linearize_ms"], s["t_sweep_ms"], s["t_cg_ms"]), flush=True)
