"""Latency regime: textbook PCG (4 kernels / iteration) vs single-reduction PCG (2) on C2-size and Madrid-size problems."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem

def run(name, g, et, loss, **kw):
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, **kw); p.set_loss(loss)
    out = {}
    for sr, gr in ((0, 0), (0, 1), (1, 0), (1, 1)):
        p.solve(g["init_aa"], pcg_single_reduction=sr, pcg_hip_graph=gr, dense_cholesky_max_cams=0)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); r, s = p.solve(g["init_aa"], pcg_single_reduction=sr, pcg_hip_graph=gr, dense_cholesky_max_cams=0); ts.append(time.perf_counter() - t)
        out[(sr, gr)] = r
        print("%-34s single_reduction=%d graph=%d %8.2f ms  %3d LM it %6d cg it  cost %.12e  (gpu pcg %.2f ms)" % (name, sr, gr, min(ts) * 1e3, s["num_iterations"], s["num_cg_iterations"], s["final_cost"], s["t_cg_ms"]))
    print("   graph vs plain launches: max rotation difference %.1e rad (same kernels, same order); single-reduction vs textbook %.1e rad" % (
        np.abs(out[(0, 1)] - out[(0, 0)]).max(), synth.angular_distance(out[(1, 1)], out[(0, 1)]).max()))

for n, e in ((400, 24000), (2000, 40000), (10000, 200000), (30000, 1000000)):
    g = synth.make_graph(n, e, 11, outlier_frac=0.1)
    run("N=%d E=%d GM" % (n, e), g, _abi.ANGLE_AXIS, LF.GemanMcClureLoss(0.1, 1.0))
    run("N=%d E=%d cov+MAGSAC" % (n, e), g, _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), cov6=g["cov6"])
