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
    for sr in (0, 1):
        p.solve(g["init_aa"], pcg_single_reduction=sr)
        t = time.perf_counter(); r, s = p.solve(g["init_aa"], pcg_single_reduction=sr); dt = time.perf_counter() - t
        out[sr] = r
        print("%-34s single_reduction=%d %8.2f ms  %3d LM it %6d cg it  cost %.12e" % (name, sr, dt * 1e3, s["num_iterations"], s["num_cg_iterations"], s["final_cost"]))
    d = synth.angular_distance(synth.align_rotations(out[1], out[0]), out[0])
    print("   mean/max rotation difference between the two: %.2e / %.2e rad" % (d.mean(), d.max()))

for n, e in ((400, 24000), (2000, 40000), (10000, 200000), (30000, 1000000)):
    g = synth.make_graph(n, e, 11, outlier_frac=0.1)
    run("N=%d E=%d GM" % (n, e), g, _abi.ANGLE_AXIS, LF.GemanMcClureLoss(0.1, 1.0))
    run("N=%d E=%d cov+MAGSAC" % (n, e), g, _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), cov6=g["cov6"])
