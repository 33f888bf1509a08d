"""Kernel trace of whole C5 solves (default schedule): run under rocprofv3 --kernel-trace; tools/r04b_solve_gaps.py reads the db."""
import sys; sys.path.insert(0, "/root/repo")
import time
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(MAGSACWeightBasedLoss(0.02))
for k in range(4):
    t = time.perf_counter(); r, s = p.solve(g["init_aa"]); dt = time.perf_counter() - t
    print("solve %d: %.2f ms, %d LM it, %d PCG it" % (k, 1e3 * dt, s["num_iterations"], s.get("cg_iterations", -1)), flush=True)
