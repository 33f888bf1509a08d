import os, sys, time
sys.path.insert(0, "/root/repo")
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(MAGSACWeightBasedLoss(0.02))
for rep in range(3):
    for sr in (0, 1):
        p.solve(g["init_aa"], pcg_single_reduction=sr)
        t = time.perf_counter(); r, s = p.solve(g["init_aa"], pcg_single_reduction=sr); dt = time.perf_counter() - t
        print("single_reduction", sr, "ms %.2f" % (1e3 * dt), "lm", s["num_iterations"], "cg", s["num_cg_iterations"], "pcg ms %.2f" % s["t_cg_ms"], "cost %.12e" % s["final_cost"], flush=True)
