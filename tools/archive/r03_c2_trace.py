"""C2 (10k cameras / 200k edges), covariance + MAGSAC: a few solves; run under rocprofv3 --kernel-trace, then tools/archive/r03_gaps.py on the db."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(10000, 200000, 7, outlier_frac=0.3)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(MAGSACWeightBasedLoss(0.02))
p.solve(g["init_aa"])
t = time.perf_counter()
for _ in range(3):
    r, s = p.solve(g["init_aa"])
print("C2 MAGSAC: %.2f ms per solve, %d LM, %d PCG iterations, pcg %.2f ms" % (1e3 * (time.perf_counter() - t) / 3, s["num_iterations"], s["num_cg_iterations"], s["t_cg_ms"]))
