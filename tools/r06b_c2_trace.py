"""C2 (10k cameras / 200k edges) as bench.py builds it -- Geman-McClure on ANGLE_AXIS, or covariances + MAGSAC -- : four solves for a kernel trace
(rocprofv3 --kernel-trace, then tools/r04b_solve_gaps.py on the database: the last solve's kernel time by name and its gaps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(10000, 200000, 11, outlier_frac=0.1)
which = sys.argv[1] if len(sys.argv) > 1 else "gm"
if which == "gm":
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS); p.set_loss(LF.GemanMcClureLoss(0.1, 1.0))
else:
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
p.solve(g["init_aa"])
for _ in range(4):
    t = time.perf_counter(); r, s = p.solve(g["init_aa"]); dt = time.perf_counter() - t
    print("C2 %s: %.3f ms, %d LM, %d PCG iterations, %d graph launches" % (which, 1e3 * dt, s["num_iterations"], s["num_cg_iterations"], s["num_graph_launches"]), flush=True)
    time.sleep(0.01)
