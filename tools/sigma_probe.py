"""sigma-consensus entry point (EstimateRotationsWithSigmaConsensus) at C5 scale: time per outer iteration."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
n, e = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
g = synth.make_graph(n, e, 2023, outlier_frac=0.3)
p = RotationProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS); p.set_loss(LF.HuberLoss(0.1))
for iters in (1, 4):
    t = time.perf_counter(); r, s = p.solve_sigma_consensus(g["init_aa"], iters, 0.02); dt = time.perf_counter() - t
    err = np.rad2deg(synth.angular_distance(synth.align_rotations(r, g["gt_aa"]), g["gt_aa"]).mean())
    print("iters_num=%d: %.1f ms total, %d outer, %d LM iterations, %d sweeps, last mean |dw| %.3g, mean error %.4f deg" % (iters, dt * 1e3, s["outer_iterations"], s["num_iterations"], s["num_residual_sweeps"], s["last_weight_change"], err))
