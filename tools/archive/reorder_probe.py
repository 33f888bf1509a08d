"""Locality relabelling (reverse Cuthill-McKee at create): hot-kernel times and solve time on a spatially coherent graph whose
ids were shuffled, with GSFM_REORDER=0 / auto, and on the uniformly random C5 graph (where it must not be adopted)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
n, e = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
for name, win in (("local window 1000, shuffled ids", 1000), ("local window 4000, shuffled ids", 4000), ("uniform random (C5)", 0)):
    g = synth.make_graph(n, e, 2023, outlier_frac=0.3, local_window=win)
    ref = None
    for mode in ("0", None):
        if mode is None: os.environ.pop("GSFM_REORDER", None)
        else: os.environ["GSFM_REORDER"] = mode
        t = time.perf_counter()
        p = RotationProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); p.set_loss(MAGSACWeightBasedLoss(0.02))
        tc = time.perf_counter() - t
        kt = p.time_kernels(g["init_aa"], reps=10)
        p.solve(g["init_aa"])
        t = time.perf_counter(); r, s = p.solve(g["init_aa"]); dt = time.perf_counter() - t
        if ref is None: ref = r
        print("%-34s reorder=%-4s create %.1f s  K1 %.0f  K2 %.0f  K3 %.0f us  solve %.1f ms (%d LM, %d PCG)  max |dR| vs reorder=0: %.1e" % (
            name, mode or "auto", tc, 1e3 * kt["k_cost"], 1e3 * kt["k_lin"], 1e3 * kt["k_matvec"], dt * 1e3, s["num_iterations"], s["num_cg_iterations"], synth.angular_distance(r, ref).max()), flush=True)
        p.close()
