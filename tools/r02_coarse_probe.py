"""Two-level preconditioner on spatially coherent graphs (GSFM_PCG_COARSE): PCG iterations, solve time and the change of the answer (dev tool).
usage: r02_coarse_probe.py [cams edges window]..."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
cases = [(20000, 400000, 400)]
if len(sys.argv) > 3:
    cases = [(int(sys.argv[k]), int(sys.argv[k + 1]), int(sys.argv[k + 2])) for k in range(1, len(sys.argv) - 2, 3)]
for cams, edges, win in cases:
    g = synth.make_graph(cams, edges, 7, outlier_frac=0.1, local_window=win)
    for name, et, loss, kw in (("Huber", _abi.ANGLE_AXIS, LF.HuberLoss(0.1), {}), ("cov+MAGSAC", _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), dict(cov6=g["cov6"]))):
        ref = None
        for coarse in ("0", "auto", "16", "64"):
            if coarse == "auto": os.environ.pop("GSFM_PCG_COARSE", None)
            else: os.environ["GSFM_PCG_COARSE"] = coarse
            p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, **kw); p.set_loss(loss)
            p.solve(g["init_aa"], max_num_iterations=2)
            t = time.perf_counter(); r, s = p.solve(g["init_aa"]); dt = time.perf_counter() - t
            if ref is None: ref = r
            d = synth.angular_distance(synth.align_rotations(r, ref), ref)
            print("%d/%d window %d %-11s aggregates %-4s %8.1f ms  %2d LM it %6d PCG it  cost %.12e  vs plain: mean dR %.1e max %.1e  gpu ms: lin %.1f sweep %.1f pcg %.1f" % (
                cams, edges, win, name, coarse, dt * 1e3, s["num_iterations"], s["num_cg_iterations"], s["final_cost"], d.mean(), d.max(), s["t_linearize_ms"], s["t_sweep_ms"], s["t_cg_ms"]), flush=True)
            p.close()
