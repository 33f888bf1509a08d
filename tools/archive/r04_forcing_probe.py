"""Forcing schedule (gsfm_rot_options::pcg_forcing) on the benchmark graph: time, PCG iterations, and distance of the answer from the
rounds-1..3 schedule (every step at cg_relative_tolerance), for both starts and a range of loose tolerances."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss, GemanMcClureLoss
from globalsfmpy_amd.solver import RotationProblem

def run(p, init, reps=3, **kw):
    p.solve(init, **kw)
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r, s = p.solve(init, **kw); ts.append(time.perf_counter() - t)
    return r, s, 1e3 * min(ts)

def report(name, p, init, gt=None, tols=(1e-5, 1e-6, 1e-7, 1e-8)):
    r0, s0, t0 = run(p, init, pcg_forcing=0)
    print("%s  forcing off: %.2f ms, %d LM / %d PCG it, cost %.12e" % (name, t0, s0["num_iterations"], s0["num_cg_iterations"], s0["final_cost"]))
    tr = p.trace()
    print("   cost changes (relative):", ["%.1e" % (abs(row[2]) / max(row[1], 1e-300)) for row in tr[1:]], " cg:", [int(row[7]) for row in tr[1:]])
    for tol in tols:
        r, s, t = run(p, init, pcg_forcing=1, pcg_forcing_tolerance=tol)
        d = synth.angular_distance(synth.align_rotations(r, r0), r0)   # (the gauge is free: compared after alignment, as BASELINE.md defines the bar)
        print("   eps %.0e rad: %.2f ms, %d LM / %d PCG it (%d inexact, %d refined), cost rel diff %.1e, vs off: mean %.2e max %.2e rad   cg: %s" % (
            tol, t, s["num_iterations"], s["num_cg_iterations"], s["num_inexact_steps"], s["num_forcing_refinements"],
            abs(s["final_cost"] - s0["final_cost"]) / s0["final_cost"], d.mean(), d.max(), [int(row[7]) for row in p.trace()[1:]]))

which = sys.argv[1:] or ["c5", "tree", "c2", "coherent"]
if "c5" in which or "tree" in which:
    g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    p.set_loss(MAGSACWeightBasedLoss(0.02))
    if "c5" in which: report("C5 (ground truth + 2 deg)", p, g["init_aa"])
    if "tree" in which:
        init, m = synth.spanning_tree_init(g, 2023)
        report("C5 (spanning-tree start)", p, init)
    p.close()
if "coherent" in which:
    g = synth.make_graph(10000, 150000, 3, outlier_frac=0.1, local_window=300)
    for coarse in ("0", None):
        if coarse is None: os.environ.pop("GSFM_PCG_COARSE", None)
        else: os.environ["GSFM_PCG_COARSE"] = coarse
        for name, et, loss, kw in (("coherent 10k/150k Huber AA", _abi.ANGLE_AXIS, __import__("globalsfmpy_amd.loss_functions", fromlist=["x"]).HuberLoss(0.1), {}), ("coherent 10k/150k MAGSAC cov", _abi.ANGLE_AXIS_COVARIANCE, MAGSACWeightBasedLoss(0.02), dict(cov6=g["cov6"]))):
            p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, **kw)
            p.set_loss(loss)
            report(name + (" [block-Jacobi]" if coarse == "0" else " [two-level]"), p, g["init_aa"])
            p.close()
    os.environ.pop("GSFM_PCG_COARSE", None)
if "c2" in which:
    g = synth.make_graph(10000, 200000, 7, outlier_frac=0.1)
    for name, et, loss, kw in (("C2 Geman-McClure", _abi.ANGLE_AXIS, GemanMcClureLoss(0.1, 1.0), {}), ("C2 MAGSAC cov", _abi.ANGLE_AXIS_COVARIANCE, MAGSACWeightBasedLoss(0.02), dict(cov6=g["cov6"]))):
        p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, **kw)
        p.set_loss(loss)
        report(name, p, g["init_aa"])
        p.close()
