"""Maximum-size check (manual, not in the test suite): a graph 15x the C5 workload, so that every per-entry plane exceeds
4 GiB (300M directed entries x 16 B) and entry indices pass 2^28.  Compares the device sweep and linearisation with the CPU
oracle over ALL edges, checks the normal-matrix symmetry, then solves.  usage: big_run.py [n_cams] [n_edges]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from oracle import pyoracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500000
e = int(sys.argv[2]) if len(sys.argv) > 2 else 150000000
t = time.perf_counter()
g = synth.make_graph(n, e, 77, outlier_frac=0.3)
print("generated %d cams / %d edges in %.0f s" % (n, e, time.perf_counter() - t), flush=True)
t = time.perf_counter()
p = RotationProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
print("device problem created in %.0f s" % (time.perf_counter() - t), flush=True)
t = time.perf_counter()
o = pyoracle.OracleProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); o.set_loss(LF.MAGSACWeightBasedLoss(0.02))
print("oracle problem created in %.0f s" % (time.perf_counter() - t), flush=True)
x = g["init_aa"]
t = time.perf_counter(); d = p.residuals(x); td = time.perf_counter() - t
t = time.perf_counter(); r = o.residuals(x); to = time.perf_counter() - t
print("sweep over all edges: device %.2f s (incl. %d MB read-back), oracle %.1f s" % (td, d["s"].nbytes // 2**20 * 4, to))
print("  max |s_dev - s_ora| / max(1, s) = %.2e   cost rel diff %.2e" % (np.max(np.abs(d["s"] - r["s"]) / np.maximum(1.0, r["s"])), abs(d["cost"] - r["cost"]) / r["cost"]), flush=True)
assert np.max(np.abs(d["s"] - r["s"]) / np.maximum(1.0, r["s"])) < 1e-10 and abs(d["cost"] - r["cost"]) < 1e-10 * r["cost"]
del d, r
ld, lo = p.linearize(x), o.linearize(x)
gd, Dd, go, Do = ld["gradient"], ld["diag_blocks"], lo["gradient"], lo["diag_blocks"]
print("linearisation: max |g_dev - g_ora| / max|g| = %.2e   max |D_dev - D_ora| / max|D| = %.2e" % (np.abs(gd - go).max() / np.abs(go).max(), np.abs(Dd - Do).max() / np.abs(Do).max()), flush=True)
assert np.abs(gd - go).max() < 1e-9 * np.abs(go).max() and np.abs(Dd - Do).max() < 1e-9 * np.abs(Do).max()
rng = np.random.default_rng(0)
u, v = rng.standard_normal((n, 3)), rng.standard_normal((n, 3))
Au, Av = p.normal_matvec(u), p.normal_matvec(v)
print("normal matrix symmetry: |v.Au - u.Av| / |v.Au| = %.2e" % (abs((v * Au).sum() - (u * Av).sum()) / abs((v * Au).sum())), flush=True)
assert abs((v * Au).sum() - (u * Av).sum()) < 1e-10 * abs((v * Au).sum())
t = time.perf_counter(); rot, s = p.solve(x); dt = time.perf_counter() - t
err = synth.angular_distance(synth.align_rotations(rot, g["gt_aa"]), g["gt_aa"])
print("solve: %.1f ms, %d LM iterations, %d PCG iterations, %s; %.3g edge-residuals/s; mean error vs ground truth %.4f deg (start %.2f deg)" % (
    dt * 1e3, s["num_iterations"], s["num_cg_iterations"], s["termination_name"], e * s["num_residual_sweeps"] / dt,
    np.rad2deg(err.mean()), np.rad2deg(synth.angular_distance(synth.align_rotations(x, g["gt_aa"]), g["gt_aa"]).mean())))
print("K1 sweep: %.1f us for %d edges" % (1e3 * p.time_sweep(x, reps=5), e))
