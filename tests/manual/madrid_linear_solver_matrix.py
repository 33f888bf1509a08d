"""Madrid under MAGSAC to convergence (63 LM iterations, trust radius at its 1e16 cap from iteration 27 on): which LINEAR SOLVER separates the device
from the oracle?  Device {exact Cholesky steps (the default at 394 cameras), PCG only} x oracle {dense, pcg}: mean / max angular distance, LM
iterations, the first row whose costs differ by more than 1e-9 relative.  (The normal matrix has a three-dimensional gauge null space that only
the LM damping D^2 / radius lifts: at radius 1e16 a factorisation works on a numerically singular matrix, a Krylov solve from zero never leaves
the range.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from oracle import pyoracle
from test_gpu_fullsize import _madrid_component
g = _madrid_component(os.path.join(ROOT, "tests", "golden"))
loss = LF.MAGSACWeightBasedLoss(0.02)
runs = {}
for kind in ("dense", "pcg"):
    o = pyoracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); o.set_loss(loss)
    o.set_linear_solver(kind)
    r, s = o.solve(g["init_aa"]); runs["oracle " + kind] = (r, s["num_iterations"], o.trace())
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); p.set_loss(loss)
r, s = p.solve(g["init_aa"]); runs["device cholesky"] = (r, s["num_iterations"], p.trace())
r, s = p.solve(g["init_aa"], dense_cholesky_max_cams=0, dense_cholesky_auto_cams=0, pcg_forcing=0); runs["device pcg"] = (r, s["num_iterations"], p.trace())
names = list(runs)
for i, a in enumerate(names):
    for b in names[i + 1:]:
        ra, ia, ta = runs[a]; rb, ib, tb = runs[b]
        d = synth.angular_distance(synth.align_rotations(ra, rb), rb)
        n = min(len(ta), len(tb))
        first = next((k for k in range(n) if abs(ta[k, 1] - tb[k, 1]) > 1e-9 * tb[k, 1]), n)
        print("%-16s vs %-16s: %.2e rad mean / %.2e max, %d / %d LM iterations, costs within 1e-9 up to row %d of %d" % (a, b, d.mean(), d.max(), ia, ib, first, n), flush=True)
