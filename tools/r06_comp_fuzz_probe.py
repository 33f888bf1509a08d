"""Component-fuzz trial 8 of seed 11 (MAGSAC + inlier weights, scenes of 77 / 414 / 147 cameras, 100 / 101 LM iterations, 2.0e-5 rad from the oracle on the
round-5 AND the round-6 library): how far apart do the device's OWN linear solvers and LM controls end on it, and the oracle's?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "manual"))
import numpy as np
import fuzz_components
from globalsfmpy_amd import synth
from globalsfmpy_amd.solver import RotationProblem
from oracle import pyoracle
(t, N, ei, ej, rel, cov, inl, init, comp, sizes, shuffled, et, loss), = list(fuzz_components.cases(30, 11, [8]))
k = len(sizes)
per = lambda x, y: max(synth.angular_distance(synth.align_rotations(x[comp == c], y[comp == c]), y[comp == c]).mean() for c in range(k))
p = RotationProblem(N, ei, ej, rel, et, cov6=cov, inlier_weight=inl); p.set_loss(loss)
runs = {}
for name, kw in (("device: component Cholesky, device LM control (default)", {}), ("device: component Cholesky, host LM control", dict(lm_device_control=0)),
                 ("device: one PCG over everything, 1e-14", dict(dense_cholesky_max_cams=0, dense_cholesky_auto_cams=0)),
                 ("device: one PCG over everything, 1e-15", dict(dense_cholesky_max_cams=0, dense_cholesky_auto_cams=0, cg_relative_tolerance=1e-15))):
    r, s = p.solve(init, **kw); runs[name] = (r, s["num_iterations"], s["final_cost"])
for name, kind in (("oracle: PCG 1e-14 (its choice above 512 cameras)", "pcg"), ("oracle: dense Cholesky of the whole batch", "dense")):
    o = pyoracle.OracleProblem(N, ei, ej, rel, et, cov6=cov, inlier_weight=inl); o.set_loss(loss); o.set_linear_solver(kind)
    r, s = o.solve(init); runs[name] = (r, s["num_iterations"], s["final_cost"])
names = list(runs)
print("trial 8 of seed 11: sizes %s, %s, error type %d" % (sizes, type(loss).__name__, et))
for n in names: print("  %-62s %3d LM iterations, final cost %.12e" % (n, runs[n][1], runs[n][2]))
print("worst component's mean angular distance (rad) between the runs:")
for i, a in enumerate(names):
    print("  " + " ".join("%8.1e" % per(runs[a][0], runs[b][0]) if j > i else "        " for j, b in enumerate(names)) + "   " + a)
