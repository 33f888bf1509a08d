import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "manual"))
import fuzz_forcing
from globalsfmpy_amd import synth
from globalsfmpy_amd.solver import RotationProblem
for spec in sys.argv[1:]:
    seed, trials = spec.split(":")
    only = [int(v) for v in trials.split(",")]
    for t, g, et, loss, init, coherent in fuzz_forcing.cases(max(only) + 1, int(seed), only):
        p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"], inlier_weight=g["inlier_weight"])
        p.set_loss(loss)
        r0, s0 = p.solve(init, pcg_forcing=0, dense_cholesky_auto_cams=0)
        print("seed %s trial %d: n=%d e=%d et=%d %s exact: %d LM %d PCG" % (seed, t, g["n_cams"], len(g["edge_i"]), et, type(loss).__name__, s0["num_iterations"], s0["num_cg_iterations"]), flush=True)
        for tol in (1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3):
            r, s = p.solve(init, pcg_forcing=0, dense_cholesky_auto_cams=0, cg_relative_tolerance=tol)
            d = synth.angular_distance(synth.align_rotations(r, r0), r0)
            print("   cg tol %.0e: LM %d PCG %5d  dR mean %.1e max %.1e" % (tol, s["num_iterations"], s["num_cg_iterations"], d.mean(), d.max()), flush=True)
        p.close()
