"""Exact Cholesky step against (forcing-schedule) PCG per LM step at 1DSfM sizes: where does the factorisation pay?  (round-3 review item 7)"""
import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
cases = [(600, 3600, 16), (800, 4800, 16), (800, 6400, 24), (1000, 6000, 16), (1500, 12000, 24), (1500, 12000, 60), (3000, 24000, 24), (1500, 45000, None), (5000, 60000, 40)]
for n, e, w in cases:
    g = synth.make_graph(n, e, seed=8, outlier_frac=0.1, **({"local_window": w} if w else {}))
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS); p.set_loss(LF.HuberLoss(0.1))
    for kw in (dict(), dict(dense_cholesky_max_cams=0), dict(dense_cholesky_max_cams=0, pcg_forcing=0), dict(dense_cholesky_max_cams=n)):
        p.solve(g["init_aa"], **kw); t = time.perf_counter(); r, s = p.solve(g["init_aa"], **kw); dt = time.perf_counter() - t
        print(n, e, w, kw, "%.2f ms" % (1e3 * dt), s["num_iterations"], "LM", s["num_dense_solves"], "dense", s["num_cg_iterations"], "cg", "t_cg %.2f ms" % s["t_cg_ms"], [int(x) for x in p.trace()[1:, 7]])
