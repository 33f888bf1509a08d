import os, sys, math
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from test_gpu_round5 import _replay_radius, _madrid
np.set_printoptions(linewidth=250, precision=6)
g = synth.make_graph(3000, 300000, seed=17, outlier_frac=0.3)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
r0, s0 = p.solve(g["init_aa"], pcg_forcing=0); print(p.trace())
r1, s1 = p.solve(g["init_aa"], verbose=1); print(s1["num_forcing_restarts"], s1["num_inexact_steps"])
g = _madrid(os.path.join(ROOT, "tests", "golden"))
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
p.set_loss(LF.SoftLOneLoss(0.1))
rd, sd = p.solve(g["init_aa"], lm_device_control=1)
td = p.trace()
rr = _replay_radius(td, sd["termination"], lambda t: float(Fraction(t) ** 3))
for k in range(len(td)):
    if rr[k] != td[k, 6]:
        t = 2.0 * td[k, 5] - 1.0
        print("row", k, "rel_dec %.17g" % td[k, 5], "radius dev %.17g replay %.17g prev %.17g" % (td[k, 6], rr[k], td[k - 1, 6]), "cube CR %.17g pow %.17g ttt %.17g" % (float(Fraction(t) ** 3), math.pow(t, 3), t * t * t))
        break
