"""Madrid (real graph, synthetic covariances) under MAGSAC to convergence: the device's two LM controls against the unperturbed CPU oracle
(round-4 review: 1.94e-6 rad; round 5 asks whether the radius law's rounding is what is left)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from oracle import pyoracle
from test_gpu_fullsize import _madrid_component
g = _madrid_component(os.path.join(ROOT, "tests", "golden"))
loss = LF.MAGSACWeightBasedLoss(0.02)
o = pyoracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); o.set_loss(loss)
ro, so = o.solve(g["init_aa"]); to = o.trace()
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); p.set_loss(loss)
for dc in (0, 1):
    r, s = p.solve(g["init_aa"], lm_device_control=dc); t = p.trace()
    d = synth.angular_distance(synth.align_rotations(r, ro), ro)
    n = min(len(t), len(to))
    first = next((k for k in range(n) if t[k, 6] != to[k, 6]), n)
    firstc = next((k for k in range(n) if abs(t[k, 1] - to[k, 1]) > 1e-9 * to[k, 1]), n)
    print("device control %d: %d LM it (oracle %d), %.2e rad mean / %.2e max from the oracle; radius column equal to the oracle's up to row %d of %d; cost within 1e-9 up to row %d"
          % (dc, s["num_iterations"], so["num_iterations"], d.mean(), d.max(), first, n, firstc))
    for k in range(max(0, first - 1), min(n, first + 3)):
        print("    row %2d: rel_dec dev %.17g oracle %.17g   radius dev %.17g oracle %.17g" % (k, t[k, 5], to[k, 5], t[k, 6], to[k, 6]))
