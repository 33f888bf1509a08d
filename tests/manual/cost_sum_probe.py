import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from oracle import pyoracle
n, e = int(sys.argv[1]), int(sys.argv[2])
g = synth.make_graph(n, e, 77, outlier_frac=0.3)
p = RotationProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
o = pyoracle.OracleProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); o.set_loss(LF.MAGSACWeightBasedLoss(0.02))
d, r = p.residuals(g["init_aa"]), o.residuals(g["init_aa"])
fd, fo = math.fsum(0.5 * d["rho"][:, 0]), math.fsum(0.5 * r["rho"][:, 0])
print("device cost %.15e  fsum(device rho) %.15e  rel %.2e" % (d["cost"], fd, abs(d["cost"] - fd) / fd))
print("oracle cost %.15e  fsum(oracle rho) %.15e  rel %.2e" % (r["cost"], fo, abs(r["cost"] - fo) / fo))
dr = d["rho"][:, 0] - r["rho"][:, 0]
print("rho0: max abs diff %.3e, sum of diffs %.3e, #edges differing by > 1e-9: %d, by > 1e-3: %d" % (np.abs(dr).max(), dr.sum(), (np.abs(dr) > 1e-9).sum(), (np.abs(dr) > 1e-3).sum()))
k = np.argsort(-np.abs(dr))[:5]
for i in k:
    print("  edge %d: s dev %.17e ora %.17e  rho dev %.12e ora %.12e  x=1000 s/(2 sigma^2) = %.9f" % (i, d["s"][i], r["s"][i], d["rho"][i, 0], r["rho"][i, 0], 1000.0 * r["s"][i] / (2 * 0.02 ** 2)))
_, s = p.solve(g["init_aa"]); _, so = (None, None)
print("cost-only sweep (trial cost path) at the start: %.15e" % p.trace()[0][1])
