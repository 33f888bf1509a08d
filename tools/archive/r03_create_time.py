import os, sys, time
sys.path.insert(0, "/root/repo")
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
for rep in range(3):
    t = time.perf_counter()
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    print("create %.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
    p.close()
