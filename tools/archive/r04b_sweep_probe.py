"""K1 / K2c timings on the benchmark graph: the covariance + MAGSAC problem (C5) and the ANGLE_AXIS problem with scalar weights (the unit- /
scalar-weight specialisations and the sigma-consensus forms).  One line per repetition; used to compare kernel variants on one box."""
import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss, TrivialLoss, SoftLOneLoss
from globalsfmpy_amd.solver import RotationProblem
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(MAGSACWeightBasedLoss(0.02))
for _ in range(reps):
    kt = p.time_kernels(g["init_aa"], reps=10)
    v = p.time_sweep_variants(g["init_aa"], reps=10)
    print("C5 cov+magsac:", {k: round(1e3 * x, 1) for k, x in kt.items()}, {k: round(1e3 * x, 1) for k, x in v.items() if x > 0}, flush=True)
p.close()
p6 = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
p6.set_loss(TrivialLoss())
p6.set_edge_weights(np.ones(len(g["edge_i"])))
for _ in range(reps):
    v = p6.time_sweep_variants(g["init_aa"], reps=10)
    print("AA scalar-weight trivial:", {k: round(1e3 * x, 1) for k, x in v.items() if x > 0}, flush=True)
p6.close()
p7 = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
p7.set_loss(SoftLOneLoss(0.1))
for _ in range(reps):
    kt = p7.time_kernels(g["init_aa"], reps=10)
    v = p7.time_sweep_variants(g["init_aa"], reps=10)
    print("AA unit-weight softl1:", {k: round(1e3 * x, 1) for k, x in kt.items()}, {k: round(1e3 * x, 1) for k, x in v.items() if x > 0}, flush=True)
