"""K2c / K3c time against the number of workgroups the column-sorted layout is dealt to (GSFM_COL_WGS, read at problem creation)."""
import os, sys; sys.path.insert(0, "/root/repo")
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
for wgs in [int(x) for x in sys.argv[1:]] or [0]:
    if wgs: os.environ["GSFM_COL_WGS"] = str(wgs)
    else: os.environ.pop("GSFM_COL_WGS", None)
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    p.set_loss(MAGSACWeightBasedLoss(0.02))
    best = {}
    for _ in range(3):
        kt = p.time_kernels(g["init_aa"], reps=10)
        for k, v in kt.items(): best[k] = min(best.get(k, 1e9), v)
    print("GSFM_COL_WGS=%d:" % wgs, {k: round(1e3 * v, 1) for k, v in best.items()}, flush=True)
    p.close()
