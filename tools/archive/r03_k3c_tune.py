#!/usr/bin/env python3
"""K2c / K3c tuning at C5: `time_kernels` of the installed library over environment settings given as KEY=VALUE[,KEY=VALUE...] arguments
(GSFM_COL_WGS = workgroups per product, GSFM_K3C_LDS_PAD / GSFM_K2C_LDS_PAD = unused dynamic LDS, i.e. fewer workgroups per CU).
Results: profiles/r03_k3c_tuning.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess
if len(sys.argv) > 1 and sys.argv[1] != "--one":
    for spec in sys.argv[1:]:
        env = dict(os.environ)
        for kv in spec.split(","):
            if kv and kv != "default":
                k, v = kv.split("="); env[k] = v
        r = subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True)
        print(spec, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
    sys.exit(0)
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(MAGSACWeightBasedLoss(0.02))
kt = p.time_kernels(g["init_aa"], reps=20)
print({k: round(1e3 * v, 1) for k, v in kt.items()})
