#!/usr/bin/env python3
"""Round-3 A/B probe on the C5 problem: K2 variants (GSFM_K2_FAST), K3 row-major vs column-sorted (GSFM_K3_COLSORT), K1 variants, one solve each."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss, TrivialLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
out = {}
for cs in ("0", "1"):
    os.environ["GSFM_K3_COLSORT"] = cs
    os.environ["GSFM_CREATE_TIMING"] = "1"
    t = time.perf_counter()
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    out["create_s_colsort" + cs] = time.perf_counter() - t
    p.set_loss(MAGSACWeightBasedLoss(0.02))
    for mode, free in (("0", "0"), ("1", "0")):
        os.environ["GSFM_K2_FAST"] = mode
        os.environ["GSFM_K2C_FREE"] = free
        kt = p.time_kernels(g["init_aa"], reps=10)
        out["colsort%s_k2fast%s_free%s" % (cs, mode, free)] = {k: round(1e3 * v, 1) for k, v in kt.items()}
        print(cs, mode, free, out["colsort%s_k2fast%s_free%s" % (cs, mode, free)], flush=True)
    os.environ["GSFM_K2C_FREE"] = "0"
    os.environ["GSFM_K2_FAST"] = "1"
    p.solve(g["init_aa"])
    t = time.perf_counter(); rot, s = p.solve(g["init_aa"]); dt = time.perf_counter() - t
    out["solve_colsort" + cs] = {"ms": 1e3 * dt, "lm": s["num_iterations"], "cg": s["num_cg_iterations"], "cost": s["final_cost"], "gpu_ms": [s["t_linearize_ms"], s["t_sweep_ms"], s["t_cg_ms"]]}
    print(out["solve_colsort" + cs], flush=True)
    if cs == "0":
        rot0 = rot
    else:
        d = synth.angular_distance(synth.align_rotations(rot, rot0), rot0)
        out["colsort_vs_rowmajor_rad"] = [float(d.mean()), float(d.max())]
    v = p.time_sweep_variants(g["init_aa"], reps=10)
    out["k1_variants_colsort" + cs] = {k: round(1e3 * x, 1) for k, x in v.items()}
    print(out["k1_variants_colsort" + cs], flush=True)
    p.close()
p6 = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
p6.set_loss(TrivialLoss())
p6.set_edge_weights(np.ones(g["edge_i"].size))
v6 = p6.time_sweep_variants(g["init_aa"], reps=10)
out["sigma_variants"] = {k: round(1e3 * x, 1) for k, x in v6.items()}
print(json.dumps(out, indent=1))
