#!/usr/bin/env python3
"""What ONE rank of an N-rank run of the C5 graph does per kernel, measured on one GPU: rank 0's share of the partition is created with
no-op collective callbacks (the numbers inside the vectors are then meaningless; the kernels, their grids and their memory traffic are
the real ones) and timed with gsfm_rot_time_kernels.  Feeds the per-rank column of DESIGN section 7's cost model."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from globalsfmpy_amd import _abi, synth, sharding
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
noop_g = _abi.ALL_GATHER_FN(lambda ctx, buf, count, stream: 0)
noop_r = _abi.ALL_REDUCE_FN(lambda ctx, buf, count, stream: 0)
worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
for world in worlds:
    if world == 1:
        p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
        init = g["init_aa"]; n_local = len(g["edge_i"])
    else:
        part = sharding.partition_cameras(g["n_cams"], g["edge_i"], g["edge_j"], world)
        ei, ej = part.relabel(g["edge_i"]), part.relabel(g["edge_j"])
        m = sharding.local_edge_mask(part, ei, ej, 0)
        sh = _abi.Shard(0, world, part.width, 0, None, noop_g, noop_r)
        p = RotationProblem(part.n_pad, ei[m], ej[m], g["rel_aa"][m], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"][m], shard=sh)
        init = part.scatter(g["init_aa"]); n_local = int(m.sum())
    p.set_loss(MAGSACWeightBasedLoss(0.02))
    kt = p.time_kernels(init, reps=20)
    print("GSFM_COL_WGS=%s " % os.environ.get("GSFM_COL_WGS", "default") + "ranks %d: rank 0 holds %d edges (%.1f %%), layout form %d; k_cost %.1f us, k_lin %.1f us, k_matvec (+ finish) %.1f us"
          % (world, n_local, 100.0 * n_local / len(g["edge_i"]), p.matvec_bytes()[1], 1e3 * kt["k_cost"], 1e3 * kt["k_lin"], 1e3 * kt["k_matvec"]), flush=True)
    if world > 1:   # a PCG run without communication: wrong numbers, real launches -- kernels + gaps of one rank's iteration
        for graph in (1, 0):
            p.solve(init, max_num_iterations=1, max_cg_iterations=400, pcg_hip_graph=graph)
            r, s = p.solve(init, max_num_iterations=1, max_cg_iterations=400, pcg_hip_graph=graph)
            print("    one LM step, PCG capped at 400: %d iterations launched, %.2f ms of PCG -> %.1f us per iteration without the collective (%s)"
                  % (s["num_pcg_launched"], s["t_cg_ms"], 1e3 * s["t_cg_ms"] / max(1, s["num_pcg_launched"]), "hipGraph replay" if graph else "plain launches"), flush=True)
    p.close()
