"""C5 solve under the PCG variants that exist on the column-sorted layout (round 5, after the 2-byte record): textbook recurrence (4 kernels per
iteration: K3c, finish, two vector kernels) against the single-reduction one (3: K3c with the entry decision, finish, one vector kernel),
host looks every 4 / 6 / 8 iterations.  Prints ms per solve (best of 3 x 5), LM / PCG iterations, final cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem

g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(MAGSACWeightBasedLoss(0.02))
init = g["init_aa"]
ref = None
for sr in (0, 1):
    for chunk in (8, 4, 6):
        opts = dict(pcg_single_reduction=sr, cg_check_interval=chunk)
        best, s = 1e9, None
        p.solve(init, **opts)
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                rot, s = p.solve(init, **opts)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 5)
        if ref is None: ref = rot
        d = synth.angular_distance(synth.align_rotations(rot, ref), ref)
        print("single_reduction=%d check_interval=%d: %.3f ms  LM %d  PCG %d (launched %d)  cost %.10e  vs first %.1e rad" % (sr, chunk, best * 1e3, s["num_iterations"], s["num_cg_iterations"], s["num_pcg_launched"], s["final_cost"], d.mean()), flush=True)
