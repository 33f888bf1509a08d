import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from globalsfmpy_amd import synth, _abi
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
np.set_printoptions(linewidth=250, precision=4)
def show(name, g, et, loss, init):
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"], inlier_weight=g["inlier_weight"])
    p.set_loss(loss)
    r0, s0 = p.solve(init, pcg_forcing=0)
    t0 = p.trace()
    r1, s1 = p.solve(init)
    t1 = p.trace()
    d = synth.angular_distance(synth.align_rotations(r1, r0), r0)
    print("%s: exact %d LM %d PCG; default %d LM %d PCG (%d inexact, %d refined) dR mean %.1e max %.1e" % (name, s0["num_iterations"], s0["num_cg_iterations"], s1["num_iterations"], s1["num_cg_iterations"], s1["num_inexact_steps"], s1["num_forcing_refinements"], d.mean(), d.max()))
    print("   [it, cost, dcost, |dx|, rel_dec, radius, cg] exact | default cg")
    for i in range(min(len(t0), len(t1))):
        a, b = t0[i], t1[i]
        print("   %3d %.10e %.3e %.3e %.4f %.2e %4d | %.3e %.4f %4d" % (a[0], a[1], a[2], a[4], a[5], a[6], a[7], b[4], b[5], b[7]))
    p.close()
g = synth.make_graph(4000, 60000, seed=44, outlier_frac=0.2)
show("t4000 magsac", g, _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), g["init_aa"])
show("t4000 softl1", g, _abi.ANGLE_AXIS, LF.SoftLOneLoss(0.1), g["init_aa"])
show("t4000 qcos huber", g, _abi.QUATERNION_COSINE, LF.HuberLoss(0.1), g["init_aa"])
g = synth.make_graph(10000, 200000, seed=1, outlier_frac=0.3)
show("C2 GM", g, _abi.ANGLE_AXIS, LF.GemanMcClureLoss(0.1, 1.0), g["init_aa"])
show("C2 magsac", g, _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), g["init_aa"])
g = synth.make_graph(100000, 10000000, seed=int(os.environ.get("SEED", "2023")), outlier_frac=0.3)
show("C5", g, _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), g["init_aa"])
sys.path.insert(0, os.path.join(ROOT, "tools"))
init_tree, _ = synth.spanning_tree_init(g, 2023)
show("C5 tree start", g, _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), init_tree)
