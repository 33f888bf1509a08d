"""Two-level preconditioner (GSFM_PCG_COARSE) on the awkward cases: every Laplacian-form error type, isolated cameras filling whole aggregates,
two disconnected coherent components, a forced coarse space on a uniformly random graph, sigma consensus.  Each against the block-Jacobi solve of
the same problem (same LM iterations, cost to 1e-10, rotations to 1e-8 rad).  usage: coarse_cases.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem


def solve(n, ei, ej, rel, et, loss, init, coarse, cov6=None, iw=None, sigma=False):
    if coarse is None: os.environ.pop("GSFM_PCG_COARSE", None)
    else: os.environ["GSFM_PCG_COARSE"] = coarse
    p = RotationProblem(n, ei, ej, rel, et, cov6=cov6, inlier_weight=iw); p.set_loss(loss)
    # (pcg_forcing=0: this run compares two PRECONDITIONERS on the exact step; loose steps depend on the preconditioner by construction)
    out = p.solve_sigma_consensus(init, 3, 0.05, pcg_forcing=0) if sigma else p.solve(init, pcg_forcing=0)
    p.close()
    return out


def run():
    bad = 0
    g = synth.make_graph(10000, 150000, 3, outlier_frac=0.1, local_window=300)
    cases = []
    for et, loss in ((_abi.ANGLE_AXIS, LF.HuberLoss(0.1)), (_abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02)), (_abi.QUATERNION_COSINE, LF.HuberLoss(0.1)),
                     (_abi.ANGLE_AXIS_COV_INLIERS, LF.SoftLOneLoss(0.1)), (_abi.ANGLE_AXIS_COVTRACE, LF.CauchyLoss(0.2)), (_abi.ROTATION_MAT_FNORM, LF.HuberLoss(0.1))):
        cases.append(("error type %d" % et, g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, loss, g["init_aa"], g["cov6"], g["inlier_weight"], False, None))
    # 1000 isolated cameras appended: whole aggregates without a single edge
    n2 = g["n_cams"] + 1000
    cases.append(("isolated cameras", n2, g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS, LF.HuberLoss(0.1), np.concatenate([g["init_aa"], np.zeros((1000, 3))]), g["cov6"], g["inlier_weight"], False, None))
    # two disconnected coherent components
    h = synth.make_graph(6000, 80000, 5, outlier_frac=0.1, local_window=200)
    cases.append(("two components", g["n_cams"] + 6000, np.concatenate([g["edge_i"], h["edge_i"] + g["n_cams"]]).astype(np.uint32), np.concatenate([g["edge_j"], h["edge_j"] + g["n_cams"]]).astype(np.uint32),
                  np.concatenate([g["rel_aa"], h["rel_aa"]]), _abi.ANGLE_AXIS, LF.HuberLoss(0.1), np.concatenate([g["init_aa"], h["init_aa"]]), None, None, False, None))
    r = synth.make_graph(9000, 150000, 9, outlier_frac=0.2)
    cases.append(("random graph, coarse forced", r["n_cams"], r["edge_i"], r["edge_j"], r["rel_aa"], _abi.ANGLE_AXIS, LF.HuberLoss(0.1), r["init_aa"], None, None, False, "64"))
    cases.append(("sigma consensus", g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS, LF.TrivialLoss(), g["init_aa"], None, None, True, None))
    for name, n, ei, ej, rel, et, loss, init, c6, iw, sigma, force in cases:
        kw = dict(cov6=c6 if c6 is not None and len(c6) == len(ei) else None, iw=iw if iw is not None and len(iw) == len(ei) else None, sigma=sigma)
        r0, s0 = solve(n, ei, ej, rel, et, loss, init, "0", **kw)
        r1, s1 = solve(n, ei, ej, rel, et, loss, init, force, **kw)
        act = np.zeros(n, bool); act[ei] = True; act[ej] = True
        d = synth.angular_distance(r1[act], r0[act]).max() if name != "two components" else max(
            synth.angular_distance(synth.align_rotations(r1[:10000], r0[:10000]), r0[:10000]).max(), synth.angular_distance(synth.align_rotations(r1[10000:], r0[10000:]), r0[10000:]).max())
        ok = s0["num_iterations"] == s1["num_iterations"] and abs(s0["final_cost"] - s1["final_cost"]) <= 1e-10 * abs(s0["final_cost"]) and d <= 1e-8 and np.array_equal(r1[~act], r0[~act])
        bad += not ok
        print("%-28s LM %2d/%2d  PCG %5d -> %5d  cost rel %.1e  max dR %.1e  %s" % (name, s0["num_iterations"], s1["num_iterations"], s0["num_cg_iterations"], s1["num_cg_iterations"],
              abs(s0["final_cost"] - s1["final_cost"]) / abs(s0["final_cost"]), d, "ok" if ok else "MISMATCH"), flush=True)
    # the coarse matrix is summed in fixed point: the same bits on every run
    ra, sa = solve(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), g["init_aa"], None, cov6=g["cov6"])
    rb, sb = solve(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), g["init_aa"], None, cov6=g["cov6"])
    same = np.array_equal(ra, rb) and sa["num_cg_iterations"] == sb["num_cg_iterations"] and sa["final_cost"] == sb["final_cost"]
    bad += not same
    print("two runs bit-identical: %s (PCG %d / %d)" % (same, sa["num_cg_iterations"], sb["num_cg_iterations"]))
    os.environ.pop("GSFM_PCG_COARSE", None)
    print("coarse cases: %d mismatches" % bad)
    return bad


if __name__ == "__main__":
    sys.exit(min(1, run()))
