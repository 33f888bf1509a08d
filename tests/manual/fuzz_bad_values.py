"""Non-finite and degenerate VALUES through the C-ABI (the argument checks themselves are in tests/test_abi.py): NaN / Inf measurements,
zero / indefinite / NaN covariances, zero and negative inlier weights, NaN initial rotations.  Nothing may crash or hang; the outcome must be
an error status or a terminated solve whose summary says so.  usage: fuzz_bad_values.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem, SolverError


def run():
    g = synth.make_graph(60, 400, 5, outlier_frac=0.1)
    bad = 0
    cases = []
    def poison(arr, val, frac=0.02, seed=0):
        a = arr.copy(); idx = np.random.default_rng(seed).choice(a.shape[0], max(1, int(frac * a.shape[0])), replace=False); a[idx] = val; return a
    for val in (np.nan, np.inf, -np.inf, 1e300, 1e-320):
        cases.append(("rel_aa=%r" % val, dict(rel_aa=poison(g["rel_aa"], val)), _abi.ANGLE_AXIS_COVARIANCE))
        cases.append(("cov6=%r" % val, dict(cov6=poison(g["cov6"], val)), _abi.ANGLE_AXIS_COVARIANCE))
        cases.append(("cov6 trace type=%r" % val, dict(cov6=poison(g["cov6"], val)), _abi.ANGLE_AXIS_COVTRACE))
        cases.append(("inlier_weight=%r" % val, dict(inlier_weight=poison(g["inlier_weight"], val)), _abi.ANGLE_AXIS_COV_INLIERS))
        cases.append(("init=%r" % val, dict(init=poison(g["init_aa"], val)), _abi.ANGLE_AXIS))
    cases.append(("cov6=0", dict(cov6=poison(g["cov6"], 0.0)), _abi.ANGLE_AXIS_COVARIANCE))
    cases.append(("cov6 negative definite", dict(cov6=poison(g["cov6"], np.array([-1e-8, -1e-8, -1e-8, 0, 0, 0]))), _abi.ANGLE_AXIS_COVARIANCE))
    cases.append(("cov6 indefinite", dict(cov6=poison(g["cov6"], np.array([1e-8, 1e-8, 1e-8, 5e-8, 0, 0]))), _abi.ANGLE_AXIS_COV_INLIERS))
    cases.append(("inlier_weight=0", dict(inlier_weight=poison(g["inlier_weight"], 0.0)), _abi.ANGLE_AXIS_INLIERS))
    cases.append(("inlier_weight<0", dict(inlier_weight=poison(g["inlier_weight"], -3.0)), _abi.ANGLE_AXIS_INLIERS))
    cases.append(("all rel_aa NaN", dict(rel_aa=np.full_like(g["rel_aa"], np.nan)), _abi.ANGLE_AXIS))
    for name, over, et in cases:
        a = dict(rel_aa=g["rel_aa"], cov6=g["cov6"], inlier_weight=g["inlier_weight"], init=g["init_aa"]); a.update(over)
        for loss in (LF.HuberLoss(0.1), LF.MAGSACWeightBasedLoss(0.02)):
            for dense in (512, 0):
                try:
                    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], a["rel_aa"], et, cov6=a["cov6"], inlier_weight=a["inlier_weight"]); p.set_loss(loss)
                    r, s = p.solve(a["init"], dense_cholesky_max_cams=dense, max_num_iterations=30)
                    finite = bool(np.isfinite(r).all())
                    verdict = "%s it=%d nonfinite=%d rotations finite=%s" % (s["termination_name"], s["num_iterations"], s["nonfinite"], finite)
                    # a run that CLAIMS convergence must have produced finite rotations
                    if s["termination_name"] in ("FUNCTION_TOLERANCE", "GRADIENT_TOLERANCE", "PARAMETER_TOLERANCE") and not finite:
                        bad += 1; verdict += "   <-- converged with non-finite rotations"
                    p.close()
                except (SolverError, ValueError) as e:
                    verdict = "refused: %s" % str(e)[:90]
                print("%-28s %-22s dense<=%-3d %s" % (name, type(loss).__name__, dense, verdict), flush=True)
    print("bad-value run: %d inconsistent outcomes" % bad)
    return bad


if __name__ == "__main__":
    sys.exit(min(1, run()))
