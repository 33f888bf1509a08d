"""Helper of tests/test_gpu_control_flow.py: a fixed set of solves (PCG steps, every termination kind that is cheap to provoke), results to an .npz.
The environment of the process selects the host-side control variant (GSFM_PHASE_TIMERS)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from globalsfmpy_amd import _abi, synth  # noqa: E402
from globalsfmpy_amd import loss_functions as LF  # noqa: E402
from globalsfmpy_amd.solver import RotationProblem  # noqa: E402


def main(out):
    res = {}

    def run(name, g, error_type, loss, init, cov=True, **opts):
        p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], error_type, cov6=g["cov6"] if cov else None)
        if loss is not None:
            p.set_loss(loss)
        r, s = p.solve(init, **opts)
        res[name + "_rot"] = r
        res[name + "_trace"] = p.trace()
        res[name + "_sum"] = np.array([s["num_iterations"], s["termination"], s["final_cost"], s["final_gradient_max_norm"], s["num_cg_iterations"],
                                       s["num_successful_steps"], s["num_unsuccessful_steps"]], dtype=np.float64)
        p.close()

    pcg = dict(dense_cholesky_max_cams=0, dense_cholesky_auto_cams=0)
    # (a) noisy graph, robust loss: ends on the function tolerance, loose and continued PCG solves on the way
    g = synth.make_graph(3000, 60000, seed=11, outlier_frac=0.2)
    run("magsac", g, _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), g["init_aa"], **pcg)
    # (b) noise-free measurements, plain least squares from a 2-degree start: ends on the GRADIENT tolerance (the test that is read late)
    g0 = synth.make_graph(800, 12000, seed=5, outlier_frac=0.0, noise=False)
    run("exact_data", g0, _abi.ANGLE_AXIS, LF.TrivialLoss(), g0["init_aa"], cov=False, **pcg)
    # (c) a far start under a non-robust loss over outliers: rejected steps, shrinking and growing radius
    g = synth.make_graph(1500, 30000, seed=23, outlier_frac=0.3)
    far = g["gt_aa"] + np.random.default_rng(3).standard_normal(g["gt_aa"].shape) * 1.0   # (the oracle: 18 accepted, 7 rejected steps, stops at the cap)
    run("far", g, _abi.ANGLE_AXIS, LF.HuberLoss(0.05), far, cov=False, max_num_iterations=25, **pcg)
    # (d) the column-sorted layout forced on a mid-size graph (K2c / K3c tasks), exact PCG schedule
    os.environ["GSFM_K3_COLSORT"] = "1"
    g = synth.make_graph(2500, 120000, seed=31, outlier_frac=0.25)
    run("colsort", g, _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), g["init_aa"], pcg_forcing=0, **pcg)
    os.environ.pop("GSFM_K3_COLSORT", None)
    # (e) exact Cholesky steps under device control (small graphs): a plain run, and one whose first factorisations break down (trust region
    # of 1e20: no damping, the gauge null space stays) so that PCG steps under host control sit between exact ones
    g = synth.make_graph(300, 6000, seed=9, outlier_frac=0.2)
    run("exact_steps", g, _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), g["init_aa"])
    g = synth.make_graph(60, 400, 3, outlier_frac=0.1)
    run("broken_factor", g, _abi.ANGLE_AXIS, LF.HuberLoss(0.1), g["init_aa"], cov=False, initial_trust_region_radius=1e20, max_trust_region_radius=1e20, pcg_forcing=0)
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1])
