#!/usr/bin/env python3
"""End to end, on the device: two-view matches -> per-edge rotation covariances (gsfm_cov_estimate) ->
uncertainty-whitened robust rotation averaging (gsfm_rot_solve, ANGLE_AXIS_COVARIANCE + MAGSAC), the pipeline of
"Revisiting Rotation Averaging: Uncertainties and Robust Losses" on a synthetic scene with heterogeneous edge quality.
usage: uncertainty_pipeline.py [n_cams] [n_edges]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from globalsfmpy_amd import _abi, covariance as cv, synth  # noqa: E402
from globalsfmpy_amd import loss_functions as LF  # noqa: E402
from globalsfmpy_amd.solver import RotationProblem  # noqa: E402


def make_scene(n_cams, n_edges, seed, outlier_frac=0.1):
    rng = np.random.Generator(np.random.PCG64(seed))
    gt_aa = 0.3 * rng.uniform(-1, 1, (n_cams, 3))
    Rw = synth.quat_to_matrix(synth.aa_to_quat(gt_aa))                  # world -> camera
    centers = np.c_[rng.uniform(-3, 3, n_cams), rng.uniform(-1, 1, n_cams), rng.uniform(-1, 1, n_cams)]
    ei, ej = synth.make_edges(rng, n_cams, n_edges)
    focal = rng.uniform(900, 1500, n_cams)
    pp = np.c_[rng.uniform(500, 700, n_cams), rng.uniform(350, 450, n_cams)]
    ptr, ms, Ks, r0, t0 = [0], [], [], [], []
    is_out = np.zeros(n_edges, dtype=bool)
    for e, (i, j) in enumerate(zip(ei, ej)):
        # heterogeneous edges: match counts 12..300, pixel noise 0.3..3 px, points spread narrow..wide
        n = int(rng.choice([12, 25, 60, 150, 300]))
        noise = float(rng.choice([0.3, 0.7, 1.5, 3.0]))
        spread = float(rng.choice([0.4, 1.0, 2.5]))
        X = np.c_[rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n), rng.uniform(6, 12, n)]  # world points in front of the rig
        Xi = (Rw[i] @ (X - centers[i]).T).T
        Xj = (Rw[j] @ (X - centers[j]).T).T
        xi = np.c_[focal[i] * Xi[:, 0] / Xi[:, 2] + pp[i, 0], focal[i] * Xi[:, 1] / Xi[:, 2] + pp[i, 1]] + noise * rng.standard_normal((n, 2))
        xj = np.c_[focal[j] * Xj[:, 0] / Xj[:, 2] + pp[j, 0], focal[j] * Xj[:, 1] / Xj[:, 2] + pp[j, 1]] + noise * rng.standard_normal((n, 2))
        Rij = Rw[j] @ Rw[i].T
        t = Rw[i] @ (centers[i] - centers[j])                          # X_j = R_ij (X_i + t)
        t /= np.linalg.norm(t)
        q = synth.matrix_to_quat(Rij[None])[0]
        init = synth.quat_to_aa(synth.quat_mul(synth.aa_to_quat(0.01 * rng.standard_normal(3)), q))   # a two-view estimate
        if rng.uniform() < outlier_frac and e >= n_cams - 1:             # gross mismatch: wrong relative rotation, matches of another pair
            is_out[e] = True
            init = synth.quat_to_aa(synth.random_unit_quat(rng, 1)[0])
        ms.append(np.c_[xi, xj]); Ks.append([focal[i], pp[i, 0], pp[i, 1], focal[j], pp[j, 0], pp[j, 1]])
        r0.append(init); t0.append(t + 0.02 * rng.standard_normal(3)); ptr.append(ptr[-1] + n)
    return {"n_cams": n_cams, "edge_i": ei, "edge_j": ej, "gt_aa": gt_aa, "match_ptr": np.array(ptr, dtype=np.uint64),
            "matches": np.ascontiguousarray(np.vstack(ms)), "intrinsics": np.array(Ks), "rot0": np.array(r0), "trans0": np.array(t0),
            "is_outlier": is_out}


def run(n_cams=200, n_edges=3000, seed=4, verbose=True):
    s = make_scene(n_cams, n_edges, seed)
    est = cv.estimate_rotation_covariances(s["match_ptr"], s["matches"], s["intrinsics"], s["rot0"], s["trans0"])
    ok = est["status"] == 0
    rel = np.where(ok[:, None], est["rotation"], s["rot0"])
    cov6 = cv.cov_to_cov6(est["cov"])
    cov6[~ok] = np.array([1e-6, 1e-6, 1e-6, 0, 0, 0])
    rng = np.random.default_rng(1)
    init = synth.quat_to_aa(synth.quat_mul(synth.aa_to_quat(np.deg2rad(3.0) * rng.standard_normal((n_cams, 3))), synth.aa_to_quat(s["gt_aa"])))
    out = {}
    for name, et, loss, kw in (("covariance-whitened MAGSAC", _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02), {"cov6": cov6}),
                               ("unit-weight SoftL1", _abi.ANGLE_AXIS, LF.SoftLOneLoss(0.1), {})):
        p = RotationProblem(n_cams, s["edge_i"], s["edge_j"], rel, et, **kw)
        p.set_loss(loss)
        r, summ = p.solve(init)
        err = np.rad2deg(synth.angular_distance(synth.align_rotations(r, s["gt_aa"]), s["gt_aa"]))
        out[name] = {"mean_deg": float(err.mean()), "median_deg": float(np.median(err)), "iterations": summ["num_iterations"]}
        if verbose:
            print("%-28s mean %.4f deg  median %.4f deg  (%d LM iterations)" % (name, err.mean(), np.median(err), summ["num_iterations"]))
    out["covariance_kernel_ms"] = est["kernel_ms"]
    out["edges_with_covariance"] = int(ok.sum())
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    e = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    print(run(n, e))
