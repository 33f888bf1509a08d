"""Evaluation metrics of the rotation stage (reference src/compare_reconstructions.cpp:7-16,147-177,
197-296 and src/read_colmap_posegraph.cpp:5-53): host C++ checked against scipy on CPU."""
import os
import sys

import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "globalsfmpy_amd"))
sfm = pytest.importorskip("GlobalSfMpy")


def _scene(n, seed, noise=0.01, outliers=0):
    rng = np.random.default_rng(seed)
    gt = R.from_rotvec(0.6 * rng.uniform(-1, 1, (n, 3)))
    A = R.from_rotvec([0.3, -0.5, 0.2])
    est = gt * A.inv() * R.from_rotvec(noise * rng.standard_normal((n, 3)))   # est_i * A ~= gt_i
    est_v = est.as_rotvec()
    if outliers:
        est_v[:outliers] = R.random(outliers, random_state=seed).as_rotvec()
    return gt.as_rotvec(), est_v, A.as_rotvec()


def test_angular_difference_is_the_geodesic_angle():
    rng = np.random.default_rng(1)
    for _ in range(50):
        a, b = R.from_rotvec(rng.uniform(-2, 2, 3)), R.from_rotvec(rng.uniform(-2, 2, 3))
        want = (a.inv() * b).magnitude()
        assert abs(sfm.AngularDifference(a.as_rotvec(), b.as_rotvec()) - want) < 1e-12
    assert sfm.AngularDifference([0.1, 0.2, 0.3], [0.1, 0.2, 0.3]) < 1e-15
    assert abs(sfm.AngularDifference([0, 0, 0], [np.pi, 0, 0]) - np.pi) < 1e-12


def test_noise_free_alignment_is_exact():
    gt, est, A = _scene(40, 2, noise=0.0)
    aligned, s = sfm.AlignRotations(list(gt), list(est))
    assert s["converged"] and s["final_cost"] < 1e-20
    assert np.abs(np.array(s["alignment"]) - A).max() < 1e-9
    assert np.abs(np.array(aligned) - gt).max() < 1e-9


def test_robust_alignment_minimum_agrees_with_scipy_cauchy_least_squares():
    gt, est, _ = _scene(120, 3, noise=0.02, outliers=15)
    Re = R.from_rotvec(est)

    def residuals(a):   # gt_i - Log(R_i Exp(a)), the reference's RotationAlignmentError
        return (gt - (Re * R.from_rotvec(a)).as_rotvec()).ravel()

    # per-block Cauchy(0.1) on |r_i|^2: scipy's loss acts per scalar residual, so minimise the block cost directly
    def cost(a):
        r = residuals(a).reshape(-1, 3)
        return 0.5 * (0.01 * np.log1p((r ** 2).sum(1) / 0.01)).sum()

    aligned, s = sfm.AlignRotations(list(gt), list(est))
    a_dev = np.array(s["alignment"])
    from scipy.optimize import minimize
    ref = minimize(cost, a_dev + 1e-3, method="BFGS", options={"gtol": 1e-12})
    assert abs(s["final_cost"] - cost(a_dev)) < 1e-12 * max(1.0, cost(a_dev))          # reports the cost it minimised
    assert cost(a_dev) <= ref.fun + 1e-12 and np.abs(a_dev - ref.x).max() < 1e-6           # a stationary point at least as good
    # the outliers do not drag the alignment: inliers end within noise of the ground truth
    err = np.array([sfm.AngularDifference(g, r) for g, r in zip(gt[15:], np.array(aligned)[15:])])
    assert err.mean() < 0.05
    # and plain L2 (scipy least_squares on the same residuals) is visibly worse on the inliers
    l2 = least_squares(residuals, np.zeros(3)).x
    err_l2 = (R.from_rotvec(gt[15:]).inv() * (Re[15:] * R.from_rotvec(l2))).magnitude()
    assert err.mean() < err_l2.mean()


def _colmap_images_txt(path, names, quats_wxyz, first_id=1):
    with open(path, "w") as f:
        f.write("# Image list with two lines of data per image:\n#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n"
                "#   POINTS2D[] as (X, Y, POINT3D_ID)\n# Number of images: %d, mean observations per image: 100\n" % len(names))
        for k, (n, q) in enumerate(zip(names, quats_wxyz)):
            f.write("%d %.17g %.17g %.17g %.17g 0.5 -1.0 2.0 1 dslr_images/%s\n1.0 2.0 -1\n" % (first_id + k, q[0], q[1], q[2], q[3], n))


def test_compare_orientations_against_colmap_poses(tmp_path):
    gt, est, _ = _scene(30, 5, noise=0.005)
    names = ["DSC_%04d.JPG" % k for k in range(30)]
    q = R.from_rotvec(gt).as_quat()                     # scipy: (x, y, z, w)
    _colmap_images_txt(str(tmp_path / "images.txt"), names, np.c_[q[:, 3], q[:, :3]], first_id=7)
    cvg = sfm.ColmapViewGraph()
    cvg.read_poses(str(tmp_path / "images.txt"))
    assert cvg.num_view == 30 and cvg.image_ids["DSC_0003.JPG"] == 10 and cvg.image_names[7] == "DSC_0000.JPG"   # directory stripped
    assert np.abs(np.array(cvg.poses[7]) - np.r_[0.5, -1.0, 2.0, gt[0]]).max() < 1e-12
    rec = sfm.Reconstruction()
    for k in range(25):                                   # five views stay unestimated
        rec.SetViewName(100 + k, names[k])
        rec.SetOrientation(100 + k, est[k])
    for k in range(25, 30):
        rec.SetViewName(100 + k, names[k])
    common = sfm.FindCommonViewsByNameColmap(cvg, rec)
    assert common == names[:25]
    info = sfm.compare_orientations_colmap(common, cvg, rec, 1.0)
    d = np.array(info.rotation_diff_when_align)
    assert info.common_camera == 25 and d.shape == (25,) and d.mean() < 0.02 and d.max() < 0.05
    # same numbers through the reconstruction-vs-reconstruction entry point
    ref = sfm.Reconstruction()
    for k in range(30):
        ref.SetViewName(k, names[k]); ref.SetOrientation(k, gt[k])
    common2 = sfm.FindCommonViewsByName(ref, rec)
    assert common2 == names[:25]
    d2 = np.array(sfm.compare_orientations(common2, ref, rec, 1.0).rotation_diff_when_align)
    assert np.abs(d - d2).max() < 1e-12
