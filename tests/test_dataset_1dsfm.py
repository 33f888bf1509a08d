"""1DSfM tracks ingestion and the CalcCovariance driver (SURVEY section 8f rows 2 + 4: io/read_1dsfm.cc:93-372,
src/uncertainty.cpp:3-33,164-198, bind_src/GlobalSfMpy.cpp:623-628).
CPU: the C++ reader against the numpy restatement on a synthetic dataset.  GPU (-m gpu): CalcCovariance against the oracle,
then the unchanged rotation-only pipeline on the dataset it completed."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from conftest import have_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "globalsfmpy_amd"))
sfm = pytest.importorskip("GlobalSfMpy")

from globalsfmpy_amd import dataset_1dsfm as ds, synth  # noqa: E402


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("synthetic_1dsfm"))
    gt = ds.write_synthetic_dataset(path, n_cams=12, n_points=600, seed=3)
    return path, gt


def test_host_reader_flattens_matches_like_the_numpy_restatement(dataset):
    path, gt = dataset
    got = sfm.Read1DSFMEdgeMatches(path)
    want = ds.read_edge_matches(path)
    assert [tuple(e) for e in got["edges"]] == want["edges"] == gt["edges"]  # the view outside cc.txt is gone
    assert np.array_equal(got["match_ptr"], want["match_ptr"])
    assert np.array_equal(got["intrinsics"], want["intrinsics"])
    assert np.array_equal(got["trans"], want["trans"])
    assert np.abs(got["rot"] - want["rot"]).max() < 1e-12
    # same matches per edge (order inside an edge is track order in both)
    assert np.array_equal(got["matches"], want["matches"])
    assert got["match_ptr"][-1] > 10000 and np.diff(got["match_ptr"].astype(np.int64)).min() >= 15


def test_view_graph_carries_common_track_counts_and_focal_lengths(dataset):
    path, gt = dataset
    vg, rec, cov = sfm.ViewGraph(), sfm.Reconstruction(), sfm.MapEdgesCovariance()
    sfm.Read1DSFM(path, rec, vg, cov)
    want = ds.read_edge_matches(path)
    assert vg.NumViews() == 12 and vg.NumEdges() == len(want["edges"])
    counts = np.diff(want["match_ptr"].astype(np.int64))
    for e, (i, j) in enumerate(want["edges"]):
        info = vg.GetEdge(i, j)
        assert info.num_verified_matches == counts[e] == info.visibility_score  # read_1dsfm.cc:359-366
        assert info.focal_length_1 == want["intrinsics"][e, 0] and info.focal_length_2 == want["intrinsics"][e, 3]
    # views without an EXIF entry fall back to 1.2 * principal point x (:347-357)
    i, j = want["edges"][0]
    assert (i % 2 == 0) and abs(vg.GetEdge(i, j).focal_length_1 - gt["focal"][i]) < 1e-9
    odd = next(e for e, (a, b) in enumerate(want["edges"]) if a % 2 == 1)
    a = want["edges"][odd][0]
    assert abs(want["intrinsics"][odd, 0] - 1.2 * gt["principal_point"][a, 0]) < 1e-9


def test_spanning_tree_initialisation_prefers_edges_with_more_common_tracks(dataset):
    path, gt = dataset
    vg, rec, cov = sfm.ViewGraph(), sfm.Reconstruction(), sfm.MapEdgesCovariance()
    sfm.Read1DSFM(path, rec, vg, cov)
    o = sfm.MapViewIdVector3d()
    assert sfm.OrientationsFromMaximumSpanningTree(vg, o)
    got = np.array([o[k] for k in range(12)])
    aligned = synth.align_rotations(got, gt["rotations_aa"])
    # chained two-view estimates (0.01 rad noise each): within a few hundredths of a radian of the truth
    assert synth.angular_distance(aligned, gt["rotations_aa"]).max() < 0.08


@pytest.fixture(scope="module")
def colmap_export(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("synthetic_colmap"))
    gt = ds.write_synthetic_colmap_export(path, n_cams=10, n_points=500, seed=2)
    return path, gt


def _colmap_builder(path):
    b = sfm.ReconstructionBuilder(sfm.ReconstructionBuilderOptions(), sfm.FeaturesAndMatchesDatabase(path + "/database"))
    sfm.AddColmapMatchesToReconstructionBuilder(path + "/two_views.txt", path + "/images/*.JPG", b)
    b.CheckView()
    return b


def test_image_size_probe_reads_jpeg_and_png_headers(tmp_path):
    ds._fake_jpeg(str(tmp_path / "a.JPG"), 6048, 4032)
    ds._fake_png(str(tmp_path / "b.png"), 1234, 567)
    assert sfm.ReadImageSize(str(tmp_path / "a.JPG")) == (6048, 4032)
    assert sfm.ReadImageSize(str(tmp_path / "b.png")) == (1234, 567)
    (tmp_path / "c.JPG").write_bytes(b"not an image")
    with pytest.raises(RuntimeError):
        sfm.ReadImageSize(str(tmp_path / "c.JPG"))


def test_colmap_two_views_reader_matches_the_numpy_restatement(colmap_export):
    path, gt = colmap_export
    b = _colmap_builder(path)
    vg, rec = b.get_view_graph(), b.get_reconstruction()
    sizes = {n: sfm.ReadImageSize(os.path.join(path, "images", n)) for n in gt["names"]}
    want = ds.read_colmap_two_views(os.path.join(path, "two_views.txt"), sizes)
    got = rec.MatchedFeatures()
    assert want["names"] == gt["names"] == [rec.ViewNames()[k] for k in range(gt["n_cams"])]   # ids by first appearance
    assert [tuple(e) for e in got["edges"]] == want["edges"] and vg.NumEdges() == len(want["edges"]) == gt["num_pairs"]
    for k in ("match_ptr", "matches", "intrinsics"):
        assert np.array_equal(got[k], want[k]), k
    assert np.abs(got["rot"] - want["rot"]).max() < 1e-15 and np.abs(got["trans"] - want["trans"]).max() < 1e-14
    # principal point = half the image size (read_colmap_posegraph.cpp:77-85)
    assert np.array_equal(got["intrinsics"][0, 1:3], gt["principal_point"][want["edges"][0][0]])
    # pairs listed larger-view-first were swapped to smaller->larger (SwapCameras): consistent with the ground truth
    R = synth.quat_to_matrix(synth.aa_to_quat(gt["rotations_aa"]))
    rel = synth.quat_to_aa(synth.matrix_to_quat(np.array([R[j] @ R[i].T for i, j in want["edges"]])))
    assert synth.angular_distance(rel, got["rot"]).max() < 0.06
    for e, (i, j) in enumerate(want["edges"]):
        info = vg.GetEdge(i, j)
        assert info.focal_length_1 == gt["focal"][i] and info.focal_length_2 == gt["focal"][j]
        assert info.num_verified_matches == info.num_homography_inliers == int(want["match_ptr"][e + 1] - want["match_ptr"][e])


def test_store_covariance_rot_needs_matches():
    with pytest.raises(RuntimeError, match="no matched features"):
        sfm.store_covariance_rot("/tmp", sfm.Reconstruction(), sfm.ViewGraph())


@pytest.mark.gpu
def test_colmap_branch_covariances_and_pipeline(colmap_export, oracle):
    path, gt = colmap_export
    # scripts/get_covariance_from_colmap.py, same calls
    b = _colmap_builder(path)
    vg, rec = b.get_view_graph(), b.get_reconstruction()
    stats = sfm.store_covariance_rot(path, rec, vg)
    em = rec.MatchedFeatures()
    E = len(em["edges"])
    assert stats["num_written"] == E == gt["num_pairs"]
    cov = sfm.MapEdgesCovariance()
    sfm.ReadCovariance(path, cov)
    want = oracle.estimate_rotation_covariances(em["match_ptr"], em["matches"], em["intrinsics"], em["rot"], em["trans"])
    for e, key in enumerate(map(tuple, em["edges"])):
        C, r = np.array(cov[key][0]), np.array(cov[key][1])
        assert np.abs(r - want["rotation"][e]).max() < 1e-8
        assert np.abs(C - want["cov"][e]).max() < 1e-5 * np.abs(want["cov"][e]).max()
    # then the use1DSfM=False branch of the pipeline on the completed export
    spec = importlib.util.spec_from_file_location("rotation_only_pipeline", os.path.join(ROOT, "examples", "rotation_only_pipeline.py"))
    pipe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pipe)
    from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
    rec2, est = pipe.sfm_pipeline(None, path, MAGSACWeightBasedLoss(0.02), sfm.RotationErrorType.ANGLE_AXIS_COVARIANCE, use1DSfM=False)
    o = rec2.EstimatedOrientations()
    got = np.array([o[k] for k in range(gt["n_cams"])])
    err = synth.angular_distance(synth.align_rotations(got, gt["rotations_aa"]), gt["rotations_aa"])
    assert err.max() < 0.05, err


@pytest.mark.skipif(have_gpu(), reason="loud-failure check is for boxes without a device")
def test_calc_covariance_fails_loudly_without_a_device(dataset):
    with pytest.raises(RuntimeError):
        sfm.CalcCovariance(dataset[0])


def test_calc_covariance_reports_unreadable_datasets(tmp_path):
    with pytest.raises(RuntimeError, match="list.txt"):
        sfm.CalcCovariance(str(tmp_path))


@pytest.mark.gpu
def test_calc_covariance_matches_the_oracle_and_completes_the_pipeline(dataset, oracle):
    path, gt = dataset
    if os.path.exists(os.path.join(path, "covariance_rot.txt")):
        os.remove(os.path.join(path, "covariance_rot.txt"))
    stats = sfm.CalcCovariance(path)
    em = ds.read_edge_matches(path)
    E = len(em["edges"])
    assert stats["num_edges"] == E and stats["num_matches"] == em["match_ptr"][-1]
    assert stats["num_written"] == E and stats["num_skipped"] == 0 and stats["num_singular"] == 0
    cov = sfm.MapEdgesCovariance()
    sfm.ReadCovariance(path, cov)
    assert len(cov) == E
    want = oracle.estimate_rotation_covariances(em["match_ptr"], em["matches"], em["intrinsics"], em["rot"], em["trans"])
    assert (want["status"] == 0).all()
    for e, key in enumerate(em["edges"]):
        C, r = cov[key]
        C, r = np.array(C), np.array(r)
        assert np.abs(r - want["rotation"][e]).max() < 1e-8                       # the refined relative rotation written beside it
        assert np.abs(C - want["cov"][e]).max() < 1e-5 * np.abs(want["cov"][e]).max()
        assert np.abs(C - C.T).max() == 0.0 and np.linalg.eigvalsh(C).min() > 0.0
    # refinement on ~200 matches beats the 0.01 rad two-view noise the file started from, for the pairs with a correct focal
    Rw = synth.quat_to_matrix(synth.aa_to_quat(gt["rotations_aa"]))
    good = [e for e, (i, j) in enumerate(em["edges"]) if i % 2 == 0 and j % 2 == 0]
    gt_rel = synth.quat_to_aa(synth.matrix_to_quat(np.array([Rw[j] @ Rw[i].T for (i, j) in em["edges"]])))
    refined = np.array([np.array(cov[k][1]) for k in em["edges"]])
    assert np.median(synth.angular_distance(refined[good], gt_rel[good])) < 0.3 * np.median(synth.angular_distance(em["rot"][good], gt_rel[good]))

    # the unchanged rotation-only pipeline now finds covariance_rot.txt and runs on it
    spec = importlib.util.spec_from_file_location("rotation_only_pipeline", os.path.join(ROOT, "examples", "rotation_only_pipeline.py"))
    pipe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pipe)
    from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
    rec, est = pipe.sfm_pipeline(None, path, MAGSACWeightBasedLoss(0.02), sfm.RotationErrorType.ANGLE_AXIS_COVARIANCE)
    o = rec.EstimatedOrientations()
    got = np.array([o[k] for k in range(12)])
    err = synth.angular_distance(synth.align_rotations(got, gt["rotations_aa"]), gt["rotations_aa"])
    assert err.max() < 0.05, err
