import hashlib

import numpy as np

from globalsfmpy_amd import synth


def test_generator_is_deterministic_and_well_formed():
    a = synth.make_graph(500, 6000, seed=42, outlier_frac=0.3)
    b = synth.make_graph(500, 6000, seed=42, outlier_frac=0.3)
    for k in ("edge_i", "edge_j", "rel_aa", "cov6", "init_aa", "gt_aa"):
        assert np.array_equal(a[k], b[k])
    assert (a["edge_i"] < a["edge_j"]).all()                       # ViewIdPair keys are (min, max)
    keys = a["edge_i"].astype(np.int64) * 500 + a["edge_j"]
    assert np.unique(keys).size == 6000                            # distinct pairs
    assert abs(a["is_outlier"].mean() - 0.3) < 1e-3
    assert not a["is_outlier"][:499].any()                         # the spanning chain stays clean
    h = hashlib.sha256(a["edge_i"].tobytes() + a["edge_j"].tobytes()).hexdigest()
    assert len(h) == 64


def test_whitened_inlier_scale():
    # s = 1e-8 n^T (kappa Sigma)^-1 n should average 1e-4 on inliers
    g = synth.make_graph(300, 5000, seed=1)
    q_gt = synth.aa_to_quat(g["gt_aa"])
    q_rel = synth.aa_to_quat(g["rel_aa"])
    err = synth.quat_mul(synth.quat_mul(q_gt[g["edge_j"]], synth.quat_conj(q_gt[g["edge_i"]])), synth.quat_conj(q_rel))
    n = synth.quat_to_aa(err)
    c = g["cov6"]
    S = np.zeros((5000, 3, 3))
    S[:, 0, 0], S[:, 1, 1], S[:, 2, 2] = c[:, 0], c[:, 1], c[:, 2]
    S[:, 0, 1] = S[:, 1, 0] = c[:, 3]; S[:, 0, 2] = S[:, 2, 0] = c[:, 4]; S[:, 1, 2] = S[:, 2, 1] = c[:, 5]
    s = 1e-8 * np.einsum("ei,eij,ej->e", n, np.linalg.inv(S), n)
    assert 0.8e-4 < s.mean() < 1.2e-4


def test_alignment_removes_the_gauge():
    g = synth.make_graph(50, 200, seed=3)
    gauge = synth.aa_to_quat(np.array([0.3, -0.2, 0.5]))
    moved = synth.quat_to_aa(synth.quat_mul(synth.aa_to_quat(g["gt_aa"]), gauge))   # R_k * G
    assert synth.angular_distance(moved, g["gt_aa"]).min() > 0.1
    back = synth.align_rotations(moved, g["gt_aa"])
    assert synth.angular_distance(back, g["gt_aa"]).max() < 1e-12


def test_spanning_tree_init_is_a_usable_start_beyond_46k_cameras():
    """OrientationsFromMaximumSpanningTree-style initialisation of the benchmark graphs: composed along a maximum spanning tree
    by match count, inlier pairs dominating.  46 341^2 overflows int32: scipy's int32 predecessor array once turned every key lookup into
    garbage at the benchmark size (mean initial error 120 degrees = random rotations)."""
    g = synth.make_graph(50000, 200000, 3, outlier_frac=0.3)
    init, m = synth.spanning_tree_init(g, 3)
    assert m[g["is_outlier"]].max() <= 150 and m[~g["is_outlier"]].min() >= 150
    d = synth.angular_distance(synth.align_rotations(init, g["gt_aa"]), g["gt_aa"])
    assert np.rad2deg(d.mean()) < 40.0, np.rad2deg(d.mean())     # a deep random tree of 1-degree measurements: tens of degrees, not 120
    # noise-free measurements: the composition is exact
    g0 = synth.make_graph(3000, 20000, 5, outlier_frac=0.2, noise=False)
    init0, _ = synth.spanning_tree_init(g0, 5)
    d0 = synth.angular_distance(synth.align_rotations(init0, g0["gt_aa"]), g0["gt_aa"])
    assert d0.max() < 1e-9
