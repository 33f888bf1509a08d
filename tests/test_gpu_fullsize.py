"""-m gpu: the BASELINE.json configurations at their own sizes.
  C2  10k cams / 200k edges, Geman-McClure: full solve against the CPU oracle.
  C3  ETH3D-terrace stand-in: few tens of views, strongly anisotropic (COLMAP-like) covariances, MAGSAC.
  C4  14 scene-sized disconnected components solved as ONE problem (per-component gauge).
  C5  100k cams / 10M edges: properties that do not need an oracle (gauge invariance, exact recovery of a
      noise-free graph, linearity and symmetry of the normal mat-vec, checksum of checksums)."""
import numpy as np
import pytest

from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF

pytestmark = pytest.mark.gpu


def _dev(g, et, loss, **kw):
    from globalsfmpy_amd.solver import RotationProblem
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"], inlier_weight=g["inlier_weight"], **kw)
    p.set_loss(loss)
    return p


def test_c2_geman_mcclure_10k_200k_against_oracle(oracle):
    g = synth.make_graph(10000, 200000, seed=202, outlier_frac=0.1)
    loss = LF.GemanMcClureLoss(0.3, 1.0)
    dev = _dev(g, _abi.ANGLE_AXIS, loss)
    ora = oracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
    ora.set_loss(loss)
    rd, sd = dev.solve(g["init_aa"])
    ro, so = ora.solve(g["init_aa"])
    assert sd["num_iterations"] == so["num_iterations"] and sd["termination"] == so["termination"]
    assert abs(sd["final_cost"] - so["final_cost"]) <= 1e-9 * so["final_cost"]
    assert synth.angular_distance(synth.align_rotations(rd, ro), ro).mean() <= 1e-6
    err = synth.angular_distance(synth.align_rotations(rd, g["gt_aa"]), g["gt_aa"])
    assert np.rad2deg(err.mean()) < 0.5


def test_c3_anisotropic_covariances_magsac_against_oracle(oracle):
    g = synth.make_graph(40, 300, seed=303, outlier_frac=0.15)
    rng = np.random.default_rng(3)
    A = synth.quat_to_matrix(synth.random_unit_quat(rng, 300))
    sig2 = (np.deg2rad(rng.uniform(0.02, 0.05, (300, 1))) * np.array([[1.0, 8.0, 40.0]])) ** 2   # 1 : 8 : 40 axis ratio
    S = 3e-4 * np.einsum("eij,ej,ekj->eik", A, sig2, A)
    g["cov6"] = np.ascontiguousarray(np.stack([S[:, 0, 0], S[:, 1, 1], S[:, 2, 2], S[:, 0, 1], S[:, 0, 2], S[:, 1, 2]], axis=1))
    loss = LF.MAGSACWeightBasedLoss(0.02)
    dev = _dev(g, _abi.ANGLE_AXIS_COVARIANCE, loss)
    ora = oracle.OracleProblem(40, g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    ora.set_loss(loss)
    a, b = dev.residuals(g["init_aa"]), ora.residuals(g["init_aa"])
    assert np.max(np.abs(a["s"] - b["s"]) / np.maximum(b["s"], 1e-30)) < 1e-10
    rd, sd = dev.solve(g["init_aa"])
    ro, so = ora.solve(g["init_aa"])
    assert abs(sd["num_iterations"] - so["num_iterations"]) <= 1
    assert abs(sd["final_cost"] - so["final_cost"]) <= 1e-5 * so["final_cost"]
    assert synth.angular_distance(synth.align_rotations(rd, ro), ro).mean() <= 1e-6


def test_c4_fourteen_disconnected_scenes_as_one_problem(oracle):
    # scene sizes after thirdparty/TheiaSfM/docs/source/performance.rst:94-112 (Trafalgar's 5288 left out for time)
    sizes = [227, 328, 332, 341, 437, 450, 553, 572, 577, 733, 789, 836, 1084, 2152]
    parts, off = [], 0
    for k, n in enumerate(sizes):
        g = synth.make_graph(n, 12 * n, seed=400 + k, outlier_frac=0.1)
        parts.append((off, g))
        off += n
    N = off
    ei = np.concatenate([g["edge_i"] + o for o, g in parts]).astype(np.uint32)
    ej = np.concatenate([g["edge_j"] + o for o, g in parts]).astype(np.uint32)
    rel = np.concatenate([g["rel_aa"] for _, g in parts])
    cov = np.concatenate([g["cov6"] for _, g in parts])
    init = np.concatenate([g["init_aa"] for _, g in parts])
    from globalsfmpy_amd.solver import RotationProblem
    loss = LF.HuberLoss(0.1)
    dev = RotationProblem(N, ei, ej, rel, _abi.ANGLE_AXIS_COVTRACE, cov6=cov)
    dev.set_loss(loss)
    ora = oracle.OracleProblem(N, ei, ej, rel, _abi.ANGLE_AXIS_COVTRACE, cov6=cov)
    ora.set_loss(loss)
    rd, sd = dev.solve(init)
    ro, so = ora.solve(init)
    assert sd["num_iterations"] == so["num_iterations"]
    assert abs(sd["final_cost"] - so["final_cost"]) <= 1e-8 * so["final_cost"]
    for o, g in parts:   # every component has its own gauge
        sl = slice(o, o + g["n_cams"])
        assert synth.angular_distance(synth.align_rotations(rd[sl], ro[sl]), ro[sl]).mean() <= 1e-6
        err = synth.angular_distance(synth.align_rotations(rd[sl], g["gt_aa"]), g["gt_aa"])
        assert np.rad2deg(err.mean()) < 3.0   # accuracy sanity only (Huber, sparse scenes, 10 % outliers)


@pytest.fixture(scope="module")
def c5():
    return synth.make_graph(100000, 10000000, seed=2023, outlier_frac=0.3)


def test_c5_sweep_is_gauge_invariant_and_sums_its_edges(c5):
    dev = _dev(c5, _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02))
    a = dev.residuals(c5["init_aa"])
    assert abs(0.5 * a["rho"][:, 0].sum() - a["cost"]) <= 1e-12 * a["cost"]          # checksum of checksums
    G = synth.aa_to_quat(np.array([0.4, -0.7, 0.2]))
    moved = synth.quat_to_aa(synth.quat_mul(synth.aa_to_quat(c5["init_aa"]), G))   # R_k G: same relative rotations
    b = dev.residuals(moved)
    assert np.max(np.abs(a["s"] - b["s"]) / np.maximum(a["s"], 1e-12)) < 1e-9
    assert abs(a["cost"] - b["cost"]) <= 1e-10 * a["cost"]


def test_c5_normal_matvec_is_linear_and_symmetric(c5):
    dev = _dev(c5, _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02))
    lin = dev.linearize(c5["init_aa"])
    rng = np.random.default_rng(1)
    u, v = rng.standard_normal((2, c5["n_cams"], 3))
    Au, Av = dev.normal_matvec(u), dev.normal_matvec(v)
    Auv = dev.normal_matvec(0.3 * u - 1.7 * v)
    scale = np.abs(Au).max() + np.abs(Av).max()
    assert np.abs(Auv - (0.3 * Au - 1.7 * Av)).max() <= 1e-12 * scale
    assert abs(np.sum(u * Av) - np.sum(v * Au)) <= 1e-11 * abs(np.sum(u * Av))
    assert np.sum(u * Au) > 0                                                       # J^T J is positive semi-definite
    # the diagonal blocks reported by linearize are the diagonal of the operator: e_k^T A e_k
    k = 12345
    e = np.zeros((c5["n_cams"], 3)); e[k, 1] = 1.0
    assert abs(dev.normal_matvec(e)[k, 1] - lin["diag_blocks"][k, 1, 1]) <= 1e-10 * lin["diag_blocks"][k, 1, 1]


def test_c5_noise_free_graph_is_recovered_exactly():
    g = synth.make_graph(100000, 10000000, seed=77, noise=False, init_noise_deg=3.0)
    dev = _dev(g, _abi.ANGLE_AXIS, LF.SoftLOneLoss(0.1))
    lin = dev.linearize(g["gt_aa"])
    assert np.abs(lin["gradient"]).max() < 1e-9 and lin["cost"] < 1e-18             # the ground truth is a stationary point
    r, s = dev.solve(g["init_aa"])
    err = synth.angular_distance(synth.align_rotations(r, g["gt_aa"]), g["gt_aa"])
    assert err.max() < 1e-8
    assert s["final_cost"] < 1e-12 * s["initial_cost"]
