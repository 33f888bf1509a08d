"""-m gpu: the round-3 kernel forms against the forms they replace and against the CPU oracle.

* K2's fast path (losses with rho'' <= 0: only the row camera's Jacobian, rho' alone) against the general path (GSFM_K2_FAST=0) and the
  oracle, in the row-major and in the column-sorted layout;
* K2c / K3c, linearisation and mat-vec on the column-sorted layout of the directed entries (GSFM_K3_COLSORT=1 forces it on graphs far below
  its size threshold) against the row-major kernels and the oracle, down to whole solves;
* the per-edge outputs of the sweep in the problem's own edge order (gsfm_rot_edge_order);
* sigma consensus with the weights fused into the inner solve's first cost sweep / linearisation.
"""
import os

import numpy as np
import pytest

from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem

pytestmark = pytest.mark.gpu


class _Env(object):
    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(1e-300, float(np.max(np.abs(b))))


FAST_CASES = [
    (_abi.ANGLE_AXIS_COVARIANCE, lambda: LF.MAGSACWeightBasedLoss(0.02)),
    (_abi.ANGLE_AXIS_COVARIANCE, lambda: LF.HuberLoss(0.05)),
    (_abi.ANGLE_AXIS, lambda: LF.SoftLOneLoss(0.1)),
    (_abi.ANGLE_AXIS, lambda: None),
    (_abi.ANGLE_AXIS_COVTRACE, lambda: LF.GemanMcClureLoss(0.3, 1.0)),
    (_abi.ANGLE_AXIS_INLIERS, lambda: LF.TukeyLoss(1.0)),
    (_abi.QUATERNION_COSINE, lambda: LF.HuberLoss(0.1)),
    (_abi.QUATERNION_COSINE, lambda: LF.MAGSACWeightBasedLoss(0.3)),
]


@pytest.mark.parametrize("case", range(len(FAST_CASES)))
def test_k2_fast_path_equals_the_general_path_and_the_oracle(oracle, case):
    et, mk = FAST_CASES[case]
    g = synth.make_graph(n_cams=300, n_edges=6000, seed=31 + case, outlier_frac=0.25, full_so3=True)
    x = g["init_aa"]
    v = np.random.default_rng(case).standard_normal((g["n_cams"], 3))
    ora = oracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"], inlier_weight=g["inlier_weight"])
    ora.set_loss(mk())
    lo = ora.linearize(x)
    yo = ora.normal_matvec(v)
    got = {}
    for fast in (0, 1):
        for colsort in (0, 1):
            with _Env(GSFM_K2_FAST=fast, GSFM_K3_COLSORT=colsort):
                dev = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"], inlier_weight=g["inlier_weight"])
                dev.set_loss(mk())
                ld = dev.linearize(x)
                got[(fast, colsort)] = (ld["gradient"], ld["diag_blocks"], dev.normal_matvec(v))
                dev.close()
    for key in ((1, 0), (0, 1), (1, 1)):
        for k in range(3):
            assert _rel(got[key][k], got[(0, 0)][k]) < 1e-12, (key, k)     # sqrt(rho')^2 vs rho', another summation order: rounding only
    for key in got:
        assert _rel(got[key][0], lo["gradient"]) < 1e-9 and _rel(got[key][1], lo["diag_blocks"]) < 1e-9 and _rel(got[key][2], yo) < 1e-9, key


@pytest.mark.parametrize("et,loss,n,e", [
    (_abi.ANGLE_AXIS_COVARIANCE, lambda: LF.MAGSACWeightBasedLoss(0.02), 3000, 90000),
    (_abi.ANGLE_AXIS, lambda: LF.SoftLOneLoss(0.1), 1100, 20000),        # rows: 2 full blocks + a ragged one
    (_abi.QUATERNION_COSINE, lambda: LF.HuberLoss(0.1), 700, 9000),
    (_abi.ANGLE_AXIS_COVARIANCE, lambda: LF.ComposedLoss(LF.CauchyLoss(0.3), LF.SoftLOneLoss(0.2)), 513, 4000),   # general K2 path feeding K3c
])
def test_column_sorted_matvec_equals_the_row_major_form(oracle, et, loss, n, e):
    g = synth.make_graph(n_cams=n, n_edges=e, seed=5, outlier_frac=0.2)
    # a repeated camera pair (one copy reversed in role is impossible with i < j: repeat it as is) and an isolated camera
    ei = np.concatenate([g["edge_i"], g["edge_i"][:3]]); ej = np.concatenate([g["edge_j"], g["edge_j"][:3]])
    rel = np.concatenate([g["rel_aa"], g["rel_aa"][:3] * 0.9]); c6 = np.concatenate([g["cov6"], g["cov6"][:3]])
    keep = (ei != n - 1) & (ej != n - 1)
    ei, ej, rel, c6 = ei[keep], ej[keep], rel[keep], c6[keep]
    v = np.random.default_rng(1).standard_normal((n, 3))
    res = {}
    for mode in (0, 1):
        with _Env(GSFM_K3_COLSORT=mode, GSFM_PCG_COARSE=0):
            dev = RotationProblem(n, ei, ej, rel, et, cov6=c6)
            dev.set_loss(loss())
            assert dev.matvec_bytes()[1] == (2 if mode else 1)
            dev.linearize(g["init_aa"])
            y = dev.normal_matvec(v)
            rot, s = dev.solve(g["init_aa"], dense_cholesky_max_cams=0, pcg_single_reduction=0)
            res[mode] = (y, rot, s)
            dev.close()
    assert _rel(res[1][0], res[0][0]) < 1e-13
    assert np.all(res[1][0][n - 1] == 0.0) or np.allclose(res[1][0][n - 1], res[0][0][n - 1], rtol=1e-13)   # the isolated camera: only its diagonal term
    ora = oracle.OracleProblem(n, ei, ej, rel, et, cov6=c6)
    ora.set_loss(loss())
    ora.linearize(g["init_aa"])
    assert _rel(res[1][0], ora.normal_matvec(v)) < 1e-9
    s0, s1 = res[0][2], res[1][2]
    assert s0["num_iterations"] == s1["num_iterations"] and s0["termination"] == s1["termination"]
    assert abs(s0["final_cost"] - s1["final_cost"]) <= 1e-11 * s0["final_cost"]
    d = synth.angular_distance(synth.align_rotations(res[1][1], res[0][1]), res[0][1])
    assert d.mean() < 1e-9, d.mean()
    assert np.array_equal(res[1][1][n - 1], g["init_aa"][n - 1])   # untouched views never move


def test_column_sorted_solve_matches_the_oracle(oracle):
    g = synth.make_graph(n_cams=2500, n_edges=60000, seed=9, outlier_frac=0.3)
    with _Env(GSFM_K3_COLSORT=1):
        dev = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    dev.set_loss(LF.MAGSACWeightBasedLoss(0.02))
    assert dev.matvec_bytes()[1] == 2
    ora = oracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    ora.set_loss(LF.MAGSACWeightBasedLoss(0.02))
    ro, so = ora.solve(g["init_aa"])
    # the exact schedule (every step at 1e-12) to 1e-9 in cost, as in round 3; the default schedule (loose early steps, round 4) lands ~1e-8 rad from
    # it -- and the last iterate is not a stationary point (Ceres stops at function_tolerance 1e-6), so its cost follows linearly: 1e-7
    for kw, cost_bar in ((dict(pcg_forcing=0), 1e-9), (dict(), 1e-7)):
        rd, sd = dev.solve(g["init_aa"], pcg_single_reduction=0, **kw)
        assert sd["num_iterations"] == so["num_iterations"] and sd["termination"] == so["termination"]
        assert abs(sd["final_cost"] - so["final_cost"]) <= cost_bar * so["final_cost"]
        d = synth.angular_distance(synth.align_rotations(rd, ro), ro)
        assert d.mean() <= 1e-6, d.mean()       # the north-star bar (observed: orders of magnitude below)


def test_edge_order_is_the_order_of_the_device_side_planes(oracle):
    g = synth.make_graph(n_cams=5000, n_edges=60000, seed=4, outlier_frac=0.1)
    dev = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    dev.set_loss(LF.HuberLoss(0.05))
    order = dev.edge_order()
    assert order.shape == (60000,) and np.array_equal(np.sort(order), np.arange(60000))
    # bucketed by (camera block of first, camera block of second), 2048 cameras per block
    key = (g["edge_i"][order] // 2048).astype(np.int64) * 1000 + g["edge_j"][order] // 2048
    assert np.all(np.diff(key) >= 0)
    out = dev.residuals(g["init_aa"], want_residuals=True)
    ora = oracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    ora.set_loss(LF.HuberLoss(0.05))
    want = ora.residuals(g["init_aa"], want_residuals=True)
    assert _rel(out["s"], want["s"]) < 1e-12 and _rel(out["rho"], want["rho"]) < 1e-11 and _rel(out["residuals"], want["residuals"]) < 1e-12


@pytest.mark.parametrize("loss", [None, "huber", "composed"])
def test_sigma_consensus_with_fused_weights_matches_the_oracle(oracle, loss):
    mk = {None: lambda: None, "huber": lambda: LF.HuberLoss(0.3), "composed": lambda: LF.ComposedLoss(LF.CauchyLoss(0.5), LF.SoftLOneLoss(0.4))}[loss]
    g = synth.make_graph(n_cams=400, n_edges=6000, seed=21, outlier_frac=0.2)
    dev = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
    ora = oracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
    dev.set_loss(mk()); ora.set_loss(mk())
    rd, sd = dev.solve_sigma_consensus(g["init_aa"], 6, 0.05)
    ro, so = ora.solve_sigma_consensus(g["init_aa"], 6, 0.05)
    assert sd["outer_iterations"] == so["outer_iterations"]
    assert abs(sd["last_weight_change"] - so["last_weight_change"]) <= 1e-9 * max(1.0, so["last_weight_change"])
    assert sd["num_iterations"] == so["num_iterations"]
    d = synth.angular_distance(synth.align_rotations(rd, ro), ro)
    assert d.mean() <= 1e-6, d.mean()
    # a second call starts from zero weights again (the reference's zero-initialised last_weights), whatever the first one left in the planes
    rd2, sd2 = dev.solve_sigma_consensus(g["init_aa"], 6, 0.05)
    assert np.array_equal(rd, rd2) and sd2["last_weight_change"] == sd["last_weight_change"]


@pytest.mark.parametrize("n,e,hub", [(70000, 300000, 40000),      # 17 camera bits: the hub row saturates the 6-bit count field of K3c's record
                                     (600000, 1200000, 3000)])    # 20 camera bits: no room for counts in the word, all from the 2-byte plane
def test_column_sorted_record_widths_and_saturated_row_counts(oracle, n, e, hub):
    g = synth.make_graph(n_cams=n, n_edges=e, seed=13, outlier_frac=0.1)
    rng = np.random.default_rng(3)
    others = rng.choice(np.arange(6, n, dtype=np.uint32), size=hub, replace=False)
    ei = np.concatenate([g["edge_i"], np.full(hub, 5, dtype=np.uint32)]); ej = np.concatenate([g["edge_j"], others])
    rel = np.concatenate([g["rel_aa"], 0.3 * rng.standard_normal((hub, 3))]); c6 = np.concatenate([g["cov6"], g["cov6"][:hub]])
    v = rng.standard_normal((n, 3))
    y = {}
    for mode in (0, 1):
        with _Env(GSFM_K3_COLSORT=mode, GSFM_PCG_COARSE=0, GSFM_REORDER=0):
            dev = RotationProblem(n, ei, ej, rel, _abi.ANGLE_AXIS_COVARIANCE, cov6=c6)
            dev.set_loss(LF.HuberLoss(0.05))
            assert dev.matvec_bytes()[1] == (2 if mode else 1)
            lin = dev.linearize(g["init_aa"])
            y[mode] = (dev.normal_matvec(v), lin["gradient"], lin["diag_blocks"])
            dev.close()
    for k in range(3):
        assert _rel(y[1][k], y[0][k]) < 1e-12, k
    ora = oracle.OracleProblem(n, ei, ej, rel, _abi.ANGLE_AXIS_COVARIANCE, cov6=c6)
    ora.set_loss(LF.HuberLoss(0.05))
    ora.linearize(g["init_aa"])
    assert _rel(y[1][0], ora.normal_matvec(v)) < 1e-9


@pytest.mark.parametrize("et,loss,n,e", [
    (_abi.ANGLE_AXIS_COVARIANCE, lambda: LF.MAGSACWeightBasedLoss(0.02), 3000, 90000),
    (_abi.QUATERNION_COSINE, lambda: LF.HuberLoss(0.1), 1100, 15000),
])
def test_single_reduction_pcg_on_the_column_sorted_layout(oracle, et, loss, n, e):
    """K3c inside the single-reduction recurrence (k_mv_col_cg + k_mv_col_finish + k_cg2_step: three launches per iteration) against the
    textbook recurrence on the same layout and against the oracle."""
    g = synth.make_graph(n_cams=n, n_edges=e, seed=17, outlier_frac=0.2)
    res = {}
    with _Env(GSFM_K3_COLSORT=1, GSFM_PCG_COARSE=0):
        dev = RotationProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"])
    dev.set_loss(loss())
    assert dev.matvec_bytes()[1] == 2
    for sr in (0, 1):
        for graph in (1, 0):
            res[(sr, graph)] = dev.solve(g["init_aa"], dense_cholesky_max_cams=0, pcg_single_reduction=sr, pcg_hip_graph=graph)
    for graph in (1, 0):
        assert np.array_equal(res[(1, graph)][0], res[(1, 1)][0]) and np.array_equal(res[(0, graph)][0], res[(0, 1)][0])   # replayed or launched: same bits
    (r0, s0), (r1, s1) = res[(0, 1)], res[(1, 1)]
    assert s0["num_iterations"] == s1["num_iterations"] and s0["termination"] == s1["termination"]
    assert abs(s1["num_cg_iterations"] - s0["num_cg_iterations"]) <= 2 * s0["num_iterations"]
    assert abs(s0["final_cost"] - s1["final_cost"]) <= 1e-11 * s0["final_cost"]
    assert synth.angular_distance(synth.align_rotations(r1, r0), r0).mean() < 1e-9
    ora = oracle.OracleProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"])
    ora.set_loss(loss())
    ora.set_linear_solver("pcg")
    ro, so = ora.solve(g["init_aa"])
    assert s1["num_iterations"] == so["num_iterations"] and s1["termination"] == so["termination"]
    assert synth.angular_distance(synth.align_rotations(r1, ro), ro).mean() <= 1e-6


def test_exact_cholesky_step_from_the_column_sorted_layout():
    """dense_cholesky_max_cams must be honoured whatever layout the entries are in: the dense matrix assembled from the body-frame blocks of
    the column-sorted layout (k_dense_assemble_col) against the one assembled from the row-major planes."""
    g = synth.make_graph(n_cams=450, n_edges=9000, seed=23, outlier_frac=0.25)
    out = {}
    for mode in (0, 1):
        with _Env(GSFM_K3_COLSORT=mode):
            dev = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
        dev.set_loss(LF.MAGSACWeightBasedLoss(0.02))
        assert dev.matvec_bytes()[1] == (2 if mode else 1)
        out[mode] = dev.solve(g["init_aa"])
        dev.close()
    (r0, s0), (r1, s1) = out[0], out[1]
    assert s0["num_dense_solves"] == s0["num_iterations"] > 0 and s1["num_dense_solves"] == s1["num_iterations"] and s1["num_cg_iterations"] == 0
    assert s0["num_iterations"] == s1["num_iterations"] and s0["termination"] == s1["termination"]
    assert abs(s0["final_cost"] - s1["final_cost"]) <= 1e-11 * s0["final_cost"]
    assert synth.angular_distance(synth.align_rotations(r1, r0), r0).mean() < 1e-9


@pytest.mark.parametrize("et", [_abi.ANGLE_AXIS, _abi.ANGLE_AXIS_INLIERS])
def test_set_edge_weights_in_both_layouts(oracle, et):
    """gsfm_rot_set_edge_weights (scalar weights replaced after creation; ANGLE_AXIS is promoted to a scalar-weight problem) gathers the
    caller's per-edge weights into the entry planes in THEIR order -- row-major or column-sorted -- and into the cost planes."""
    g = synth.make_graph(n_cams=900, n_edges=14000, seed=29, outlier_frac=0.2)
    w = np.random.default_rng(5).uniform(0.2, 3.0, size=14000)
    ora = oracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, inlier_weight=g["inlier_weight"])
    ora.set_loss(LF.HuberLoss(0.1))
    ora.set_edge_weights(w)
    ora.set_linear_solver("pcg")
    ro, so = ora.solve(g["init_aa"])
    lo = ora.linearize(g["init_aa"])
    for mode in (0, 1):
        with _Env(GSFM_K3_COLSORT=mode, GSFM_PCG_COARSE=0):
            dev = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, inlier_weight=g["inlier_weight"])
        dev.set_loss(LF.HuberLoss(0.1))
        dev.set_edge_weights(w)
        assert dev.matvec_bytes()[1] == (2 if mode else 1)
        ld = dev.linearize(g["init_aa"])
        assert _rel(ld["gradient"], lo["gradient"]) < 1e-9 and _rel(ld["diag_blocks"], lo["diag_blocks"]) < 1e-9 and abs(ld["cost"] - lo["cost"]) <= 1e-12 * lo["cost"]
        rd, sd = dev.solve(g["init_aa"])
        assert sd["num_iterations"] == so["num_iterations"] and sd["termination"] == so["termination"]
        assert synth.angular_distance(synth.align_rotations(rd, ro), ro).mean() <= 1e-6
        dev.close()
