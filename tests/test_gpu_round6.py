"""Round-6 items: the 24-byte measurement planes (three quaternion components on the covariance-whitened problems) on the rotations that
stress their decode, the column-sorted layout without its density condition, the component_rest option and the freeze rule of the component
step, the weak-scaling line of bench.py."""
import os
import sys

import numpy as np
import pytest

from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


class _Env:
    def __init__(self, **kw): self.kw = {k: str(v) for k, v in kw.items()}
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


@pytest.mark.parametrize("et", [_abi.ANGLE_AXIS_COVARIANCE, _abi.ANGLE_AXIS_COV_INLIERS])
def test_three_component_measurement_planes_on_the_rotations_that_stress_the_decode(oracle, et):
    """csrc/kernels.hpp, qrel_three: on W_MATRIX problems of a million edges or more the measured rotation is stored as three quaternion components, the largest one
    dropped (its index in bit 62 of the first two, its sign in bit 62 of the third) and rebuilt by a square root.  Measurements built to hit
    every dropped index and both signs: half-turns about each axis and about diagonals (w ~ 0: x, y or z is the largest), angles of
    pi +- 1e-9 and pi +- 1e-4, rotation vectors LONGER than pi (ceres::AngleAxisToQuaternion then gives w < 0), tiny and exactly-zero
    rotations, ties between two components.  Per-edge s against the oracle at 1e-12 (the bar of test_gpu_parity), gradient and diagonal
    blocks at 1e-9, on both layouts of the directed entries."""
    rng = np.random.default_rng(12)
    n = 400
    g = synth.make_graph(n, 6000, seed=61, outlier_frac=0.2)
    rel = g["rel_aa"].copy()
    axes = [np.array(a, dtype=float) / np.linalg.norm(a) for a in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [0, 1, 1], [1, 1, 1], [1, -1, 0], [-1, 0, 0], [0, -1, 1])]
    angles = [np.pi, np.pi - 1e-9, np.pi + 1e-9, np.pi - 1e-4, np.pi + 1e-4, 1.5 * np.pi, 1.9 * np.pi, 2.0 * np.pi - 1e-6, 0.5 * np.pi, 2.0 * np.arccos(0.5), 1e-9, 1e-200, 0.0,
              2.0 * np.arctan2(np.sqrt(0.5), np.sqrt(0.5))]
    k = 0
    for a in axes:
        for t in angles:
            rel[k] = a * t; k += 1
    for _ in range(200):   # uniformly random rotations scaled past pi now and then
        v = rng.standard_normal(3); v /= np.linalg.norm(v)
        rel[k] = v * rng.uniform(0.0, 2.0 * np.pi); k += 1
    x = g["init_aa"] + 0.3 * rng.standard_normal(g["init_aa"].shape)
    ora = oracle.OracleProblem(n, g["edge_i"], g["edge_j"], rel, et, cov6=g["cov6"], inlier_weight=g["inlier_weight"])
    ora.set_loss(LF.MAGSACWeightBasedLoss(0.02))
    ro, lo = ora.residuals(x, want_residuals=True), ora.linearize(x)
    small = RotationProblem(n, g["edge_i"], g["edge_j"], rel, et, cov6=g["cov6"], inlier_weight=g["inlier_weight"])
    assert small.sweep_bytes()[1] == 8 + 32 + 48   # below a million edges the full quaternion stays (the small configurations keep their last bits)
    small.close()
    for colsort in (0, 1):
        with _Env(GSFM_K3_COLSORT=colsort, GSFM_QREL3=1):   # (forced: by itself the compact form starts at a million edges)
            dev = RotationProblem(n, g["edge_i"], g["edge_j"], rel, et, cov6=g["cov6"], inlier_weight=g["inlier_weight"])
        assert dev.sweep_bytes()[1] == 8 + 24 + 48
        dev.set_loss(LF.MAGSACWeightBasedLoss(0.02))
        rd, ld = dev.residuals(x, want_residuals=True), dev.linearize(x)
        scale = np.maximum(1.0, ro["s"])
        worst = int(np.argmax(np.abs(rd["s"] - ro["s"]) / scale))
        print("layout %d: max |s_dev - s_ora| / max(1, s) = %.2e (edge %d), residual vectors %.2e" % (
            colsort, (np.abs(rd["s"] - ro["s"]) / scale).max(), worst, (np.abs(rd["residuals"] - ro["residuals"]) / np.sqrt(scale)[:, None]).max()))
        assert (np.abs(rd["s"] - ro["s"]) / scale).max() <= 1e-12
        assert (np.abs(rd["residuals"] - ro["residuals"]) / np.sqrt(scale)[:, None]).max() <= 1e-12
        assert np.abs(ld["gradient"] - lo["gradient"]).max() <= 1e-9 * np.abs(lo["gradient"]).max()
        assert np.abs(ld["diag_blocks"] - lo["diag_blocks"]).max() <= 1e-9 * np.abs(lo["diag_blocks"]).max()
        dev.close()


def test_a_non_finite_measurement_fails_a_covariance_problem_like_ceres(oracle):
    """The three-component planes must not launder a NaN measurement into a finite quaternion (bit 62 of a NaN is set: qrel_encode stores a
    pattern that decodes to NaN instead): FAILURE at iteration 0, the rotations untouched -- as test_gpu_parity checks for the 32-byte planes."""
    g = synth.make_graph(300, 3000, 5, outlier_frac=0.1)
    for bad in (np.nan, np.inf):
        rel = g["rel_aa"].copy(); rel[17, 1] = bad
        for cls in (RotationProblem, oracle.OracleProblem):
            with _Env(GSFM_QREL3=1):
                p = cls(g["n_cams"], g["edge_i"], g["edge_j"], rel, _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
            p.set_loss(LF.HuberLoss(0.1))
            r, s = p.solve(g["init_aa"])
            assert s["termination_name"] == "FAILURE" and s["num_iterations"] == 0 and np.array_equal(r, g["init_aa"])


def test_a_sparse_large_graph_takes_the_column_sorted_layout_and_solves_as_the_row_major_one():
    """Round 6: the column-sorted layout no longer asks for 512 rows x mean degree >= cameras / 2 (profiles/r06_density_rule.txt: it wins at
    every density measured).  70 000 cameras / 560 000 edges -- 1.12 M directed entries, 8 k per row block against 70 k cameras, a sixteenth of
    the old threshold -- takes K2c / K3c by default and gives the row-major solve's iterations and rotations."""
    g = synth.make_graph(70000, 560000, seed=33, outlier_frac=0.2)
    out = {}
    for mode in (None, 0):
        env = {} if mode is None else {"GSFM_K3_COLSORT": mode}
        with _Env(**env):
            p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
        p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
        form = int(p.matvec_bytes()[1])
        r, s = p.solve(g["init_aa"], pcg_forcing=0)
        out[mode] = (form, r, s)
        p.close()
    (f_d, r_d, s_d), (f_r, r_r, s_r) = out[None], out[0]
    print("default: layout form %d, %d LM / %d PCG; row-major: form %d, %d LM / %d PCG" % (f_d, s_d["num_iterations"], s_d["num_cg_iterations"], f_r, s_r["num_iterations"], s_r["num_cg_iterations"]))
    assert f_d == 2 and f_r == 1
    assert s_d["num_iterations"] == s_r["num_iterations"] and abs(s_d["num_cg_iterations"] - s_r["num_cg_iterations"]) <= 0.02 * s_r["num_cg_iterations"] + 2
    assert abs(s_d["final_cost"] - s_r["final_cost"]) <= 1e-9 * s_r["final_cost"]
    assert synth.angular_distance(synth.align_rotations(r_d, r_r), r_r).max() <= 1e-8


def test_component_rest_option_and_the_freeze_rule(oracle):
    """gsfm_rot_options::component_rest (round 6; a documented departure from Ceres' single global stopping rule): 1 = a factorised component whose
    exact step has fallen below 1e-10 rad -- and has at least halved against the previous measurement, or the trust radius is not below its
    initial value -- is put to rest; 0 = every component is solved in every LM iteration, as the reference does.  Both within the bar of the
    oracle per component, same LM iterations; with the option off nothing is skipped (more PCG iterations on the large components' floor)."""
    from test_gpu_round5 import _batch_of_scenes
    sizes = (300, 700, 120, 450, 64)
    N, ei, ej, rel, cov, init, comp = _batch_of_scenes(sizes, 700, None)
    dev = RotationProblem(N, ei, ej, rel, _abi.ANGLE_AXIS_COVTRACE, cov6=cov); dev.set_loss(LF.HuberLoss(0.1))
    r1, s1 = dev.solve(init)
    r0, s0 = dev.solve(init, component_rest=0)
    r1b, s1b = dev.solve(init)   # (the state of the first solve must not leak into the third: frozen / stepmax are reset per solve)
    ora = oracle.OracleProblem(N, ei, ej, rel, _abi.ANGLE_AXIS_COVTRACE, cov6=cov); ora.set_loss(LF.HuberLoss(0.1))
    ro, so = ora.solve(init)
    print("component_rest 1: %d LM / %d PCG; 0: %d LM / %d PCG; oracle %d LM" % (s1["num_iterations"], s1["num_cg_iterations"], s0["num_iterations"], s0["num_cg_iterations"], so["num_iterations"]))
    assert s1["num_iterations"] == s0["num_iterations"] == so["num_iterations"]
    assert s0["num_cg_iterations"] >= s1["num_cg_iterations"]
    assert np.array_equal(r1, r1b) and s1["num_cg_iterations"] == s1b["num_cg_iterations"]
    for c in range(len(sizes)):
        m = comp == c
        for r in (r1, r0):
            assert synth.angular_distance(synth.align_rotations(r[m], ro[m]), ro[m]).mean() <= 1e-6, c


@pytest.mark.parametrize("case", ["single_small", "single_one_tile", "single_odd", "components", "components_all_small"])
def test_one_launch_per_block_column_gives_the_fused_steps_bits(case):
    """dense_kernels.hpp, k_chol_look (round 6): the exact step's factorisation as one launch per block column -- the panel of column k + 1
    beside the update with column k, one elimination per block row -- against the fused step kernel it replaces (GSFM_CHOL_FUSED=1), for the
    single dense matrix (solver_dense.hpp: 1, 2, an odd and an even number of block columns) and for the components of a disconnected graph
    factorised side by side (solver_components.hpp): every tile receives the same updates in the same order from the same instructions, so
    whole solves agree to the last bit -- rotations, cost, iteration counts."""
    from test_gpu_round5 import _batch_of_scenes
    if case.startswith("single"):
        n = {"single_small": 150, "single_one_tile": 10, "single_odd": 331}[case]
        g = synth.make_graph(n, 12 * n if n > 20 else 30, seed=77 + n, outlier_frac=0.1)
        N, ei, ej, rel, cov, init = g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], g["cov6"], g["init_aa"]
    else:
        sizes = (300, 700, 120, 450, 64) if case == "components" else (200, 260, 150, 90, 21)
        N, ei, ej, rel, cov, init, _ = _batch_of_scenes(sizes, 900)
    out = []
    for fused in (1, 0):
        with _Env(GSFM_CHOL_FUSED=fused):
            p = RotationProblem(N, ei, ej, rel, _abi.ANGLE_AXIS_COVTRACE, cov6=cov)
            p.set_loss(LF.HuberLoss(0.1))
            r, s = p.solve(init)
            out.append((r, s, p.trace().copy()))
    (rf, sf, tf), (rl, sl, tl) = out
    print("%s: %d cameras, %d LM iterations, %d exact steps, final cost %.17g" % (case, N, sl["num_iterations"], sl["num_dense_solves"], sl["final_cost"]))
    assert sl["num_dense_solves"] == sl["num_iterations"] > 0 and sf["num_dense_solves"] == sf["num_iterations"]
    assert sf["num_iterations"] == sl["num_iterations"] and sf["final_cost"] == sl["final_cost"]
    assert np.array_equal(tf, tl)
    assert np.array_equal(rf, rl)
