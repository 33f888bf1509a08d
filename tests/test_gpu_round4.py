"""-m gpu: round 4 -- the forcing schedule of the PCG solves (gsfm_rot_options::pcg_forcing)."""
import numpy as np
import pytest

from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF

pytestmark = pytest.mark.gpu


def _problem(g, et, loss):
    from globalsfmpy_amd.solver import RotationProblem
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"] if et == _abi.ANGLE_AXIS_COVARIANCE else None)
    p.set_loss(loss)
    return p


@pytest.mark.parametrize("single_reduction", [0, 1])
@pytest.mark.parametrize("n,e,et,loss", [(3000, 40000, _abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02)), (2000, 30000, _abi.ANGLE_AXIS, LF.HuberLoss(0.1)),
                                         (1500, 20000, _abi.ROTATION_MAT_FNORM, LF.CauchyLoss(0.2))])
def test_a_continued_solve_is_bit_for_bit_the_uninterrupted_one(n, e, et, loss, single_reduction):
    """pcg_forcing = 2 stops every PCG solve at the loose criterion, evaluates the step, then CONTINUES the solve to the tight tolerance and
    evaluates again.  The device state at a stop is resumable (kernels.hpp: CgScalars::done_seen, the mat-vec-entry decision of the
    single-reduction recurrence), so the whole LM trajectory must equal the one with the schedule off: same bits, same PCG iteration counts."""
    g = synth.make_graph(n, e, seed=31, outlier_frac=0.15)
    p = _problem(g, et, loss)
    kw = dict(dense_cholesky_max_cams=0, pcg_single_reduction=single_reduction)
    r0, s0 = p.solve(g["init_aa"], pcg_forcing=0, **kw)
    t0 = p.trace()
    r2, s2 = p.solve(g["init_aa"], pcg_forcing=2, **kw)
    t2 = p.trace()
    assert s2["num_forcing_refinements"] > 0 and s2["num_inexact_steps"] == 0
    assert np.array_equal(t0, t2), (t0[:, 7], t2[:, 7])
    assert np.array_equal(r0, r2) and s0["final_cost"] == s2["final_cost"] and s0["num_cg_iterations"] == s2["num_cg_iterations"]


@pytest.mark.parametrize("et,loss", [(_abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02)), (_abi.ANGLE_AXIS, LF.SoftLOneLoss(0.1)), (_abi.QUATERNION_COSINE, LF.HuberLoss(0.1))])
def test_forcing_schedule_reaches_the_answer_of_the_exact_schedule(oracle, et, loss):
    """Default schedule against pcg_forcing = 0 and against the oracle on a 4000-camera graph: same LM iterations, rotations two orders inside the
    parity bar, no gauge drift (compared WITHOUT alignment: the loose steps' gauge component is removed), and fewer PCG iterations."""
    g = synth.make_graph(4000, 60000, seed=44, outlier_frac=0.2)
    p = _problem(g, et, loss)
    r0, s0 = p.solve(g["init_aa"], pcg_forcing=0)
    r1, s1 = p.solve(g["init_aa"])
    d = synth.angular_distance(r1, r0)
    print("et %d: %d -> %d PCG iterations (%d inexact steps, %d continued); rotations vs the exact schedule: mean %.2e max %.2e rad (no alignment)"
          % (et, s0["num_cg_iterations"], s1["num_cg_iterations"], s1["num_inexact_steps"], s1["num_forcing_refinements"], d.mean(), d.max()))
    assert s1["num_forcing_restarts"] == 0 and s1["num_inexact_steps"] > 0 and s1["num_cg_iterations"] < s0["num_cg_iterations"]
    assert s1["num_iterations"] == s0["num_iterations"] and s1["termination"] == s0["termination"]
    assert d.mean() <= 1e-8 and d.max() <= 1e-6
    ora = oracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"] if et == _abi.ANGLE_AXIS_COVARIANCE else None)
    ora.set_loss(loss)
    ro, so = ora.solve(g["init_aa"])
    assert s1["num_iterations"] == so["num_iterations"]
    assert synth.angular_distance(synth.align_rotations(r1, ro), ro).mean() <= 1e-6


def test_fast_linearisation_path_is_refused_for_a_loss_whose_second_derivative_turns_positive(oracle):
    """Geman-McClure with a NEGATIVE sigma^2 has rho'' > 0 for s < -a^2 sigma^2 ... the alpha = 0 fast path of K2 would silently drop the
    Corrector's second-order term there; the host decides eligibility from kind and parameter signs (prepare_loss), so the device takes the
    general path and matches the oracle's Jets (round-3 advisor)."""
    from globalsfmpy_amd.solver import RotationProblem
    g = synth.make_graph(300, 3000, seed=12, outlier_frac=0.1)
    loss = LF.GemanMcClureLoss(0.5, -0.001)
    dev = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS); dev.set_loss(loss)
    ora = oracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS); ora.set_loss(loss)
    a, b = dev.linearize(g["init_aa"]), ora.linearize(g["init_aa"])
    scale = np.abs(b["gradient"]).max()
    assert np.abs(a["gradient"] - b["gradient"]).max() <= 1e-9 * scale
    assert np.abs(a["diag_blocks"] - b["diag_blocks"]).max() <= 1e-9 * np.abs(b["diag_blocks"]).max()


def test_exact_steps_take_over_where_pcg_costs_more_than_a_factorisation(oracle):
    """dense_cholesky_auto_cams (default 5333): a graph beyond dense_cholesky_max_cams starts on PCG and switches to exact Cholesky steps -- the
    reference's own linear solver, estimator.cpp:300 -- once a PCG-solved step has cost more GPU time than a factorisation of its size takes.
    A spatially coherent 800-camera graph (hundreds of block-Jacobi iterations per step) switches after its first step; a uniformly random one
    (two dozen iterations) never does; both agree with the oracle, and asking for PCG only (dense_cholesky_max_cams = 0) is respected."""
    hard = synth.make_graph(800, 4800, seed=8, outlier_frac=0.1, local_window=16)     # a chain-like graph: 150-270 block-Jacobi iterations per step
    easy = synth.make_graph(1500, 45000, seed=9, outlier_frac=0.1)
    for g, switches in ((hard, True), (easy, False)):
        p = _problem(g, _abi.ANGLE_AXIS, LF.HuberLoss(0.1))
        r, s = p.solve(g["init_aa"])
        rp, sp = p.solve(g["init_aa"], dense_cholesky_max_cams=0)
        print("%s: %d LM iterations, %d exact steps, %d PCG iterations (PCG only: %d)" % ("coherent" if switches else "random", s["num_iterations"], s["num_dense_solves"],
              s["num_cg_iterations"], sp["num_cg_iterations"]))
        assert sp["num_dense_solves"] == 0
        if switches:
            assert 0 < s["num_dense_solves"] < s["num_iterations"] and s["num_cg_iterations"] < sp["num_cg_iterations"]
        else:
            assert s["num_dense_solves"] == 0
        ora = oracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
        ora.set_loss(LF.HuberLoss(0.1))
        ro, so = ora.solve(g["init_aa"])
        assert s["num_iterations"] == so["num_iterations"]
        assert synth.angular_distance(synth.align_rotations(r, ro), ro).mean() <= 1e-6


@pytest.mark.parametrize("et,loss", [(_abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02)), (_abi.ANGLE_AXIS, LF.SoftLOneLoss(0.1)), (_abi.QUATERNION_COSINE, LF.HuberLoss(0.1)),
                                     (_abi.ROTATION_MAT_FNORM, LF.CauchyLoss(0.3))])
def test_lm_control_on_the_device_follows_the_host_loop_bit_for_bit(et, loss):
    """Exact steps (300 cameras): the one-lane control kernel takes the trust-region decisions from the device scalars and the accept path runs
    predicated on it (lm_device_control = 1, one read-back per iteration) -- the same formulas in the same order as the host loop, hence the same
    trace (cost, cost change, gradient norm, step norm, relative decrease, radius per iteration), rotations and summary, rejected steps included."""
    g = synth.make_graph(300, 3000, seed=77, outlier_frac=0.25)
    p = _problem(g, et, loss)
    init = g["init_aa"] + 0.3 * np.random.default_rng(1).standard_normal(g["init_aa"].shape)   # a far start: rejected and invalid steps happen
    out = []
    for dc in (0, 1):
        r, s = p.solve(init, lm_device_control=dc)
        out.append((r, s, p.trace()))
    (r0, s0, t0), (r1, s1, t1) = out
    assert s0["num_dense_solves"] == s0["num_iterations"] > 0
    assert np.array_equal(t0, t1), (t0[:, :3], t1[:, :3])
    assert np.array_equal(r0, r1)
    for k in ("num_iterations", "num_successful_steps", "num_unsuccessful_steps", "termination", "final_cost", "num_residual_sweeps", "num_linearizations", "iters_to_1e6", "final_radius"):
        assert s0[k] == s1[k], k
