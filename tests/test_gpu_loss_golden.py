"""Device-level pin of the HIP loss program against the reference's own vectors: every row of
tests/golden/loss_vectors.json -- (s, rho, rho', rho'') recorded by importing the reference's
scripts/loss_functions.py:47-458 (tests/golden/make_loss_vectors.py) -- goes through the device routines
the sweeps use (gsfm_rot_loss_eval -> loss_eval<LM> / loss_value<LM> of csrc/loss_dev.hpp), at the same
tolerances the oracle is held to in test_oracle_golden.py.  The grid contains the MAGSAC cut, table-cell
midpoints and edges (banker's-rounding ties of loss_functions.py:304), the Huber / Tukey knees, s = 0 and
s up to 1e2, so the rows are fed as exact doubles, not through a residual."""
import json
import math
import os

import numpy as np
import pytest

from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from test_oracle_golden import _build, _cases

pytestmark = pytest.mark.gpu


def _tiny_problem():
    g = synth.make_graph(n_cams=8, n_edges=16, seed=3)
    return RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)


def _assert_rows(got, rows, rtol, what):
    want = rows[:, 1:]
    finite = np.where(np.isfinite(want), np.abs(want), 0.0)
    atol = 1e-14 * finite.max(axis=0)          # rho = w(0) - w(s): differences of near-equal table values cancel
    for k in range(got.shape[1]):
        w, g = want[:, k], got[:, k]
        inf = ~np.isfinite(w)
        assert np.array_equal(np.isnan(w), np.isnan(g)) and np.array_equal(w[inf & ~np.isnan(w)], g[inf & ~np.isnan(w)]), what
        err = np.abs(g[~inf] - w[~inf])
        ok = (err <= rtol * np.maximum(np.abs(w[~inf]), 1e-300) + 1e-300) | (err <= atol[k])
        assert ok.all(), (what, k, rows[~inf][~ok][:3], g[~inf][~ok][:3])


def test_every_reference_loss_row_through_the_device_loss_program(golden_dir):
    dev = _tiny_problem()
    n_rows, modes, n_fast = 0, set(), 0
    for case in _cases(golden_dir):
        obj = _build(case["program"]) if case["program"] else getattr(LF, case["class"])(*case["args"])
        dev.set_loss(obj)                                   # native descriptor -> prepare_loss -> device program
        rows = np.asarray(case["rows"], dtype=np.float64)
        rho3, val, fast = dev.loss_eval(rows[:, 0], with_fast_rho1=True)
        rtol = 1e-9 if getattr(obj, "use_weight_inverse", False) else 1e-12   # same bars as the oracle's test
        what = (case["class"], case["args"])
        _assert_rows(rho3, rows, rtol, what)
        # K2's fast path (losses whose rho'' is never positive: the cheap single leaves and the nu = 3 MAGSAC weight loss) evaluates
        # rho' alone, with a host-precomputed constant factor: the same recorded rows, the same bar.  It must exist exactly for the
        # losses that can never need the Corrector's alpha term, i.e. never for a recorded row with rho'' > 0.
        if np.isfinite(fast).all():
            n_fast += 1
            assert not (rows[:, 3] > 0).any(), what
            want1 = np.stack([rows[:, 0], rows[:, 2]], axis=1)
            _assert_rows(fast[:, None], want1, rtol, what + ("fast rho'",))
        else:
            assert np.isnan(fast).all(), what
        # the solver's cost-only specialisation (trial-cost sweeps): rho alone; for MAGSAC nu = 3 it evaluates the table entry
        # as exp(-x / 1000) of the same quantised cell
        _assert_rows(val[:, None], rows[:, :2], rtol, what + ("value",))
        n_rows += len(rows)
        modes.add(case["class"])
    assert n_rows > 2000 and {"MAGSACWeightBasedLoss", "MAGSACWeightBasedLoss4", "MAGSACWeightBasedLoss9", "HuberLoss", "TolerantLoss"} <= modes
    assert n_fast >= 5


def test_magsac_tie_rounding_rows_hit_the_reference_cell(golden_dir):
    """loss_functions.py:304 rounds 1000 s / (2 sigma^2) with Python's round() (ties to even).  At an exact tie the two
    neighbouring cells differ by ~1e-3 relative in rho', far above the 1e-12 bar, so these rows fail unless the device
    rounds the same way (rint, not round)."""
    dev = _tiny_problem()
    sigma = 0.02
    dev.set_loss(LF.MAGSACWeightBasedLoss(sigma))
    ssm2 = 2.0 * sigma * sigma
    ties = np.array([(k + 0.5) * ssm2 / 1000.0 for k in range(0, 4000, 7)])
    ties = ties[np.abs(1000.0 * ties / ssm2 - np.floor(1000.0 * ties / ssm2) - 0.5) == 0.0]   # exact .5 in double arithmetic
    assert len(ties) > 100
    rho3, _, fast = dev.loss_eval(ties, with_fast_rho1=True)
    ref = LF.MAGSACWeightBasedLoss(sigma)
    for s, got, f1 in zip(ties, rho3, fast):
        out = [0.0, 0.0, 0.0]
        ref.Evaluate(float(s), out)                          # pinned to the reference by test_oracle_golden.py
        assert np.allclose(got, out, rtol=1e-12, atol=0.0), (s, got, out)
        assert math.isclose(f1, out[1], rel_tol=1e-12), (s, f1, out)   # K2's fast rho' lands in the same cell
        x = round(1000.0 * float(s) / ssm2)
        assert x % 2 == 0                                    # the tie went to the even cell


def test_star_graph_feeds_the_same_rows_through_the_residual_sweep(golden_dir, oracle):
    """Plumbing check of the same vectors end to end: a star graph whose edge k has |log(R_j R_i^T R_ij^T)|^2 = s_k
    (all cameras at the identity, measurement = rotation by sqrt(s_k) about x), read back through gsfm_rot_residuals.
    s is reproduced to ~1 ulp only, so rows within 1e-9 of a MAGSAC cell boundary are left to the exact test above."""
    for cls, args in (("MAGSACWeightBasedLoss", (0.02, False)), ("HuberLoss", (0.1,)), ("GemanMcClureLoss", (0.1, 2.0)), ("SoftLOneLoss", (0.1,))):
        case = next(c for c in _cases(golden_dir) if c["class"] == cls and tuple(c["args"]) == args and not c["program"])
        rows = np.asarray(case["rows"], dtype=np.float64)
        rows = rows[(rows[:, 0] > 0) & (rows[:, 0] < 9.0)]
        if cls.startswith("MAGSAC"):
            cell = 1000.0 * rows[:, 0] / (2 * args[0] ** 2)
            rows = rows[np.abs(cell - np.floor(cell) - 0.5) > 1e-6]
        n = len(rows)
        assert n > 20
        ei = np.zeros(n, dtype=np.uint32); ej = np.arange(1, n + 1, dtype=np.uint32)
        rel = np.zeros((n, 3)); rel[:, 0] = np.sqrt(rows[:, 0])
        dev = RotationProblem(n + 1, ei, ej, rel, _abi.ANGLE_AXIS)
        dev.set_loss(getattr(LF, cls)(*args))
        out = dev.residuals(np.zeros((n + 1, 3)))
        assert np.allclose(out["s"], rows[:, 0], rtol=1e-14, atol=0)
        # rho' is Lipschitz in s away from the cell edges: 1e-14 in s -> <= 1e-10 here
        assert np.allclose(out["rho"], rows[:, 1:], rtol=1e-9, atol=1e-13 * np.abs(rows[:, 1:]).max()), cls
        assert math.isclose(out["cost"], 0.5 * rows[:, 1].sum(), rel_tol=1e-9)


def test_theia_residual_known_answers_through_the_device(golden_dir):
    """Theia's pairwise_rotation_error_test.cc:87-139 (the only residual-level vectors the reference tree holds), through the device: a
    two-camera problem per case, residual read back with gsfm_rot_residuals (the scalar weight enters as an ANGLE_AXIS_INLIERS weight),
    Theia's own tolerance 1e-12.  The squared norm also comes out of the stand-alone edge sweep (gsfm_rot_edge_sq_norms)."""
    from globalsfmpy_amd import solver
    kats = json.load(open(os.path.join(golden_dir, "residual_kats.json")))
    assert len(kats["cases"]) >= 4
    for c in kats["cases"]:
        rot = np.array([c["rotation1"], c["rotation2"]], dtype=np.float64)
        rel = np.array([c["relative_rotation"]], dtype=np.float64)
        p = RotationProblem(2, [0], [1], rel, _abi.ANGLE_AXIS_INLIERS, inlier_weight=np.array([c["weight"]]))
        out = p.residuals(rot, want_residuals=True)
        want = np.asarray(c["expected"], dtype=np.float64)
        assert np.abs(out["residuals"][0] - want).max() < kats["tolerance"], c["name"]
        assert abs(out["s"][0] - float(want @ want)) < 1e-12 * max(1.0, float(want @ want)), c["name"]
        sweep = solver.edge_sq_norms(2, [0], [1], rel, rot)
        w2 = c["weight"] ** 2
        assert abs(sweep["s"][0] * w2 - float(want @ want)) < 1e-12 * max(1.0, float(want @ want)), c["name"]
