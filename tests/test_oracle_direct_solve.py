"""The oracle's PCG stand-in for SPARSE_NORMAL_CHOLESKY, pinned against a DIRECT sparse solve above the dense-Cholesky sizes.

The reference solves every graph with a sparse Cholesky factorisation of the damped normal equations
(/root/reference/src/GSfM_nonlinear_rotation_estimator.cpp:299-305, ceres SPARSE_NORMAL_CHOLESKY).  Oracle and product both switch to block-
Jacobi PCG beyond 512 cameras, so above that size every device-vs-oracle comparison is PCG against PCG by the same author.  Here the
oracle's linear systems of the first LM iterations -- exactly as its solver saw them: (J^T J + D^2) y = J^T r with the Corrector and the
Jacobi column scaling applied, near-singular (three-dimensional gauge null space, damping 1e-4 and less) -- are handed to scipy's SuperLU and
the PCG answer must equal the direct one to 1e-10 of the step.  Sizes: a spatially coherent graph of C2's size (10k cameras / 200k edges:
the factor of a uniformly random graph of that size fills in completely -- SuperLU did not finish in 20 minutes on it, which is SURVEY 8d's
remark about CHOLMOD on C5 in small), and uniformly random graphs of 2 000 cameras and of a Trafalgar-sized component of C4 (5 288 cameras,
15 864 unknowns: a dense LAPACK Cholesky).  (-m gpu twin: tests/test_gpu_direct_solve.py holds the DEVICE's LM step against the same
direct solve.)"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF


def direct_step(sysk, edge_i, edge_j, n_cams, sparse):
    """y solving (J^T J + diag(D)^2) y = rhs by sparse LU, from a captured system (oracle/pyoracle.py captured_step)."""
    E, R, _ = sysk["Ji"].shape
    rows = np.repeat(np.arange(E * R), 3)
    ci = (3 * edge_i.astype(np.int64)[:, None, None] + np.arange(3)[None, None, :] + np.zeros((1, R, 1), dtype=np.int64)).ravel()
    cj = (3 * edge_j.astype(np.int64)[:, None, None] + np.arange(3)[None, None, :] + np.zeros((1, R, 1), dtype=np.int64)).ravel()
    J = sp.coo_matrix((np.concatenate([sysk["Ji"].ravel(), sysk["Jj"].ravel()]), (np.concatenate([rows, rows]), np.concatenate([ci, cj]))),
                      shape=(E * R, 3 * n_cams)).tocsr()
    A = (J.T @ J + sp.diags(sysk["D"] ** 2)).tocsc()
    assert np.allclose(J.T @ sysk["rt"].ravel(), sysk["rhs"], rtol=1e-12, atol=1e-12 * np.abs(sysk["rhs"]).max())
    if sparse:
        lu = spla.splu(A)
        solve = lu.solve
    else:
        import scipy.linalg as sla
        cf = sla.cho_factor(A.toarray(), lower=True, overwrite_a=True, check_finite=False)
        solve = lambda b: sla.cho_solve(cf, b, check_finite=False)   # noqa: E731
    y = solve(sysk["rhs"])
    y = y + solve(sysk["rhs"] - A @ y)      # one step of iterative refinement: the matrix is weakly damped
    return y, A


CASES = [   # name, cameras, edges, error type, loss, outlier fraction, local window (0 = uniformly random graph), sparse LU / dense Cholesky
    ("C2-sized coherent 10k/200k, Geman-McClure", 10000, 200000, _abi.ANGLE_AXIS, lambda: LF.GemanMcClureLoss(0.1, 1.0), 0.1, 120, True),
    ("C2-sized coherent 10k/200k, covariance + MAGSAC", 10000, 200000, _abi.ANGLE_AXIS_COVARIANCE, lambda: LF.MAGSACWeightBasedLoss(0.02), 0.1, 120, True),
    ("random 2000/40k, covariance + MAGSAC", 2000, 40000, _abi.ANGLE_AXIS_COVARIANCE, lambda: LF.MAGSACWeightBasedLoss(0.02), 0.3, 0, False),
    ("Trafalgar-sized random 5288/80k, quaternion Huber", 5288, 80000, _abi.QUATERNION_COSINE, lambda: LF.HuberLoss(0.1), 0.3, 0, False),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_oracle_pcg_step_equals_a_sparse_direct_solve(oracle, case):
    name, n, e, et, mk, outl, window, sparse = CASES[case]
    g = synth.make_graph(n, e, 11, outlier_frac=outl, local_window=window)
    o = oracle.OracleProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"])
    o.set_loss(mk())
    o.set_linear_solver("pcg")
    o.capture_steps(3)
    _, s = o.solve(g["init_aa"], max_num_iterations=3)
    assert s["num_iterations"] == 3
    for k in range(3):
        sysk = o.captured_step(k)
        assert sysk["cg"] > 0                                     # it WAS the PCG path
        y, A = direct_step(sysk, g["edge_i"], g["edge_j"], n, sparse)
        rel = np.linalg.norm(sysk["y"] - y) / np.linalg.norm(y)
        res = np.linalg.norm(A @ sysk["y"] - sysk["rhs"]) / np.linalg.norm(sysk["rhs"])
        print("%s, LM iteration %d: PCG %d iterations, |y_pcg - y_direct| / |y_direct| = %.2e, PCG residual %.1e, damping min %.1e" % (name, k + 1, sysk["cg"], rel, res, (sysk["D"] ** 2).min()))
        assert rel <= 1e-10, (name, k, rel)
