"""Test helper: how far does the ORACLE's own answer move when its inputs move by one unit in the last place?

With the MAGSAC losses the objective is a staircase in s (table cell = 2 sigma^2 / 1000, loss_functions.py:304), and on
slow-converging graphs the last LM iterations run at a trust radius of 1e10 and more; there a 1-ulp change of the
measurements flips table cells, accepted steps turn into rejected ones, and the converged rotations move by 1e-5..1e-3
rad -- for the oracle's exact Cholesky against itself.  No implementation can agree with another one more closely than
that, so the convergence-level assertions for those configurations are calibrated against this spread; the 1e-6 rad
bar of north_star is asserted wherever the problem is well-posed (every non-staircase loss to convergence, and the
staircase losses over the first iterations)."""
import numpy as np

from globalsfmpy_amd import synth


def ulp_perturbed(rel_aa, rng):
    """Every measurement component moved by -1, 0 or +1 ulp (relative 2.2e-16)."""
    return rel_aa * (1.0 + rng.integers(-1, 2, rel_aa.shape) * 2.220446049250313e-16)


def oracle_spread(make_oracle, rel_aa, x0, ref_rot, n_runs=3, seed=0, **solve_kw):
    """make_oracle(rel_aa) -> OracleProblem with loss and linear solver set.  Returns (mean-distance list, max-distance list,
    iteration-count list) of `n_runs` oracle solves on 1-ulp-perturbed measurements against `ref_rot`."""
    rng = np.random.default_rng(seed)
    means, maxs, iters = [], [], []
    for _ in range(n_runs):
        r, s = make_oracle(ulp_perturbed(rel_aa, rng)).solve(x0, **solve_kw)
        d = synth.angular_distance(synth.align_rotations(r, ref_rot), ref_rot)
        means.append(float(d.mean())); maxs.append(float(d.max())); iters.append(int(s["num_iterations"]))
    return means, maxs, iters
