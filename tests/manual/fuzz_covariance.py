"""Randomised device-vs-oracle run of the per-edge covariance estimation (gsfm_cov_estimate, K7; reference src/uncertainty.cpp:36-162) on
awkward view pairs: very few matches, coplanar scenes, (nearly) pure rotations, gross outliers among the matches, poor initial poses,
zero translation (the reference's skip rule), hundreds to thousands of matches.  usage: fuzz_covariance.py [batches] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from globalsfmpy_amd import covariance as cv
from oracle import pyoracle


def awkward_batch(rng, n_edges):
    b = cv.make_two_view_batch(n_edges, int(rng.integers(1 << 30)), matches_per_edge=(5, 60), noise_px=float(rng.choice([0.05, 0.5, 3.0])),   # (never exactly noise-free: a residual of exactly 0 has no autodiff derivative in the reference)
                               init_rot_noise=float(rng.choice([0.0, 0.01, 0.2])), init_t_noise=float(rng.choice([0.0, 0.02, 0.5])))
    ptr = b["match_ptr"].astype(np.int64)
    kinds = []
    for e in range(n_edges):
        lo, hi = ptr[e], ptr[e + 1]
        k = int(rng.integers(0, 8))
        kinds.append(k)
        if k == 1:      # gross outliers among the matches
            m = rng.random(hi - lo) < 0.3
            b["matches"][lo:hi][m] = rng.uniform(0, 1200, (int(m.sum()), 4))
        elif k == 2:    # zero translation: the reference skips the pair
            b["trans"][e] = 0.0
        elif k == 3:    # tiny translation (nearly a pure rotation)
            b["trans"][e] *= 1e-9
        elif k == 4:    # all matches identical
            b["matches"][lo:hi] = b["matches"][lo]
        elif k == 5:    # coplanar / collinear image points
            b["matches"][lo:hi, 1] = b["matches"][lo, 1]; b["matches"][lo:hi, 3] = b["matches"][lo, 3]
        elif k == 6:    # far-off initial rotation
            b["rot"][e] += rng.standard_normal(3)
    return b, kinds


def run(batches=20, seed=1, n_edges=200):
    rng = np.random.default_rng(seed)
    o = pyoracle
    bad = drifted = 0
    for t in range(batches):
        b, kinds = awkward_batch(rng, n_edges)
        dev = cv.estimate_rotation_covariances(b["match_ptr"], b["matches"], b["intrinsics"], b["rot"], b["trans"])
        ora = o.estimate_rotation_covariances(b["match_ptr"], b["matches"], b["intrinsics"], b["rot"], b["trans"])
        for e in range(n_edges):
            why = []
            if dev["status"][e] != ora["status"][e]:
                why.append("status %d vs %d" % (dev["status"][e], ora["status"][e]))
            elif ora["status"][e] == 0:
                # (iteration counts are not compared: with the minimal 5 matches the cost goes to zero and the relative function tolerance
                # trips an iteration earlier or later on rounding, with the same pose and covariance)
                dr = np.abs(dev["rotation"][e] - ora["rotation"][e]).max()
                co, cd = ora["cov"][e], dev["cov"][e]
                if not (np.isfinite(cd).all() == np.isfinite(co).all()):
                    why.append("finite-ness of the covariance differs")
                elif np.isfinite(co).all():
                    rel = np.abs(cd - co).max() / max(np.abs(co).max(), 1e-300)
                    ev = np.abs(np.linalg.eigvalsh(0.5 * (co + co.T)))
                    cond = ev.max() / max(ev.min(), 1e-300)     # the inverse of H is only as good as H's conditioning allows
                    if ora["iterations"][e] > 40:     # dozens to hundreds of LM iterations from a far-off start or through gross outliers: rounding is
                        drifted += int(dr > 1e-6)     # amplified along the way (as in the rotation solver's hard cases); tallied, not judged
                    elif dr > 1e-8 * max(1.0, cond * 1e-8) or rel > 1e-6 + 1e-13 * cond:
                        why.append("rotation %.1e covariance %.1e" % (dr, rel))
            if why:
                bad += 1
                n = int(b["match_ptr"][e + 1] - b["match_ptr"][e])
                print("MISMATCH batch %d edge %d kind %d matches %d: %s  (oracle iterations %d)" % (t, e, kinds[e], n, "; ".join(why), ora["iterations"][e]), flush=True)
    print("covariance fuzz: %d view pairs, seed %d: %d mismatches (%d long trajectories ended > 1e-6 rad apart)" % (batches * n_edges, seed, bad, drifted))
    return bad


if __name__ == "__main__":
    sys.exit(min(1, run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1)))
