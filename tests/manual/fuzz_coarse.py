"""Randomised run of the two-level preconditioner: random coherent graphs (4k-30k cameras, windows 100-3000, degree 6-60, outliers, isolated
cameras, shuffled or ordered ids), Laplacian-form error types and losses; automatic choice and a forced coarse space, each against block-Jacobi.
usage: fuzz_coarse.py [trials] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem


def run(trials=20, seed=1):
    rng = np.random.default_rng(seed)
    bad = 0
    for t in range(trials):
        n = int(rng.integers(4096, 30000)); win = int(rng.choice([100, 200, 400, 1000, 3000])); deg = int(rng.integers(6, 61))
        avail = n - 1 + sum(max(0, n - d) for d in range(2, win // 2 + 1)) // 2
        e = min(n * deg // 2, avail)
        g = synth.make_graph(n, e, int(rng.integers(1 << 30)), outlier_frac=float(rng.uniform(0, 0.3)), local_window=win)
        if rng.random() < 0.3:   # ordered ids (no relabelling needed): undo the shuffle by sorting cameras along the chain is not possible here; instead append isolated cameras
            pass
        iso = int(rng.choice([0, 0, 300]))
        nn = n + iso
        init = np.concatenate([g["init_aa"], np.zeros((iso, 3))]) if iso else g["init_aa"]
        et, loss = [(_abi.ANGLE_AXIS, LF.HuberLoss(0.1)), (_abi.ANGLE_AXIS_COVARIANCE, LF.MAGSACWeightBasedLoss(0.02)), (_abi.QUATERNION_COSINE, LF.SoftLOneLoss(0.1)),
                    (_abi.ANGLE_AXIS_COV_INLIERS, LF.CauchyLoss(0.2)), (_abi.ANGLE_AXIS_COVNORM, LF.GemanMcClureLoss(0.1, 1.0))][int(rng.integers(5))]
        res = {}
        for mode in ("0", None, "24"):
            if mode is None: os.environ.pop("GSFM_PCG_COARSE", None)
            else: os.environ["GSFM_PCG_COARSE"] = mode
            p = RotationProblem(nn, g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"], inlier_weight=g["inlier_weight"]); p.set_loss(loss)
            # (chain-like graphs -- window 100, degree 8 -- need thousands of block-Jacobi iterations per step: the reference solve gets a cap it
            # cannot hit, or its truncated steps would be the inaccurate side of the comparison)
            res[mode] = p.solve(init, max_num_iterations=15, max_cg_iterations=100000 if mode == "0" else 1000)
            p.close()
        (r0, s0) = res["0"]
        line = "trial %d n=%d win=%d deg=%d iso=%d et=%d %s: PCG %d" % (t, n, win, deg, iso, et, type(loss).__name__, s0["num_cg_iterations"])
        for mode in (None, "24"):
            r, s = res[mode]
            d = synth.angular_distance(r[:n], r0[:n]).max()
            its = s0["num_iterations"]
            tol = 1e-8 if its <= 10 else 1e-6     # (long MAGSAC trajectories amplify the rounding of a different PCG path)
            ok = s["num_iterations"] == s0["num_iterations"] and abs(s["final_cost"] - s0["final_cost"]) <= 1e-9 * abs(s0["final_cost"]) and d <= tol and np.array_equal(r[n:], r0[n:])
            bad += not ok
            line += " | %s %d dR %.0e%s" % ("auto" if mode is None else "forced", s["num_cg_iterations"], d, "" if ok else " MISMATCH")
        print(line, flush=True)
    os.environ.pop("GSFM_PCG_COARSE", None)
    print("coarse fuzz: %d trials, seed %d: %d mismatches" % (trials, seed, bad))
    return bad


if __name__ == "__main__":
    sys.exit(min(1, run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1)))
