"""Randomised differential run: device (HIP, through the C-ABI) against the CPU oracle on small adversarial graphs -- rotations over all of
SO(3) (angles up to pi), repeated camera pairs, isolated cameras, zero and pi relative rotations, cameras started on the cut locus, every
error type, every loss.  tests/test_gpu_fuzz.py runs a short fixed-seed pass; by hand:
    python tests/manual/fuzz_differential.py [trials] [seed]        (DESIGN.md section 2 for the tally)
Residuals, s, rho, cost, gradient, diagonal blocks and the normal-equation mat-vec are compared on every trial; full solves on the
smooth losses.  A solve that disagrees is re-run on the ORACLE with its inputs moved by one ulp: hard cases (large initial error, trust
radius at its 1e16 cap, i.e. vanishing damping on a matrix with a gauge null space) amplify rounding by x3 per LM iteration, for any
implementation; a disagreement counts only if it is beyond ten times the oracle's own spread AND the two cost
traces neither coincide until the damping vanishes nor drift apart smoothly from rounding level (chaotic LM iteration on a hard case).  FUZZ_ONLY=t1,t2 replays single trials verbosely."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from oracle import pyoracle

ETS = [_abi.QUATERNION_NORM, _abi.ROTATION_MAT_FNORM, _abi.QUATERNION_COSINE, _abi.ANGLE_AXIS_COVARIANCE, _abi.ANGLE_AXIS, _abi.ANGLE_AXIS_INLIERS,
       _abi.ANGLE_AXIS_COV_INLIERS, _abi.ANGLE_AXIS_COVTRACE, _abi.ANGLE_AXIS_COVNORM]


def random_loss(rng):
    a = float(np.exp(rng.uniform(np.log(1e-3), np.log(3.0))))
    k = int(rng.integers(0, 16))
    if k == 0: return LF.TrivialLoss()
    if k == 1: return LF.HuberLoss(a)
    if k == 2: return LF.SoftLOneLoss(a)
    if k == 3: return LF.CauchyLoss(a)
    if k == 4: return LF.ArctanLoss(a)
    if k == 5: return LF.TolerantLoss(a, a * float(rng.uniform(0.05, 0.5)))
    if k == 6: return LF.TukeyLoss(a)
    if k == 7: return LF.LOneHalfLoss(a)
    if k == 8: return LF.LTwoLoss(a, 1.0)
    if k == 9: return LF.GemanMcClureLoss(a, float(rng.uniform(0.1, 2.0)))
    if k == 10: return LF.MAGSACWeightBasedLoss(float(rng.uniform(0.005, 0.5)))
    if k == 11: return LF.MAGSACWeightBasedLoss4(float(rng.uniform(0.005, 0.5)))
    if k == 12: return LF.MAGSACWeightBasedLoss9(float(rng.uniform(0.005, 0.5)))
    if k == 13: return LF.ScaledLoss(LF.HuberLoss(a), float(rng.uniform(0.1, 5.0)))
    if k == 14: return LF.ComposedLoss(LF.SoftLOneLoss(a), LF.HuberLoss(a * 2))
    return None   # Ceres NULL loss


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if not np.array_equal(np.isfinite(a), np.isfinite(b)):
        return float("inf")      # one side non-finite where the other is not
    m = np.isfinite(b)           # (both non-finite in the same places: the reference's Corrector does the same, e.g. rho' = 0 < rho'')
    if not m.any():
        return 0.0
    return float(np.abs(a[m] - b[m]).max() / max(1e-300, np.abs(b[m]).max()))


def run(trials=200, seed=1, quick=False):
    """Returns the number of unexplained mismatches.  quick: full solves only where convergence is fast (small initial error)."""
    rng = np.random.default_rng(seed)
    bad = 0
    for t in range(trials):
        n = int(rng.integers(3, 200))
        e = int(rng.integers(n - 1, min(n * (n - 1) // 2, 12 * n) + 1))
        gseed, outl, full, inoise, noisy = int(rng.integers(1 << 30)), float(rng.uniform(0, 0.4)), bool(rng.integers(2)), float(rng.choice([0.0, 2.0, 30.0, 120.0])), bool(rng.integers(4))
        g = synth.make_graph(n, e, gseed, outlier_frac=outl, full_so3=full, init_noise_deg=inoise, noise=noisy)
        ei, ej, rel, c6, iw = g["edge_i"].copy(), g["edge_j"].copy(), g["rel_aa"].copy(), g["cov6"].copy(), g["inlier_weight"].copy()
        mode = int(rng.integers(0, 5))
        if mode == 1 and e > 4:      # repeated camera pairs, one of them reversed
            k = int(rng.integers(1, min(e, 20)))
            ei = np.concatenate([ei, ej[:k]]); ej = np.concatenate([ej, g["edge_i"][:k]])
            rel = np.concatenate([rel, -rel[:k]]); c6 = np.concatenate([c6, c6[:k]]); iw = np.concatenate([iw, iw[:k]])
        if mode == 2:                # isolated cameras at the end and in the middle of the index range
            n += 3
        if mode == 3:                # exact zero / exact pi relative rotations
            rel[: max(1, e // 10)] = 0.0
            rel[-1] = np.array([np.pi, 0.0, 0.0])
        init = g["init_aa"] if n == g["n_cams"] else np.concatenate([g["init_aa"], 0.1 * rng.standard_normal((n - g["n_cams"], 3))])
        if mode == 4:                # cameras started exactly at the identity / at an angle of pi
            init = init.copy(); init[0] = 0.0; init[-1] = np.array([0.0, np.pi, 0.0])
        et = ETS[int(rng.integers(len(ETS)))]
        loss = random_loss(rng)
        v = rng.standard_normal((n, 3))
        only = os.environ.get("FUZZ_ONLY")
        if only and t not in [int(x) for x in only.split(",")]:
            continue
        dev = RotationProblem(n, ei, ej, rel, et, cov6=c6, inlier_weight=iw); dev.set_loss(loss)
        ora = pyoracle.OracleProblem(n, ei, ej, rel, et, cov6=c6, inlier_weight=iw); ora.set_loss(loss)
        tag = "trial %d n=%d e=%d mode=%d et=%d loss=%s" % (t, n, len(ei), mode, et, type(loss).__name__)
        a, b = dev.residuals(init, want_residuals=True), ora.residuals(init, want_residuals=True)
        la, lb = dev.linearize(init), ora.linearize(init)
        staircase = "MAGSAC" in type(loss).__name__    # table look-up in s: an edge within rounding of a cell boundary lands in either cell
        # scales: residual entries that are pure rounding noise (a noise-free graph evaluated at the truth) are compared absolutely
        r_scale = max(1.0, float(np.abs(b["residuals"]).max()))
        jmax = float(np.sqrt(np.abs(np.asarray(lb["diag_blocks"])).max()))
        rho1 = np.asarray(b["rho"])[:, 1]
        sr = float(np.sqrt(max(1.0, np.abs(rho1[np.isfinite(rho1)]).max()))) if np.isfinite(rho1).any() else 1.0   # (the robustified residual is sqrt(rho') r)
        g_scale = jmax * sr * max(float(np.sqrt(b["s"].max())), 1e-9) + 1e-300   # ~ |J| |r|, floored where r itself is rounding noise
        def scaled(x, y, scale):
            x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
            if not np.array_equal(np.isfinite(x), np.isfinite(y)):
                return float("inf")
            m = np.isfinite(y)
            return float(np.abs(x[m] - y[m]).max() / scale) if m.any() else 0.0
        errs = {"residuals": scaled(a["residuals"], b["residuals"], r_scale), "s": scaled(a["s"], b["s"], max(1.0, float(b["s"].max()))),
                "rho": 0.0 if staircase else relerr(a["rho"], b["rho"]),
                "cost": abs(a["cost"] - b["cost"]) / max(1.0, abs(b["cost"])),
                "gradient": 0.0 if staircase else max(0.0, scaled(la["gradient"], lb["gradient"], g_scale) - 1e-13 * jmax / g_scale),   # (absolute floor: |J| x rounding of r)
                "blocks": 0.0 if staircase else relerr(la["diag_blocks"], lb["diag_blocks"]),
                "matvec": 0.0 if staircase else relerr(dev.normal_matvec(v), ora.normal_matvec(v))}
        if staircase:   # at most a handful of edges may sit on a cell boundary
            flips = int((np.abs(a["rho"] - b["rho"]).max(axis=1) > 1e-9 * np.abs(b["rho"]).max()).sum())
            errs["rho"] = 0.0 if flips <= max(2, len(ei) // 200) else float(flips)
        # L1/2: rho ~ s^(1/4) is not Lipschitz at 0 -- a noise-level s of 1e-31 that differs by its own size moves the cost by 1e-9 per edge
        tol = {"residuals": 1e-11, "s": 1e-11, "rho": 1e-9, "cost": 1e-7 if type(loss).__name__ == "LOneHalfLoss" else 1e-11, "gradient": 1e-8, "blocks": 1e-8, "matvec": 1e-8}
        worst = [k for k in errs if not errs[k] <= tol[k]]
        if worst:
            bad += 1
            print("MISMATCH", tag, {k: errs[k] for k in worst}, flush=True)
            if only:
                gd, go = np.asarray(la["gradient"]), np.asarray(lb["gradient"])
                print("  device gradient: nan rows", np.flatnonzero(~np.isfinite(gd).all(axis=1))[:10], " oracle gradient: nan rows", np.flatnonzero(~np.isfinite(go).all(axis=1))[:10])
                dd = np.nan_to_num(np.abs(gd - go).max(axis=1)); k = int(np.argmax(dd)); print("  worst camera", k, gd[k], go[k])
                inc = np.flatnonzero((ei == k) | (ej == k))
                for q in inc[:40]:
                    print("   edge %d (%d,%d) s dev %.17g ora %.17g rho dev %s ora %s" % (q, ei[q], ej[q], a["s"][q], b["s"][q], a["rho"][q], b["rho"][q]))
        # full solve on the convex-ish losses (the staircase losses are chaotic by construction: tests/sensitivity.py)
        hard = inoise > 2.0 or full or mode == 4     # slow convergence: dozens of LM iterations, trust radius at its cap
        if type(loss).__name__ in ("TrivialLoss", "HuberLoss", "SoftLOneLoss", "NoneType") and mode != 3 and not (quick and hard):
            rd, sd = dev.solve(init, max_num_iterations=60)
            ro, so = ora.solve(init, max_num_iterations=60)
            act = np.zeros(n, bool); act[ei] = True; act[ej] = True
            dist = synth.angular_distance(synth.align_rotations(rd[act], ro[act]), ro[act]).mean() if act.sum() > 2 else 0.0
            dc = abs(sd["final_cost"] - so["final_cost"]) / max(1e-30, abs(so["final_cost"]), 1e-12)
            if not (dist <= 1e-6 and (dc <= 1e-6 or so["final_cost"] < 1e-20) and sd["termination"] == so["termination"]):
                # how far does the oracle move against ITSELF when its measurements move by one ulp?  (tests/sensitivity.py)
                spread = []
                for k in range(3):
                    rel_p = rel * (1.0 + np.random.default_rng(100 + k).integers(-1, 2, rel.shape) * 2.220446049250313e-16)
                    op = pyoracle.OracleProblem(n, ei, ej, rel_p, et, cov6=c6, inlier_weight=iw); op.set_loss(loss)
                    init_p = init * (1.0 + np.random.default_rng(200 + k).integers(-1, 2, init.shape) * 2.220446049250313e-16)   # (the start too: a camera exactly on the cut locus |angle| = pi)
                    rp, sp = op.solve(init_p, max_num_iterations=60); op.close()
                    spread.append((float(synth.angular_distance(synth.align_rotations(rp[act], ro[act]), ro[act]).mean()) if act.sum() > 2 else 0.0, sp["num_iterations"]))
                print("   oracle vs 1-ulp-perturbed oracle: mean dR / iterations", ["%.2e / %d" % x for x in spread], flush=True)
                if only:
                    td, to = dev.trace(), ora.trace()
                    for k in range(max(len(td), len(to))):
                        a_ = td[k] if k < len(td) else None; b_ = to[k] if k < len(to) else None
                        print("   it %2d dev cost %.15e |g| %.3e |dx| %.3e rho %.3e rad %.2e cg %d | ora cost %.15e |g| %.3e |dx| %.3e rho %.3e rad %.2e" % (
                            k, *( (a_[1], a_[3], a_[4], a_[5], a_[6], int(a_[7])) if a_ is not None else (0, 0, 0, 0, 0, 0)), *((b_[1], b_[3], b_[4], b_[5], b_[6]) if b_ is not None else (0, 0, 0, 0, 0))))
                    print("   dense solves", sd["num_dense_solves"], "cg", sd["num_cg_iterations"])
                if max(x[0] for x in spread) * 10 >= dist:
                    print("   -> within the oracle's own sensitivity", flush=True)
                    dev.close(); ora.close()
                    continue
                # otherwise the two trajectories must at least coincide for as long as the step is well-posed, i.e. until the trust radius
                # has grown past 1e13 (damping below 1e-13: from there rounding grows x3 per iteration and flips borderline decisions)
                td, to = dev.trace(), ora.trace()
                m = min(len(td), len(to))
                well = [k for k in range(m) if td[k][6] < 1e13 and to[k][6] < 1e13]
                k_last = max(well) if well else 0
                dev_cost, ora_cost = np.array([td[k][1] for k in range(k_last + 1)]), np.array([to[k][1] for k in range(k_last + 1)])
                agree = float(np.abs(dev_cost - ora_cost).max() / max(np.abs(ora_cost).max(), 1e-300))
                # or: a hard non-convex case (no robust loss over 30 % outliers, rotations all over SO(3)) whose LM iteration is itself chaotic --
                # the difference starts at rounding level and grows by a steady factor per iteration, with no jump anywhere (a defect in one
                # of the two implementations would show as a step from 1e-13 to something large at the iteration where it bites)
                d = [abs(td[k][1] - to[k][1]) / max(abs(to[k][1]), 1e-300) for k in range(m)]
                smooth = m >= 6 and max(d[:3]) <= 1e-12
                if only:
                    print("   relative cost difference per iteration:", " ".join("%.0e" % x for x in d), flush=True)
                run_max = max(d[0], 1e-15)
                for k in range(1, m):   # from rounding level up to 1e-9 (beyond that borderline accept / reject decisions flip and the costs jump by a step)
                    # (ratios are taken from 1e-11 up: below that the difference is a handful of ulps and its size is luck -- trial 1101 of seed 112 under
                    # the forced layout goes 2e-12 -> 8e-9 with one device build and 2e-11 -> 1e-8 with another in the same 14-rad step across the cut locus)
                    if d[k] > 1000.0 * max(run_max, 1e-11) or (run_max <= 1e-9 < d[k] and d[k] > 1e-6):
                        smooth = False
                    run_max = max(run_max, d[k])
                    if run_max > 1e-9:
                        break
                if smooth:
                    grow = [d[k] / d[k - 1] for k in range(1, m) if 1e-14 < d[k - 1] < 1e-5 and d[k] > 0]
                    print("   -> chaotic iteration: the cost difference grows smoothly from %.0e, x%.1f per iteration (median), no jump" % (max(d[:3]), float(np.median(grow)) if grow else 0.0), flush=True)
                    dev.close(); ora.close()
                    continue
                if k_last >= 3 and agree <= 1e-9:
                    print("   -> trajectories coincide (cost to %.1e) for the %d iterations before the damping vanishes; the rest is rounding amplified by the near-singular steps" % (agree, k_last), flush=True)
                    dev.close(); ora.close()
                    continue
                bad += 1
                print("SOLVE MISMATCH", tag, "mean dR %.2e  cost %.9e vs %.9e  it %d vs %d  term %s vs %s" % (dist, sd["final_cost"], so["final_cost"],
                      sd["num_iterations"], so["num_iterations"], sd["termination_name"], so["termination_name"]), flush=True)
        dev.close(); ora.close()
    print("fuzz: %d trials, seed %d: %d unexplained mismatches" % (trials, seed, bad))
    return bad


if __name__ == "__main__":
    sys.exit(min(1, run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1)))
