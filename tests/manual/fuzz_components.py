"""Randomised run of the per-component step of disconnected view graphs (round 5, csrc/solver_components.hpp): 2-6 scenes of 40-1500 cameras batched
as one problem -- contiguous or interleaved in one random numbering, near or far starts, every error type, smooth losses and MAGSAC -- solved with
the DEFAULT options (small components factorised side by side, PCG on the large ones, components at rest once converged, absolute floor of the
tolerance) against the CPU oracle, component by component (each has its own gauge).  A trial passes with the oracle's LM iteration count, no
capped step and every component within 1e-6 rad (mean); trials whose ORACLE moves by a comparable amount under 1-ulp perturbations of the
measurements -- or when started 1e-13 rad away, the size of what separates two correct linear solvers -- are reported as ill-posed.
usage: python tests/manual/fuzz_components.py [trials] [seed] [only: trial,trial...]   (FUZZ_PROBE=1: the chosen trials row by row against the one-PCG path)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem

ETS = [_abi.ROTATION_MAT_FNORM, _abi.QUATERNION_COSINE, _abi.ANGLE_AXIS_COVARIANCE, _abi.ANGLE_AXIS, _abi.ANGLE_AXIS_INLIERS, _abi.ANGLE_AXIS_COV_INLIERS,
       _abi.ANGLE_AXIS_COVTRACE, _abi.ANGLE_AXIS_COVNORM]


def cases(trials, seed, only=None):
    """The trials of run(trials, seed), generated without touching the device: (t, N, ei, ej, rel, cov, inl, init, comp, sizes, shuffled, et, loss)."""
    rng = np.random.default_rng(seed)
    for t in range(trials):
        k = int(rng.integers(2, 7))
        sizes = [int(np.exp(rng.uniform(np.log(40), np.log(1500)))) for _ in range(k)]
        scenes = [synth.make_graph(n, min(int(n * rng.uniform(6, 20)), 2 * n * (n - 1) // 5), int(rng.integers(1, 1 << 30)), outlier_frac=float(rng.uniform(0, 0.3))) for n in sizes]   # (at most 80 % of the pairs: a 40-camera scene has 780)
        offs = np.cumsum([0] + sizes)
        N = int(offs[-1])
        ei = np.concatenate([g["edge_i"] + o for o, g in zip(offs, scenes)]).astype(np.int64)
        ej = np.concatenate([g["edge_j"] + o for o, g in zip(offs, scenes)]).astype(np.int64)
        rel = np.concatenate([g["rel_aa"] for g in scenes]); cov = np.concatenate([g["cov6"] for g in scenes]); inl = np.concatenate([g["inlier_weight"] for g in scenes])
        init = np.concatenate([g["init_aa"] for g in scenes])
        comp = np.concatenate([np.full(n, c) for c, n in enumerate(sizes)])
        if rng.random() < 0.3:
            init = init + float(rng.uniform(0.03, 0.2)) * rng.standard_normal(init.shape)
        shuffled = rng.random() < 0.5
        if shuffled:
            perm = rng.permutation(N)
            ei, ej = perm[ei], perm[ej]
            sw = ei > ej
            rel = rel.copy(); rel[sw] = -rel[sw]
            ei, ej = np.where(sw, ej, ei), np.where(sw, ei, ej)
            inv = np.empty(N, dtype=np.int64); inv[perm] = np.arange(N)
            init, comp = init[inv], comp[inv]
        et = ETS[int(rng.integers(len(ETS)))]
        a = float(np.exp(rng.uniform(np.log(0.05), np.log(1.0))))
        loss = [LF.HuberLoss(a), LF.SoftLOneLoss(a), LF.CauchyLoss(a), LF.GemanMcClureLoss(a, 1.0), LF.TrivialLoss(), LF.TukeyLoss(max(a, 0.3)),
                LF.MAGSACWeightBasedLoss(0.02)][int(rng.integers(7))]
        if only is not None and t not in only:
            continue
        yield t, N, ei.astype(np.uint32), ej.astype(np.uint32), rel, cov, inl, init, comp, sizes, shuffled, et, loss


def run(trials=30, seed=1, only=None, **solve_kw):
    from oracle import pyoracle
    from sensitivity import ulp_perturbed
    bad = 0
    for t, N, ei, ej, rel, cov, inl, init, comp, sizes, shuffled, et, loss in cases(trials, seed, only):
        k = len(sizes)
        p = RotationProblem(N, ei, ej, rel, et, cov6=cov, inlier_weight=inl); p.set_loss(loss)
        rd, sd = p.solve(init, **solve_kw)
        if os.environ.get("FUZZ_PROBE"):   # one trial under the lens: the component step against the device's own one-PCG-over-everything path, row by row
            tr1 = p.trace()
            ra, sa = p.solve(init, dense_cholesky_max_cams=0, dense_cholesky_auto_cams=0)
            tr2 = p.trace()
            per = lambda x, y: np.array([synth.angular_distance(synth.align_rotations(x[comp == c], y[comp == c]), y[comp == c]).mean() for c in range(k)])
            print("   component step %d it, one PCG %d it; per component, component step vs one PCG: %s" % (sd["num_iterations"], sa["num_iterations"], " ".join("%.1e" % v for v in per(rd, ra))))
            os.makedirs(os.path.join(ROOT, "gpurun_out", "r05"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "r05", "comp_probe_%d_%d.npz" % (seed, t)), component_step=rd, one_pcg=ra, comp=comp, trace_component_step=tr1, trace_one_pcg=tr2)
            print("   [it, cost, |dx|, rel_dec, radius, cg] component step | one PCG")
            for i in range(min(len(tr1), len(tr2))):
                a1, a2 = tr1[i], tr2[i]
                print("   %3d %.12e %.3e %.6f %.2e %4d | %.12e %.3e %.6f %.2e %4d%s" % (a1[0], a1[1], a1[4], a1[5], a1[6], a1[7], a2[1], a2[4], a2[5], a2[6], a2[7], "" if abs(a1[1] - a2[1]) <= 1e-9 * abs(a2[1]) else "  <"))
        p.close()
        o = pyoracle.OracleProblem(N, ei, ej, rel, et, cov6=cov, inlier_weight=inl); o.set_loss(loss)
        ro, so = o.solve(init)
        if os.environ.get("FUZZ_PROBE"):
            print("   oracle %d it; per component, one PCG vs oracle: %s; component step vs oracle: %s" % (so["num_iterations"], " ".join("%.1e" % v for v in per(ra, ro)), " ".join("%.1e" % v for v in per(rd, ro))))
        d = np.array([synth.angular_distance(synth.align_rotations(rd[comp == c], ro[comp == c]), ro[comp == c]).mean() for c in range(k)])
        ok = sd["num_iterations"] == so["num_iterations"] and sd["num_pcg_capped_steps"] == 0 and d.max() <= 1e-6
        verdict = "ok" if ok else "MISMATCH"
        if not ok:   # is the problem well-posed?  the oracle against itself on measurements moved by one ulp
            o2 = pyoracle.OracleProblem(N, ei, ej, ulp_perturbed(rel, np.random.default_rng(100)), et, cov6=cov, inlier_weight=inl); o2.set_loss(loss)
            r2, s2 = o2.solve(init)
            spread = max(synth.angular_distance(synth.align_rotations(r2[comp == c], ro[comp == c]), ro[comp == c]).mean() for c in range(k))
            if spread >= 0.1 * d.max() or s2["num_iterations"] != so["num_iterations"]:
                verdict = "ill-posed (oracle vs 1-ulp oracle: %.1e rad, %d it)" % (spread, s2["num_iterations"])
            else:
                # ... or under a perturbation the size of what separates two correct linear solvers (the oracle's global PCG at 1e-14 against exact
                # factorisations: steps that differ by ~1e-13 rad): the same oracle started 1e-13 rad (rms per component of the vector) away
                r3, s3 = o.solve(init + 1e-13 * np.random.default_rng(101).standard_normal(init.shape))
                spread3 = max(synth.angular_distance(synth.align_rotations(r3[comp == c], ro[comp == c]), ro[comp == c]).mean() for c in range(k))
                if spread3 >= 0.1 * d.max() or s3["num_iterations"] != so["num_iterations"]:
                    verdict = "ill-posed (oracle started 1e-13 rad away: %.1e rad, %d it)" % (spread3, s3["num_iterations"])
                else:
                    # ... or under the accuracy of the linear solve itself: the oracle's answer is its PCG(1e-14)'s, not the reference's Cholesky's (too slow on
                    # the CPU at this size).  The device's one-PCG-over-everything path -- the oracle's algorithm -- at 1e-14 and at 1e-15:
                    p2 = RotationProblem(N, ei, ej, rel, et, cov6=cov, inlier_weight=inl); p2.set_loss(loss)
                    ra, sa = p2.solve(init, dense_cholesky_max_cams=0, dense_cholesky_auto_cams=0)
                    rb, sb = p2.solve(init, dense_cholesky_max_cams=0, dense_cholesky_auto_cams=0, cg_relative_tolerance=1e-15)
                    p2.close()
                    spread4 = max(synth.angular_distance(synth.align_rotations(rb[comp == c], ra[comp == c]), ra[comp == c]).mean() for c in range(k))
                    if spread4 >= 0.1 * d.max() or sa["num_iterations"] != sb["num_iterations"]:
                        verdict = "ill-posed (one PCG over everything at 1e-14 against 1e-15: %.1e rad, %d / %d it)" % (spread4, sa["num_iterations"], sb["num_iterations"])
                    else:
                        # ... or the ORACLE's linear solver is the one that is off (round 6): beyond 512 cameras the oracle iterates (PCG to 1e-14 over all
                        # components), the reference factorises (estimator.cpp:299-305) and so does the device, component by component.  Where a dense
                        # Cholesky of the whole batch is affordable on the CPU, the oracle solves once more with it: the device must follow THAT run.
                        dd = None
                        if N <= 2500:
                            o3 = pyoracle.OracleProblem(N, ei, ej, rel, et, cov6=cov, inlier_weight=inl); o3.set_loss(loss); o3.set_linear_solver("dense")
                            r4, s4 = o3.solve(init)
                            dd = np.array([synth.angular_distance(synth.align_rotations(rd[comp == c], r4[comp == c]), r4[comp == c]).mean() for c in range(k)])
                            do = max(synth.angular_distance(synth.align_rotations(ro[comp == c], r4[comp == c]), r4[comp == c]).mean() for c in range(k))
                        if dd is not None and dd.max() <= 1e-6 and s4["num_iterations"] == sd["num_iterations"]:
                            verdict = "ok against the oracle's EXACT Cholesky steps (%.1e rad, %d it); the oracle's PCG(1e-14) run is %.1e rad / %d it from its own Cholesky run" % (dd.max(), s4["num_iterations"], do, so["num_iterations"])
                        elif dd is not None and (do >= 0.1 * d.max() or s4["num_iterations"] != so["num_iterations"]):   # (the oracle's own two solvers part company: by distance, or by LM iteration count)
                            verdict = "ill-posed (the oracle's Cholesky against its own PCG(1e-14): %.1e rad, %d / %d it; device vs its Cholesky run %.1e rad)" % (do, s4["num_iterations"], so["num_iterations"], dd.max())
                        else:
                            bad += 1
        print("trial %3d sizes %-32s %s et=%d %-24s %s LM %2d/%2d component steps %2d PCG %5d  worst component %.1e rad  %s" % (
            t, sizes, "shuffled" if shuffled else "in order", et, type(loss).__name__, "", sd["num_iterations"], so["num_iterations"], sd["num_dense_solves"],
            sd["num_cg_iterations"], d.max(), verdict), flush=True)
    print("component fuzz: %d trials, seed %d: %d mismatches" % (trials, seed, bad))
    return bad


if __name__ == "__main__":
    sys.exit(min(1, run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1, only=[int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None)))
