"""Round-5 probe of the forcing schedule's known misses (profiles/r04_fuzz_forcing.txt trials 77 / 94 of seed 1, r04b_kappa_sweep.txt trials 1 / 35
of seed 9): how does the distance of a schedule's answer from the exact schedule's scale with the allowed deviation per step, and where do the two
trajectories part?
usage: python tools/r05_forcing_probe.py seed:trial[,trial...] [seed:trial...]   (environment PROBE_EPS="1e-8,1e-9,..." overrides the sweep)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "manual"))
import fuzz_forcing
from globalsfmpy_amd import synth
from globalsfmpy_amd.solver import RotationProblem


def main():
    np.set_printoptions(linewidth=250, precision=4)
    eps_list = [float(v) for v in os.environ.get("PROBE_EPS", "1e-8,1e-9,1e-10,1e-11").split(",")]
    for spec in sys.argv[1:]:
        seed, trials = spec.split(":")
        dense = seed.endswith("d")
        seed = seed.rstrip("d")
        only = [int(v) for v in trials.split(",")]
        for t, g, et, loss, init, coherent in fuzz_forcing.cases(max(only) + 1, int(seed), only, dense):
            p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"], inlier_weight=g["inlier_weight"])
            p.set_loss(loss)
            r0, s0 = p.solve(init, pcg_forcing=0)
            t0 = p.trace()
            print("seed %s trial %d: n=%d e=%d %s et=%d %s  exact schedule: %d LM it, term %d, %d PCG it, %.1f ms" % (
                seed, t, g["n_cams"], len(g["edge_i"]), "coherent" if coherent else "random", et, type(loss).__name__, s0["num_iterations"], s0["termination"],
                s0["num_cg_iterations"], s0["t_total_ms"]), flush=True)
            for tol in (1e-13, 1e-11, 1e-10):   # the exact schedule's own sensitivity to the linear solve's tolerance
                r, s = p.solve(init, pcg_forcing=0, cg_relative_tolerance=tol)
                d = synth.angular_distance(synth.align_rotations(r, r0), r0)
                print("   exact schedule at cg tol %.0e: LM %d PCG %5d  dR mean %.1e max %.1e" % (tol, s["num_iterations"], s["num_cg_iterations"], d.mean(), d.max()), flush=True)
            first = None
            for eps in eps_list:
                r, s = p.solve(init, pcg_forcing_tolerance=abs(eps))
                tr = p.trace()
                d = synth.angular_distance(synth.align_rotations(r, r0), r0)
                print("   forcing eps %.0e: LM %d PCG %5d inexact %d refined %d  dR mean %.1e max %.1e  %.1f ms" % (
                    eps, s["num_iterations"], s["num_cg_iterations"], s["num_inexact_steps"], s["num_forcing_refinements"], d.mean(), d.max(), s["t_total_ms"]), flush=True)
                if first is None:
                    first = tr
            if os.environ.get("PROBE_TRACE", "1") != "0":
                k = min(len(t0), len(first))
                print("   [it, cost, dcost, |g|, |dx|, rel_dec, radius, cg]  exact | default")
                for i in range(k):
                    a, b = t0[i], first[i]
                    print("   %3d %.10e %.3e %.2e %.3e %.4f %.2e %4d | %.10e %.3e %.3e %.4f %.2e %4d" % (
                        a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], b[1], b[2], b[4], b[5], b[6], b[7]))
            p.close()


if __name__ == "__main__":
    main()
