"""Randomised three-way run of the loss programs: device (gsfm_rot_loss_eval: loss_eval<LM> / loss_value<LM> of csrc/loss_dev.hpp), the CPU
oracle's interpreter and the Python classes (globalsfmpy_amd/loss_functions.py, pinned to the reference's recorded vectors) on random
NESTED programs -- ScaledLoss / ComposedLoss trees up to the interpreter's node and stack limits -- over squared norms from 0 and 1e-300 to
1e+20.  usage: fuzz_loss.py [programs] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from oracle import pyoracle


def leaf(rng):
    a = float(np.exp(rng.uniform(np.log(1e-3), np.log(10.0))))
    k = int(rng.integers(0, 13))
    return [LF.TrivialLoss, lambda: LF.HuberLoss(a), lambda: LF.SoftLOneLoss(a), lambda: LF.CauchyLoss(a), lambda: LF.ArctanLoss(a),
            lambda: LF.TolerantLoss(a, a * float(rng.uniform(0.05, 0.5))), lambda: LF.TukeyLoss(a), lambda: LF.LOneHalfLoss(a), lambda: LF.LTwoLoss(a, 1.0),
            lambda: LF.GemanMcClureLoss(a, float(rng.uniform(0.1, 2.0))), lambda: LF.MAGSACWeightBasedLoss(float(rng.uniform(0.005, 0.5))),
            lambda: LF.MAGSACWeightBasedLoss4(float(rng.uniform(0.005, 0.5))), lambda: LF.MAGSACWeightBasedLoss9(float(rng.uniform(0.005, 0.5)))][k]()


def tree(rng, depth):
    if depth == 0 or rng.random() < 0.3:
        return leaf(rng)
    if rng.random() < 0.5:
        return LF.ScaledLoss(tree(rng, depth - 1), float(np.exp(rng.uniform(-3, 3))))
    return LF.ComposedLoss(tree(rng, depth - 1), tree(rng, depth - 1))


def run(programs=300, seed=1):
    rng = np.random.default_rng(seed)
    g = synth.make_graph(n_cams=8, n_edges=16, seed=3)
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
    s_grid = np.concatenate([[0.0, 1e-300, 1e-200, 1e-100], np.exp(rng.uniform(np.log(1e-30), np.log(1e20), 60))])
    bad = skipped = 0
    for t in range(programs):
        loss = tree(rng, int(rng.integers(0, 4)))
        nodes = loss.native_program()
        if nodes is None or len(nodes) > 16:
            skipped += 1
            continue
        try:
            p.set_loss(loss)
        except Exception as e:      # beyond the interpreter's stack depth: must be refused, not mis-evaluated
            skipped += 1
            continue
        s = np.concatenate([s_grid, np.exp(rng.uniform(np.log(1e-8), np.log(1e2), 40))])
        rho3, val = p.loss_eval(s)
        for k, sk in enumerate(s):
            want = pyoracle.loss_eval(nodes, float(sk))
            buf = [0.0, 0.0, 0.0]
            try:
                loss.Evaluate(float(sk), buf)
            except (ZeroDivisionError, OverflowError, ValueError):
                buf = None      # the Python class raises where IEEE arithmetic gives inf / nan: the compiled sides are compared with each other
            for name, got, ref in (("device vs oracle", rho3[k], want),) + ((("oracle vs python", want, np.array(buf)),) if buf is not None else ()):
                for c in range(3):
                    a_, b_ = float(got[c]), float(ref[c])
                    if np.isnan(a_) and np.isnan(b_):
                        continue
                    if np.isinf(a_) or np.isinf(b_):
                        ok = a_ == b_
                    else:
                        # relative 1e-9, or absolute at 1e-12 of the row's scale: the reference's own formulas cancel (Tukey's 1 - (1 - s/a^2)^3 at
                        # small s, f'' g'^2 + f' g'' of a composition at huge s, MAGSAC's w(0) - w(s)), and the device contracts a*b+c into FMAs
                        # where the host compilers do not -- both sides are then equally far from the exact value
                        fin = np.abs(np.asarray(ref, dtype=np.float64)); fin = fin[np.isfinite(fin)]
                        ok = abs(a_ - b_) <= 1e-9 * max(abs(b_), abs(a_)) + 1e-12 * max(1.0, float(fin.max()) if fin.size else 1.0)
                    if not ok:
                        bad += 1
                        print("MISMATCH program %d (%s) s=%.17g component %d: %s %.17g vs %.17g   nodes %s" % (t, type(loss).__name__, sk, c, name, a_, b_, nodes), flush=True)
                        break
            # cost-only routine against the full one
            if np.isfinite(rho3[k][0]) and not abs(val[k] - rho3[k][0]) <= 1e-12 * max(abs(rho3[k][0]), 1e-300) + 1e-300:
                bad += 1
                print("MISMATCH program %d s=%.17g: loss_value %.17g vs loss_eval %.17g  nodes %s" % (t, sk, val[k], rho3[k][0], nodes), flush=True)
    p.close()
    print("loss fuzz: %d programs (%d beyond the interpreter's limits and refused), seed %d: %d mismatches" % (programs, skipped, seed, bad))
    return bad


if __name__ == "__main__":
    sys.exit(min(1, run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1)))
