#!/usr/bin/env python3
"""Generates tests/golden/loss_vectors.json and gamma_samples.json.

Runs ONLY in the authoring container (needs /root/reference).  It imports the reference's own
scripts/loss_functions.py against a stub `GlobalSfMpy` module (LossFunction base, tgamma and the
constants/tables parsed from include/gamma_values.cpp) and records, per loss class and parameter
set, (s, rho, rho', rho'') on a grid.  Only these input/output vectors are committed; no reference
source travels.
"""
import json
import math
import os
import re
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_gamma_values():
    txt = open(os.path.join(REF, "include/gamma_values.cpp")).read()
    out = {}
    for nu in (3, 4, 9):
        for name in ("nu", "C", "sigma_quantile", "upper_incomplete_gamma_of_k", "precision_of_stored_gamma"):
            m = re.search(r"constexpr double %s%d\s*=\s*([^;]+);" % (name, nu), txt)
            out["%s%d" % (name, nu)] = float(m.group(1))
        m = re.search(r"constexpr int stored_gamma_number%d\s*=\s*(\d+);" % nu, txt)
        out["stored_gamma_number%d" % nu] = int(m.group(1))
        m = re.search(r"stored_gamma_values%d\s*=\s*\{([^}]*)\}" % nu, txt, re.S)
        vals = [float(v) for v in m.group(1).replace("\n", "").split(",") if v.strip()]
        assert len(vals) == out["stored_gamma_number%d" % nu], (nu, len(vals))
        out["stored_gamma_values%d" % nu] = vals
    return out


def main():
    g = parse_gamma_values()
    stub = types.ModuleType("GlobalSfMpy")

    class LossFunction(object):
        def __init__(self):
            pass
    stub.LossFunction = LossFunction
    stub.tgamma = math.gamma
    for k, v in g.items():
        setattr(stub, k, v)
    sys.modules["GlobalSfMpy"] = stub
    sys.path.insert(0, os.path.join(REF, "scripts"))
    import loss_functions as L  # the reference's file

    def grid(extra=()):
        s = [0.0]
        s += [10.0 ** e for e in range(-12, 3)]
        s += [3.3e-7, 2.5e-5, 7.7e-4, 4.2e-3, 0.0123, 0.37, 1.9, 42.0]
        s += list(extra)
        return sorted(set(s))

    cases = []

    def add(name, args, obj, extra=(), program=None):
        rows = []
        for s in grid(extra):
            out = [0.0, 0.0, 0.0]
            obj.Evaluate(s, out)
            rows.append([s, out[0], out[1], out[2]])
        cases.append({"class": name, "args": list(args), "program": program, "rows": rows})

    def knees(a):
        b = a * a
        return [b, b * (1 - 1e-9), b * (1 + 1e-9), 0.5 * b, 2 * b]

    for a in (0.1, 1.0, 0.02):
        add("TrivialLoss", [], L.TrivialLoss()) if a == 0.1 else None
        add("HuberLoss", [a], L.HuberLoss(a), knees(a))
        add("SoftLOneLoss", [a], L.SoftLOneLoss(a), knees(a))
        add("CauchyLoss", [a], L.CauchyLoss(a), knees(a))
        add("ArctanLoss", [a], L.ArctanLoss(a), knees(a))
        add("TukeyLoss", [a], L.TukeyLoss(a), knees(a))
        add("LOneHalfLoss", [a], L.LOneHalfLoss(a), [0.01, 0.0099, 0.0101])
    for a, b in ((0.5, 0.1), (0.01, 0.002), (1.0, 0.02)):
        add("TolerantLoss", [a, b], L.TolerantLoss(a, b), [a, a + 36.7 * b, a + 36.8 * b, a + 30 * b])
    for a, s2 in ((1.0, 1.0), (0.1, 2.0), (0.05, 0.25)):
        add("LTwoLoss", [a, s2], L.LTwoLoss(a, s2))
        add("GemanMcClureLoss", [a, s2], L.GemanMcClureLoss(a, s2), knees(a))

    def magsac_extra(sigma, nu):
        q = g["sigma_quantile%d" % nu]
        cut = q * q * sigma * sigma
        cell = 2 * sigma * sigma / 1000.0
        ex = [cut, cut * (1 - 1e-12), cut * (1 + 1e-12), 2 * cut]
        for x in (0, 1, 2, 7, 100, 1234, 3000, 5000):
            ex += [x * cell, (x + 0.5) * cell, (x + 0.49999) * cell, (x + 0.50001) * cell]
        return [v for v in ex if v >= 0]

    for sigma in (0.02, 0.1, 1.0):
        for inv in (False, True):
            add("MAGSACWeightBasedLoss", [sigma, inv], L.MAGSACWeightBasedLoss(sigma, inv), magsac_extra(sigma, 3))
            add("MAGSACWeightBasedLoss4", [sigma, inv], L.MAGSACWeightBasedLoss4(sigma, inv), magsac_extra(sigma, 4))
            add("MAGSACWeightBasedLoss9", [sigma, inv], L.MAGSACWeightBasedLoss9(sigma, inv), magsac_extra(sigma, 9))
    # defaults (inverse flag differs per class, reference :286,:345,:403)
    add("MAGSACWeightBasedLoss", [0.02], L.MAGSACWeightBasedLoss(0.02), magsac_extra(0.02, 3))
    add("MAGSACWeightBasedLoss4", [0.02], L.MAGSACWeightBasedLoss4(0.02), magsac_extra(0.02, 4))
    add("MAGSACWeightBasedLoss9", [0.02], L.MAGSACWeightBasedLoss9(0.02), magsac_extra(0.02, 9))

    # combinators: program = nested description rebuilt by the test with the build's own classes
    add("ScaledLoss", [], L.ScaledLoss(L.HuberLoss(0.1), 2.5), knees(0.1),
        program=["Scaled", ["HuberLoss", 0.1], 2.5])
    add("ComposedLoss", [], L.ComposedLoss(L.CauchyLoss(0.3), L.SoftLOneLoss(0.2)), knees(0.2),
        program=["Composed", ["CauchyLoss", 0.3], ["SoftLOneLoss", 0.2]])
    add("ComposedLoss", [], L.ComposedLoss(L.ScaledLoss(L.TukeyLoss(1.5), 0.7), L.ComposedLoss(L.HuberLoss(0.5), L.ArctanLoss(2.0))),
        program=["Composed", ["Scaled", ["TukeyLoss", 1.5], 0.7], ["Composed", ["HuberLoss", 0.5], ["ArctanLoss", 2.0]]])
    add("ScaledLoss", [], L.ScaledLoss(L.MAGSACWeightBasedLoss(0.02), 3.0), magsac_extra(0.02, 3),
        program=["Scaled", ["MAGSACWeightBasedLoss", 0.02], 3.0])

    json.dump({"generator": "tests/golden/make_loss_vectors.py", "source": "reference scripts/loss_functions.py (imported)",
               "cases": cases}, open(os.path.join(HERE, "loss_vectors.json"), "w"))

    samples = {}
    for nu in (3, 4, 9):
        vals = g["stored_gamma_values%d" % nu]
        n = len(vals)
        idx = sorted(set(list(range(0, 40)) + list(range(0, n, 97)) + list(range(n - 40, n))))
        samples[str(nu)] = {
            "n": n,
            "C": g["C%d" % nu], "sigma_quantile": g["sigma_quantile%d" % nu],
            "upper_incomplete_gamma_of_k": g["upper_incomplete_gamma_of_k%d" % nu],
            "precision": g["precision_of_stored_gamma%d" % nu],
            "index": idx, "value": [vals[i] for i in idx],
        }
    json.dump({"generator": "tests/golden/make_loss_vectors.py", "source": "reference include/gamma_values.cpp (sampled)",
               "tables": samples}, open(os.path.join(HERE, "gamma_samples.json"), "w"))
    print("cases:", len(cases), "rows:", sum(len(c["rows"]) for c in cases))


if __name__ == "__main__":
    main()
