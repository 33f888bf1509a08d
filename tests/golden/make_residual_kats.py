#!/usr/bin/env python3
"""Generates tests/golden/residual_kats.json: the four known-answer cases of Theia's
pairwise_rotation_error_test.cc:87-139 (SmallRotation, NontrivialRotation,
OneHundredEightyDegreeRotation x2, Weight), with the expected residual recomputed by an
independent tool (scipy.spatial.transform.Rotation), exactly as the Theia test computes its ground
truth: gt = weight * log(R2 R1^T Rrel^T).  Tolerance of the Theia test: 1e-12 (:79)."""
import json
import os

import numpy as np
from scipy.spatial.transform import Rotation as R

HERE = os.path.dirname(os.path.abspath(__file__))


def euler_xyz(ax, ay, az):
    # Eigen: AngleAxis(x, UnitX) * AngleAxis(y, UnitY) * AngleAxis(z, UnitZ)
    return R.from_rotvec([np.deg2rad(ax), 0, 0]) * R.from_rotvec([0, np.deg2rad(ay), 0]) * R.from_rotvec([0, 0, np.deg2rad(az)])


def case(name, rel, weight, r1, r2):
    err = r2 * r1.inv() * rel.inv()
    return {"name": name, "rotation1": r1.as_rotvec().tolist(), "rotation2": r2.as_rotvec().tolist(),
            "relative_rotation": rel.as_rotvec().tolist(), "weight": weight,
            "expected": (weight * err.as_rotvec()).tolist()}


def main():
    I = R.identity()
    z = lambda d: R.from_rotvec([0, 0, np.deg2rad(d)])  # noqa: E731
    cases = [
        case("SmallRotation", z(1.0), 1.0, I, z(2.0)),
        case("NontrivialRotation", euler_xyz(5.9, 1.8, 7.6), 1.0, I, euler_xyz(5.3, 1.2, 8.1)),
        case("OneHundredEightyDegreeRotation", z(-179.0), 1.0, I, z(179.0)),
        case("OneHundredEightyDegreeRotationSwapped", z(-179.0), 1.0, z(179.0), I),
        case("Weight", euler_xyz(5.9, 1.8, 7.6), 2.0, I, euler_xyz(5.3, 1.2, 8.1)),
    ]
    json.dump({"generator": "tests/golden/make_residual_kats.py", "source": "Theia pairwise_rotation_error_test.cc:87-139 inputs; expected via scipy Rotation",
               "tolerance": 1e-12, "cases": cases}, open(os.path.join(HERE, "residual_kats.json"), "w"), indent=1)
    for c in cases:
        print(c["name"], c["expected"])


if __name__ == "__main__":
    main()
