#!/usr/bin/env python3
"""Generates tests/golden/pipeline_trace.json.

Runs ONLY in the authoring container (needs /root/reference).  Imports the reference's scripts/sfm_pipeline.py against a
recording stand-in of the compiled module (tests/pipeline_recorder.py) and calls
`sfm_with_1dsfm_dataset(..., onlyRotationAvg=True)`; the fixture is the ordered list of module calls it issued
(callee name, kinds of the positional arguments) plus the YAML keys its __main__ block reads.  No reference source travels."""
import json
import os
import re
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import pipeline_recorder as pr  # noqa: E402


def main():
    log = []
    stub = pr.make_module(log)
    # the gamma constants loss_functions.py reads at import time: any numbers do, the trace does not depend on them
    for nu in (3, 4, 9):
        for name in ("nu", "C", "sigma_quantile", "upper_incomplete_gamma_of_k", "precision_of_stored_gamma"):
            setattr(stub, "%s%d" % (name, nu), float(nu) if name == "nu" else 1.0)
        setattr(stub, "stored_gamma_number%d" % nu, 2)
        setattr(stub, "stored_gamma_values%d" % nu, [1.0, 0.5])
    sys.modules["GlobalSfMpy"] = stub
    sys.path.insert(0, os.path.join(REF, "scripts"))
    import sfm_pipeline as ref  # noqa: E402
    del log[:]  # default-argument evaluation at import is not part of the run
    loss = ref.MAGSACWeightBasedLoss(0.02)
    ref.sfm_with_1dsfm_dataset("flags.yaml", "/data/scene", loss, ref.HuberLoss(0.1),
                               stub.RotationErrorType.ANGLE_AXIS_COVARIANCE, stub.PositionErrorType.BASELINE, onlyRotationAvg=True)
    calls_1dsfm = list(log)
    del log[:]
    ref.sfm_pipeline("flags.yaml", "/data/scene", loss, ref.HuberLoss(0.1), stub.RotationErrorType.ANGLE_AXIS_COVARIANCE,
                     stub.PositionErrorType.BASELINE, onlyRotationAvg=True, use1DSfM=False)
    calls_colmap = list(log)
    # scripts/get_covariance_from_colmap.py is a flat script (argparse + exit()): its module calls, in textual order
    cov_src = open(os.path.join(REF, "scripts", "get_covariance_from_colmap.py")).read()
    cov_calls = re.findall(r"(?:sfm|reconstruction_builder)\.(\w+)\(", cov_src)
    src = open(os.path.join(REF, "scripts", "sfm_pipeline.py")).read()
    main_block = src[src.index("if __name__ == '__main__':"):]
    yaml_keys = re.findall(r"config\['([^']+)'\]", main_block)
    main_calls = re.findall(r"sfm\.(\w+)\(", main_block)
    out = {"source": "scripts/sfm_pipeline.py: sfm_with_1dsfm_dataset(flagfile, path, MAGSACWeightBasedLoss(0.02), HuberLoss(0.1), "
                     "ANGLE_AXIS_COVARIANCE, BASELINE, onlyRotationAvg=True)",
           "calls": calls_1dsfm, "calls_colmap": calls_colmap, "get_covariance_from_colmap_calls": cov_calls, "main_yaml_keys": yaml_keys, "main_module_calls": main_calls,
           "main_defaults": {"rotation_loss": "MAGSACWeightBasedLoss(0.02)", "position_loss": "HuberLoss(0.1)",
                             "rotation_error_type": "ANGLE_AXIS_COVARIANCE", "position_error_type": "BASELINE"}}
    with open(os.path.join(HERE, "pipeline_trace.json"), "w") as f:
        json.dump(out, f, indent=1)
    for c in calls_colmap:
        print(c)
    print(cov_calls)
    print(yaml_keys, main_calls)


if __name__ == "__main__":
    main()
