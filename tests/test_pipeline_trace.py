"""this repo's driver (examples/rotation_only_pipeline.py) issues the same module calls, in the same
order, with the same argument kinds as the reference's scripts/sfm_pipeline.py run with onlyRotationAvg=True.  The
reference's trace is the committed fixture tests/golden/pipeline_trace.json (made by tests/golden/make_pipeline_trace.py);
ours is recorded here with the same recorder, in a subprocess so that the stand-in module is the first `GlobalSfMpy`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RUNNER = r"""
import importlib.util, json, os, sys
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
import pipeline_recorder as pr
log = []
stub = pr.make_module(log)
sys.modules["GlobalSfMpy"] = stub
spec = importlib.util.spec_from_file_location("rotation_only_pipeline", os.path.join(root, "examples", "rotation_only_pipeline.py"))
drv = importlib.util.module_from_spec(spec); spec.loader.exec_module(drv)
del log[:]
drv.sfm_pipeline("flags.yaml", "/data/scene", drv.MAGSACWeightBasedLoss(0.02), stub.RotationErrorType.ANGLE_AXIS_COVARIANCE)
print("TRACE " + json.dumps(log))
del log[:]
drv.sfm_pipeline("flags.yaml", "/data/scene", drv.MAGSACWeightBasedLoss(0.02), stub.RotationErrorType.ANGLE_AXIS_COVARIANCE, use1DSfM=False)
print("COLMAP " + json.dumps(log))
"""


@pytest.fixture(scope="module")
def reference_trace(golden_dir):
    with open(os.path.join(golden_dir, "pipeline_trace.json")) as f:
        return json.load(f)


def test_driver_issues_the_reference_call_sequence(reference_trace):
    out = subprocess.run([sys.executable, "-c", _RUNNER, ROOT], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    ours = json.loads(next(l for l in out.stdout.splitlines() if l.startswith("TRACE "))[6:])
    ref = reference_trace["calls"]
    assert [c[0] for c in ours] == [c[0] for c in ref]
    assert ours == ref  # argument kinds too: objects are named after the call that produced them
    colmap = json.loads(next(l for l in out.stdout.splitlines() if l.startswith("COLMAP "))[7:])
    assert colmap == reference_trace["calls_colmap"]  # the use1DSfM=False branch (sfm_pipeline.py:38-47)


def test_compiled_module_answers_every_call_of_the_trace(reference_trace):
    sys.path.insert(0, os.path.join(ROOT, "globalsfmpy_amd"))
    sfm = pytest.importorskip("GlobalSfMpy")
    objs = {"ReconstructionBuilderOptions()": sfm.ReconstructionBuilderOptions()}
    objs["GlobalReconstructionEstimator()"] = sfm.GlobalReconstructionEstimator(objs["ReconstructionBuilderOptions()"].reconstruction_estimator_options)
    objs["ReconstructionBuilder()"] = sfm.ReconstructionBuilder(objs["ReconstructionBuilderOptions()"], sfm.Reconstruction(), sfm.ViewGraph())
    for name, _, _ in reference_trace["calls"]:
        if "()." in name:
            owner, meth = name.split("().")
            assert hasattr(objs[owner + "()"], meth), name
        else:
            assert hasattr(sfm, name), name
    assert hasattr(objs["GlobalReconstructionEstimator()"], "orientations")
    for name, _, _ in reference_trace["calls_colmap"]:
        if "()." not in name:
            assert hasattr(sfm, name), name
    # scripts/get_covariance_from_colmap.py: module functions and builder methods it calls
    for name in reference_trace["get_covariance_from_colmap_calls"]:
        assert hasattr(sfm, name) or hasattr(objs["ReconstructionBuilder()"], name), name
    # the __main__ block (:114-148), including the PLY export (a rotation-only reconstruction has views but no points)
    missing = [n for n in reference_trace["main_module_calls"] if not hasattr(sfm, n)]
    assert missing == []
    d = reference_trace["main_defaults"]
    assert d["rotation_loss"] == "MAGSACWeightBasedLoss(0.02)" and d["rotation_error_type"] == "ANGLE_AXIS_COVARIANCE"
    assert hasattr(sfm.RotationErrorType, d["rotation_error_type"]) and hasattr(sfm.PositionErrorType, d["position_error_type"])


def test_write_ply_file_of_a_rotation_only_reconstruction(tmp_path):
    """Theia io/write_ply_file.cc:74-123: header + one vertex per track and per ESTIMATED view (green, at its position)."""
    sys.path.insert(0, os.path.join(ROOT, "globalsfmpy_amd"))
    sfm = pytest.importorskip("GlobalSfMpy")
    rec = sfm.Reconstruction()
    for v in range(5):
        rec.SetViewName(v, "img%d.jpg" % v)
    o = sfm.MapViewIdVector3d()
    for v in (0, 2, 4):
        o[v] = [0.1 * v, 0.0, 0.0]
    sfm.SetOrientations(o, rec)
    path = tmp_path / "out.ply"
    assert sfm.WritePlyFile(str(path), rec, 2)
    lines = path.read_text().splitlines()
    assert lines[:3] == ["ply", "format ascii 1.0", "element vertex 3"] and lines[9] == "end_header"
    assert lines[10:] == ["0 0 0 0 255 0"] * 3
    assert not sfm.WritePlyFile(str(tmp_path / "no_such_dir" / "x.ply"), rec, 2)


def test_yaml_keys_of_the_reference_main_block_are_the_ones_our_loader_knows(reference_trace, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "globalsfmpy_amd"))
    sfm = pytest.importorskip("GlobalSfMpy")
    assert reference_trace["main_yaml_keys"] == ["1dsfm_dataset_directory", "output_reconstruction", "glog_directory", "v", "log_to_stderr"]
    flags = tmp_path / "flags.yaml"
    flags.write_text("1dsfm_dataset_directory: /data/x\noutput_reconstruction: out\nglog_directory: ./log\nv: 1\nlog_to_stderr: true\n"
                     "num_threads: 16\nmin_num_two_view_inliers: 30\nrotation_filtering_max_difference_degrees: 15.0\n")
    o = sfm.ReconstructionBuilderOptions()
    sfm.load_1DSFM_config(str(flags), o)
    assert o.num_threads == 16
