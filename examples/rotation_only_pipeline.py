#!/usr/bin/env python3
"""Counterpart of the reference's scripts/sfm_pipeline.py run with onlyRotationAvg=True (:23-70, :114-148):
the same module calls in the same order with the same argument kinds, on a 1DSfM-style dataset directory
(EGs.txt, cc.txt, covariance_rot.txt) or, with use1DSfM=False, on a COLMAP export (two_views.txt, images/).
usage: rotation_only_pipeline.py <dataset_dir> [flags.yaml]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "globalsfmpy_amd"))  # like sys.path.append('../build')
sys.path.insert(0, os.path.dirname(HERE))

import GlobalSfMpy as sfm  # noqa: E402
from globalsfmpy_amd.loss_functions import *  # noqa: E402,F401,F403


def sfm_pipeline(flagfile, dataset_dir, robust_loss, error_type, use1DSfM=True):
    """Rotation-only run.  Returns (reconstruction with the estimated orientations, the estimator for its summary)."""
    def options():
        o = sfm.ReconstructionBuilderOptions()
        if flagfile:
            sfm.load_1DSFM_config(flagfile, o)
        return o

    if use1DSfM:   # EGs.txt + cc.txt (+ tracks) and covariance_rot.txt in one directory
        opts = options()
        scene, graph, edge_cov = sfm.Reconstruction(), sfm.ViewGraph(), sfm.MapEdgesCovariance()
        sfm.Read1DSFM(dataset_dir, scene, graph, edge_cov)
        builder = sfm.ReconstructionBuilder(opts, scene, graph)
    else:          # COLMAP export: two_views.txt + images/ + covariance_rot.txt (sfm_pipeline.py:38-47)
        store = sfm.FeaturesAndMatchesDatabase(dataset_dir + "/database")
        opts = options()
        edge_cov = sfm.MapEdgesCovariance()
        sfm.ReadCovariance(dataset_dir, edge_cov)
        builder = sfm.ReconstructionBuilder(opts, store)
        sfm.AddColmapMatchesToReconstructionBuilder(dataset_dir + "/two_views.txt", dataset_dir + "/images/*.JPG", builder)
    builder.CheckView()
    graph, scene = builder.get_view_graph(), builder.get_reconstruction()
    solver = sfm.GlobalReconstructionEstimator(opts.reconstruction_estimator_options)
    solver.FilterInitialViewGraphAndCalibrateCameras(graph, scene)
    assert solver.EstimateGlobalRotationsUncertainty(robust_loss, edge_cov, error_type), solver.LastError()
    sfm.SetOrientations(solver.orientations, scene)
    return scene, solver


if __name__ == "__main__":
    dataset = sys.argv[1]
    flags = sys.argv[2] if len(sys.argv) > 2 else None
    sfm.InitGlog(0, True, "./log")
    if not os.path.exists(os.path.join(dataset, "covariance_rot.txt")):  # sfm_pipeline.py:136-137
        print("covariance_rot.txt missing: estimating per-edge covariances on the device:", sfm.CalcCovariance(dataset))
    scene, solver = sfm_pipeline(flags, dataset, MAGSACWeightBasedLoss(0.02), sfm.RotationErrorType.ANGLE_AXIS_COVARIANCE)  # noqa: F405
    print("estimated %d orientations; solver summary: %s" % (len(scene.EstimatedOrientations()), solver.LastSummary()))
    sfm.WriteReconstruction(scene, os.path.join(dataset, "rotations_out.txt"))
    sfm.WritePlyFile(os.path.join(dataset, "rotations_out.ply"), scene, 2)   # sfm_pipeline.py:146
    sfm.StopGlog()
