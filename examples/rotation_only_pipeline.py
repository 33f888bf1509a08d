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


def sfm_pipeline(flagfile, dataset_path, loss_func, rotation_error_type, use1DSfM=True):
    if use1DSfM:
        options = sfm.ReconstructionBuilderOptions()
        if flagfile:
            sfm.load_1DSFM_config(flagfile, options)
        reconstruction = sfm.Reconstruction()
        view_graph = sfm.ViewGraph()
        rot_covariances = sfm.MapEdgesCovariance()
        sfm.Read1DSFM(dataset_path, reconstruction, view_graph, rot_covariances)
        reconstruction_builder = sfm.ReconstructionBuilder(options, reconstruction, view_graph)
    else:  # COLMAP export: two_views.txt + images/ + covariance_rot.txt (sfm_pipeline.py:38-47)
        database = sfm.FeaturesAndMatchesDatabase(dataset_path + "/database")
        options = sfm.ReconstructionBuilderOptions()
        if flagfile:
            sfm.load_1DSFM_config(flagfile, options)
        rot_covariances = sfm.MapEdgesCovariance()
        sfm.ReadCovariance(dataset_path, rot_covariances)
        reconstruction_builder = sfm.ReconstructionBuilder(options, database)
        sfm.AddColmapMatchesToReconstructionBuilder(dataset_path + "/two_views.txt", dataset_path + "/images/*.JPG", reconstruction_builder)
    reconstruction_builder.CheckView()
    view_graph = reconstruction_builder.get_view_graph()
    reconstruction = reconstruction_builder.get_reconstruction()
    estimator = sfm.GlobalReconstructionEstimator(options.reconstruction_estimator_options)
    estimator.FilterInitialViewGraphAndCalibrateCameras(view_graph, reconstruction)
    assert estimator.EstimateGlobalRotationsUncertainty(loss_func, rot_covariances, rotation_error_type), estimator.LastError()
    sfm.SetOrientations(estimator.orientations, reconstruction)
    return reconstruction, estimator


if __name__ == "__main__":
    dataset = sys.argv[1]
    flags = sys.argv[2] if len(sys.argv) > 2 else None
    sfm.InitGlog(0, True, "./log")
    if not os.path.exists(os.path.join(dataset, "covariance_rot.txt")):  # sfm_pipeline.py:136-137
        print("covariance_rot.txt missing: estimating per-edge covariances on the device:", sfm.CalcCovariance(dataset))
    rec, est = sfm_pipeline(flags, dataset, MAGSACWeightBasedLoss(0.02), sfm.RotationErrorType.ANGLE_AXIS_COVARIANCE)  # noqa: F405
    print("estimated %d orientations; solver summary: %s" % (len(rec.EstimatedOrientations()), est.LastSummary()))
    sfm.WriteReconstruction(rec, os.path.join(dataset, "rotations_out.txt"))
    sfm.StopGlog()
