"""MI355X-native robust rotation averaging, drop-in for GlobalSfMpy's
GSfMNonlinearRotationEstimator path."""
from . import _abi  # noqa: F401
from ._abi import (QUATERNION_NORM, ROTATION_MAT_FNORM, QUATERNION_COSINE, ANGLE_AXIS_COVARIANCE, ANGLE_AXIS,  # noqa: F401
                   ANGLE_AXIS_INLIERS, ANGLE_AXIS_COV_INLIERS, ANGLE_AXIS_COVTRACE, ANGLE_AXIS_COVNORM)

__all__ = ["_abi"]
