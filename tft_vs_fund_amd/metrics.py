"""Host-side mirrors of the reference's scalar error metrics (numpy; they are evaluation code,
not part of the GPU hot path).  ReprError itself runs on the GPU: Context.repr_error."""
import numpy as np


def AngError(R_t_true, R_t_est):
    """auxiliar_functions/AngError.m:21-28: rotation angle of R_true' R_est and angle between the
    translation directions, degrees, abs(acos(.)) with NO clamping (NaN when the argument drifts above 1)."""
    R_true, t_true = R_t_true[:, 0:3], R_t_true[:, 3]
    R_est, t_est = R_t_est[:, 0:3], R_t_est[:, 3]
    with np.errstate(invalid="ignore"):
        rot_err = abs(180 * np.arccos((np.trace(R_true.T @ R_est) - 1) / 2) / np.pi)
        t_err = abs(180 * np.arccos(np.dot(t_est / np.linalg.norm(t_est), t_true / np.linalg.norm(t_true))) / np.pi)
    return rot_err, t_err
