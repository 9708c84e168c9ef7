"""Host-side mirrors of the reference's scalar error metrics (numpy; they are evaluation code,
not part of the GPU hot path).  ReprError itself runs on the GPU: Context.repr_error."""
import numpy as np


def AngError(R_t_true, R_t_est):
    """auxiliar_functions/AngError.m:21-28: rotation angle of R_true' R_est and angle between the translation
    directions, degrees, abs(acos(.)) with NO clamping -- an argument that rounding pushed above 1 gives MATLAB's
    complex acos, whose magnitude is a tiny angle."""
    R_true, t_true = R_t_true[:, 0:3], R_t_true[:, 3]
    R_est, t_est = R_t_est[:, 0:3], R_t_est[:, 3]
    rot_err = float(abs(180 * np.arccos(complex((np.trace(R_true.T @ R_est) - 1) / 2)) / np.pi))
    t_err = float(abs(180 * np.arccos(complex(np.dot(t_est / np.linalg.norm(t_est), t_true / np.linalg.norm(t_true)))) / np.pi))
    return rot_err, t_err


def AngError_batch(R_t_true, R_t_est):
    """AngError for a batch: R_t_true (3,4), R_t_est (B,3,4) -> rot_err (B,), t_err (B,) in degrees."""
    R_t_est = np.asarray(R_t_est, dtype=np.float64)
    c = (np.einsum("ij,bij->b", R_t_true[:, :3], R_t_est[:, :, :3]) - 1) / 2
    t = R_t_est[:, :, 3] / np.linalg.norm(R_t_est[:, :, 3], axis=1, keepdims=True)
    d = t @ (R_t_true[:, 3] / np.linalg.norm(R_t_true[:, 3]))
    return np.abs(180 * np.arccos(c.astype(complex)) / np.pi), np.abs(180 * np.arccos(d.astype(complex)) / np.pi)
