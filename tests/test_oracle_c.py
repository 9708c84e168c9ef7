"""The plain-C restatement (oracle/tft_oracle_c.c: explicit 4N x 27 system, one-sided Jacobi SVDs,
four-candidate recover_R_t) against the committed golden vectors of the numpy/LAPACK oracle."""
import os

import numpy as np
import pytest

from oracle import c_oracle
from helpers import rel_err_T, rel_err, golden_cases


def test_c_oracle_matches_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "synthetic_linear.npz"))
    for ci, pre in golden_cases(g):
        C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
        if C.shape[1] > 300:
            C = C[:1]
        out = c_oracle.linear_tft_pose_batch(C, CalM, reconst=True, threads=2)
        assert np.all(out["status"] == 0)
        tol = 1e-9 if C.shape[1] >= 12 else 1e-6
        for b in range(C.shape[0]):
            assert rel_err_T(out["T"][b], g[pre + "tft_T"][b]) < tol
            assert rel_err(out["R_t_2"][b], g[pre + "tft_Rt2"][b]) < tol
            assert rel_err(out["R_t_3"][b], g[pre + "tft_Rt3"][b]) < tol
            assert rel_err(out["Reconst"][b], g[pre + "tft_Rec"][b]) < tol


def test_c_oracle_too_few_points():
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, _, _ = generate_scene_batch(1, 6, noise=1.0, seed=1)
    assert c_oracle.linear_tft_pose_batch(C, CalM)["status"][0] == 1


@pytest.mark.parametrize("N,sigma,seed", [(8, 1.0, 11), (9, 2.0, 12), (12, 1.0, 13), (60, 0.5, 14), (200, 1.0, 15)])
def test_c_oracle_linear_f_matches_numpy_oracle(N, sigma, seed):
    """Second, LAPACK-free pin of the numpy oracle for LinearFPoseEstimation (one-sided Jacobi SVDs, explicit N x 9 system,
    all four candidates): independent arithmetic, same reference algorithm.  A cheirality-vote tie is resolved by the unspecified
    signs of svd(E): such a triplet is compared with the best of the sign conventions."""
    from oracle import tft_oracle as O
    from tft_vs_fund_amd.scenes import generate_scene_batch
    from helpers import pose_err, pose_err_any_convention
    B = 6
    C, CalM, _, _ = generate_scene_batch(B, N, noise=sigma, seed=seed)
    out = c_oracle.linear_f_pose_batch(C, CalM, reconst=True, threads=2)
    assert np.all(out["status"] == 0)
    tol = 1e-9 if N >= 12 else 1e-6
    for b in range(B):
        ob = {k: out[k][b] for k in ("T", "R_t_2", "R_t_3")}
        Cb = C[b].T.copy()
        ref = O.LinearFPoseEstimation(Cb, CalM)
        e = pose_err(ob, ref)
        if e >= tol:
            e = pose_err_any_convention(ob, O.LinearFPoseEstimation, Cb, CalM)[1]
        assert e < tol, (N, b, e)
        if pose_err(ob, ref) < tol:
            assert rel_err(out["Reconst"][b], ref[2]) < max(tol, 1e-8)
