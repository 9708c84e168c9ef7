"""
Nordberg's diverging trials on real data (results/real_fountain_trials.json: mean ReprError 155 px, median 2.7) are the REFERENCE's
iteration, not a kernel defect.

tests/golden/nordberg_divergent.npz (tools/nordberg_divergence_extract.py on the MI355X, then tools/nordberg_divergence_check.py on the CPU):
of 7 000 (triplet, trial) problems of the fountain-P11 list (70 triplets x 100 noise trials, sigma = 0.5 px, 100-inlier samples) 62 end
with a Nordberg ReprError above 50 px, in 7 triplets; the worst trial of each is stored with
  * the 50-digit evaluation of NordbergTFTPoseEstimation.m:47-222 on Gauss_Helmert.m:38-83 (oracle/gh_mp_oracle.py) under all eight sign
    conventions of linearTFT's singular vectors (NordbergTFTPoseEstimation.m:73-78 builds its rotations from them): ReprError > 50 px
    in 8 of 8 conventions for every trial, exit "objective rose" at iteration 2 (3 once) -- the first Gauss-Helmert step lands far
    off (the axis-angle parameters divide by the rotation angle, :131,:185) and Gauss_Helmert.m:75-80 keeps it;
  * the numpy/LAPACK oracle under the same conventions: the same picture;
  * the HIP kernel's result: within 3e-8 ... 2e-6 of the nearest convention, same iteration count; Ressl on the same inputs: 1 - 14 px.
"""
import os

import numpy as np
import pytest

from helpers import rel_err_T, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "nordberg_divergent.npz")


def _repr_all(O, CalM, R2, R3, inl):
    Ps = [CalM[0:3] @ np.eye(3, 4), CalM[3:6] @ R2, CalM[6:9] @ R3]
    return float(O.ReprError(Ps, inl.T.copy()))


def test_fixture_says_every_convention_of_the_exact_iteration_diverges():
    g = np.load(FIX)
    assert int(g["n_divergent"]) == 62 and int(g["n_total"]) == 7000 and g["Corresp"].shape == (7, 100, 6)
    assert np.all(g["mp_repr"] > 50.0) and np.all(g["np_repr"] > 50.0)       # 50-digit iteration and LAPACK oracle, [trial, convention]
    assert np.all(g["mp_reason"] == "rose") and np.all(g["mp_iter"] <= 3) and np.array_equal(g["mp_iter"], g["np_iter"])
    assert np.all(g["gpu_nord_repr"] > 50.0) and np.all(g["gpu_ressl_repr"] < 20.0) and np.all(g["gpu_nord_status"] == 0)


def test_lapack_oracle_diverges_on_the_stored_trials():
    """The restatement of the reference in numpy/LAPACK arithmetic (stand-in for MATLAB's), default sign convention, run here."""
    from oracle import tft_oracle as O
    g = np.load(FIX)
    off = g["inlier_offsets"]
    for b in range(g["Corresp"].shape[0]):
        R2, R3, _, T, it, dbg = O.NordbergTFTPoseEstimation(g["Corresp"][b].T.copy(), g["CalM"][b], True)
        e = _repr_all(O, g["CalM"][b], R2, R3, g["inliers"][off[b]:off[b + 1]])
        assert e > 50.0 and dbg["reason"] == "rose" and it == int(g["np_iter"][b, 0]), (b, e, it, dbg["reason"])
        assert abs(e - g["np_repr"][b, 0]) < 1e-3 * e


@pytest.mark.gpu
def test_kernel_reproduces_the_diverging_iteration(gpu_ctx):
    from oracle import tft_oracle as O
    g = np.load(FIX)
    off = g["inlier_offsets"]
    C = np.ascontiguousarray(g["Corresp"]); CalM = np.ascontiguousarray(g["CalM"])
    out = gpu_ctx.pose_batch("NordbergTFTPoseEstimation", C, CalM, reconst=False)
    res = gpu_ctx.pose_batch("ResslTFTPoseEstimation", C, CalM, reconst=False)
    assert np.all(out["status"] == 0) and np.all(res["status"] == 0)
    for b in range(C.shape[0]):
        devs = [max(rel_err_T(out["T"][b], g["mp_T"][b, c]), rel_err(out["R_t_2"][b], g["mp_Rt2"][b, c]), rel_err(out["R_t_3"][b], g["mp_Rt3"][b, c]))
                for c in range(8)]
        c0 = int(np.argmin(devs))
        # (the diverged step amplifies the start's rounding: 1e-9 on the converging scenes of tests/test_gpu_gh_noise.py, 1e-5 here)
        assert devs[c0] < 1e-5 and int(out["iter"][b]) == int(g["mp_iter"][b, c0]), (b, devs[c0], out["iter"][b], g["mp_iter"][b])
        inl = g["inliers"][off[b]:off[b + 1]]
        assert _repr_all(O, CalM[b], out["R_t_2"][b], out["R_t_3"][b], inl) > 50.0
        assert _repr_all(O, CalM[b], res["R_t_2"][b], res["R_t_3"][b], inl) < 20.0      # the input is fine: Ressl's parameterisation converges on it
