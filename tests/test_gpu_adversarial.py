"""Degenerate and non-finite inputs through every pose method, minimal samples (exact kernels) and larger batches: nothing may hang,
and a triplet whose outputs are not finite must say so in its status (the reference's NaN / Inf breaks, Gauss_Helmert.m:53,63;
unassigned R_f, R_t_from_TFT.m:101)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

pytestmark = pytest.mark.gpu

METHODS = ["LinearTFTPoseEstimation", "LinearFPoseEstimation", "OptimFPoseEstimation", "ResslTFTPoseEstimation", "NordbergTFTPoseEstimation",
           "FaugPapaTFTPoseEstimation", "PiPoseEstimation", "PiColPoseEstimation"]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("rows", [1, 0])                                        # four triplets per wavefront / one (TFF_OPT_ROWS; the default picks by batch size)
@pytest.mark.parametrize("N", [7, 8, 9, 12, 64, 200])
def test_degenerate_inputs_never_return_silent_garbage(N, rows):
    import torch
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    ctx = api.Context(0)
    ctx.set_rows(rows)
    C, CalM, _, _ = generate_scene_batch(16, N, noise=1.0, seed=N)
    cases = {}
    c = C.copy(); c[0] = 0.0; cases["all zeros"] = c
    c = C.copy(); c[1] = c[1][:1]; cases["all correspondences identical"] = c
    c = C.copy(); c[2, :, 2:4] = c[2, :, 0:2]; c[2, :, 4:6] = c[2, :, 0:2]; cases["three identical views"] = c
    c = C.copy(); c[3, 0, 0] = np.nan; cases["one NaN"] = c
    c = C.copy(); c[4, 1, 3] = np.inf; cases["one Inf"] = c
    c = C.copy(); t = np.linspace(0, 1, N); c[5, :, 0] = 100 + 500 * t; c[5, :, 1] = 50 + 250 * t; cases["view 1 collinear"] = c
    c = C.copy(); c[6] *= 1e150; cases["huge coordinates"] = c
    c = C.copy(); c[7] *= 1e-150; cases["tiny coordinates"] = c
    c = C.copy(); c[8, : N // 2] = c[8, N // 2: 2 * (N // 2)]; cases["half duplicated"] = c
    for name, c in cases.items():
        for m in METHODS:
            if N < 8 and m in ("LinearFPoseEstimation", "OptimFPoseEstimation"):
                continue
            out = ctx.pose_batch(m, c, CalM, reconst=True)
            torch.cuda.synchronize()
            st = np.asarray(out["status"])
            for k in ("T", "R_t_2", "R_t_3"):
                v = np.asarray(out[k]).reshape(len(st), -1)
                silent = (~np.isfinite(v).all(axis=1)) & (st == 0)
                assert not silent.any(), (N, name, m, k, np.nonzero(silent)[0].tolist())
    # non-finite pose hypotheses through the inlier count
    R2 = np.tile(np.eye(3, 4), (4, 1, 1)); R3 = R2.copy(); R3[1, 0, 3] = np.nan; R3[2] = 0.0
    cnt = ctx.inlier_count(C[0], CalM, R2, R3, 1.0)
    torch.cuda.synchronize()
    assert cnt.shape[0] == 4
