"""The MEX drop-in's call pattern (reference: experiments.m:108, `[R_t_2,R_t_3,Reconst,T,iter] = Method(Corresp,CalM)` on ONE
triplet): every method called with B = 1 on a FRESH context with the library's default options -- no TFF_OPT_* set, so the route is whatever
`capi.hip::rows_for` picks for a batch of one -- against the committed goldens: the linear methods and OptimF against the oracle fixtures
(1e-9 / 1e-8), the five Gauss-Helmert methods against the 50-digit fixtures at 1e-9 with the SAME iteration count
(TFT_methods/ResslTFTPoseEstimation.m:47-105, Optimization/Gauss_Helmert.m:49-82)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from helpers import rel_err_T, rel_err, golden_cases   # noqa: E402
from test_gpu_gh_noise import _dev_conventions          # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fresh_ctx():
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.build import build_library
    build_library()
    return api.Context(0)                                                    # library defaults, nothing set


@pytest.mark.parametrize("method,key,tol", [("LinearTFTPoseEstimation", "tft", 1e-9), ("LinearFPoseEstimation", "f", 1e-9)])
def test_linear_methods_one_triplet_per_call(fresh_ctx, golden_dir, method, key, tol):
    g = np.load(os.path.join(golden_dir, "synthetic_linear.npz"))
    n = 0
    for ci, pre in golden_cases(g):
        C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
        if C.shape[1] < 12 or pre + key + "_T" not in g.files:
            continue
        for b in range(min(C.shape[0], 4)):
            out = fresh_ctx.pose_batch(method, C[b:b + 1], CalM, reconst=True)
            assert out["status"][0] == 0 and out["iter"][0] == 0
            assert rel_err_T(out["T"][0], g[pre + key + "_T"][b]) < tol, (ci, b)
            assert rel_err(out["R_t_2"][0], g[pre + key + "_Rt2"][b]) < tol and rel_err(out["R_t_3"][0], g[pre + key + "_Rt3"][b]) < tol, (ci, b)
            assert rel_err(out["Reconst"][0], g[pre + key + "_Rec"][b]) < tol, (ci, b)
            n += 1
    assert n >= 4


def test_optim_f_one_triplet_per_call(fresh_ctx, golden_dir):
    from test_gpu_parity import _optimf_check
    g = np.load(os.path.join(golden_dir, "optimf.npz"))
    n = flips = 0
    for ci, pre in golden_cases(g):
        C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
        if C.shape[1] < 12:
            continue
        for b in range(min(C.shape[0], 4)):
            out = fresh_ctx.pose_batch("OptimFPoseEstimation", C[b:b + 1], CalM, reconst=True)
            assert out["status"][0] == 0
            flips += _optimf_check(out, 0, g[pre + "optimf_T"][b], g[pre + "optimf_Rt2"][b], g[pre + "optimf_Rt3"][b], g[pre + "optimf_Rec"][b],
                                   g[pre + "optimf_iter"][b], (ci, b)) != 0
            n += 1
    assert n >= 4 and flips <= 1


@pytest.mark.parametrize("method,fixture", [("ResslTFTPoseEstimation", "gh_mp.npz"), ("NordbergTFTPoseEstimation", "gh_mp_nordberg.npz"),
                                            ("FaugPapaTFTPoseEstimation", "gh_mp_faugpapa.npz"), ("PiPoseEstimation", "gh_mp_pi.npz"),
                                            ("PiColPoseEstimation", "gh_mp_picol.npz")])
def test_gauss_helmert_methods_one_triplet_per_call(fresh_ctx, golden_dir, method, fixture):
    g = np.load(os.path.join(golden_dir, fixture))
    n = 0
    for ci, pre in golden_cases(g):
        C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
        for b in range(min(C.shape[0], 6)):
            out = fresh_ctx.pose_batch(method, C[b:b + 1], CalM, reconst=False)
            if int(out["status"][0]) != 0:                                   # PiCol: 'minimal param could not be found' under some convention of the fixture
                assert pre + "mp4_iter" in g.files and (g[pre + "mp4_iter"][b] < 0).any(), (method, ci, b)
                continue
            if pre + "mp4_iter" in g.files and (g[pre + "mp4_iter"][b] < 0).any():
                continue
            d, dit = _dev_conventions(out["T"][0], out["R_t_2"][0], out["R_t_3"][0], out["iter"][0], g, pre, b)
            assert dit == 0 and d < 1e-9, (method, ci, b, dit, d)
            n += 1
    assert n >= 12
