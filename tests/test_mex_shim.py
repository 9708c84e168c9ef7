"""matlab/tftfund_mex.c has no MATLAB to build against in the build container or on the GPU box.  These tests put it through a C compiler
all the same, against tests/mex_stub/mex.h -- a stand-in holding only the types and documented prototypes of the MX / MEX calls the gateway
uses (test infrastructure; the real build uses MATLAB's header) -- and, on the GPU box, run its mexFunction end to end through a minimal
implementation of those calls (tests/mex_stub/mex_stub.c) against libtftfund.so."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "mex_stub")
MEX = os.path.join(ROOT, "matlab", "tftfund_mex.c")


def test_mex_gateway_compiles_cleanly():
    """syntax, prototypes and format strings: gcc -Wall -Wextra -Werror in C99 and C11 (the stand-in's mexErrMsgIdAndTxt carries the printf
    format attribute, so every %s / %d of the 23 error messages is checked against its arguments)"""
    for std in ("c99", "c11"):
        subprocess.run(["gcc", "-std=" + std, "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wformat=2", "-I" + STUB, "-I" + os.path.join(ROOT, "include"), MEX],
                       check=True)


def test_mex_gateway_binds_every_host_entry_point_it_names():
    """the eight `tff_<method>_pose_batch_host` symbols, `tff_pose_batch_host_multi` and `tff_bundle_adjust_views_batch_host` it calls are declared in
    include/tftfund.h with the argument lists it uses (the compile above) and exported by the library (tests/test_capi_symbols.py)"""
    src = open(MEX).read()
    hdr = open(os.path.join(ROOT, "include", "tftfund.h")).read()
    import re
    called = set(re.findall(r"\b(tff_[a-z0-9_]+)\s*[;(]", src)) | set(re.findall(r"fn = (tff_[a-z0-9_]+);", src))
    assert len(called) >= 14
    for name in called:
        assert re.search(r"\b%s\s*\(" % name, hdr), name


@pytest.mark.gpu
def test_mex_gateway_runs_end_to_end_through_the_stub(tmp_path):
    """mexFunction('linear_tft' / 'ressl_tft' / 'linear_f', Corresp 6 x N x B, CalM 9 x 3) through the stub implementation of the MX calls:
    outputs equal the C ABI's bit for bit, a too-small sample raises the MATLAB error the reference raises (linearF.m:36)."""
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.build import build_library
    from tft_vs_fund_amd.scenes import generate_scene_batch, calm_colmajor
    so = build_library()
    exe = str(tmp_path / "mex_driver")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + STUB, "-I" + os.path.join(ROOT, "include"), MEX, os.path.join(STUB, "mex_stub.c"),
                    "-o", exe, "-L" + os.path.dirname(so), "-ltftfund", "-Wl,-rpath," + os.path.dirname(so)], check=True)
    ctx = api.Context(0)
    B, N = 5, 40
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=77)
    blob = C.tobytes() + calm_colmajor(CalM).tobytes()
    for mex_name, method in (("linear_tft", "LinearTFTPoseEstimation"), ("ressl_tft", "ResslTFTPoseEstimation"), ("linear_f", "LinearFPoseEstimation")):
        r = subprocess.run([exe, mex_name, str(B), str(N)], input=blob, capture_output=True, check=True)
        got = np.frombuffer(r.stdout, dtype=np.float64)
        assert got.size == B * (12 + 12 + 27 + 1)
        ref = ctx.pose_batch(method, C, CalM, reconst=True)
        Rt2 = got[:12 * B].reshape(B, 4, 3).transpose(0, 2, 1); Rt3 = got[12 * B:24 * B].reshape(B, 4, 3).transpose(0, 2, 1)
        T = got[24 * B:51 * B].reshape(B, 3, 3, 3).transpose(0, 3, 2, 1); it = got[51 * B:]
        assert np.array_equal(Rt2, np.asarray(ref["R_t_2"])) and np.array_equal(Rt3, np.asarray(ref["R_t_3"]))
        assert np.array_equal(T, np.asarray(ref["T"])) and np.array_equal(it, np.asarray(ref["iter"], dtype=np.float64))
    C1 = C[:1, :6].copy()                                                          # six correspondences: linearF needs eight
    r = subprocess.run([exe, "linear_f", "1", "6"], input=C1.tobytes() + calm_colmajor(CalM).tobytes(), capture_output=True)
    assert r.returncode == 3 and b"tftfund:tooFew" in r.stderr


@pytest.mark.gpu
def test_mex_gateway_bundle_adjustment_for_four_views(tmp_path):
    """mexFunction('bundle_adjustment', Corresp 8 x N, CalM 12 x 3, R_t_0 12 x 4) through the stub: MATLAB's arrays go to
    tff_bundle_adjust_views_batch_host as they are; the result equals the oracle's restatement of BundleAdjustment.m (1e-9, same iter); a problem with
    fewer than two complete views raises the MATLAB error where the reference stops (BundleAdjustment.m:73-74)."""
    import warnings
    from oracle import ba_oracle as BA
    from tft_vs_fund_amd.build import build_library
    from tft_vs_fund_amd.scenes import generate_multiview_scene
    so = build_library()
    exe = str(tmp_path / "mex_driver")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + STUB, "-I" + os.path.join(ROOT, "include"), MEX, os.path.join(STUB, "mex_stub.c"),
                    "-o", exe, "-L" + os.path.dirname(so), "-ltftfund", "-Wl,-rpath," + os.path.dirname(so)], check=True)
    M, N = 4, 30
    C, CalM, R_t, X = generate_multiview_scene(M, N, noise=1.0, seed=41)
    sc = np.linalg.norm(R_t[3:6, 3]); R_t[:, 3] /= sc
    R0 = R_t.copy(); R0[3:, 3] *= 1.01
    blob = np.ascontiguousarray(C.T).tobytes() + np.ascontiguousarray(CalM.T).tobytes() + np.ascontiguousarray(R0.T).tobytes()   # column-major, as MATLAB holds them
    r = subprocess.run([exe, "bundle_adjustment", str(M), str(N)], input=blob, capture_output=True, check=True)
    got = np.frombuffer(r.stdout, dtype=np.float64)
    assert got.size == 12 * M + 3 * N + 2
    Rk = got[:12 * M].reshape(4, 3 * M).T; Xk = got[12 * M:12 * M + 3 * N].reshape(N, 3).T; it, err = int(got[-2]), got[-1]
    Ro, Xo, ito, erro = BA.BundleAdjustment(CalM, R0, C, None)
    assert it == ito and abs(err - erro) <= 1e-9 * erro
    assert np.abs(Rk - Ro).max() < 1e-9 * np.abs(Ro).max() and np.abs(Xk - Xo).max() < 1e-9 * np.abs(Xo).max()
    C2 = C[:4].copy(); C2[2, 0] = np.nan                                           # two views, one of them incomplete
    blob = np.ascontiguousarray(C2.T).tobytes() + np.ascontiguousarray(CalM[:6].T).tobytes() + np.ascontiguousarray(R0[:6].T).tobytes()
    r = subprocess.run([exe, "bundle_adjustment", "2", str(N)], input=blob, capture_output=True)
    assert r.returncode == 3 and b"tftfund:views" in r.stderr
