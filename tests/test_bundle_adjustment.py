"""BundleAdjustment (Optimization/BundleAdjustment.m).  Parity is unpinned (MATLAB + closed-source lsqnonlin):
the oracle (oracle/ba_oracle.py) is pinned by finite differences and by an independent optimiser (MINPACK through scipy) at the
converged optimum; the HIP kernel is compared with the oracle's Levenberg-Marquardt statement and with the goldens."""
import ctypes
import os

import numpy as np
import pytest

from oracle import ba_oracle as BA
from oracle import tft_oracle as O
from tft_vs_fund_amd.scenes import generate_scene_batch, generate_multiview_scene, calm_colmajor
from helpers import rel_err, golden_cases


def _start(C, CalM, b):
    R2, R3, Rec, T, _ = O.LinearTFTPoseEstimation(C[b].T.copy(), CalM)
    return np.vstack([np.eye(3, 4), R2, R3]), Rec


@pytest.mark.parametrize("N,sigma", [(12, 1.0), (40, 2.0)])
def test_oracle_jacobian_and_converged_optimum(N, sigma):
    from scipy.optimize import least_squares
    C, CalM, Rt0, _ = generate_scene_batch(1, N, noise=sigma, seed=5 + N)
    R_t_0, Rec = _start(C, CalM, 0)
    Rt, Recn, it, err, d = BA.BundleAdjustment(CalM, R_t_0, C[0].T.copy(), Rec, True)
    f, J = BA.bundleadjustment_LM(d["x0"], d["Corresp_n"], d["CalM_n"])
    h = 1e-6
    for k in np.random.default_rng(1).integers(0, d["x0"].size, 8):          # analytic Jacobian (BundleAdjustment.m:180-195) vs central differences
        e = np.zeros(d["x0"].size); e[k] = h
        fd = (BA.bundleadjustment_LM(d["x0"] + e, d["Corresp_n"], d["CalM_n"], False) - BA.bundleadjustment_LM(d["x0"] - e, d["Corresp_n"], d["CalM_n"], False)) / (2 * h)
        assert np.abs(fd - J[:, k]).max() < 1e-7
    fun = lambda x: BA.bundleadjustment_LM(x, d["Corresp_n"], d["CalM_n"], False)
    sol = least_squares(fun, d["x0"], jac=lambda x: BA.bundleadjustment_LM(x, d["Corresp_n"], d["CalM_n"])[1], method="lm", xtol=1e-13, ftol=1e-13)
    assert err <= d["cost0"] and abs(err - np.linalg.norm(sol.fun)) < 1e-5 * err      # same optimum as MINPACK, to the FunctionTolerance
    assert 1 <= it <= 20 and abs(np.linalg.norm(Rt[3:6, 3]) - 1) < 1e-12                 # |t2| = 1 (:112)
    r_lin, _ = O.AngError(Rt0[1], R_t_0[6:9]); r_ba, _ = O.AngError(Rt0[1], Rt[6:9])
    assert r_ba < r_lin + 0.05


def test_oracle_noise_free_is_a_fixed_point():
    C, CalM, Rt0, _ = generate_scene_batch(1, 25, noise=0.0, seed=2)
    R_t_0, Rec = _start(C, CalM, 0)
    Rt, Recn, it, err = BA.BundleAdjustment(CalM, R_t_0, C[0].T.copy(), None)
    assert err < 1e-10 and np.abs(Rt - R_t_0).max() < 1e-8


def test_emulated_kernel_matches_oracle(golden_dir):
    from emu import emu_build
    lib = emu_build.load()
    g = np.load(os.path.join(golden_dir, "ba.npz"))
    _p = lambda a: ctypes.c_void_p(a.ctypes.data) if a is not None else None
    for pre, with_rec in (("c0_", True), ("c2_", False), ("c5_", True)):
        C, CalM = g[pre + "Corresp"][:2], g[pre + "CalM"]
        B, N, _ = C.shape
        calm = calm_colmajor(CalM)
        r2 = np.ascontiguousarray(g[pre + "Rt2_in"][:2].transpose(0, 2, 1)).reshape(B, 12)
        r3 = np.ascontiguousarray(g[pre + "Rt3_in"][:2].transpose(0, 2, 1)).reshape(B, 12)
        x0 = np.ascontiguousarray(g[pre + "Rec_in"][:2].transpose(0, 2, 1)) if with_rec else None
        o2 = np.zeros((B, 12)); o3 = np.zeros((B, 12)); rec = np.zeros((B, N, 3)); it = np.zeros(B, dtype=np.int32); err = np.zeros(B); st = np.zeros(B, dtype=np.int32)
        lib.emu_bundle_adjust(_p(calm), ctypes.c_long(0), _p(r2), _p(r3), _p(C), ctypes.c_long(B), ctypes.c_int(N), _p(x0), _p(o2), _p(o3), _p(rec), _p(it), _p(err), _p(st))
        sfx = "" if with_rec else "_tri"
        for b in range(B):
            assert st[b] == 0 and it[b] == int(g[pre + "iter" + sfx][b])
            assert abs(err[b] - g[pre + "err" + sfx][b]) <= 1e-9 * g[pre + "err" + sfx][b] + 1e-12
            assert rel_err(o2[b].reshape(4, 3).T, g[pre + "Rt2" + sfx][b]) < 1e-9 and rel_err(o3[b].reshape(4, 3).T, g[pre + "Rt3" + sfx][b]) < 1e-9
            if with_rec:
                assert rel_err(rec[b].T, g[pre + "Rec"][b]) < 1e-9


def _views_case(M, N, seed, with_x0=False, moved_frame=False, nan_view=None):
    """An M-view problem near its optimum: ground truth scaled to |t2| = 1, poses perturbed by ~0.5 degrees / 1 %, optionally expressed in a frame in
    which camera 1 is not [I|0] (BundleAdjustment.m:80-86) and with one observation missing in view `nan_view`."""
    rng = np.random.default_rng(1000 * M + seed)
    C, CalM, R_t, X = generate_multiview_scene(M, N, noise=1.0, seed=10 * M + seed)
    sc = np.linalg.norm(R_t[3:6, 3]); R_t[:, 3] /= sc; X = X / sc
    R0 = R_t.copy()
    for j in range(1, M):
        w = 0.01 * rng.standard_normal(3); th = np.linalg.norm(w); k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R0[3 * j:3 * j + 3, :3] = (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx) @ R_t[3 * j:3 * j + 3, :3]
        R0[3 * j:3 * j + 3, 3] = R_t[3 * j:3 * j + 3, 3] * (1 + 0.01 * rng.standard_normal(3))
    X0 = X * (1 + 0.01 * rng.standard_normal(X.shape)) if with_x0 else None
    if moved_frame:
        a = 0.3
        G = np.eye(4); G[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]; G[:3, 3] = [0.2, -0.1, 0.4]
        R0 = np.vstack([R0[3 * j:3 * j + 3] @ G for j in range(M)])
        if X0 is not None:
            X0 = (np.linalg.inv(G) @ np.vstack([X0, np.ones(N)]))[:3]
    if nan_view is not None:
        C = C.copy(); C[2 * nan_view + 1, N // 2] = np.nan
    return CalM, R0, C, X0


def _oracle_ba(CalM, R0, C, X0):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                                          # numpy's mean over a view with a NaN (that IS the reference's behaviour)
        return BA.BundleAdjustment(CalM, R0, C, X0)


VIEWS_CASES = [(2, 12, False, False, None), (2, 30, True, True, None), (4, 40, False, True, None), (4, 20, True, False, 2), (5, 70, False, False, 0),
               (6, 25, True, True, None), (6, 16, False, False, 1), (3, 30, False, True, 2)]


def test_oracle_missing_observation_drops_the_whole_view():
    """BundleAdjustment.m:28-29 promises points "not seen in image m"; what the code does with a NaN is decided by Normalize2Ddata.m:34-37 (`mean` over the
    view) and :165: the whole view is skipped.  Pin of that reading: the four-view problem with one NaN in view 3 gives the poses of views 2, 4 and the
    points of the three-view problem (views 1, 2, 4), and camera 3 comes back with its initial pose."""
    CalM, R0, C, _ = _views_case(4, 30, 5, nan_view=2)
    R_t, X, it, err = _oracle_ba(CalM, R0, C, None)
    keep = [0, 1, 3]
    rows3 = np.concatenate([np.arange(3 * j, 3 * j + 3) for j in keep]); rows2 = np.concatenate([np.arange(2 * j, 2 * j + 2) for j in keep])
    R3, X3, it3, err3 = BA.BundleAdjustment(CalM[rows3], R0[rows3], C[rows2], None)
    assert it == it3 and abs(err - err3) < 1e-9 * err3
    assert rel_err(R_t[rows3], R3) < 1e-9 and rel_err(X, X3) < 1e-9
    sc = np.linalg.norm(R_t[6:9, 3]) / np.linalg.norm(R0[6:9, 3])              # (the common scale 1/|t2| of :112 is all that touches its translation)
    assert rel_err(R_t[6:9, :3], R0[6:9, :3]) < 1e-12 and rel_err(R_t[6:9, 3], sc * R0[6:9, 3]) < 1e-12
    with pytest.raises(ValueError):                                             # one complete view left: nothing to triangulate from (:73-74)
        Cn = C[0:4].copy(); Cn[0, 0] = np.nan
        _oracle_ba(CalM[0:6], R0[0:6], Cn, None)


def _run_emulated_views(lib, M, CalM, R0, C, X0):
    _p = lambda a: ctypes.c_void_p(a.ctypes.data) if a is not None else None
    N = C.shape[1]
    calm = np.ascontiguousarray(CalM.T).reshape(-1); rt = np.ascontiguousarray(R0.T).reshape(-1); Cc = np.ascontiguousarray(C.T)
    x0 = np.ascontiguousarray(X0.T) if X0 is not None else None
    out = np.zeros(12 * M); rec = np.zeros((N, 3)); it = np.zeros(1, dtype=np.int32); err = np.zeros(1); st = np.zeros(1, dtype=np.int32)
    assert lib.emu_bundle_adjust_views(M, _p(calm), ctypes.c_long(0), _p(rt), _p(Cc), ctypes.c_long(1), ctypes.c_int(N), _p(x0), _p(out), _p(rec), _p(it), _p(err), _p(st)) == 0
    return out.reshape(4, 3 * M).T, rec.T, int(it[0]), float(err[0]), int(st[0])


@pytest.mark.parametrize("M,N,with_x0,moved,nan_view", VIEWS_CASES[:3] + VIEWS_CASES[5:])                # (the GPU test runs all of them)
def test_emulated_views_kernel_matches_oracle(M, N, with_x0, moved, nan_view):
    """k_bundle_adjust_views<M> (csrc/ba_views_kernel.h) on the lane emulator against BundleAdjustment.m as restated: 2 .. 6 views, with and without
    Reconst0, first camera [I|0] or not, one view with a missing observation."""
    from emu import emu_build
    lib = emu_build.load()
    CalM, R0, C, X0 = _views_case(M, N, 3, with_x0, moved, nan_view)
    Ro, Xo, ito, erro = _oracle_ba(CalM, R0, C, X0)
    Rk, Xk, itk, errk, st = _run_emulated_views(lib, M, CalM, R0, C, X0)
    assert st == 0 and itk == ito and abs(errk - erro) <= 1e-9 * erro
    assert rel_err(Rk, Ro) < 1e-9 and rel_err(Xk, Xo) < 1e-9
    assert np.array_equal(Rk[0:3], np.eye(3, 4))


def test_emulated_views_kernel_too_few_views_and_three_view_twin(golden_dir):
    from emu import emu_build
    lib = emu_build.load()
    CalM, R0, C, X0 = _views_case(2, 12, 4, nan_view=1)                         # two views, one incomplete, nothing to triangulate from: the reference stops (:73-74)
    Rk, Xk, itk, errk, st = _run_emulated_views(lib, 2, CalM, R0, C, None)
    assert st == 1 and np.isnan(Rk).all() and np.isnan(Xk).all() and np.isnan(errk)
    g = np.load(os.path.join(golden_dir, "ba.npz"))                             # the goldens of the three-view kernel through the general one
    pre = "c0_"
    R0 = np.vstack([np.eye(3, 4), g[pre + "Rt2_in"][0], g[pre + "Rt3_in"][0]])
    Rk, Xk, itk, errk, st = _run_emulated_views(lib, 3, g[pre + "CalM"], R0, g[pre + "Corresp"][0].T.copy(), g[pre + "Rec_in"][0])
    assert st == 0 and itk == int(g[pre + "iter"][0]) and abs(errk - g[pre + "err"][0]) <= 1e-9 * g[pre + "err"][0]
    assert rel_err(Rk[3:6], g[pre + "Rt2"][0]) < 1e-9 and rel_err(Rk[6:9], g[pre + "Rt3"][0]) < 1e-9 and rel_err(Xk, g[pre + "Rec"][0]) < 1e-9


@pytest.mark.gpu
def test_gpu_bundle_adjustment_views(gpu_ctx):
    """tff_bundle_adjust_views_batch_dev against the oracle: every case of the emulator test, each as a batch (the case itself + the same problem with
    other noise), and the _host entry point; too few views -> TFF_ST_TOO_FEW + NaN; the reference-shaped wrapper."""
    import torch
    from tft_vs_fund_amd import api
    for M, N, with_x0, moved, nan_view in VIEWS_CASES + [(3, 200, False, False, None), (6, 300, True, False, None)]:
        cases = [_views_case(M, N, sd, with_x0, moved, nan_view) for sd in (3, 8, 9)]
        CalM = cases[0][0]
        R0 = np.stack([c[1] for c in cases]); C = np.stack([np.ascontiguousarray(c[2].T) for c in cases])
        X0 = np.stack([c[3] for c in cases]) if with_x0 else None
        out = gpu_ctx.bundle_adjust_views(CalM, R0, C, X0)
        torch.cuda.synchronize()
        for b, c in enumerate(cases):
            Ro, Xo, ito, erro = _oracle_ba(*c)
            assert int(out["status"][b]) == 0 and int(out["iter"][b]) == ito and abs(float(out["repr_err"][b]) - erro) <= 1e-9 * erro
            assert rel_err(out["R_t"][b].cpu().numpy(), Ro) < 1e-9 and rel_err(out["Reconst"][b].cpu().numpy(), Xo) < 1e-9
    # one calibration per problem (calm_stride = 9 M): the same bits as the shared calibration
    cases = [_views_case(4, 40, sd, False, True, None) for sd in (3, 8, 9)]
    R0 = np.stack([c[1] for c in cases]); C = np.stack([np.ascontiguousarray(c[2].T) for c in cases])
    shared = gpu_ctx.bundle_adjust_views(cases[0][0], R0, C, None)
    per_item = gpu_ctx.bundle_adjust_views(np.stack([cases[0][0]] * 3), R0, C, None)
    for k in ("R_t", "Reconst", "iter", "repr_err", "status"):
        assert torch.equal(shared[k], per_item[k]), k
    # host-pointer entry point, MATLAB layouts straight through
    M, N = 4, 40
    CalM, R0, Cm, _ = _views_case(M, N, 3, False, True, None)
    calm = np.ascontiguousarray(CalM.T).reshape(-1); rt = np.ascontiguousarray(R0.T).reshape(-1); Cc = np.ascontiguousarray(Cm.T)
    o = np.zeros(12 * M); rec = np.zeros((N, 3)); it = np.zeros(1, dtype=np.int32); err = np.zeros(1); st = np.zeros(1, dtype=np.int32)
    _p = lambda a: ctypes.c_void_p(a.ctypes.data)
    assert gpu_ctx.lib.tff_bundle_adjust_views_batch_host(gpu_ctx.handle, M, _p(calm), 0, _p(rt), _p(Cc), 1, N, None, _p(o), _p(rec), _p(it), _p(err), _p(st)) == 0
    Ro, Xo, ito, erro = _oracle_ba(CalM, R0, Cm, None)
    assert st[0] == 0 and it[0] == ito and rel_err(o.reshape(4, 3 * M).T, Ro) < 1e-9 and rel_err(rec.T, Xo) < 1e-9
    assert gpu_ctx.lib.tff_bundle_adjust_views_batch_host(gpu_ctx.handle, 7, _p(calm), 0, _p(rt), _p(Cc), 1, N, None, _p(o), _p(rec), _p(it), _p(err), _p(st)) != 0
    # too few complete views
    CalM2, R02, C2, _ = _views_case(2, 12, 4, nan_view=1)
    out = gpu_ctx.bundle_adjust_views(CalM2, R02[None], np.ascontiguousarray(C2.T)[None], None)
    assert int(out["status"][0]) == 1 and bool(torch.isnan(out["R_t"]).all()) and bool(torch.isnan(out["Reconst"]).all())
    with pytest.raises(ValueError):
        api.BundleAdjustment(CalM2, R02, C2)
    R_t, Rec, it1, err1 = api.BundleAdjustment(CalM, R0, Cm)
    assert it1 == ito and abs(err1 - erro) <= 1e-9 * erro and rel_err(R_t, Ro) < 1e-9


@pytest.mark.gpu
def test_gpu_bundle_adjustment_golden(gpu_ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "ba.npz"))
    e = np.load(os.path.join(golden_dir, "epfl.npz"))
    for ci, pre in golden_cases(g):
        C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
        for sfx, x0 in (("", g[pre + "Rec_in"]), ("_tri", None)):
            out = gpu_ctx.bundle_adjust(CalM, g[pre + "Rt2_in"], g[pre + "Rt3_in"], C, x0)
            st = out["status"].cpu().numpy(); it = out["iter"].cpu().numpy(); err = out["repr_err"].cpu().numpy()
            assert np.all(st == 0) and np.array_equal(it, g[pre + "iter" + sfx])
            assert np.all(np.abs(err - g[pre + "err" + sfx]) <= 1e-9 * g[pre + "err" + sfx] + 1e-12)
            assert rel_err(out["R_t_2"].cpu().numpy(), g[pre + "Rt2" + sfx]) < 1e-9 and rel_err(out["R_t_3"].cpu().numpy(), g[pre + "Rt3" + sfx]) < 1e-9
            if sfx == "":
                assert rel_err(out["Reconst"].cpu().numpy(), g[pre + "Rec"]) < 1e-9
    for n in range(int(e["count"])):
        pre = "t%d_" % n
        Cs = np.ascontiguousarray(e[pre + "sample"][:, :50].T)[None]
        out = gpu_ctx.bundle_adjust(e[pre + "CalM"], e[pre + "tft_Rt2"][None], e[pre + "tft_Rt3"][None], Cs, None)
        assert int(out["status"][0]) == 0 and int(out["iter"][0]) == int(g[pre + "ba_iter"])
        assert abs(float(out["repr_err"][0]) - float(g[pre + "ba_err"])) < 1e-8 * float(g[pre + "ba_err"])
        assert rel_err(out["R_t_3"][0].cpu().numpy(), g[pre + "ba_Rt3"]) < 1e-8


@pytest.mark.gpu
def test_gpu_bundle_adjustment_full_size_and_wrapper(gpu_ctx):
    """A 4000 x 100 batch refined from the linear TFT poses: every triplet converges, the residual norm never rises, the mean pose
    error drops; the reference-shaped single call agrees with the batch."""
    import torch
    from tft_vs_fund_amd import api
    B, N = 4000, 100
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=1.0, seed=808)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    lin = gpu_ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=True)
    out = gpu_ctx.bundle_adjust(calm, lin["R_t_2"].contiguous(), lin["R_t_3"].contiguous(), d, lin["Reconst"].contiguous())
    torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    it = out["iter"].cpu().numpy()
    assert it.min() >= 1 and it.max() <= 30

    def rot_err(Rt):
        R = Rt.cpu().numpy()[:, :, :3]
        c = (np.einsum("ij,bij->b", Rt0[1][:, :3], R) - 1) / 2
        return np.degrees(np.arccos(np.clip(c, -1, 1)))
    assert rot_err(out["R_t_3"]).mean() < 0.8 * rot_err(lin["R_t_3"]).mean()
    assert np.abs(np.linalg.norm(out["R_t_2"].cpu().numpy()[:, :, 3], axis=1) - 1).max() < 1e-12
    R_t_0 = np.vstack([np.eye(3, 4), lin["R_t_2"][0].cpu().numpy(), lin["R_t_3"][0].cpu().numpy()])
    R_t, Rec, it1, err1 = api.BundleAdjustment(CalM, R_t_0, C[0].T.copy(), lin["Reconst"][0].cpu().numpy())
    assert it1 == it[0] and abs(err1 - float(out["repr_err"][0])) < 1e-9 * err1 and rel_err(R_t[6:9], out["R_t_3"][0].cpu().numpy()) < 1e-9   # (the wrapper runs the general kernel)
    # a first camera other than [I|0]: the wrapper changes coordinates as BundleAdjustment.m:80-86 does
    a = 0.3
    G = np.eye(4); G[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]]); G[:3, 3] = [0.2, -0.1, 0.4]
    R_t_g = np.vstack([R_t_0[3 * j:3 * j + 3] @ G for j in range(3)])
    Xg = np.linalg.inv(G) @ np.vstack([lin["Reconst"][0].cpu().numpy(), np.ones(N)])
    R_t2, Rec2, it2, err2 = api.BundleAdjustment(CalM, R_t_g, C[0].T.copy(), Xg[:3])
    assert it2 == it1 and abs(err2 - err1) < 1e-9 * err1 and rel_err(R_t2, R_t) < 1e-8
    with pytest.raises(ValueError):
        api.BundleAdjustment(CalM, R_t_0[0:6], C[0].T.copy())
