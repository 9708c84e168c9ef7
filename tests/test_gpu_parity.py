"""
Parity gate on the MI355X: the HIP path, called through the C ABI, against
(1) the committed golden fixtures (oracle outputs on synthetic and EPFL inputs),
(2) the oracle on fresh seeded inputs at sizes it finishes in seconds, and
(3) size-independent properties at BASELINE.json's full size (B=10 000, N=200).

Tolerance: BASELINE.json's north_star asks 1e-6 relative on T / F entries and
recovered R, t.  The linear paths are fp64 end to end and agree to ~1e-11, so the
tests hold them to 1e-9 (T up to its free global sign).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from helpers import rel_err_T, rel_err, golden_cases, votes_match   # noqa: E402

TOL = 1e-9
TOL_MINIMAL = 1e-6   # N < 12: minimal-sample geometry is ill-conditioned; the north_star bound applies


def _oracle():
    from oracle import tft_oracle as O
    return O


def test_native_library_is_loaded(gpu_ctx):
    assert gpu_ctx.lib.tff_version() >= 100
    maps = open("/proc/self/maps").read()
    assert "libtftfund.so" in maps


@pytest.mark.parametrize("solver", ["invit", "jacobi"])
def test_linear_tft_golden_synthetic(gpu_ctx, golden_dir, solver):
    g = np.load(os.path.join(golden_dir, "synthetic_linear.npz"))
    gpu_ctx.set_solver(solver)
    try:
        for ci, pre in golden_cases(g):
            C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
            out = gpu_ctx.pose_batch("LinearTFTPoseEstimation", C, CalM, reconst=True)
            assert np.all(out["status"] == 0) and np.all(out["iter"] == 0)
            tol = TOL if C.shape[1] >= 12 else TOL_MINIMAL
            for b in range(C.shape[0]):
                assert rel_err_T(out["T"][b], g[pre + "tft_T"][b]) < tol, (ci, b)
                assert rel_err(out["R_t_2"][b], g[pre + "tft_Rt2"][b]) < tol, (ci, b)
                assert rel_err(out["R_t_3"][b], g[pre + "tft_Rt3"][b]) < tol, (ci, b)
                assert rel_err(out["Reconst"][b], g[pre + "tft_Rec"][b]) < tol, (ci, b)
    finally:
        gpu_ctx.set_solver("invit")


def test_linear_tft_golden_intermediates(gpu_ctx, golden_dir):
    import torch
    g = np.load(os.path.join(golden_dir, "synthetic_linear.npz"))
    for ci, pre in golden_cases(g):
        C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
        N = C.shape[1]
        out = gpu_ctx.pose_batch("LinearTFTPoseEstimation", torch.from_numpy(C).cuda(), torch.from_numpy(CalM).cuda(),
                                 reconst=False, debug=True)
        dbg = out["debug"].cpu().numpy()
        for b in range(C.shape[0]):
            Tlin = dbg[b, 33:60].reshape(3, 3, 3, order="F")           # constrained linearTFT tensor
            assert rel_err_T(Tlin, g[pre + "dbg_lin_T"][b]) < (TOL if N >= 12 else TOL_MINIMAL)
            # votes: ALL FOUR candidate scores equal the reference's (homogeneous DLT point per correspondence and candidate,
            # R_t_from_TFT.m:96-101).  Their order depends on the signs svd(E) gives U(:,3) and V(:,3), which MATLAB leaves
            # unspecified: one of the four possible orders must match entry by entry.
            for k, key in ((60, "dbg_votes2"), (64, "dbg_votes3")):
                assert votes_match(dbg[b, k:k + 4], g[pre + key][b]), (ci, b, dbg[b, k:k + 4], g[pre + key][b])
            assert abs(dbg[b, 68] - g[pre + "dbg_lam"][b]) < (TOL if N >= 12 else TOL_MINIMAL) * abs(g[pre + "dbg_lam"][b])
            assert int(g[pre + "dbg_rankE"][b]) == 15


def test_linear_tft_golden_epfl(gpu_ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "epfl.npz"))
    for n in range(int(g["count"])):
        pre = "t%d_" % n
        Cs = g[pre + "sample"]
        out = gpu_ctx.pose_batch("LinearTFTPoseEstimation", np.ascontiguousarray(Cs.T)[None], g[pre + "CalM"], reconst=True)
        assert out["status"][0] == 0
        assert rel_err_T(out["T"][0], g[pre + "tft_T"]) < TOL
        assert rel_err(out["R_t_2"][0], g[pre + "tft_Rt2"]) < TOL
        assert rel_err(out["R_t_3"][0], g[pre + "tft_Rt3"]) < TOL
        assert rel_err(out["Reconst"][0], g[pre + "tft_Rec"]) < 1e-8


@pytest.mark.parametrize("N,sigma,seed", [(7, 1.0, 1), (9, 2.0, 2), (33, 0.5, 3), (64, 1.0, 4), (65, 1.0, 5),
                                          (200, 1.0, 6), (257, 0.0, 7), (1000, 1.0, 8), (1500, 1.0, 9)])
def test_linear_tft_vs_oracle_seeded(gpu_ctx, N, sigma, seed):
    from tft_vs_fund_amd.scenes import generate_scene_batch
    O = _oracle()
    B = 5
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=sigma, seed=seed)
    out = gpu_ctx.pose_batch("LinearTFTPoseEstimation", C, CalM, reconst=True)
    assert np.all(out["status"] == 0)
    tol = TOL if N >= 12 else TOL_MINIMAL
    for b in range(B):
        R2, R3, Rec, T, _ = O.LinearTFTPoseEstimation(C[b].T.copy(), CalM)
        assert rel_err_T(out["T"][b], T) < tol
        assert rel_err(out["R_t_2"][b], R2) < tol and rel_err(out["R_t_3"][b], R3) < tol
        assert rel_err(out["Reconst"][b], Rec) < tol


@pytest.mark.parametrize("method", ["LinearTFTPoseEstimation", "LinearFPoseEstimation", "OptimFPoseEstimation"])
@pytest.mark.parametrize("N", [12, 60, 200])
def test_collinear_camera_centres_take_the_exact_tiers(gpu_ctx, method, N):
    """experiments.m option 'angle' up to 180 degrees: collinear camera centres.  The slices of the (calibrated) tensor are then nearly of rank
    one, the 3 x 3 null vectors of R_t_from_TFT.m:47-55 are determined to eps sigma_1 / (sigma_2 - sigma_3) by the reference's svd but only to
    the SQUARE of that by the fast tier's formed Gram matrix (R_t_3 4.7e-7 off at N = 200 before small_la.h::null3 learnt to report it):
    the fast path has to hand such triplets to the exact kernel.  Required: the default route equals the exact kernel (TFF_OPT_SOLVER = 1)
    to 1e-10, and both sit at the oracle within the conditioning of the configuration (measured 2.6e-9; gate 2e-8)."""
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    O = _oracle()
    B = 24
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=300 + N, angle=180)
    out = gpu_ctx.pose_batch(method, C, CalM, reconst=False)
    ectx = api.Context(0, solver="jacobi")
    ectx.set_rows(False)                                    # the one-triplet exact kernel for all: the kernel that redoes what the default route flags
    exact = ectx.pose_batch(method, C, CalM, reconst=False)
    assert np.all(out["status"] == 0) and np.all(exact["status"] == 0)
    tol = 2e-8 if method != "OptimFPoseEstimation" else 2e-7
    if method == "LinearTFTPoseEstimation":                 # the exact tiers with four triplets per wavefront (TFF_OPT_SOLVER = 1, rows on): a second
        ectx.set_rows(True)                                 # implementation of the same tiers, equal within the conditioning of the configuration
        rex = ectx.pose_batch(method, C, CalM, reconst=False)
        assert np.all(rex["status"] == 0)
        for b in range(B):
            assert rel_err_T(rex["T"][b], exact["T"][b]) < tol and rel_err(rex["R_t_2"][b], exact["R_t_2"][b]) < tol and rel_err(rex["R_t_3"][b], exact["R_t_3"][b]) < tol, b
    for b in range(B):
        assert rel_err_T(out["T"][b], exact["T"][b]) < 1e-10 and rel_err(out["R_t_2"][b], exact["R_t_2"][b]) < 1e-10
        assert rel_err(out["R_t_3"][b], exact["R_t_3"][b]) < 1e-10, (b, rel_err(out["R_t_3"][b], exact["R_t_3"][b]))
        ref = getattr(O, method)(C[b].T.copy(), CalM)
        assert rel_err_T(out["T"][b], ref[3]) < tol and rel_err(out["R_t_2"][b], ref[0]) < tol and rel_err(out["R_t_3"][b], ref[1]) < tol, b


def test_per_triplet_calibration_and_drop_in_wrapper(gpu_ctx):
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    O = _oracle()
    Ca, CalA, _, _ = generate_scene_batch(2, 50, noise=1.0, seed=21, focalL=50.0)
    Cb, CalB, _, _ = generate_scene_batch(2, 50, noise=1.0, seed=22, focalL=80.0)
    C = np.concatenate([Ca, Cb]); Cal = np.stack([CalA, CalA, CalB, CalB])
    out = gpu_ctx.pose_batch("LinearTFTPoseEstimation", C, Cal, reconst=False)
    for b in range(4):
        R2, R3, _, T, _ = O.LinearTFTPoseEstimation(C[b].T.copy(), Cal[b])
        assert rel_err(out["R_t_2"][b], R2) < TOL and rel_err(out["R_t_3"][b], R3) < TOL and rel_err_T(out["T"][b], T) < TOL
    # reference-shaped single call: Corresp 6xN, CalM 9x3 -> (R_t_2, R_t_3, Reconst, T, iter)
    R2, R3, Rec, T, it = api.LinearTFTPoseEstimation(C[0].T.copy(), Cal[0])
    o2, o3, orec, oT, oit = O.LinearTFTPoseEstimation(C[0].T.copy(), Cal[0])
    assert R2.shape == (3, 4) and Rec.shape == (3, 50) and T.shape == (3, 3, 3) and it == oit == 0
    assert rel_err(R2, o2) < TOL and rel_err(Rec, orec) < TOL and rel_err_T(T, oT) < TOL
    with pytest.raises(ValueError):
        api.LinearTFTPoseEstimation(C[0].T[:, :6].copy(), Cal[0])          # N < 7: experiments.m:99


def test_edge_cases(gpu_ctx):
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, _, _ = generate_scene_batch(3, 6, noise=1.0, seed=1)
    out = gpu_ctx.pose_batch("LinearTFTPoseEstimation", C, CalM, reconst=True)
    assert np.all(out["status"] == 1) and np.all(np.isnan(out["T"]))
    out = gpu_ctx.pose_batch("LinearTFTPoseEstimation", np.zeros((0, 10, 6)), CalM, reconst=True)     # empty batch
    assert out["T"].shape == (0, 3, 3, 3)
    C, CalM, _, _ = generate_scene_batch(2, 20, noise=1.0, seed=2)
    C[1, 3, 2] = np.nan
    out = gpu_ctx.pose_batch("LinearTFTPoseEstimation", C, CalM, reconst=False)
    assert out["status"][0] == 0 and out["status"][1] in (2, 3)


@pytest.mark.parametrize("method", ["LinearFPoseEstimation", "OptimFPoseEstimation", "ResslTFTPoseEstimation", "NordbergTFTPoseEstimation",
                                    "FaugPapaTFTPoseEstimation", "PiPoseEstimation", "PiColPoseEstimation"])
def test_bad_inputs_are_reported_per_triplet(gpu_ctx, method):
    """NaN coordinates, a degenerate triplet (every correspondence identical) and too few points: reported through `status` for that
    triplet only (MATLAB would throw or return NaN), the other triplets of the batch are unaffected, nothing hangs."""
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, _, _ = generate_scene_batch(4, 24, noise=1.0, seed=21)
    ref = gpu_ctx.pose_batch(method, C, CalM, reconst=True)
    Cb = C.copy()
    Cb[1, 5, 3] = np.nan
    Cb[2, :, :] = Cb[2, 0:1, :]
    out = gpu_ctx.pose_batch(method, Cb, CalM, reconst=True)
    assert out["status"][1] != 0 and out["status"][2] != 0
    for b in (0, 3):
        assert out["status"][b] == ref["status"][b]
        if ref["status"][b] == 0:
            assert np.array_equal(out["R_t_3"][b], ref["R_t_3"][b]) and np.array_equal(out["T"][b], ref["T"][b])
    few = gpu_ctx.pose_batch(method, C[:, :6], CalM, reconst=True)
    assert np.all(few["status"] == 1) and np.all(np.isnan(few["T"]))


def test_bundle_adjust_bad_inputs(gpu_ctx):
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, _, _ = generate_scene_batch(3, 20, noise=1.0, seed=22)
    lin = gpu_ctx.pose_batch("LinearTFTPoseEstimation", C, CalM, reconst=True)
    Cb = C.copy(); Cb[1, 2, 0] = np.nan
    out = gpu_ctx.bundle_adjust(CalM, lin["R_t_2"], lin["R_t_3"], Cb, lin["Reconst"])
    st = out["status"].cpu().numpy()
    assert st[1] != 0 and st[0] == 0 and st[2] == 0


def test_full_size_properties(gpu_ctx):
    """B=10 000, N=200 (BASELINE.json configs[1]) through the device-pointer ABI."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B, N = 10000, 200
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=1.0, seed=1234)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    out = gpu_ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=True)
    torch.cuda.synchronize()
    st = out["status"].cpu().numpy()
    T = out["T"].cpu().numpy(); R2 = out["R_t_2"].cpu().numpy(); R3 = out["R_t_3"].cpu().numpy()
    Rec = out["Reconst"].cpu().numpy()
    assert np.all(st == 0)
    assert np.allclose(np.sqrt((T.reshape(B, -1) ** 2).sum(1)), 1.0, atol=1e-12)          # transform_TFT.m:49
    for R in (R2[:, :, :3], R3[:, :, :3]):
        assert np.abs(np.einsum("bij,bkj->bik", R, R) - np.eye(3)).max() < 1e-9
        assert np.abs(np.linalg.det(R) - 1).max() < 1e-9
    assert np.abs(np.linalg.norm(R2[:, :, 3], axis=1) - 1).max() < 1e-12                  # |t2| = 1
    # accuracy against the scene's ground truth (sigma = 1 px): sub-degree rotations
    cosr = (np.einsum("ij,bij->b", Rt0[0][:, :3], R2[:, :, :3]) - 1) / 2
    assert np.degrees(np.arccos(np.clip(cosr, -1, 1))).max() < 2.0
    # reprojection of Reconst through the recovered cameras: a few pixels RMS at sigma = 1
    K = CalM[0:3]
    X = np.concatenate([Rec, np.ones((B, 1, N))], axis=1)
    err2 = 0
    for Pm, cols in ((np.broadcast_to(K @ np.eye(3, 4), (B, 3, 4)), slice(0, 2)), (K @ R2, slice(2, 4)), (K @ R3, slice(4, 6))):
        x = np.einsum("bij,bjn->bin", Pm, X)
        err2 = err2 + ((x[:, :2] / x[:, 2:3] - C[:, :, cols].transpose(0, 2, 1)) ** 2).sum(1)
    rms = np.sqrt(err2.mean(axis=1) / 3)
    assert rms.max() < 10.0 and np.median(rms) < 3.0
    # batch independence: a triplet computed alone gives bit-identical results
    sub = gpu_ctx.pose_batch("LinearTFTPoseEstimation", d[4321:4322].contiguous(), calm, reconst=True)
    assert torch.equal(sub["T"][0], out["T"][4321]) and torch.equal(sub["R_t_3"][0], out["R_t_3"][4321])
    # host-pointer and device-pointer entry points agree bit for bit
    h = gpu_ctx.pose_batch("LinearTFTPoseEstimation", C[:64], CalM, reconst=True)
    assert np.array_equal(h["T"], T[:64]) and np.array_equal(h["Reconst"], Rec[:64])
    # order of the correspondences is immaterial (up to rounding)
    perm = np.random.default_rng(0).permutation(N)
    p = gpu_ctx.pose_batch("LinearTFTPoseEstimation", np.ascontiguousarray(C[:64][:, perm]), CalM, reconst=True)
    for b in range(64):
        assert rel_err_T(p["T"][b], T[b]) < 1e-9 and rel_err(p["R_t_3"][b], R3[b]) < 1e-9
    assert rel_err(p["Reconst"][:, :, np.argsort(perm)], Rec[:64]) < 1e-8
    # the two eigen-solvers agree
    gpu_ctx.set_solver("jacobi")
    try:
        j = gpu_ctx.pose_batch("LinearTFTPoseEstimation", d[:512].contiguous(), calm, reconst=False)
        jT = j["T"].cpu().numpy()
        assert max(rel_err_T(jT[b], T[b]) for b in range(512)) < 1e-9
    finally:
        gpu_ctx.set_solver("invit")


# ---------------------------------------------------------------------------
# LinearFPoseEstimation (F_methods/LinearFPoseEstimation.m, linearF.m, TFT_from_P.m)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("solver", ["invit", "jacobi"])
def test_linear_f_golden_synthetic(gpu_ctx, golden_dir, solver):
    g = np.load(os.path.join(golden_dir, "synthetic_linear.npz"))
    gpu_ctx.set_solver(solver)
    try:
        for ci, pre in golden_cases(g):
            C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
            out = gpu_ctx.pose_batch("LinearFPoseEstimation", C, CalM, reconst=True)
            if C.shape[1] < 8:
                assert np.all(out["status"] == 1)                       # linearF.m:35-37
                continue
            assert np.all(out["status"] == 0) and np.all(out["iter"] == 0)
            tol = TOL if C.shape[1] >= 12 else TOL_MINIMAL
            for b in range(C.shape[0]):
                assert rel_err_T(out["T"][b], g[pre + "f_T"][b]) < tol, (ci, b)
                assert rel_err(out["R_t_2"][b], g[pre + "f_Rt2"][b]) < tol, (ci, b)
                assert rel_err(out["R_t_3"][b], g[pre + "f_Rt3"][b]) < tol, (ci, b)
                assert rel_err(out["Reconst"][b], g[pre + "f_Rec"][b]) < tol, (ci, b)
    finally:
        gpu_ctx.set_solver("invit")


def test_linear_f_golden_epfl(gpu_ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "epfl.npz"))
    for n in range(int(g["count"])):
        pre = "t%d_" % n
        Cs = g[pre + "sample"]
        out = gpu_ctx.pose_batch("LinearFPoseEstimation", np.ascontiguousarray(Cs.T)[None], g[pre + "CalM"], reconst=True)
        assert out["status"][0] == 0
        assert rel_err_T(out["T"][0], g[pre + "f_T"]) < TOL
        assert rel_err(out["R_t_2"][0], g[pre + "f_Rt2"]) < TOL
        assert rel_err(out["R_t_3"][0], g[pre + "f_Rt3"]) < TOL
        assert rel_err(out["Reconst"][0], g[pre + "f_Rec"]) < 1e-8


@pytest.mark.parametrize("N,sigma,seed", [(8, 1.0, 1), (9, 2.0, 2), (64, 1.0, 4), (65, 1.0, 5), (200, 1.0, 6), (257, 0.0, 7), (1500, 1.0, 9)])
def test_linear_f_vs_oracle_seeded(gpu_ctx, N, sigma, seed):
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    O = _oracle()
    B = 5
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=sigma, seed=seed)
    out = gpu_ctx.pose_batch("LinearFPoseEstimation", C, CalM, reconst=True)
    assert np.all(out["status"] == 0)
    tol = TOL if N >= 12 else TOL_MINIMAL
    for b in range(B):
        R2, R3, Rec, T, _ = O.LinearFPoseEstimation(C[b].T.copy(), CalM)
        assert rel_err_T(out["T"][b], T) < tol
        assert rel_err(out["R_t_2"][b], R2) < tol and rel_err(out["R_t_3"][b], R3) < tol
        assert rel_err(out["Reconst"][b], Rec) < tol
    if N == 9:
        R2, R3, Rec, T, it = api.LinearFPoseEstimation(C[0].T.copy(), CalM)      # reference-shaped call
        o2, _, _, oT, _ = O.LinearFPoseEstimation(C[0].T.copy(), CalM)
        assert rel_err(R2, o2) < tol and rel_err_T(T, oT) < tol and it == 0
        with pytest.raises(ValueError):
            api.LinearFPoseEstimation(C[0].T[:, :7].copy(), CalM)               # linearF.m:35-37


def test_linear_f_full_size_properties(gpu_ctx):
    """configs[2] second half: the same 10k x 200 batch through LinearFPoseEstimation."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B, N = 10000, 200
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=1.0, seed=4321)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    out = gpu_ctx.pose_batch("LinearFPoseEstimation", d, calm, reconst=False)
    torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    T = out["T"].cpu().numpy(); R2 = out["R_t_2"].cpu().numpy(); R3 = out["R_t_3"].cpu().numpy()
    assert np.allclose(np.sqrt((T.reshape(B, -1) ** 2).sum(1)), 1.0, atol=1e-12)          # TFT_from_P.m:33
    for R in (R2[:, :, :3], R3[:, :, :3]):
        assert np.abs(np.einsum("bij,bkj->bik", R, R) - np.eye(3)).max() < 1e-9
        assert np.abs(np.linalg.det(R) - 1).max() < 1e-9
    assert np.abs(np.linalg.norm(R2[:, :, 3], axis=1) - 1).max() < 1e-12
    cosr = (np.einsum("ij,bij->b", Rt0[1][:, :3], R3[:, :, :3]) - 1) / 2
    assert np.degrees(np.arccos(np.clip(cosr, -1, 1))).max() < 3.0
    # T is exactly the trifocal tensor of the returned cameras: its trilinearities vanish on reprojected points
    O = _oracle()
    K = CalM[0:3]
    for b in (0, 17, 9999):
        Tref = O.TFT_from_P(K @ np.eye(3, 4), K @ R2[b], K @ R3[b])
        assert rel_err_T(T[b], Tref) < 1e-9
    h = gpu_ctx.pose_batch("LinearFPoseEstimation", C[:64], CalM, reconst=False)
    assert np.array_equal(h["T"], T[:64])


# ---------------------------------------------------------------------------
# ResslTFTPoseEstimation + Gauss_Helmert (TFT_methods/ResslTFTPoseEstimation.m, Optimization/Gauss_Helmert.m)
#
# The parity gate for Ressl is tests/test_gpu_gh_noise.py: the kernel reproduces a 50-digit evaluation of the reference's
# iteration to 1e-9 with identical iteration counts.  THIS file compares with the LAPACK-backed numpy oracle
# (oracle/tft_oracle.py), whose own fp64 evaluation of pinv(W + 1e-12 I) -- every correspondence gets one direction of weight
# ~1e12, A'WA cancels ten digits -- deviates from that exact iteration by the amounts measured in profiles/r2_gh_noise_mp.txt:
#   same stopping iteration as the exact evaluation: median 7e-7 (N = 200) .. 4e-6 (N = 12);
#   a different stopping iteration (~40 % of the scenes: the exit test "objective rose" compares objectives that agree to
#   ~1e-9 once the iteration stagnates): up to 3e-5 (N = 200), 1e-4 (N = 60), 1e-3 (N = 12).
# The tolerances below are that reference-noise envelope with a factor ~3 of head room (asserted against the fixture in
# test_gpu_gh_noise.py::test_kernel_is_no_noisier_than_the_lapack_evaluation).  Nordberg has its own 50-digit gate there too (and
# one more source of non-uniqueness: the signs of linearTFT's singular vectors, tests/helpers.py::oracle_under_epipole_conventions,
# change its iterates by up to 4e-4 -- inside this envelope); FaugPapa and the Pi methods share the weight blocks and are held to
# the same envelope (no extended-precision restatement of their callbacks exists).
# ---------------------------------------------------------------------------
def _ressl_tol(N, same_iterations=True):
    if same_iterations:
        return 2e-3 if N < 50 else 1e-4
    return 1e-2 if N < 50 else 2e-3


GH_METHODS = [("ResslTFTPoseEstimation", "ressl"), ("NordbergTFTPoseEstimation", "nordberg"), ("FaugPapaTFTPoseEstimation", "faugpapa")]


@pytest.mark.parametrize("method,key", GH_METHODS)
def test_gh_golden_synthetic(gpu_ctx, golden_dir, method, key):
    g = np.load(os.path.join(golden_dir, "synthetic_gh.npz"))
    worst = {}
    for ci, pre in golden_cases(g):
        C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
        N = C.shape[1]
        out = gpu_ctx.pose_batch(method, C, CalM, reconst=True)
        assert np.all(out["status"] == 0)
        for b in range(C.shape[0]):
            dit = int(out["iter"][b]) - int(g[pre + key + "_iter"][b])
            assert abs(dit) <= 5
            e = max(rel_err_T(out["T"][b], g[pre + key + "_T"][b]), rel_err(out["R_t_2"][b], g[pre + key + "_Rt2"][b]),
                    rel_err(out["R_t_3"][b], g[pre + key + "_Rt3"][b]))
            worst[(N, dit)] = max(worst.get((N, dit), 0), e)
            assert e < _ressl_tol(N, dit == 0), (ci, b, dit, e)
            assert rel_err(out["Reconst"][b], g[pre + key + "_Rec"][b]) < 10 * _ressl_tol(N, dit == 0)
    print(method, "worst relative deviation from the dense oracle by (N, iteration difference):", worst)


@pytest.mark.parametrize("method,key", GH_METHODS)
def test_gh_noise_free_is_exact(gpu_ctx, method, key):
    """sigma = 0: the linear solution already satisfies every constraint; Gauss-Helmert stops at once and the
    ground truth is recovered to rounding (the known answer the oracle itself is pinned with)."""
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, Rt0, _ = generate_scene_batch(8, 40, noise=0.0, seed=3)
    out = gpu_ctx.pose_batch(method, C, CalM, reconst=False)
    assert np.all(out["status"] == 0) and np.all(out["iter"] <= 2)
    s = np.linalg.norm(Rt0[0][:, 3])
    for b in range(8):
        assert np.abs(out["R_t_2"][b][:, :3] - Rt0[0][:, :3]).max() < 1e-8
        assert np.abs(out["R_t_3"][b][:, :3] - Rt0[1][:, :3]).max() < 1e-8
        assert np.abs(out["R_t_3"][b][:, 3] - Rt0[1][:, 3] / s).max() < 1e-7


@pytest.mark.parametrize("method,key", GH_METHODS)
def test_gh_vs_oracle_metrics_and_iterations(gpu_ctx, method, key):
    """Downstream metrics (AngError, ReprError: what experiments.m:112-120 records) and the iteration histogram
    against the dense oracle on seeded scenes."""
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    O = _oracle()
    B, N = 12, 60
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=1.0, seed=77)
    out = gpu_ctx.pose_batch(method, C, CalM, reconst=True)
    assert np.all(out["status"] == 0)
    dit = []
    for b in range(B):
        R2, R3, Rec, T, it = getattr(O, method)(C[b].T.copy(), CalM)
        dit.append(int(out["iter"][b]) - it)
        for k, (Rg, Ro) in enumerate(((out["R_t_2"][b], R2), (out["R_t_3"][b], R3))):
            rg, tg = O.AngError(Rt0[k], Rg); ro, to = O.AngError(Rt0[k], Ro)
            assert abs(rg - ro) < 0.02 + 0.02 * ro and abs(tg - to) < 0.02 + 0.02 * to   # degrees; an iteration flip moves the pose by ~1e-4 rad
        P = lambda Ra, Rb: [CalM[0:3] @ np.eye(3, 4), CalM[3:6] @ Ra, CalM[6:9] @ Rb]
        eg = O.ReprError(P(out["R_t_2"][b], out["R_t_3"][b]), C[b].T.copy(), out["Reconst"][b])
        eo = O.ReprError(P(R2, R3), C[b].T.copy(), Rec)
        assert abs(eg - eo) < 1e-2 * eo                                            # late GH steps more or less
    assert max(abs(d) for d in dit) <= 5 and sum(1 for d in dit if d == 0) >= B // 3
    R2, R3, Rec, T, it = getattr(api, method)(C[0].T.copy(), CalM)              # reference-shaped call
    assert it == int(out["iter"][0]) and R2.shape == (3, 4) and T.shape == (3, 3, 3)


@pytest.fixture(scope="module")
def full_size_batch():
    """BASELINE.json configs[1] / configs[2]: 10 000 synthetic triplets of 200 correspondences, sigma = 1 px"""
    from tft_vs_fund_amd.scenes import generate_scene_batch
    return generate_scene_batch(10000, 200, noise=1.0, seed=99)


@pytest.mark.parametrize("method", ["ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "FaugPapaTFTPoseEstimation"])
def test_gh_improves_on_linear_and_full_size(gpu_ctx, method, full_size_batch):
    """configs[2] at its stated size: the 10 000 x 200 batch through the Gauss-Helmert methods; every triplet finishes, the
    refinement must not be worse than its linear initialisation in mean pose error."""
    import torch
    C, CalM, Rt0, _ = full_size_batch
    B = C.shape[0]
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    lin = gpu_ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=False)
    res = gpu_ctx.pose_batch(method, d, calm, reconst=False)
    torch.cuda.synchronize()
    assert int((res["status"] != 0).sum()) == 0
    it = res["iter"].cpu().numpy()
    assert it.min() >= 1 and it.max() <= 30

    def rot_err(Rt):
        R = Rt.cpu().numpy()[:, :, :3]
        c = (np.einsum("ij,bij->b", Rt0[1][:, :3], R) - 1) / 2
        return np.degrees(np.arccos(np.clip(c, -1, 1)))
    assert rot_err(res["R_t_3"]).mean() <= rot_err(lin["R_t_3"]).mean() * 1.02


@pytest.mark.parametrize("method,fixture", [("ResslTFTPoseEstimation", "gh_mp.npz"), ("FaugPapaTFTPoseEstimation", "gh_mp_faugpapa.npz")])
def test_full_size_batch_contains_the_extended_precision_scenes(gpu_ctx, golden_dir, method, fixture, full_size_batch):
    """configs[2]: the N = 200 scenes of the 50-digit fixtures embedded at scattered positions of the 10 000 x 200 batch come out bit-identical
    to their single launches (no cross-triplet state, whatever workgroup or spill slice a triplet lands on) -- and therefore within 1e-9
    of the 50-digit iteration with its iteration count."""
    g = np.load(os.path.join(golden_dir, fixture))
    pre = [p for _, p in golden_cases(g) if int(g[p + "meta"][0]) == 200][0]
    Cg, CalM = g[pre + "Corresp"], g[pre + "CalM"]
    C = full_size_batch[0].copy()
    assert np.array_equal(CalM, full_size_batch[1])
    pos = np.linspace(17, C.shape[0] - 23, Cg.shape[0]).astype(int)
    C[pos] = Cg
    big = gpu_ctx.pose_batch(method, C, CalM, reconst=False)
    small = gpu_ctx.pose_batch(method, Cg, CalM, reconst=False)
    assert np.all(np.asarray(big["status"]) == 0)
    for k in ("T", "R_t_2", "R_t_3", "iter"):
        assert np.array_equal(np.asarray(big[k])[pos], np.asarray(small[k])), k
    for b in range(Cg.shape[0]):
        assert int(small["iter"][b]) == int(g[pre + "mp_iter"][b])
        assert max(rel_err_T(small["T"][b], g[pre + "mp_T"][b]), rel_err(small["R_t_2"][b], g[pre + "mp_Rt2"][b]), rel_err(small["R_t_3"][b], g[pre + "mp_Rt3"][b])) < 1e-9


# ---------------------------------------------------------------------------
# OptimFPoseEstimation (F_methods/OptimFPoseEstimation.m, optimF.m): SURVEY 8(f) rank 1.
# One epipolar equation per correspondence -> scalar weight blocks, no 1e12-weighted directions:
# this Gauss-Helmert problem is well conditioned and parity is to rounding, iteration counts included
# (a stagnation-exit flip stays possible in principle; the tests allow it with the looser bound).
# ---------------------------------------------------------------------------
def _optimf_check(out, b, gT, gR2, gR3, gRec, git, where):
    dit = int(out["iter"][b]) - int(git)
    assert abs(dit) <= 1, where
    tol = 1e-8 if dit == 0 else 1e-4
    assert rel_err_T(out["T"][b], gT) < tol, where
    assert rel_err(out["R_t_2"][b], gR2) < tol and rel_err(out["R_t_3"][b], gR3) < tol, where
    if gRec is not None:
        assert rel_err(out["Reconst"][b], gRec) < 10 * tol, where
    return dit


@pytest.mark.parametrize("solver", ["invit", "jacobi"])
def test_optim_f_golden(gpu_ctx, golden_dir, solver):
    g = np.load(os.path.join(golden_dir, "optimf.npz"))
    e = np.load(os.path.join(golden_dir, "epfl.npz"))
    gpu_ctx.set_solver(solver)
    flips = 0
    try:
        for ci, pre in golden_cases(g):
            C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
            out = gpu_ctx.pose_batch("OptimFPoseEstimation", C, CalM, reconst=True)
            assert np.all(out["status"] == 0)
            for b in range(C.shape[0]):
                flips += _optimf_check(out, b, g[pre + "optimf_T"][b], g[pre + "optimf_Rt2"][b], g[pre + "optimf_Rt3"][b],
                                       g[pre + "optimf_Rec"][b], g[pre + "optimf_iter"][b], (ci, b)) != 0
        for n in range(int(e["count"])):
            pre = "t%d_" % n
            out = gpu_ctx.pose_batch("OptimFPoseEstimation", np.ascontiguousarray(e[pre + "sample"].T)[None], e[pre + "CalM"], reconst=True)
            assert out["status"][0] == 0
            flips += _optimf_check(out, 0, g[pre + "optimf_T"], g[pre + "optimf_Rt2"], g[pre + "optimf_Rt3"], g[pre + "optimf_Rec"],
                                   g[pre + "optimf_iter"], ("epfl", n)) != 0
    finally:
        gpu_ctx.set_solver("invit")
    assert flips <= 2


@pytest.mark.parametrize("N,sigma,seed", [(8, 1.0, 1), (9, 2.0, 2), (64, 1.0, 4), (65, 1.0, 5), (200, 1.0, 6), (257, 0.0, 7), (1500, 1.0, 9)])
def test_optim_f_vs_oracle_seeded(gpu_ctx, N, sigma, seed):
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    O = _oracle()
    B = 4
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=sigma, seed=seed)
    out = gpu_ctx.pose_batch("OptimFPoseEstimation", C, CalM, reconst=True)
    assert np.all(out["status"] == 0)
    for b in range(B):
        R2, R3, Rec, T, it = O.OptimFPoseEstimation(C[b].T.copy(), CalM)
        _optimf_check(out, b, T, R2, R3, Rec, it, (N, b))
    if N == 9:
        R2, R3, Rec, T, it = api.OptimFPoseEstimation(C[0].T.copy(), CalM)      # reference-shaped call
        assert it == int(out["iter"][0]) and rel_err(R2, out["R_t_2"][0]) < 1e-12
        with pytest.raises(ValueError):
            api.OptimFPoseEstimation(C[0].T[:, :7].copy(), CalM)                # optimF.m:36-38


def test_optim_f_full_size_improves_on_linear_f(gpu_ctx):
    """10k x 200 batch: every pair refined, iteration counts in the range the oracle shows (2..12 for it1 + it2),
    rotations orthonormal, mean pose error not worse than LinearFPoseEstimation's."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B, N = 10000, 200
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=1.0, seed=4321)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    lin = gpu_ctx.pose_batch("LinearFPoseEstimation", d, calm, reconst=False)
    res = gpu_ctx.pose_batch("OptimFPoseEstimation", d, calm, reconst=False)
    torch.cuda.synchronize()
    assert int((res["status"] != 0).sum()) == 0
    it = res["iter"].cpu().numpy()
    assert it.min() >= 2 and it.max() <= 40
    R3 = res["R_t_3"].cpu().numpy()[:, :, :3]
    assert np.abs(np.einsum("bij,bkj->bik", R3, R3) - np.eye(3)).max() < 1e-9
    assert np.abs(np.linalg.norm(res["R_t_2"].cpu().numpy()[:, :, 3], axis=1) - 1).max() < 1e-9   # |t2| = 1

    def rot_err(Rt):
        R = Rt.cpu().numpy()[:, :, :3]
        c = (np.einsum("ij,bij->b", Rt0[1][:, :3], R) - 1) / 2
        return np.degrees(np.arccos(np.clip(c, -1, 1)))
    assert rot_err(res["R_t_3"]).mean() <= rot_err(lin["R_t_3"]).mean()


# ---------------------------------------------------------------------------
# PiPoseEstimation / PiColPoseEstimation (TFT_methods/Pi*.m): SURVEY 8(f) rank 2.
# Same Gauss-Helmert noise level as Ressl (rank-3 B blocks -> 1e12 weights).  The Pi matrices are built from null
# vectors whose sign / basis the reference leaves to svd; Pi is invariant to them, PiCol is not
# (PiColPoseEstimation.m:93-94), so the oracle runs under the convention that reproduces the kernel's start.
# ---------------------------------------------------------------------------
PI_METHODS = [("PiPoseEstimation", None), ("PiColPoseEstimation", 180)]


@pytest.mark.parametrize("method,angle", PI_METHODS)
@pytest.mark.parametrize("N,sigma,seed", [(12, 1.0, 3), (40, 0.0, 4), (60, 1.0, 5), (200, 1.0, 6)])
def test_pi_methods_vs_oracle_in_kernel_convention(gpu_ctx, method, angle, N, sigma, seed):
    import torch
    from helpers import oracle_in_kernel_convention
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B = 4
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=sigma, seed=seed, angle=angle)
    out = gpu_ctx.pose_batch(method, torch.from_numpy(C).cuda(), torch.from_numpy(CalM).cuda(), reconst=True, debug=True)
    torch.cuda.synchronize()
    st = out["status"].cpu().numpy(); it_g = out["iter"].cpu().numpy()
    assert np.all(st == 0)
    ip = out["init_p"].cpu().numpy(); ix = out["init_x"].cpu().numpy()
    T = out["T"].cpu().numpy(); R2 = out["R_t_2"].cpu().numpy(); R3 = out["R_t_3"].cpu().numpy(); Rec = out["Reconst"].cpu().numpy()
    for b in range(B):
        (o2, o3, oRec, oT, it, d), dev = oracle_in_kernel_convention(method, C[b].T.copy(), CalM, ip[b], ix[b])
        dit = int(it_g[b]) - it                                                 # stagnation exits flip often for these models
        assert abs(dit) <= 5, (b, it_g[b], it)
        # PiCol's KKT matrix has singular values at pinv's truncation threshold (Gauss_Helmert.m:67): a step may or may not
        # include such a direction, on top of the 1e12-weight noise every trilinearity model has
        tol = 1e-8 if sigma == 0 else _ressl_tol(N, dit == 0) * ((5 if dit == 0 else 25) if angle else 1)
        assert rel_err_T(T[b], oT) < tol and rel_err(R2[b], o2) < tol and rel_err(R3[b], o3) < tol, (b, dit)
        assert rel_err(Rec[b], oRec) < 10 * tol


def test_pi_golden_lapack_convention(gpu_ctx, golden_dir):
    """PiPoseEstimation against goldens computed under LAPACK's sign conventions (synthetic + EPFL samples)."""
    g = np.load(os.path.join(golden_dir, "pi.npz"))
    e = np.load(os.path.join(golden_dir, "epfl.npz"))
    worst = {}
    for ci, pre in golden_cases(g, "p"):
        C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
        N, sigma = C.shape[1], float(g[pre + "meta"][1])
        out = gpu_ctx.pose_batch("PiPoseEstimation", C, CalM, reconst=True)
        assert np.all(out["status"] == 0)
        for b in range(C.shape[0]):
            dit = int(out["iter"][b]) - int(g[pre + "pi_iter"][b])
            assert abs(dit) <= 5
            err = max(rel_err_T(out["T"][b], g[pre + "pi_T"][b]), rel_err(out["R_t_2"][b], g[pre + "pi_Rt2"][b]), rel_err(out["R_t_3"][b], g[pre + "pi_Rt3"][b]))
            worst[(N, dit)] = max(worst.get((N, dit), 0), err)
            assert err < (1e-8 if sigma == 0 else _ressl_tol(N, dit == 0)), (ci, b, dit, err)
    epfl_dev = []
    for n in range(int(e["count"])):
        pre = "t%d_" % n
        out = gpu_ctx.pose_batch("PiPoseEstimation", np.ascontiguousarray(e[pre + "sample"].T)[None], e[pre + "CalM"], reconst=True)
        assert out["status"][0] == 0
        assert abs(int(out["iter"][0]) - int(g[pre + "pi_iter"])) <= 5
        epfl_dev.append(max(rel_err_T(out["T"][0], g[pre + "pi_T"]), rel_err(out["R_t_3"][0], g[pre + "pi_Rt3"])))
    # real data: noisier matches, ill-conditioned triplets where the first step is rounding-dominated: bound the bulk
    assert np.median(epfl_dev) < 5e-3 and sum(d > 5e-2 for d in epfl_dev) <= 2, epfl_dev
    print("Pi worst relative deviation from the LAPACK-convention oracle by (N, iteration difference):", worst)


@pytest.mark.parametrize("method,angle", PI_METHODS)
def test_pi_methods_noise_free_and_wrappers(gpu_ctx, method, angle):
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, Rt0, _ = generate_scene_batch(6, 40, noise=0.0, seed=13, angle=angle)
    out = gpu_ctx.pose_batch(method, C, CalM, reconst=False)
    assert np.all(out["status"] == 0) and np.all(out["iter"] <= 3)
    s = np.linalg.norm(Rt0[0][:, 3])
    for b in range(6):
        assert np.abs(out["R_t_2"][b][:, :3] - Rt0[0][:, :3]).max() < 1e-8 and np.abs(out["R_t_3"][b][:, :3] - Rt0[1][:, :3]).max() < 1e-8
        assert np.abs(out["R_t_3"][b][:, 3] - Rt0[1][:, 3] / s).max() < 1e-7
    R2, R3, Rec, T, it = getattr(api, method)(C[0].T.copy(), CalM)              # reference-shaped call
    assert it == int(out["iter"][0]) and R2.shape == (3, 4) and T.shape == (3, 3, 3) and Rec.shape == (3, 40)
    with pytest.raises(ValueError):
        getattr(api, method)(C[0].T[:, :6].copy(), CalM)


@pytest.mark.parametrize("method,angle,B", [("PiPoseEstimation", None, 10000), ("PiColPoseEstimation", 180, 10000)])     # BASELINE's batch size
def test_pi_methods_full_size_statistics(gpu_ctx, method, angle, B):
    """N = 200 batches: every triplet finishes, the iteration counts stay in the oracle's range, and the mean pose
    error is comparable with the linear method's (the refinement imposes the minimal parameterisation; on these scenes
    it does not degrade the linear solution by more than a few per cent, as in the oracle)."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    N = 200
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=1.0, seed=99, angle=angle)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    lin = gpu_ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=False)
    res = gpu_ctx.pose_batch(method, d, calm, reconst=False)
    torch.cuda.synchronize()
    st = res["status"].cpu().numpy()
    assert int((st != 0).sum()) <= (0 if angle is None else B // 50)            # PiCol: 'minimal param could not be found' is possible
    it = res["iter"].cpu().numpy()[st == 0]
    assert it.min() >= 1 and it.max() <= 60

    def rot_err(Rt):
        R = Rt.cpu().numpy()[st == 0][:, :, :3]
        c = (np.einsum("ij,bij->b", Rt0[1][:, :3], R) - 1) / 2
        return np.degrees(np.arccos(np.clip(c, -1, 1)))
    assert rot_err(res["R_t_3"]).mean() <= rot_err(lin["R_t_3"]).mean() * 1.25


def test_singular_kkt_takes_the_pinv_path(gpu_ctx):
    """Collinear camera centres make the KKT matrix of the generic parameterisations numerically singular in a fraction of the
    triplets (Pi: about a third at N = 100): Gauss_Helmert.m:67's pinv truncates there, and so do the kernels (eigen-decomposition
    fall-back of the elimination) -- every triplet returns a finite pose, none is reported as rank deficient."""
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, Rt0, _ = generate_scene_batch(600, 100, noise=1.0, seed=7, angle=180)
    for method in ("PiPoseEstimation", "NordbergTFTPoseEstimation", "ResslTFTPoseEstimation"):
        out = gpu_ctx.pose_batch(method, C, CalM, reconst=False)
        assert np.all(out["status"] == 0), (method, np.unique(out["status"], return_counts=True))
        assert np.all(np.isfinite(out["R_t_3"])) and np.all(np.isfinite(out["T"]))
        R = out["R_t_3"][:, :, :3]
        assert np.abs(np.einsum("bij,bkj->bik", R, R) - np.eye(3)).max() < 1e-9


def test_iterative_methods_at_large_n_use_the_global_workspace(gpu_ctx):
    """N = 1500: the per-correspondence state of the iterative methods (up to 26 N doubles) exceeds the 160 KB of LDS and lives in
    a global workspace instead; pinv(W) really truncates at this N (tolerance 4N eps(lambda_max) > 1e-12: eigen-decomposition path).
    Ressl is checked against the block-structured restatement (the dense oracle would pinv a 6000 x 6000 matrix)."""
    from oracle import gh_block_oracle as G
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B, N = 6, 1500
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=1.0, seed=321)
    errs = {}
    for method in ("ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "FaugPapaTFTPoseEstimation", "PiPoseEstimation", "OptimFPoseEstimation"):
        out = gpu_ctx.pose_batch(method, C, CalM, reconst=True)
        assert np.all(out["status"] == 0), method
        R = out["R_t_3"][:, :, :3]
        c = (np.einsum("ij,bij->b", Rt0[1][:, :3], R) - 1) / 2
        errs[method] = np.degrees(np.arccos(np.clip(c, -1, 1))).mean()
        assert errs[method] < 0.3, (method, errs[method])                    # sigma = 1 px at N = 1500: ~0.1 degree
        if method == "ResslTFTPoseEstimation":
            for b in range(2):
                R2, R3, Rec, T, it = G.ResslTFTPoseEstimation_blocks(C[b].T.copy(), CalM)
                dit = int(out["iter"][b]) - it
                assert abs(dit) <= 5
                tol = 1e-4 if dit == 0 else 2e-3
                assert rel_err_T(out["T"][b], T) < tol and rel_err(out["R_t_3"][b], R3) < tol

@pytest.mark.parametrize("method,N", [("OptimFPoseEstimation", 500), ("ResslTFTPoseEstimation", 500), ("NordbergTFTPoseEstimation", 400),
                                      ("FaugPapaTFTPoseEstimation", 200), ("PiPoseEstimation", 300), ("PiColPoseEstimation", 200),
                                      ("PiPoseEstimation", 64), ("OptimFPoseEstimation", 1500),
                                      ("ResslTFTPoseEstimation", 200), ("NordbergTFTPoseEstimation", 200)])   # (N = 200: xi in LDS, W+ in the slices -- FLAG_XI_IN_LDS)
def test_spilled_batches_are_bit_identical_to_single_triplets(gpu_ctx, method, N):
    """Configurations whose per-correspondence state lives in global spill slices (for occupancy, or because it exceeds the LDS):
    every triplet of a batch -- neighbouring blocks running concurrently on adjacent slices -- must come out bit-identical to the same
    triplet launched alone.  (A slice sized without the kernels' alignment pads once let OptimF's v overlap the next block's xi.)"""
    from tft_vs_fund_amd.scenes import generate_scene_batch
    # N = 1500: slices of 96 KB, the 512 MB workspace holds 5461 of them, so a batch of 6000 also runs the grid-stride loop (blocks
    # out of step with their neighbours -- the condition under which an overlap shows)
    B = 6000 if N == 1500 else 96
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=77 + N)
    full = gpu_ctx.pose_batch(method, C, CalM, reconst=True)
    for b in list(range(0, 6)) + [B // 2, B - 1]:
        one = gpu_ctx.pose_batch(method, C[b:b + 1], CalM, reconst=True)
        for k in ("T", "R_t_2", "R_t_3", "Reconst", "iter", "status"):
            assert np.array_equal(np.asarray(full[k][b]), np.asarray(one[k][0]), equal_nan=True), (method, N, b, k)


@pytest.mark.parametrize("method", ["ResslTFTPoseEstimation", "FaugPapaTFTPoseEstimation", "PiPoseEstimation", "OptimFPoseEstimation"])
def test_spill_only_if_needed_route_agrees_with_the_default(gpu_ctx, method):
    """TFF_OPT_SPILL = 1 keeps the per-correspondence state of the iterative methods in LDS whenever it fits (the default moves it to global
    slices when that buys occupancy): same arithmetic, so the same results and iteration counts."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B, N = 96, 200
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=314)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    ref = gpu_ctx.pose_batch(method, d, calm, reconst=False)
    gpu_ctx.set_spill_only_if_needed(True)
    try:
        out = gpu_ctx.pose_batch(method, d, calm, reconst=False)
    finally:
        gpu_ctx.set_spill_only_if_needed(False)
    torch.cuda.synchronize()
    assert torch.equal(out["status"], ref["status"]) and torch.equal(out["iter"], ref["iter"])
    for k in ("T", "R_t_2", "R_t_3"):
        assert (out[k] - ref[k]).abs().max().item() < 1e-9 * max(1.0, ref[k].abs().max().item()), k


def test_exact_fixup_pass_finds_every_retry_in_a_large_batch(gpu_ctx):
    """Minimal noisy samples with the whole-batch routing switched off (TFF_OPT_EXACT_BELOW = 0): the fast tiers flag the triplets
    they cannot finish or certify (status ST_RETRY inside the library) and the exact kernel, whose fixed 1024-block grid strides
    over the status array, redoes them.  None may be left behind, and the result must not depend on the route."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B = 40000
    C, CalM, _, _ = generate_scene_batch(B, 7, noise=3.0, seed=5)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    gpu_ctx.set_exact_below(0)
    try:
        out = gpu_ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=False, debug=True)
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_exact_below(12)
    st = out["status"].cpu().numpy(); dbg = out["debug"].cpu().numpy()
    assert np.all(st == 0)
    redone = dbg[:, 69] >= 10000                                           # the exact kernel stamps its iteration count + 10000
    assert 0 < redone.sum() < B and redone[1024:].sum() >= 1               # both routes ran, also beyond the first grid-full of triplets
    assert np.all(np.isfinite(out["T"].cpu().numpy()))
    ref = gpu_ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=False)   # default: the whole batch through the exact kernel
    torch.cuda.synchronize()
    T0 = ref["T"].cpu().numpy(); T1 = out["T"].cpu().numpy()
    eT = np.array([rel_err_T(T1[b], T0[b]) for b in range(0, B, 7)])
    # the fast tiers that were NOT flagged agree with the exact kernel (that is what the flags are for)
    assert np.quantile(eT, 0.999) < 1e-7 and eT.max() < 1e-5, (eT.max(), np.quantile(eT, 0.999))
    e3 = np.abs(out["R_t_3"].cpu().numpy() - ref["R_t_3"].cpu().numpy()).reshape(B, -1).max(axis=1)
    assert (e3 > 1e-6).mean() < 2e-3, (e3 > 1e-6).mean()                    # only cheirality ties / rounding-level sign decisions may differ


def test_all_120_epfl_list_triplets_against_ground_truth_and_oracle(gpu_ctx, golden_dir):
    """The reference's real-data lists (70 + 50 triplets, experiments_real.m:31-35,78), one deterministic 100-inlier sample each (the
    samples of tests/test_oracle_pins.py), through the HIP path: the seven methods of experiments_real.m:62 against the `.camera`
    ground truth (per-method medians), the two linear methods against the numpy oracle triplet by triplet at 1e-9 (svd(E)-sign ties
    under any convention)."""
    from test_oracle_pins import _epfl_list_samples, _gt_err
    from helpers import pose_err_any_convention
    O = _oracle()
    samples = _epfl_list_samples(golden_dir)
    meths = ["LinearTFTPoseEstimation", "ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "FaugPapaTFTPoseEstimation",
             "PiPoseEstimation", "LinearFPoseEstimation", "OptimFPoseEstimation"]
    by_n = {}
    for k, smp in enumerate(samples):
        by_n.setdefault(smp[2].shape[1], []).append(k)
    out = {m: [None] * len(samples) for m in meths}
    for n, ks in by_n.items():                                                   # one batched call per method and sample size, per-triplet calibration
        C = np.ascontiguousarray(np.stack([samples[k][2].T for k in ks]))
        CalM = np.stack([samples[k][3] for k in ks])
        for m in meths:
            r = gpu_ctx.pose_batch(m, C, CalM, reconst=False)
            for j, k in enumerate(ks):
                out[m][k] = dict(T=np.asarray(r["T"][j]), R_t_2=np.asarray(r["R_t_2"][j]), R_t_3=np.asarray(r["R_t_3"][j]), status=int(r["status"][j]))
    for m in meths:
        errs = {"fountain": [], "herzjesu": []}
        for k, (dataset, ti, S, CalM, Rt0) in enumerate(samples):
            if out[m][k]["status"] != 0:
                continue
            errs[dataset].append(_gt_err(Rt0, (out[m][k]["R_t_2"], out[m][k]["R_t_3"])))
        for dataset, v in errs.items():
            a = np.array(v)
            assert len(v) >= (68 if dataset == "fountain" else 48), (m, dataset, len(v))
            assert np.median(a[:, 0]) < (0.3 if dataset == "fountain" else 0.8), (m, dataset, np.median(a[:, 0]))
            assert np.median(a[:, 1]) < (0.9 if dataset == "fountain" else 1.5), (m, dataset, np.median(a[:, 1]))
    for m in ("LinearTFTPoseEstimation", "LinearFPoseEstimation"):
        worst = 0.0
        for k, (dataset, ti, S, CalM, Rt0) in enumerate(samples):
            e0, eb = pose_err_any_convention(out[m][k], getattr(O, m), S.copy(), CalM)
            worst = max(worst, eb)
            assert eb < 1e-9, (m, dataset, ti, e0, eb)
        print("%s on the 120 list triplets: worst deviation from the oracle %.2e" % (m, worst))
