"""
CPU coverage of the HIP kernel *logic*: the kernels of tft_vs_fund_amd/csrc are
compiled by g++ against tests/emu/hip_emu.h (one thread per lane, barriers for
the cross-lane primitives) and compared with the oracle.  This is test
infrastructure -- the shipped library has no CPU path -- and the real parity
gate is tests/test_gpu_parity.py on the MI355X.
"""
import ctypes
import os

import numpy as np
import pytest

from oracle import tft_oracle as O
from tft_vs_fund_amd.scenes import generate_scene_batch, calm_colmajor
from helpers import rel_err_T, rel_err
from emu import emu_build

FLAG_JACOBI = 2


@pytest.fixture(scope="module")
def emu():
    return emu_build.load()


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def run_linear_tft(lib, C, CalM, flags=0, reconst=True, entry="emu_linear_tft_pose", debug=True):
    B, N, _ = C.shape
    calm = calm_colmajor(CalM)
    Rt2 = np.zeros((B, 12)); Rt3 = np.zeros((B, 12)); T = np.zeros((B, 27))
    Rec = np.zeros((B, N, 3)) if reconst else None
    it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32); dbg = np.zeros((B, 128))
    getattr(lib, entry)(_p(C), _p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N), ctypes.c_int(flags),
                            _p(Rt2), _p(Rt3), _p(T), _p(Rec), _p(it), _p(st), _p(dbg) if debug else None)
    return dict(R_t_2=Rt2.reshape(B, 4, 3).transpose(0, 2, 1), R_t_3=Rt3.reshape(B, 4, 3).transpose(0, 2, 1),
                T=T.reshape(B, 3, 3, 3).transpose(0, 3, 2, 1), Reconst=None if Rec is None else Rec.transpose(0, 2, 1),
                iter=it, status=st, debug=dbg)


@pytest.mark.parametrize("N,sigma,flags", [(7, 1.0, 0), (12, 1.0, 0), (70, 0.0, 0), (130, 1.0, 0), (12, 1.0, FLAG_JACOBI)])
def test_linear_tft_kernel_matches_oracle(emu, N, sigma, flags):
    B = 2
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=sigma, seed=100 + N)
    out = run_linear_tft(emu, C, CalM, flags)
    assert np.all(out["status"] == 0) and np.all(out["iter"] == 0)
    for b in range(B):
        R2, R3, Rec, T, _ = O.LinearTFTPoseEstimation(C[b].T.copy(), CalM)
        tol = 1e-9                                            # fp64; expected 1e-11 or better
        assert rel_err_T(out["T"][b], T) < tol
        assert rel_err(out["R_t_2"][b], R2) < tol and rel_err(out["R_t_3"][b], R3) < tol
        assert rel_err(out["Reconst"][b], Rec) < tol
        votes = out["debug"][b, 60:68]
        assert sorted(np.abs(votes[0:4])) == [0, 0, 2 * N, 2 * N] or sigma > 0


@pytest.mark.parametrize("entry,N", [("emu_linear_tft_pose", 9), ("emu_linear_f_pose", 9), ("emu_optim_f_pose", 10), ("emu_ressl_tft_pose", 9)])
def test_grid_stride_loop_leaves_no_state_between_triplets(emu, entry, N):
    """One block taking several triplets through the same LDS (grid capped at 1) gives bit-identical results to one block per triplet."""
    B = 2
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=40 + N)
    ref = run_linear_tft(emu, C, CalM, entry=entry)
    emu.emu_set_grid_cap(1)
    try:
        out = run_linear_tft(emu, C, CalM, entry=entry)
    finally:
        emu.emu_set_grid_cap(0)
    for k in ("T", "R_t_2", "R_t_3", "Reconst", "iter", "status"):
        assert np.array_equal(ref[k], out[k], equal_nan=True), (entry, k)


def test_too_few_points_sets_status(emu):
    C, CalM, _, _ = generate_scene_batch(1, 6, noise=1.0, seed=1)
    out = run_linear_tft(emu, C, CalM)
    assert out["status"][0] == 1 and np.all(np.isnan(out["T"][0]))


@pytest.mark.parametrize("N,sigma,flags", [(8, 1.0, 0), (12, 1.0, 0), (70, 0.0, 0), (130, 1.0, 0), (12, 1.0, FLAG_JACOBI)])
def test_linear_f_kernel_matches_oracle(emu, N, sigma, flags):
    B = 2
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=sigma, seed=200 + N)
    out = run_linear_tft(emu, C, CalM, flags, entry="emu_linear_f_pose")
    assert np.all(out["status"] == 0) and np.all(out["iter"] == 0)
    for b in range(B):
        R2, R3, Rec, T, _ = O.LinearFPoseEstimation(C[b].T.copy(), CalM)
        tol = 1e-9 if N >= 12 else 1e-6
        assert rel_err_T(out["T"][b], T) < tol
        assert rel_err(out["R_t_2"][b], R2) < tol and rel_err(out["R_t_3"][b], R3) < tol
        assert rel_err(out["Reconst"][b], Rec) < tol


def test_linear_f_needs_8_points(emu):
    C, CalM, _, _ = generate_scene_batch(1, 7, noise=1.0, seed=1)
    out = run_linear_tft(emu, C, CalM, entry="emu_linear_f_pose")
    assert out["status"][0] == 1 and np.all(np.isnan(out["T"][0]))


@pytest.mark.parametrize("entry", ["emu_optim_f_pose", "emu_optim_f_pose_staged"])
def test_optim_f_kernel_matches_golden(emu, golden_dir, entry):
    """OptimFPoseEstimation kernel (linearF start + per-pair Gauss-Helmert with scalar weight blocks): the
    epipolar model is well conditioned, so iteration counts and results match the dense oracle to rounding.
    emu_optim_f_pose: the fused one-triplet kernel (small batches); emu_optim_f_pose_staged: the three stages of csrc/optimf_rows_kernel.h
    (linear stage and pose tail four triplets per wavefront, the refinement with all 54 sums of an iteration in one sweep)."""
    import os
    g = np.load(os.path.join(golden_dir, "optimf.npz"))
    for pre, flags, nb in (("c0_", 0, 2), ("c3_", 0, 1), ("c1_", FLAG_JACOBI, 1)):
        if flags and entry.endswith("staged"):
            continue
        C, CalM = g[pre + "Corresp"][:nb], g[pre + "CalM"]
        out = run_linear_tft(emu, C, CalM, flags, entry=entry)
        assert np.all(out["status"] == 0)
        for b in range(nb):
            assert int(out["iter"][b]) == int(g[pre + "optimf_iter"][b])
            assert rel_err_T(out["T"][b], g[pre + "optimf_T"][b]) < 1e-8
            assert rel_err(out["R_t_2"][b], g[pre + "optimf_Rt2"][b]) < 1e-8 and rel_err(out["R_t_3"][b], g[pre + "optimf_Rt3"][b]) < 1e-8
            assert rel_err(out["Reconst"][b], g[pre + "optimf_Rec"][b]) < 1e-8
    e = np.load(os.path.join(golden_dir, "epfl.npz"))
    out = run_linear_tft(emu, np.ascontiguousarray(e["t3_sample"].T)[None], e["t3_CalM"], entry=entry)
    assert out["status"][0] == 0 and int(out["iter"][0]) == int(g["t3_optimf_iter"])
    assert rel_err_T(out["T"][0], g["t3_optimf_T"]) < 1e-8 and rel_err(out["R_t_3"][0], g["t3_optimf_Rt3"]) < 1e-8
    C, CalM, _, _ = generate_scene_batch(1, 7, noise=1.0, seed=1)
    out = run_linear_tft(emu, C, CalM, entry=entry)
    assert out["status"][0] == 1 and np.all(np.isnan(out["T"][0]))          # optimF.m:36-38


def test_optim_f_staged_matches_fused_and_oracle(emu):
    """Five triplets (a ragged last wavefront in the two row-layout stages), N = 40: the staged route against the fused kernel and the oracle --
    same iteration counts, results within the Gauss-Helmert loop's amplification of the start's rounding."""
    B, N = 5, 40
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=21)
    st = run_linear_tft(emu, C, CalM, entry="emu_optim_f_pose_staged", debug=False)
    fu = run_linear_tft(emu, C, CalM, entry="emu_optim_f_pose", debug=False)
    assert np.all(st["status"] == 0) and np.array_equal(st["iter"], fu["iter"]) and np.all(st["iter"] >= 2)
    assert np.abs(st["R_t_2"] - fu["R_t_2"]).max() < 1e-8 and np.abs(st["R_t_3"] - fu["R_t_3"]).max() < 1e-8
    for b in (0, B - 1):
        R2, R3, Rec, T, it = O.OptimFPoseEstimation(C[b].T.copy(), CalM)
        assert int(it) == int(st["iter"][b])
        assert rel_err_T(st["T"][b], T) < 1e-8 and rel_err(st["R_t_2"][b], R2) < 1e-8 and rel_err(st["R_t_3"][b], R3) < 1e-8
        assert rel_err(st["Reconst"][b], Rec) < 1e-8


@pytest.mark.parametrize("refine", [0, 1])
def test_linear_f_block_kernel(emu, refine):
    """k_linear_f: linearF / optimF for the pairs (1,2), (1,3) (the tff_linear_f_batch_dev building block)."""
    B, N = 2, 20
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=77)
    F21 = np.zeros((B, 9)); F31 = np.zeros((B, 9)); it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
    emu.emu_linear_f(_p(C), ctypes.c_long(B), ctypes.c_int(N), ctypes.c_int(refine), _p(F21), _p(F31), _p(it), _p(st))
    assert np.all(st == 0)
    for b in range(B):
        Cb = C[b].T.copy()
        if refine:
            (f21, i1), (f31, i2) = O.optimF(Cb[0:2], Cb[2:4]), O.optimF(Cb[0:2], Cb[4:6])
            assert it[b] == i1 + i2
        else:
            f21, f31 = O.linearF(Cb[0:2], Cb[2:4]), O.linearF(Cb[0:2], Cb[4:6])
        assert rel_err_T(F21[b].reshape(3, 3).T, f21) < 1e-8 and rel_err_T(F31[b].reshape(3, 3).T, f31) < 1e-8


def test_gh_workgroup_path_matches_fused_kernel(emu):
    """The default Gauss-Helmert path (gh_wg_kernel.h: k_gh_linear -> k_gh_block, four wavefronts per triplet -> k_gh_finish)
    against the fused single-wavefront kernel: same per-correspondence arithmetic, sums grouped per wavefront."""
    C, CalM, _, _ = generate_scene_batch(1, 13, noise=1.0, seed=53)
    B, N = 1, 13
    calm = calm_colmajor(CalM)
    Rt2 = np.zeros((B, 12)); Rt3 = np.zeros((B, 12)); T = np.zeros((B, 27)); Rec = np.zeros((B, N, 3))
    it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
    emu.emu_gh_wg_pose(ctypes.c_int(0), _p(C), _p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N), ctypes.c_int(0),
                       _p(Rt2), _p(Rt3), _p(T), _p(Rec), _p(it), _p(st))
    ref = run_linear_tft(emu, C, CalM, entry="emu_ressl_tft_pose")
    assert st[0] == 0 and ref["status"][0] == 0 and abs(int(it[0]) - int(ref["iter"][0])) <= 2
    tol = 2e-3 if it[0] == ref["iter"][0] else 1e-2                          # Gauss-Helmert noise level at N ~ 12
    assert rel_err_T(T.reshape(3, 3, 3).transpose(2, 1, 0), ref["T"][0]) < tol
    assert rel_err(Rt3.reshape(4, 3).T, ref["R_t_3"][0]) < tol and rel_err(Rec[0].T, ref["Reconst"][0]) < 10 * tol
    C7, _, _, _ = generate_scene_batch(1, 6, noise=1.0, seed=5)               # too few points: status 1, NaN outputs
    Rt2 = np.zeros((1, 12)); Rt3 = np.zeros((1, 12)); T = np.zeros((1, 27)); Rec = np.zeros((1, 6, 3))
    emu.emu_gh_wg_pose(ctypes.c_int(0), _p(C7), _p(calm), ctypes.c_long(0), ctypes.c_long(1), ctypes.c_int(6), ctypes.c_int(0),
                       _p(Rt2), _p(Rt3), _p(T), _p(Rec), _p(it), _p(st))
    assert st[0] == 1 and np.all(np.isnan(T))


def test_faugpapa_workgroup_path_matches_fused_kernel(emu):
    """FaugPapa through k_gh_block: the pseudo-inverse workspace overlays D | H | Y there (gh_wg_carve) and the solve runs on one
    wavefront of the four; against the fused single-wavefront kernel (separate workspace)."""
    B, N = 1, 9
    C, CalM, _, _ = generate_scene_batch(B, N, noise=0.5, seed=77)
    calm = calm_colmajor(CalM)
    Rt2 = np.zeros((B, 12)); Rt3 = np.zeros((B, 12)); T = np.zeros((B, 27)); Rec = np.zeros((B, N, 3))
    it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
    emu.emu_gh_wg_pose(ctypes.c_int(2), _p(C), _p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N), ctypes.c_int(0),
                       _p(Rt2), _p(Rt3), _p(T), _p(Rec), _p(it), _p(st))
    ref = run_linear_tft(emu, C, CalM, entry="emu_faugpapa_tft_pose")
    assert st[0] == 0 and ref["status"][0] == 0 and abs(int(it[0]) - int(ref["iter"][0])) <= 2
    tol = 2e-3 if it[0] == ref["iter"][0] else 2e-2
    assert rel_err_T(T.reshape(3, 3, 3).transpose(2, 1, 0), ref["T"][0]) < tol
    assert rel_err(Rt3.reshape(4, 3).T, ref["R_t_3"][0]) < tol


def test_wave_eigh_ql_against_lapack(emu):
    """wave_eigh_ql (Householder tridiagonalisation + implicit QL on one wavefront) and the truncated pseudo-inverse built on it:
    eigenvalues, orthogonality, residual and pinv(M) b against numpy on generic, singular and KKT-graded matrices."""
    rng = np.random.default_rng(0)
    for n in (12, 39):                                                                       # the sizes in use are 20 ... 39; pinv's n * eps margin is thin below ~10
        cases = []
        if n < 39:
            G = rng.standard_normal((n, n - 3)); cases.append(G @ G.T)                     # singular: three eigenvalues truncated
            cases.append(np.diag(rng.standard_normal(n)))
        u = (2 * n) // 3; c = n - u                                                        # [N C'; C 1e-12 I], N ~ 1e13 and rank deficient
        J = rng.standard_normal((u - 1, u)); K = np.zeros((n, n))
        K[:u, :u] = 1e13 * J.T @ J; K[u:, :u] = rng.standard_normal((c, u)); K[:u, u:] = K[u:, :u].T; K[u:, u:] = 1e-12 * np.eye(c)
        cases.append(K)
        Ms = np.stack(cases); bs = rng.standard_normal((len(cases), n))
        Maug = np.concatenate([Ms, bs[:, :, None]], axis=2).copy()
        lam = np.zeros((len(cases), n)); vecs = np.zeros((len(cases), n, n)); sol = np.zeros((len(cases), n))
        emu.emu_eigh(_p(Maug), ctypes.c_long(len(cases)), ctypes.c_int(n), _p(lam), _p(vecs), _p(sol))
        for i, M in enumerate(Ms):
            nrm = np.linalg.norm(M, 2)
            w, VV = np.linalg.eigh(M)
            V = vecs[i].T
            assert np.abs(np.sort(lam[i]) - w).max() < 1e-14 * nrm
            assert np.abs(V.T @ V - np.eye(n)).max() < 1e-13
            assert np.abs(M @ V - V * lam[i]).max() < 1e-14 * nrm
            tol = n * np.spacing(np.abs(w).max())
            if np.min(np.abs(np.abs(w) - tol)) > 0.5 * tol:                                # no eigenvalue sits at the truncation threshold
                keep = np.abs(w) > tol
                xref = VV[:, keep] @ ((VV[:, keep].T @ bs[i]) / w[keep])
                # graded KKT case: eigenvalues of O(1) next to |M| ~ 1e13 carry an absolute error eps * |M| ~ 1e-3 in ANY backward-stable
                # solver (LAPACK included), which is what the multiplier block sees; the parameter block does not
                head = n if i < len(cases) - 1 else (2 * n) // 3
                assert np.abs(sol[i][:head] - xref[:head]).max() < 1e-9 * np.abs(xref).max()
                assert np.abs(sol[i] - xref).max() < 2e-2 * np.abs(xref).max()


@pytest.mark.parametrize("collinear,angle", [(0, None), (1, 180)])
def test_pi_workgroup_path_matches_fused_kernel(emu, collinear, angle):
    """k_pi_block (four wavefronts per triplet, pi_wg_kernel.h) against the fused single-wavefront Pi kernels."""
    B, N = 1, 14
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=61, angle=angle)
    calm = calm_colmajor(CalM)
    Rt2 = np.zeros((B, 12)); Rt3 = np.zeros((B, 12)); T = np.zeros((B, 27)); Rec = np.zeros((B, N, 3))
    it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
    emu.emu_pi_wg_pose(ctypes.c_int(collinear), _p(C), _p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N), ctypes.c_int(0),
                       _p(Rt2), _p(Rt3), _p(T), _p(Rec), _p(it), _p(st))
    ref = run_linear_tft(emu, C, CalM, entry="emu_picol_pose" if collinear else "emu_pi_pose")
    assert st[0] == 0 and ref["status"][0] == 0 and abs(int(it[0]) - int(ref["iter"][0])) <= 5
    tol = (2e-3 if it[0] == ref["iter"][0] else 1e-2) * (10 if collinear else 1)
    assert rel_err_T(T.reshape(3, 3, 3).transpose(2, 1, 0), ref["T"][0]) < tol
    assert rel_err(Rt3.reshape(4, 3).T, ref["R_t_3"][0]) < tol and rel_err(Rec[0].T, ref["Reconst"][0]) < 10 * tol


def run_pi_debug(lib, collinear, C, CalM, flags=0):
    B, N, _ = C.shape
    calm = calm_colmajor(CalM)
    Rt2 = np.zeros((B, 12)); Rt3 = np.zeros((B, 12)); T = np.zeros((B, 27)); Rec = np.zeros((B, N, 3))
    it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32); ip = np.zeros((B, 27)); ix = np.zeros((B, 6 * N))
    lib.emu_pi_pose_debug(ctypes.c_int(collinear), _p(C), _p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N), ctypes.c_int(flags),
                          _p(Rt2), _p(Rt3), _p(T), _p(Rec), _p(it), _p(st), _p(ip), _p(ix))
    return dict(R_t_2=Rt2.reshape(B, 4, 3).transpose(0, 2, 1), R_t_3=Rt3.reshape(B, 4, 3).transpose(0, 2, 1),
                T=T.reshape(B, 3, 3, 3).transpose(0, 3, 2, 1), Reconst=Rec.transpose(0, 2, 1), iter=it, status=st, init_p=ip, init_x=ix)


@pytest.mark.parametrize("method,collinear,angle,N,sigma", [("PiPoseEstimation", 0, None, 12, 1.0), ("PiPoseEstimation", 0, None, 30, 0.0),
                                                            ("PiColPoseEstimation", 1, 180, 14, 1.0), ("PiColPoseEstimation", 1, 180, 30, 0.0)])
def test_pi_kernels_match_oracle_under_their_sign_convention(emu, method, collinear, angle, N, sigma):
    """Pi / PiCol kernels: the start of the Gauss-Helmert iteration is the oracle's under one of the svd sign
    conventions the reference leaves open (tests/helpers.py), and from there results agree to the Gauss-Helmert
    noise level (1e12-weighted normal equations, see test_gpu_parity.py)."""
    from helpers import oracle_in_kernel_convention
    B = 1 if collinear else 2                                               # the 38x38 pseudo-inverse is slow under emulation
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=sigma, seed=900 + N, angle=angle)
    out = run_pi_debug(emu, collinear, C, CalM)
    assert np.all(out["status"] == 0)
    for b in range(B):
        (R2, R3, Rec, T, it, d), dev = oracle_in_kernel_convention(method, C[b].T.copy(), CalM, out["init_p"][b], out["init_x"][b])
        dit = int(out["iter"][b]) - it
        assert abs(dit) <= 5
        tol = 1e-8 if sigma == 0 else (2e-3 if dit == 0 else 1e-2)
        assert rel_err_T(out["T"][b], T) < tol and rel_err(out["R_t_2"][b], R2) < tol and rel_err(out["R_t_3"][b], R3) < tol
        assert rel_err(out["Reconst"][b], Rec) < 10 * tol


def test_pi_kernel_against_lapack_convention_golden(emu, golden_dir):
    """PiPoseEstimation is invariant to those sign choices (they relabel an equivalent problem): the kernel also agrees
    with the golden output computed under LAPACK's conventions."""
    import os
    g = np.load(os.path.join(golden_dir, "pi.npz"))
    C, CalM = g["p2_Corresp"][:1], g["p2_CalM"]                              # N = 50, sigma = 1
    out = run_linear_tft(emu, C, CalM, entry="emu_pi_pose")
    assert np.all(out["status"] == 0)
    for b in range(1):
        dit = int(out["iter"][b]) - int(g["p2_pi_iter"][b])
        assert abs(dit) <= 5
        tol = 1e-4 if dit == 0 else 2e-3
        assert rel_err_T(out["T"][b], g["p2_pi_T"][b]) < tol and rel_err(out["R_t_3"][b], g["p2_pi_Rt3"][b]) < tol


def test_ressl_kernel_matches_block_checker(emu, golden_dir):
    """Gauss-Helmert kernel on one N = 12 triplet: against the same-block-algebra restatement
    (oracle/gh_block_oracle.py) and the dense oracle's golden output.  Tolerances: see
    tests/test_gpu_parity.py (the algorithm is rounding-sensitive by construction)."""
    import os
    from oracle import gh_block_oracle as G
    g = np.load(os.path.join(golden_dir, "synthetic_gh.npz"))
    C, CalM = g["c1_Corresp"][1:2], g["c1_CalM"]
    out = run_linear_tft(emu, C, CalM, entry="emu_ressl_tft_pose")
    assert out["status"][0] == 0
    R2, R3, Rec, T, it = G.ResslTFTPoseEstimation_blocks(C[0].T.copy(), CalM)
    assert abs(int(out["iter"][0]) - it) <= 1 and abs(int(out["iter"][0]) - int(g["c1_ressl_iter"][1])) <= 2
    assert rel_err_T(out["T"][0], T) < 2e-3 and rel_err(out["R_t_3"][0], R3) < 2e-3
    assert rel_err_T(out["T"][0], g["c1_ressl_T"][1]) < 1e-2 and rel_err(out["R_t_3"][0], g["c1_ressl_Rt3"][1]) < 1e-2


@pytest.mark.parametrize("deficient", ["P2", "P3", "none"])
def test_nordberg_projective_fixup_executes(emu, deficient):
    """NordbergTFTPoseEstimation.m:56-62: when P3(:,1:3) (else P2(:,1:3)) of the linear solution has rank 2, the cameras are
    transformed by H = [I 0; null(.)' 1] before the parameterisation.  No correspondence set reaches that branch through the
    whole pipeline (it needs sigma_3 <= 3 eps(sigma_1) in linearTFT's output), so the kernel's NordbergModel::init is fed
    cameras with an EXACTLY rank-deficient block and compared with the oracle's nordberg_param0 on the same cameras."""
    rng = np.random.default_rng(11)
    A = rng.standard_normal((3, 3)); B = rng.standard_normal((3, 3))
    a = rng.standard_normal(3); a /= np.linalg.norm(a)
    b = rng.standard_normal(3); b /= np.linalg.norm(b)
    if deficient == "P2":
        A[:, 2] = 2.0 * A[:, 0] - 0.5 * A[:, 1]                               # exact rank 2 (up to one rounding per entry)
    if deficient == "P3":
        B[:, 1] = 0.25 * B[:, 0] + 3.0 * B[:, 2]
    if deficient != "none":
        M = A if deficient == "P2" else B
        s = np.linalg.svd(M, compute_uv=False)
        if not s[2] <= 3 * np.spacing(s[0]):                                  # one more projection makes it rank 2 to the last bit rank() looks at
            U, sv, Vt = np.linalg.svd(M); sv[2] = 0.0; M[:] = (U * sv) @ Vt
        assert O.rank(M) == 2
    P1 = np.eye(3, 4); P2 = np.hstack([A, a.reshape(3, 1)]); P3 = np.hstack([B, b.reshape(3, 1)])
    T = np.stack([np.outer(A[:, i], b) - np.outer(a, B[:, i]) for i in range(3)], axis=2)       # T_i = a_i e31' - e21 b_i'   (linearTFT.m:87-91)
    T = T / np.linalg.norm(T)
    refs = [O.nordberg_param0(T, P1, P2, P3, sg) for sg in ((1.0, -1.0) if deficient != "none" else (1.0,))]
    _, oP2, oP3, _ = refs[0]
    if deficient != "none":
        assert O.rank(oP2[:, 0:3]) == 3 and O.rank(oP3[:, 0:3]) == 3 and (np.abs(oP2 - P2).max() > 1e-3 or np.abs(oP3 - P3).max() > 1e-3)
    t = np.ascontiguousarray(T.transpose(2, 1, 0)).reshape(27)               # vec order j + 3k + 9i
    pa = np.concatenate([A.reshape(9, order="F"), B.reshape(9, order="F")])
    epi = np.concatenate([a, b])
    p = np.zeros(19); bad = ctypes.c_int(-1)
    emu.emu_nordberg_init(_p(t), _p(pa), _p(epi), _p(p), ctypes.byref(bad))
    assert bad.value == 0
    # null(.)'s sign is svd's choice: the kernel must reproduce the oracle's parameters under one of the two signs
    # (rotation vectors directly; the normalised sparse tensor part up to its global sign)
    dev = [max(np.abs(p[0:9] - r[3][0:9]).max(), min(np.abs(p[9:19] - r[3][9:19]).max(), np.abs(p[9:19] + r[3][9:19]).max())) for r in refs]
    assert min(dev) < 1e-8, dev


def test_pinv_shortcut_equals_jacobi_branch_on_graded_blocks(emu):
    """pinv_one_null_packed (Cholesky of W + nn', pi_kernel.h) against the Jacobi eigen-decomposition branch it replaces, on weight
    blocks whose kept eigenvalues reach down to ~1e-6 of the largest while ONE direction sits under MATLAB's pinv tolerance
    (Gauss_Helmert.m:52,57: W = B B' + 1e-12 I, the shift already applied by the caller)."""
    emu.emu_pinv_one_null.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(12)
    worst = 0.0
    for trial in range(200):
        Q, _ = np.linalg.qr(rng.standard_normal((4, 4)))
        lam = np.array([10 ** rng.uniform(2, 4), 10 ** rng.uniform(-1, 2), 10 ** rng.uniform(-3, -1), 10 ** rng.uniform(-14, -11)])
        W = (Q * (lam + 1e-12)) @ Q.T
        W = 0.5 * (W + W.T)
        tolW = 4 * 200 * np.spacing(lam[0])                                      # E N eps(|W|): between lam[3] and lam[2]
        assert lam[3] + 1e-12 < tolW < lam[2]
        a = np.zeros(10); b = np.zeros(10); ok = ctypes.c_int(0)
        emu.emu_pinv_one_null(_p(np.ascontiguousarray(W)), float(tolW), _p(a), _p(b), ctypes.byref(ok))
        assert ok.value == 1, trial
        ref = (Q[:, :3] / (lam[:3] + 1e-12)) @ Q[:, :3].T                        # the pseudo-inverse itself, from the construction
        refp = np.array([ref[i, j] for i in range(4) for j in range(i + 1)])
        scale = np.abs(refp).max()
        # both branches within the conditioning of the problem (1e-16 lam_1 / lam_3 ~ 1e-9 at worst) of the construction
        assert np.abs(a - refp).max() < 1e-8 * scale, (trial, np.abs(a - refp).max() / scale)
        assert np.abs(b - refp).max() < 1e-8 * scale, (trial, np.abs(b - refp).max() / scale)
        worst = max(worst, np.abs(a - b).max() / scale)
    assert worst < 1e-8
    # no direction under the tolerance: the shortcut declines (the caller then takes the Jacobi branch)
    W = np.diag([100.0, 10.0, 1.0, 0.5])
    a = np.zeros(10); b = np.zeros(10); ok = ctypes.c_int(1)
    emu.emu_pinv_one_null(_p(W), 1e-9, _p(a), _p(b), ctypes.byref(ok))
    assert ok.value == 0


@pytest.mark.parametrize("model,fixture", [("ressl", "gh_mp.npz"), ("nordberg", "gh_mp_nordberg.npz"), ("pi", "gh_mp_pi.npz")])
def test_emulated_workgroup_kernels_reproduce_the_50_digit_iteration(emu, model, fixture):
    """The GPU-less twin of tests/test_gpu_gh_noise.py: the workgroup Gauss-Helmert kernels (factored weights), compiled against the
    lane emulator, on the first two N = 12 scenes of the extended-precision fixtures -- T, R_t_2, R_t_3 within 1e-9 of the 50-digit
    evaluation (observed <= 6e-11) and the same iteration count; Nordberg under one of the eight sign conventions of linearTFT's
    singular vectors (tests/helpers.py::oracle_under_epipole_conventions)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture))
    pre = "c0_"
    C = np.ascontiguousarray(g[pre + "Corresp"][:2]); CalM = g[pre + "CalM"]
    B, N, _ = C.shape
    calm = calm_colmajor(CalM)
    Rt2 = np.zeros((B, 12)); Rt3 = np.zeros((B, 12)); T = np.zeros((B, 27)); it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
    if model == "pi":
        emu.emu_pi_wg_pose(ctypes.c_int(0), _p(C), _p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N), ctypes.c_int(0),
                           _p(Rt2), _p(Rt3), _p(T), None, _p(it), _p(st))
    else:
        emu.emu_gh_wg_pose(ctypes.c_int(0 if model == "ressl" else 1), _p(C), _p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N),
                           ctypes.c_int(0), _p(Rt2), _p(Rt3), _p(T), None, _p(it), _p(st))
    assert np.all(st == 0)
    if model == "nordberg":
        # the serial part of the initial parameters ran in k_nordberg_init, one triplet per lane (as the C ABI); flag 4096: on lane 0 of the block
        # kernel's owner wavefront, as the fused kernel does -- same function, same results
        T0 = np.zeros_like(T); R20 = np.zeros_like(Rt2); R30 = np.zeros_like(Rt3); it0 = np.zeros_like(it); st0 = np.zeros_like(st)
        emu.emu_gh_wg_pose(ctypes.c_int(1), _p(C), _p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N), ctypes.c_int(4096),
                           _p(R20), _p(R30), _p(T0), None, _p(it0), _p(st0))
        assert np.array_equal(it0, it) and np.array_equal(st0, st)
        assert np.abs(T0 - T).max() < 1e-12 and np.abs(R30 - Rt3).max() < 1e-12 * max(1.0, np.abs(Rt3).max())
    Tt = T.reshape(B, 3, 3, 3).transpose(0, 3, 2, 1); R2 = Rt2.reshape(B, 4, 3).transpose(0, 2, 1); R3 = Rt3.reshape(B, 4, 3).transpose(0, 2, 1)
    for b in range(B):
        if pre + "mp4_T" in g.files:
            cand = [(max(rel_err_T(Tt[b], g[pre + "mp4_T"][b, c]), rel_err(R2[b], g[pre + "mp4_Rt2"][b, c]), rel_err(R3[b], g[pre + "mp4_Rt3"][b, c])),
                     int(it[b]) - int(g[pre + "mp4_iter"][b, c])) for c in range(g[pre + "mp4_T"].shape[1])]
            d, dit = min(cand)
        else:
            d = max(rel_err_T(Tt[b], g[pre + "mp_T"][b]), rel_err(R2[b], g[pre + "mp_Rt2"][b]), rel_err(R3[b], g[pre + "mp_Rt3"][b]))
            dit = int(it[b]) - int(g[pre + "mp_iter"][b])
        assert dit == 0 and d < 1e-9, (model, b, d, dit)


# ---- FaugPapa's block kernel and its pseudo-inverse solver (csrc/gh_fp_kernel.h, csrc/wave_trid.h) ----------------------------------
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_emulated_tridiagonal_pinv_solver_matches_lapack(emu, variant):
    """wave_pinv_solve_trid (x = pinv(S) b under MATLAB's truncation, without eigenvectors) on KKT-like indefinite matrices of order
    20 .. 32 with a cluster of tiny eigenvalues under the tolerance: the three code paths (reduction in registers / LDS bursts, eigenpairs
    in the pivot / minor form) against numpy's eigh at 1e-11, same number of kept eigenvalues."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import proto_trid_pinv as P
    rng = np.random.default_rng(5 + variant)
    for n in (31, 32, 20):
        B = 2                                                                                   # (the GPU suite runs the solver on thousands of systems; this is the CPU smoke of each code path)
        Ms, tols, refs, keeps = [], [], [], []
        for b in range(B):
            S, rhs, tol = P.random_kkt(rng, n - 12, 12)
            Ms.append(np.hstack([S, rhs[:, None]])); tols.append(tol)
            lam, V = np.linalg.eigh(S)
            k = np.abs(lam) > tol
            refs.append(V[:, k] @ ((V[:, k].T @ rhs) / lam[k])); keeps.append(int(k.sum()))
        Maug = np.ascontiguousarray(np.stack(Ms)); tol = np.array(tols)
        sol = np.zeros((B, n)); kept = np.zeros(B, dtype=np.int32); fail = np.zeros(B, dtype=np.int32)
        emu.emu_trid_pinv(_p(Maug), _p(tol), ctypes.c_long(B), ctypes.c_int(n), _p(sol), _p(kept), _p(fail), ctypes.c_int(variant))
        assert not fail.any() and kept.tolist() == keeps
        for b in range(B):
            assert np.linalg.norm(sol[b] - refs[b]) <= 1e-11 * np.linalg.norm(refs[b]), (n, b)
        # the numpy twin of the kernel's algorithm (tools/proto_trid_pinv.py) agrees too
        x, nk = P.pinv_solve_sym(Ms[0][:, :n], Ms[0][:, n], tols[0])
        assert nk == keeps[0] and np.linalg.norm(x - refs[0]) <= 1e-10 * np.linalg.norm(refs[0])


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_emulated_tridiagonal_pinv_solver_reports_clustered_eigenvalues(emu, variant):
    """Two kept eigenvalues 1e-9 |T| apart: the eigenvectors come from independent iterations and need not be orthogonal inside the cluster.
    The solver must either still be right or say so (fail = 1: the caller takes the orthogonalising eigen-decomposition) -- never silently
    wrong.  Well separated spectra (the test above) must not trip the guard."""
    rng = np.random.default_rng(11 + variant)
    n, B = 24, 3
    Ms, tols, refs = [], [], []
    for b in range(B):
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam = np.concatenate([-np.geomspace(1e-3, 1.0, 6), np.geomspace(1e-2, 50.0, n - 6)])
        lam[10] = lam[9] * (1.0 + (1e-9 if b % 2 == 0 else 3e-13) * 50.0 / lam[9])          # a pair 1e-9 |T| (3e-13 |T|) apart
        S = (Q * lam) @ Q.T
        S = 0.5 * (S + S.T)
        rhs = rng.standard_normal(n)
        Ms.append(np.hstack([S, rhs[:, None]])); tols.append(1e-9)
        w, V = np.linalg.eigh(S)
        refs.append(V @ ((V.T @ rhs) / w))
    Maug = np.ascontiguousarray(np.stack(Ms)); tol = np.array(tols)
    sol = np.zeros((B, n)); kept = np.zeros(B, dtype=np.int32); fail = np.zeros(B, dtype=np.int32)
    emu.emu_trid_pinv(_p(Maug), _p(tol), ctypes.c_long(B), ctypes.c_int(n), _p(sol), _p(kept), _p(fail), ctypes.c_int(variant))
    assert np.all(kept == n)
    for b in range(B):
        err = np.linalg.norm(sol[b] - refs[b]) / np.linalg.norm(refs[b])
        assert fail[b] == 1 or err <= 1e-9, (b, err, fail[b])


def test_emulated_faugpapa_block_kernel_reproduces_the_extended_precision_iteration(emu, golden_dir):
    """k_fp_block (FaugPapaTFTPoseEstimation.m:48-153 on Gauss_Helmert.m:38-83, factored form) on an N = 12 scene of the 50-digit fixture:
    1e-9 and the same iteration count; nothing handed back to the generic kernel."""
    g = np.load(os.path.join(golden_dir, "gh_mp_faugpapa.npz"))
    C = np.ascontiguousarray(g["c0_Corresp"][:1]); CalM = g["c0_CalM"]                      # (one scene: 25 s on the lane emulator; the GPU suite runs all 48)
    B, N, _ = C.shape
    calm = calm_colmajor(CalM)
    Rt2 = np.zeros((B, 12)); Rt3 = np.zeros((B, 12)); T = np.zeros((B, 27)); it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
    handed = emu.emu_fp_pose(_p(C), _p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N), ctypes.c_int(0), _p(Rt2), _p(Rt3), _p(T), None, _p(it), _p(st))
    assert handed == 0 and np.all(st == 0)
    R2 = Rt2.reshape(B, 4, 3).transpose(0, 2, 1); R3 = Rt3.reshape(B, 4, 3).transpose(0, 2, 1); Tt = T.reshape(B, 3, 3, 3).transpose(0, 3, 2, 1)
    for b in range(B):
        assert int(it[b]) == int(g["c0_mp_iter"][b])
        d = max(rel_err_T(Tt[b], g["c0_mp_T"][b]), rel_err(R2[b], g["c0_mp_Rt2"][b]), rel_err(R3[b], g["c0_mp_Rt3"][b]))
        assert d < 1e-9, (b, d)


def test_emulated_picol_block_kernel_reproduces_the_extended_precision_iteration(emu, golden_dir):
    """k_pi_block<PiColModel> (PiColPoseEstimation.m:50-218; 5 x 5 weight blocks with two deflated near-null directions each,
    pi_wg_kernel.h::pinv_block_deflated2) on an N = 12 scene of the 50-digit fixture (six or seven iterations under every convention): 1e-9 and
    the same iteration count under one of the four sign conventions of the start (tests/helpers.py::kernel_null_convention)."""
    g = np.load(os.path.join(golden_dir, "gh_mp_picol.npz"))
    pick = [6]
    C = np.ascontiguousarray(g["c0_Corresp"][pick]); CalM = g["c0_CalM"]
    B, N, _ = C.shape
    calm = calm_colmajor(CalM)
    Rt2 = np.zeros((B, 12)); Rt3 = np.zeros((B, 12)); T = np.zeros((B, 27)); it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
    emu.emu_pi_wg_pose(ctypes.c_int(1), _p(C), _p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N), ctypes.c_int(0),
                       _p(Rt2), _p(Rt3), _p(T), None, _p(it), _p(st))
    assert np.all(st == 0)
    R2 = Rt2.reshape(B, 4, 3).transpose(0, 2, 1); R3 = Rt3.reshape(B, 4, 3).transpose(0, 2, 1); Tt = T.reshape(B, 3, 3, 3).transpose(0, 3, 2, 1)
    for b, s in enumerate(pick):
        T4, R24, R34, it4 = g["c0_mp4_T"][s], g["c0_mp4_Rt2"][s], g["c0_mp4_Rt3"][s], g["c0_mp4_iter"][s]
        d, dit, mit = min((max(rel_err_T(Tt[b], T4[c]), rel_err(R2[b], R24[c]), rel_err(R3[b], R34[c])), abs(int(it[b]) - int(it4[c])), int(it4[c]))
                          for c in range(4) if it4[c] >= 0)
        assert (mit > 1 or b > 0) and dit == 0 and d < 1e-9, (s, d, dit, mit)


@pytest.mark.parametrize("n", [9, 15, 16, 17, 27, 32])
def test_emulated_row_eigvec_dpp_form_matches_the_readlane_form_and_lapack(emu, n):
    """row_min_eigvec<n> (csrc/row_eig.h: every cross-lane operand through v_fmac_f64 row_newbcast, two matrix rows per position of a row of
    16 lanes, the four rows of 16 bit-identical replicas) against wave_min_eigvec_reg<n> (the v_readlane form, one row per lane) and
    numpy.linalg.eigh: V(:,end) of the reference's svd calls (linearTFT.m:64-67, :84, linearF.m:54-55).  Sizes in use: 9, 15, 27; 16 / 17 / 32
    are the layout's edge cases (a full lo half, a hi half of one row, both halves full).  The two forms perform the same operations per matrix
    entry in the same order -- only the reductions (norms) are summed in a different order -- so they agree to a few ulps and iterate equally."""
    rng = np.random.default_rng(n)
    B = 6
    Gs = []
    for b in range(B):
        A = rng.standard_normal((4 * n, n))
        U, sv, Vt = np.linalg.svd(A, full_matrices=False)
        sv[-1] = sv[-2] * (0.02 if b < 4 else 0.3)                                           # sigma_n / sigma_(n-1): the kernels' typical gap, and a slow one
        if b == 5: sv[-1] = 0.0                                                              # exactly singular (minimal samples)
        A = (U * sv) @ Vt
        Gs.append(A.T @ A)
    G = np.ascontiguousarray(np.stack(Gs))
    xr = np.zeros((B, n)); xl = np.zeros((B, n))
    ir = np.zeros(B, dtype=np.int32); il = np.zeros(B, dtype=np.int32); cr = np.zeros(B, dtype=np.int32); cl = np.zeros(B, dtype=np.int32)
    assert emu.emu_row_eig(_p(G), ctypes.c_long(B), ctypes.c_int(n), _p(xr), _p(xl), _p(ir), _p(il), _p(cr), _p(cl)) == 0
    for b in range(B):
        w, V = np.linalg.eigh(G[b])
        v = V[:, 0]
        assert cr[b] == 1 and cl[b] == 1, (b, ir[b], il[b])
        assert abs(np.linalg.norm(xr[b]) - 1.0) < 1e-14
        gap = (w[1] - w[0]) / w[-1]
        tol = 1e-15 / gap + 1e-13                                                            # eigenvector of a formed Gram matrix: eps |G| / gap
        assert min(np.abs(xr[b] - v).max(), np.abs(xr[b] + v).max()) < tol, (b, gap)
        assert min(np.abs(xr[b] - xl[b]).max(), np.abs(xr[b] + xl[b]).max()) < 1e-13 + tol
        assert abs(int(ir[b]) - int(il[b])) <= 1, (b, ir[b], il[b])


# ---- four triplets per wavefront (csrc/tft_rows_kernel.h) -----------------------------------------------------------------------
FLAG_DBG_ADAPTIVE = 32


@pytest.mark.parametrize("B,N,sigma", [(5, 12, 1.0), (4, 70, 0.0), (3, 130, 1.0), (6, 200, 1.0)])
def test_rows_kernel_matches_oracle(emu, B, N, sigma):
    """One triplet per row of 16 lanes: batches that do and do not fill the last wavefront (tail rows repeat the last triplet and store
    nothing), one to thirteen trips per data pass.  Production route (adaptive votes) and debug route (all four scores)."""
    C, CalM, _, _ = generate_scene_batch(B, N, noise=sigma, seed=100 + N)
    for debug in (False, True):
        out = run_linear_tft(emu, C, CalM, entry="emu_linear_tft_pose_rows", debug=debug)
        assert np.all(out["status"] == 0) and np.all(out["iter"] == 0)
        for b in range(B):
            R2, R3, Rec, T, _ = O.LinearTFTPoseEstimation(C[b].T.copy(), CalM)
            assert rel_err_T(out["T"][b], T) < 1e-9
            assert rel_err(out["R_t_2"][b], R2) < 1e-9 and rel_err(out["R_t_3"][b], R3) < 1e-9
            assert rel_err(out["Reconst"][b], Rec) < 1e-9
    # the t3 scale from the sums taken during the votes (production; debug[95] = 1: the picks are the main candidates) and from the separate
    # pass (all four scores evaluated, main candidate (R,t) whatever the data say: the pick is the other rotation in about half of the triplets)
    prod = run_linear_tft(emu, C, CalM, flags=FLAG_DBG_ADAPTIVE, entry="emu_linear_tft_pose_rows", debug=True)
    assert np.all(prod["debug"][:, 95] == 1.0)
    assert np.abs(prod["debug"][:, 68] / out["debug"][:, 68] - 1.0).max() < 1e-12
    if B >= 5:
        assert set(out["debug"][:, 95].tolist()) == {0.0, 1.0}
    nw = min(B, 2)                                           # (the one-triplet kernel as the reference: a wavefront per triplet, two are enough here)
    wave = run_linear_tft(emu, C[:nw], CalM)
    # the four cheirality scores of both essential matrices, up to the candidate order (the signs svd(E) gives U(:,3), V(:,3) permute the list)
    so, sw = out["debug"][:nw, 60:68].reshape(nw, 2, 4), wave["debug"][:, 60:68].reshape(nw, 2, 4)
    assert np.array_equal(np.sort(so, axis=2), np.sort(sw, axis=2))
    assert np.abs(out["debug"][:nw, 33:60] - wave["debug"][:, 33:60]).max() < 1e-12     # linearTFT's constrained tensor (normalised frame)


def test_rows_kernel_grid_stride_and_too_few(emu):
    C, CalM, _, _ = generate_scene_batch(9, 20, noise=1.0, seed=77)
    ref = run_linear_tft(emu, C, CalM, entry="emu_linear_tft_pose_rows", debug=False)
    emu.emu_set_grid_cap(1)                                  # one wavefront takes all three quads through the same LDS
    try:
        out = run_linear_tft(emu, C, CalM, entry="emu_linear_tft_pose_rows", debug=False)
    finally:
        emu.emu_set_grid_cap(0)
    for k in ("T", "R_t_2", "R_t_3", "Reconst", "iter", "status"):
        assert np.array_equal(ref[k], out[k], equal_nan=True), k
    C6, _, _, _ = generate_scene_batch(5, 6, noise=1.0, seed=1)
    few = run_linear_tft(emu, C6, CalM, entry="emu_linear_tft_pose_rows", debug=False)
    assert np.all(few["status"] == 1) and np.all(np.isnan(few["T"])) and np.all(np.isnan(few["R_t_3"]))


def test_rows_kernel_adaptive_votes_second_sweep(emu):
    """Three consistent correspondences of points BEHIND the three cameras, past the first 16: the candidate that looked unanimous after the
    first trip ends at 2 N - 12, so its partner's score decides and is evaluated in a second sweep.  Scores and poses must equal those of
    the one-triplet kernel, which evaluates every candidate over every correspondence."""
    from tft_vs_fund_amd.scenes import scene_cameras
    B, N = 5, 40
    C, CalM, _, _ = generate_scene_batch(B, N, noise=0.5, seed=5)
    _, Ps, _, _ = scene_cameras()
    C = C.copy()
    rng = np.random.default_rng(3)
    for b in range(B - 1):                                   # the last triplet stays clean: its row must not be disturbed by the others' second sweep
        X = np.array([0.0, -3000.0, 700.0]) + rng.uniform(-150, 150, size=(3, 3))
        for v in range(3):
            x = (Ps[v] @ np.c_[X, np.ones(3)].T).T
            assert np.all(x[:, 2] * np.sign(np.linalg.det(Ps[v][:, :3])) < 0)      # behind camera v
            C[b, 20:23, 2 * v:2 * v + 2] = x[:, :2] / x[:, 2:3]
    wave = run_linear_tft(emu, C, CalM, reconst=False)
    scores = wave["debug"][:, 60:68].reshape(B, 2, 4)
    assert np.all(np.abs(scores[:-1]).max(axis=2) == 2 * N - 12) and np.all(np.abs(scores[-1]).max(axis=1) == 2 * N)
    for flags, debug in ((0, False), (FLAG_DBG_ADAPTIVE, True)):
        out = run_linear_tft(emu, C, CalM, flags=flags, reconst=False, entry="emu_linear_tft_pose_rows", debug=debug)
        assert np.all(out["status"] == 0) and np.all(wave["status"] == 0)
        assert np.abs(out["R_t_2"] - wave["R_t_2"]).max() < 1e-9 and np.abs(out["R_t_3"] - wave["R_t_3"]).max() < 1e-9
    sweeps = out["debug"][:, 94].astype(int)
    assert np.all((sweeps[:4] & 15) == 2) and np.all(sweeps[:4] >= 48)     # the first wavefront made the second sweep, both pairs complete
    assert (sweeps[4] & 15) == 1 and sweeps[4] < 16                        # the second wavefront (the clean triplet alone) did not
    got = out["debug"][:, 60:68].reshape(B, 2, 4)
    assert np.array_equal(np.sort(got[:4], axis=2), np.sort(scores[:4], axis=2))
    # a score that was not needed is reported as 0: same pick
    assert np.array_equal(np.argmax(got[4], axis=1), np.argmax(scores[4], axis=1)) and np.all(np.abs(got[4]).max(axis=1) == 2 * N)


def test_linear_tft_kernels_under_address_and_ub_sanitizers():
    """tests/emu/emu_build.build(sanitize=True): the rows kernel, the one-triplet kernel and the exact kernel compiled with
    -fsanitize=address,undefined and run in a child process that has the sanitizer runtimes preloaded -- ragged last wavefront, grid-stride
    reuse of the LDS, a batch that sends triplets to the exact kernel.  Any out-of-bounds LDS index, use of an uninitialised overlay or signed
    overflow aborts the child."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import ctypes, sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np
        from emu import emu_build
        from test_emulated_kernels import run_linear_tft
        from tft_vs_fund_amd.scenes import generate_scene_batch
        lib = ctypes.CDLL(emu_build.build(sanitize=True))
        C, CalM, _, _ = generate_scene_batch(5, 20, noise=1.0, seed=3)
        a = run_linear_tft(lib, C, CalM, entry="emu_linear_tft_pose_rows", debug=False)
        lib.emu_set_grid_cap(1)
        b = run_linear_tft(lib, C, CalM, entry="emu_linear_tft_pose_rows", debug=True)
        lib.emu_set_grid_cap(0)
        w = run_linear_tft(lib, C, CalM)
        C7, _, _, _ = generate_scene_batch(2, 7, noise=1.0, seed=4)          # minimal samples: the exact kernel
        e = run_linear_tft(lib, C7, CalM)
        assert np.all(a["status"] == 0) and np.all(b["status"] == 0) and np.all(w["status"] == 0) and np.all(e["status"] == 0)
        assert np.abs(a["R_t_3"] - w["R_t_3"]).max() < 1e-9 and np.array_equal(a["T"], b["T"])
        print("sanitized run ok")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=emu_build.sanitizer_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "sanitized run ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("B,N,sigma", [(5, 12, 1.0), (3, 70, 0.0), (2, 130, 1.0)])
def test_rows_linear_f_kernel_matches_oracle(emu, B, N, sigma):
    """LinearFPoseEstimation with one triplet per row of 16 lanes (csrc/f_rows_kernel.h): moments of the centred coordinates scaled
    afterwards, 9 x 9 eigen-solves in the row layout, the pose tail shared with the trifocal rows kernel, T from the cameras."""
    C, CalM, _, _ = generate_scene_batch(B, N, noise=sigma, seed=100 + N)
    out = run_linear_tft(emu, C, CalM, entry="emu_linear_f_pose_rows", debug=False)
    assert np.all(out["status"] == 0) and np.all(out["iter"] == 0)
    for b in range(B):
        R2, R3, Rec, T, _ = O.LinearFPoseEstimation(C[b].T.copy(), CalM)
        assert rel_err_T(out["T"][b], T) < 1e-9
        assert rel_err(out["R_t_2"][b], R2) < 1e-9 and rel_err(out["R_t_3"][b], R3) < 1e-9 and rel_err(out["Reconst"][b], Rec) < 1e-9
    C7, _, _, _ = generate_scene_batch(3, 7, noise=1.0, seed=2)              # linearF.m:35-37
    few = run_linear_tft(emu, C7, CalM, entry="emu_linear_f_pose_rows", debug=False)
    assert np.all(few["status"] == 1) and np.all(np.isnan(few["T"]))


@pytest.mark.parametrize("B,N,sigma", [(6, 7, 1.0), (5, 9, 2.0), (2, 30, 1.0)])
def test_rows_exact_kernel_matches_oracle(emu, B, N, sigma):
    """The exact tiers with one triplet per row of 16 lanes (csrc/tft_rows_exact_kernel.h, rows_qr.h): Householder QR of the explicit 4N x 27
    system with the owner of a column publishing its chunk through LDS, inverse iteration straight from the packed R, the 27 x 15 re-solve from
    R * Up, certified null vectors, all four votes with the exact re-score behind them -- minimal samples (one chunk), two chunks, and N = 30
    (five chunks, two trips per data pass).  Agreement with the oracle and with the one-triplet exact kernel."""
    C, CalM, _, _ = generate_scene_batch(B, N, noise=sigma, seed=100 + N)
    out = run_linear_tft(emu, C, CalM, entry="emu_linear_tft_pose_rows_exact", debug=True)
    ref = run_linear_tft(emu, C, CalM, flags=FLAG_JACOBI)
    assert np.all(out["status"] == 0) and np.all(ref["status"] == 0)
    assert np.all(out["debug"][:, 69] >= 20000) and np.all(out["debug"][:, 70] >= 20000)          # this kernel's stamp: nothing was handed on
    for b in range(B):
        R2, R3, Rec, T, _ = O.LinearTFTPoseEstimation(C[b].T.copy(), CalM)
        assert rel_err_T(out["T"][b], T) < 1e-9 and rel_err(out["R_t_2"][b], R2) < 1e-9 and rel_err(out["R_t_3"][b], R3) < 1e-9
        assert rel_err(out["Reconst"][b], Rec) < 1e-9
        assert rel_err_T(out["T"][b], ref["T"][b]) < 1e-11 and rel_err(out["R_t_3"][b], ref["R_t_3"][b]) < 1e-11
    so, sw = out["debug"][:, 60:68].reshape(B, 2, 4), ref["debug"][:, 60:68].reshape(B, 2, 4)
    assert np.array_equal(np.sort(so, axis=2), np.sort(sw, axis=2))


def test_rows_qr_and_inverse_iteration_match_lapack(emu):
    """rows_qr.h on random tall systems, four per wavefront: 28 x 27 and 44 x 27 (one and two chunks), 27 x 15, N x 9; well separated and
    close smallest singular values (ratio 0.02 / 0.7); a rank-deficient system (8 x 9) returns a null vector."""
    rng = np.random.default_rng(1)
    for n, rows in ((27, 28), (27, 44), (15, 27), (9, 8), (9, 40)):
        B = 6
        A = rng.standard_normal((B, rows, n))
        for b in range(B):
            U, s, Vt = np.linalg.svd(A[b], full_matrices=False)
            if s.size == n:
                s[-1] = s[-2] * (0.02 if b % 2 == 0 else 0.7)
            A[b] = (U * s) @ Vt
        A = np.ascontiguousarray(A)
        x = np.zeros((B, n)); its = np.zeros(B, dtype=np.int32); conv = np.zeros(B, dtype=np.int32)
        assert emu.emu_rows_qr(_p(A), ctypes.c_long(B), ctypes.c_int(rows), ctypes.c_int(n), _p(x), _p(its), _p(conv), None) == 0
        assert np.all(conv == 1)
        for b in range(B):
            if rows < n:
                assert np.linalg.norm(A[b] @ x[b]) < 1e-12 and abs(np.linalg.norm(x[b]) - 1) < 1e-12
            else:
                v = np.linalg.svd(A[b])[2][-1]
                assert min(np.abs(x[b] - v).max(), np.abs(x[b] + v).max()) < 1e-11
                assert its[b] <= (8 if b % 2 == 0 else 60)


@pytest.mark.parametrize("B,N,sigma", [(6, 8, 1.0), (5, 9, 2.0), (3, 20, 0.5)])
def test_rows_exact_linear_f_kernel_matches_oracle(emu, B, N, sigma):
    """LinearFPoseEstimation's exact tiers with one triplet per row of 16 lanes (csrc/f_rows_kernel.h::k_linear_f_pose_rows_exact): QR of the explicit
    N x 9 systems, linearF's own normalisation computed from the normalised points, certified null vectors and votes."""
    C, CalM, _, _ = generate_scene_batch(B, N, noise=sigma, seed=100 + N)
    out = run_linear_tft(emu, C, CalM, entry="emu_linear_f_pose_rows_exact", debug=True)
    ref = run_linear_tft(emu, C, CalM, flags=FLAG_JACOBI, entry="emu_linear_f_pose")
    assert np.all(out["status"] == 0) and np.all(out["debug"][:, 69:71] >= 20000)
    for b in range(B):
        R2, R3, Rec, T, _ = O.LinearFPoseEstimation(C[b].T.copy(), CalM)
        assert rel_err_T(out["T"][b], T) < 1e-9 and rel_err(out["R_t_2"][b], R2) < 1e-9 and rel_err(out["R_t_3"][b], R3) < 1e-9
        assert rel_err(out["Reconst"][b], Rec) < 1e-9
        assert rel_err_T(out["T"][b], ref["T"][b]) < 1e-10 and rel_err(out["R_t_3"][b], ref["R_t_3"][b]) < 1e-10


def _run_gh_wg(emu, model, C, CalM, reconst=True):
    B, N, _ = C.shape
    calm = calm_colmajor(CalM)
    Rt2 = np.zeros((B, 12)); Rt3 = np.zeros((B, 12)); T = np.zeros((B, 27)); Rec = np.zeros((B, N, 3)) if reconst else None
    it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
    emu.emu_gh_wg_pose(ctypes.c_int(model), _p(C), _p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N), ctypes.c_int(0),
                       _p(Rt2), _p(Rt3), _p(T), _p(Rec), _p(it), _p(st))
    return dict(Rt2=Rt2, Rt3=Rt3, T=T, Rec=Rec, iter=it, status=st)


@pytest.mark.parametrize("neighbour", ["nan"])          # ("collinear", a row the exact tiers redo, runs on the GPU: tests/test_gpu_rows.py)
def test_gh_finish_rows_keeps_the_rows_of_a_wavefront_independent(emu, neighbour):
    """k_gh_finish_rows (gh_rows_kernel.h) with Reconst requested: a failed triplet (status > 0) keeps its all-NaN outputs -- the pose tail that
    its row still runs on a dummy tensor must not store -- and what a triplet gets does not depend on what the other rows of its wavefront hold:
    neither on a dead row nor on a row the exact tiers have to redo (collinear camera centres: the fast null vectors report, the whole
    wavefront goes through the tail a second time and only that row may store)."""
    B, N = 3, 12
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=77)
    ref = _run_gh_wg(emu, 0, C, CalM)
    assert np.all(ref["status"] == 0)
    Cx = C.copy()
    if neighbour == "nan":
        Cx[1, 5, 2] = np.nan
    else:
        Cc, _, _, _ = generate_scene_batch(1, N, noise=1.0, seed=312, angle=180)
        Cx[1] = Cc[0]
    out = _run_gh_wg(emu, 0, Cx, CalM)
    if neighbour == "nan":
        assert out["status"][1] == 2
        assert np.all(np.isnan(out["Rec"][1])) and np.all(np.isnan(out["T"][1])) and np.all(np.isnan(out["Rt2"][1])) and np.all(np.isnan(out["Rt3"][1]))
    else:
        assert out["status"][1] == 0 and np.all(np.isfinite(out["Rec"][1]))
    for b in (0, 2):
        assert out["status"][b] == 0 and out["iter"][b] == ref["iter"][b]
        for k in ("Rec", "T", "Rt2", "Rt3"):
            assert np.array_equal(out[k][b], ref[k][b]), (neighbour, b, k)


FLAG_PRE, FLAG_PRE_GLOBAL = 8192, 16384


@pytest.mark.parametrize("B,N,sigma,extra", [(5, 12, 1.0, 0), (2, 130, 1.0, FLAG_PRE_GLOBAL), (5, 200, 1.0, 0)])
def test_moments_kernel_feeds_the_rows_kernel(emu, B, N, sigma, extra):
    """k_tft_moments (tft_moments_kernel.h: one triplet per wavefront, correspondences parked in LDS -- or, extra = FLAG_PRE_GLOBAL, re-read from
    global memory as for N beyond the LDS budget) + k_linear_tft_pose_rows<true> against the oracle at 1e-9 and against the fused rows kernel,
    whose two data passes it replaces, to rounding (same arithmetic per correspondence, sums in a different order); the normalisations and the
    linear tensor of the debug record to 1e-12.  Also one block taking all triplets through the same LDS copy (grid capped at 1)."""
    C, CalM, _, _ = generate_scene_batch(B, N, noise=sigma, seed=900 + N)
    fused = run_linear_tft(emu, C, CalM, entry="emu_linear_tft_pose_rows", debug=True)
    out = run_linear_tft(emu, C, CalM, flags=FLAG_PRE | extra, entry="emu_linear_tft_pose_rows", debug=True)
    assert np.all(out["status"] == 0) and np.all(out["iter"] == 0)
    assert np.abs(out["debug"][:, 71:80] / fused["debug"][:, 71:80] - 1.0).max() < 1e-13        # Normalize2Ddata: s, -s cx, -s cy per view
    assert np.abs(out["debug"][:, 33:60] - fused["debug"][:, 33:60]).max() < 1e-11              # linearTFT's constrained tensor
    for b in range(B):
        R2, R3, Rec, T, _ = O.LinearTFTPoseEstimation(C[b].T.copy(), CalM)
        assert rel_err_T(out["T"][b], T) < 1e-9 and rel_err(out["R_t_2"][b], R2) < 1e-9 and rel_err(out["R_t_3"][b], R3) < 1e-9
        assert rel_err(out["Reconst"][b], Rec) < 1e-9
        assert rel_err_T(out["T"][b], fused["T"][b]) < 1e-11 and rel_err(out["R_t_3"][b], fused["R_t_3"][b]) < 1e-11
    prod = run_linear_tft(emu, C, CalM, flags=FLAG_PRE | extra, entry="emu_linear_tft_pose_rows", debug=False)   # (production votes: the scale sums ride along)
    emu.emu_set_grid_cap(1)
    try:
        one = run_linear_tft(emu, C, CalM, flags=FLAG_PRE | extra, entry="emu_linear_tft_pose_rows", debug=False)
    finally:
        emu.emu_set_grid_cap(0)
    for k in ("T", "R_t_2", "R_t_3", "Reconst", "status"):
        assert np.array_equal(one[k], prod[k], equal_nan=True), k


@pytest.mark.parametrize("entry,ref_entry,n", [("emu_linear_tft_pose_rows_exact", "emu_linear_tft_pose", 7), ("emu_linear_f_pose_rows_exact", "emu_linear_f_pose", 8)])
def test_rows_exact_kernels_score_contaminated_minimal_samples_like_the_one_triplet_kernel(emu, entry, ref_entry, n):
    """Minimal samples of a scene with gross outliers (config 4): most fast votes are not certified, so the exact re-score runs -- for N <= 8 with the two
    candidates of an essential matrix side by side in a row's sixteen positions (rows_vote_exact_pair).  All eight scores equal the one-triplet exact
    kernel's, and so do the poses."""
    Ns, B = 60, 4
    Cs, CalM, _, _ = generate_scene_batch(1, Ns, noise=0.5, seed=77)
    scene = Cs[0].copy()
    rng = np.random.default_rng(5)
    bad = rng.choice(Ns, Ns // 4, replace=False)
    scene[bad, 2:6] += rng.uniform(20, 80, size=(bad.size, 4))
    C = np.ascontiguousarray(np.stack([scene[rng.choice(Ns, n, replace=False)] for _ in range(B)]))
    out = run_linear_tft(emu, C, CalM, entry=entry, debug=True)
    ref = run_linear_tft(emu, C, CalM, flags=FLAG_JACOBI, entry=ref_entry)
    live = (out["status"] == 0) & (ref["status"] == 0)
    assert live.sum() >= B - 1 and np.array_equal(out["status"] == 3, ref["status"] == 3)
    so, sw = out["debug"][:, 60:68].reshape(B, 2, 4), ref["debug"][:, 60:68].reshape(B, 2, 4)
    assert np.array_equal(so[live], sw[live])
    for b in np.nonzero(live)[0]:
        assert rel_err(out["R_t_2"][b], ref["R_t_2"][b]) < 1e-8 and rel_err(out["R_t_3"][b], ref["R_t_3"][b]) < 1e-8
