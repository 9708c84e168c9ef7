"""
What pins the oracle (oracle/tft_oracle.py), since the MATLAB reference ships
no tests or golden vectors and cannot run here ("parity unpinned"):

  * known answers that follow from the reference's own ground truth
    (generateSyntheticScene.m:113): noise-free scenes recover R_t0;
  * structural facts of linearTFT (rank(E) = 15; cheirality votes are
    {+2N, -2N, 0, 0} on clean data);
  * deterministic EPFL inlier counts under the 1-px rule of
    experiments_real.m:93-99 (obtained there with an
    independent scratch restatement);
  * the committed golden fixtures reproduce bit-for-bit-ish from the oracle
    (guards against silent drift of the oracle or of LAPACK).
"""
import os

import numpy as np
import pytest

from oracle import tft_oracle as O
from tft_vs_fund_amd.scenes import generate_scene_batch
from helpers import rel_err_T, rel_err, golden_cases

METHODS = ["LinearTFTPoseEstimation", "LinearFPoseEstimation", "ResslTFTPoseEstimation",
           "NordbergTFTPoseEstimation", "FaugPapaTFTPoseEstimation", "OptimFPoseEstimation", "PiPoseEstimation"]


@pytest.mark.parametrize("method", METHODS)
def test_noise_free_scene_recovers_ground_truth(method):
    C, CalM, Rt0, X = generate_scene_batch(2, 40, noise=0.0, seed=3)
    for b in range(2):
        R2, R3, Rec, T, it = getattr(O, method)(C[b].T.copy(), CalM)[:5]
        s = np.linalg.norm(Rt0[0][:, 3])                      # reference fixes |t2| = 1
        assert np.max(np.abs(R2[:, :3] - Rt0[0][:, :3])) < 1e-9
        assert np.max(np.abs(R3[:, :3] - Rt0[1][:, :3])) < 1e-9
        assert np.max(np.abs(R2[:, 3] - Rt0[0][:, 3] / s)) < 1e-9
        assert np.max(np.abs(R3[:, 3] - Rt0[1][:, 3] / s)) < 1e-9
        # Reconst is in camera-1 coordinates at scale 1/s
        P = [CalM[0:3] @ np.eye(3, 4), CalM[3:6] @ R2, CalM[6:9] @ R3]
        assert O.ReprError(P, C[b].T.copy(), Rec) < 1e-8


def test_rank_E_is_15_and_votes_are_clean():
    C, CalM, Rt0, X = generate_scene_batch(3, 30, noise=0.0, seed=5)
    for b in range(3):
        Cb = C[b].T.copy()
        x1, N1 = O.Normalize2Ddata(Cb[0:2]); x2, N2 = O.Normalize2Ddata(Cb[2:4]); x3, N3 = O.Normalize2Ddata(Cb[4:6])
        T, P1, P2, P3, d = O.linearTFT(x1, x2, x3, return_debug=True)
        assert d["rankE"] == 15
        Tn = O.transform_TFT(T, N1, N2, N3, 1)
        _, _, dd = O.R_t_from_TFT(Tn, CalM, Cb, return_debug=True)
        for v in (dd["votes2"], dd["votes3"]):
            assert sorted(v) == [-60, 0, 0, 60]
        # T is a valid trifocal tensor of the returned cameras
        assert rel_err_T(O.TFT_from_P(P1, P2, P3), T / np.linalg.norm(T)) < 1e-9


def test_normalize2ddata_properties():
    rng = np.random.default_rng(0)
    p = rng.normal(size=(2, 50)) * 300 + 700
    q, Nm = O.Normalize2Ddata(p)
    assert q.shape == (2, 50)                                  # quirk: 2xN, not homogeneous
    assert np.allclose(q.mean(axis=1), 0, atol=1e-12)
    assert abs(np.mean(np.sqrt(np.sum(q ** 2, axis=0))) - np.sqrt(2)) < 1e-12


def test_linearF_needs_8_points():
    C, CalM, _, _ = generate_scene_batch(1, 7, noise=1.0, seed=1)
    with pytest.raises(ValueError):
        O.linearF(C[0].T[0:2], C[0].T[2:4])


EPFL_INLIERS = [1360, 1253, 1250, 85, 1222, 1037, 920, 39]
EPFL_TOTAL = [1400, 1306, 1302, 95, 1482, 1267, 1117, 97]


def test_epfl_inlier_counts(golden_dir):
    g = np.load(os.path.join(golden_dir, "epfl.npz"))
    assert int(g["count"]) == 8
    for n in range(8):
        pre = "t%d_" % n
        Corresp, CalM, Rt0 = g[pre + "Corresp_all"], g[pre + "CalM"], g[pre + "Rt0"]
        assert Corresp.shape[1] == EPFL_TOTAL[n]
        Ps = [CalM[0:3] @ np.eye(3, 4), CalM[3:6] @ Rt0[0], CalM[6:9] @ Rt0[1]]
        Rec0 = O.triangulation3D(Ps, Corresp)
        Rec0 = Rec0[0:3] / Rec0[3:4]
        resid = O.project3Dpoints(Rec0, Ps) - Corresp
        n_in = int(np.sum(np.sum(np.abs(resid) > 1.0, axis=0) == 0))       # experiments_real.m:98
        assert n_in == EPFL_INLIERS[n] == int(g[pre + "n_inliers"])


def test_golden_linear_reproduces(golden_dir):
    g = np.load(os.path.join(golden_dir, "synthetic_linear.npz"))
    for ci, pre in golden_cases(g):
        C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
        b = 0
        R2, R3, Rec, T, it = O.LinearTFTPoseEstimation(C[b].T.copy(), CalM)
        assert rel_err_T(T, g[pre + "tft_T"][b]) < 1e-10
        assert rel_err(R2, g[pre + "tft_Rt2"][b]) < 1e-10 and rel_err(R3, g[pre + "tft_Rt3"][b]) < 1e-10
        if C.shape[1] >= 8:
            R2, R3, Rec, T, it = O.LinearFPoseEstimation(C[b].T.copy(), CalM)
            assert rel_err_T(T, g[pre + "f_T"][b]) < 1e-10
            assert rel_err(R2, g[pre + "f_Rt2"][b]) < 1e-10 and rel_err(R3, g[pre + "f_Rt3"][b]) < 1e-10


def test_golden_optimf_reproduces_and_refines(golden_dir):
    """optimf.npz reproduces from the oracle; on noisy scenes optimF's Gauss-Helmert step lowers the
    reprojection error of its linearF start (what the refinement is for, OptimFPoseEstimation.m:48-49)."""
    g = np.load(os.path.join(golden_dir, "optimf.npz"))
    for ci, pre in golden_cases(g):
        C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
        if C.shape[1] > 200:
            continue
        Cb = C[0].T.copy()
        R2, R3, Rec, T, it = O.OptimFPoseEstimation(Cb, CalM)
        assert it == int(g[pre + "optimf_iter"][0])
        assert rel_err_T(T, g[pre + "optimf_T"][0]) < 1e-9 and rel_err(R3, g[pre + "optimf_Rt3"][0]) < 1e-9
        if g[pre + "meta"][1] > 0 and C.shape[1] >= 50:
            L2, L3, LRec, _, _ = O.LinearFPoseEstimation(Cb, CalM)
            P = lambda Ra, Rb: [CalM[0:3] @ np.eye(3, 4), CalM[3:6] @ Ra, CalM[6:9] @ Rb]
            assert O.ReprError(P(R2, R3), Cb) < O.ReprError(P(L2, L3), Cb)


def test_optimF_two_view_properties():
    """optimF.m:34-109: the refined F is rank 2, satisfies the epipolar constraint exactly on noise-free
    points after one Gauss-Helmert iteration, and needs 8 correspondences (:36-38)."""
    C, CalM, Rt0, _ = generate_scene_batch(1, 30, noise=0.0, seed=9)
    Cb = C[0].T.copy()
    F, it = O.optimF(Cb[0:2], Cb[2:4])
    assert it == 1 and np.linalg.svd(F, compute_uv=False)[2] < 1e-12 * np.linalg.norm(F)
    x1 = np.vstack([Cb[0:2], np.ones(30)]); x2 = np.vstack([Cb[2:4], np.ones(30)])
    assert np.max(np.abs(np.einsum("in,ij,jn->n", x2, F, x1))) < 1e-9 * np.linalg.norm(F) * 1e6
    with pytest.raises(ValueError):
        O.optimF(Cb[0:2, :7], Cb[2:4, :7])


def test_picol_noise_free_collinear_scene_recovers_ground_truth():
    """PiColPoseEstimation is the variant for collinear camera centres (generateSyntheticScene.m:45-50, angle = 180)."""
    C, CalM, Rt0, X = generate_scene_batch(2, 40, noise=0.0, seed=4, angle=180)
    s = np.linalg.norm(Rt0[0][:, 3])
    for b in range(2):
        R2, R3, Rec, T, it = O.PiColPoseEstimation(C[b].T.copy(), CalM)
        assert np.max(np.abs(R2[:, :3] - Rt0[0][:, :3])) < 1e-8 and np.max(np.abs(R3[:, :3] - Rt0[1][:, :3])) < 1e-8
        assert np.max(np.abs(R3[:, 3] - Rt0[1][:, 3] / s)) < 1e-7


@pytest.mark.parametrize("name,angle", [("Pi", None), ("PiCol", 180)])
def test_pi_callbacks_are_consistent(name, angle):
    """The GH callbacks of the Pi methods: f vanishes at the exact reprojections of a noise-free scene, B and C are
    the true Jacobians (finite differences), and so is A -- except PiColPoseEstimation.m:186, whose sign for
    dA(ind2+4)/dpi21 the restatement keeps as the reference has it."""
    func = O._pi_constraintsGH if name == "Pi" else O._picol_constraintsGH
    fn = O.PiPoseEstimation if name == "Pi" else O.PiColPoseEstimation
    E = 4 if name == "Pi" else 5
    C, CalM, _, _ = generate_scene_batch(1, 15, noise=0.0, seed=3, angle=angle)
    p0, xe = fn(C[0].T.copy(), CalM, init_only=True)
    f = func(xe, p0)[0]
    assert np.abs(f).max() < 1e-12
    rng = np.random.default_rng(0)
    x = xe + 0.01 * rng.standard_normal(xe.shape); p = p0 + 0.01 * rng.standard_normal(27)
    f, g, A, B, Cm, D = func(x, p)
    h = 1e-6
    An = np.zeros_like(A); Cn = np.zeros_like(Cm); Bn = np.zeros_like(B)
    for k in range(27):
        dp = np.zeros(27); dp[k] = h
        fp, gp = func(x, p + dp)[:2]; fm, gm = func(x, p - dp)[:2]
        An[:, k] = (fp - fm) / (2 * h); Cn[:, k] = (gp - gm) / (2 * h)
    for k in range(x.size):
        dx = np.zeros(x.size); dx[k] = h
        Bn[:, k] = (func(x + dx, p)[0] - func(x - dx, p)[0]) / (2 * h)
    assert np.abs(B - Bn).max() < 1e-7 and np.abs(Cm - Cn).max() < 1e-7
    bad = {(int(r % E), int(c)) for r, c in np.argwhere(np.abs(A - An) > 1e-6)}
    assert bad == (set() if name == "Pi" else {(3, 0), (3, 1), (3, 2)})


def test_golden_pi_reproduces(golden_dir):
    g = np.load(os.path.join(golden_dir, "pi.npz"))
    for key, fn, prefix in (("pi", O.PiPoseEstimation, "p"), ("picol", O.PiColPoseEstimation, "q")):
        for ci, pre in golden_cases(g, prefix):
            C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
            if C.shape[1] > 100:
                continue
            R2, R3, Rec, T, it = fn(C[0].T.copy(), CalM)
            assert it == int(g[pre + key + "_iter"][0])
            assert rel_err_T(T, g[pre + key + "_T"][0]) < 1e-9 and rel_err(R3, g[pre + key + "_Rt3"][0]) < 1e-9


def test_golden_epfl_linear_quality(golden_dir):
    """On real data the linear TFT pose is within a few degrees of the EPFL ground truth."""
    g = np.load(os.path.join(golden_dir, "epfl.npz"))
    for n in range(8):
        pre = "t%d_" % n
        for m in ("tft", "f", "ressl"):
            r2, t2 = O.AngError(g[pre + "Rt0"][0], g[pre + m + "_Rt2"])
            r3, t3 = O.AngError(g[pre + "Rt0"][1], g[pre + m + "_Rt3"])
            # AngError does not clamp acos: a NaN means the argument drifted above 1, i.e. ~0 degrees
            assert np.nan_to_num(max(r2, r3)) < 1.0 and np.nan_to_num(max(t2, t3)) < 2.0
            assert float(g[pre + m + "_repr_all"]) < 6.5


def test_epfl_ground_truth_pins_seven_methods(golden_dir):
    """Known answers the reference's own data implies (experiments_real.m:86-91): every method the reference runs on real data
    (methods_to_test = [1:5,7:8], experiments_real.m:63 -- PiCol is for collinear centres and is excluded there) recovers the
    `.camera` ground-truth poses of every fixture triplet from its 100-inlier sample: rotations within 2 degrees, translation
    directions within 2.5 degrees (observed: <= 1.51 / 1.77).  An independent pin of the restatement -- still not a reference run:
    the MATLAB code itself cannot be executed here."""
    g = np.load(os.path.join(golden_dir, "epfl.npz"))
    meths = ["LinearTFTPoseEstimation", "ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "FaugPapaTFTPoseEstimation",
             "PiPoseEstimation", "LinearFPoseEstimation", "OptimFPoseEstimation"]
    for n in range(int(g["count"])):
        pre = "t%d_" % n
        Cs, CalM, Rt0 = g[pre + "sample"], g[pre + "CalM"], g[pre + "Rt0"]
        for m in meths:
            out = getattr(O, m)(Cs.copy(), CalM)
            r2, t2 = O.AngError(Rt0[0], out[0])
            r3, t3 = O.AngError(Rt0[1], out[1])
            # AngError does not clamp acos: a NaN means the argument drifted above 1, i.e. ~0 degrees
            assert np.nan_to_num(max(r2, r3)) < 2.0 and np.nan_to_num(max(t2, t3)) < 2.5, (str(g[pre + "name"]), m, r2, r3, t2, t3)


def _epfl_list_samples(golden_dir):
    """The reference's own real-data lists (experiments_real.m:31-35,78: the first 70 triplets of fountain-P11 and the first 50 of Herz-Jesu-P8 by
    match count) from the inputs-only fixture: inliers by the 1-px rule against the `.camera` ground truth (:93-99, oracle triangulation),
    a deterministic 100-inlier sample each (Philox keyed by the triplet's position in its list)."""
    from tft_vs_fund_amd import experiments as X
    out = []
    for dataset, n_trip in (("fountain", 70), ("herzjesu", 50)):
        for ti, tr in enumerate(X.load_epfl_all(os.path.join(golden_dir, "epfl_all.npz"), dataset, n_trip)):
            C, CalM, Rt0 = tr["Corresp"], tr["CalM"], tr["R_t0"]
            P0 = [CalM[0:3] @ np.eye(3, 4), CalM[3:6] @ Rt0[0], CalM[6:9] @ Rt0[1]]
            Xh = O.triangulation3D(P0, C.copy())
            Xh = Xh / Xh[3:4]
            proj = np.concatenate([(P @ Xh)[0:2] / (P @ Xh)[2:3] for P in P0])
            Ci = C[:, np.sum(np.abs(proj - C) > 1.0, axis=0) == 0]
            n = min(100, Ci.shape[1])
            if n < 8:
                continue
            rng = np.random.Generator(np.random.Philox(key=[7, ti]))
            out.append((dataset, ti, Ci[:, np.sort(rng.choice(Ci.shape[1], size=n, replace=False))], CalM, Rt0))
    return out


def _gt_err(Rt0, out):
    r2, t2 = O.AngError(Rt0[0], out[0])
    r3, t3 = O.AngError(Rt0[1], out[1])
    return np.nan_to_num(max(r2, r3)), np.nan_to_num(max(t2, t3))       # AngError does not clamp acos: NaN = ~0 degrees


def test_epfl_ground_truth_pins_all_120_triplets_of_the_references_lists(golden_dir):
    """The only pin the reference itself holds for pose VALUES is the `.camera` ground truth of its two real-data sets (experiments_real.m:86-91).
    The oracle's linear methods on ALL 120 triplets of the reference's lists, and the Gauss-Helmert family on every sixth, against it: medians
    and maxima per data set (observed: LinearTFT rot median 0.13 / 0.41 deg, max 1.5 / 1.8; translation direction median 0.44 / 0.72, max 3.8 / 4.7)."""
    samples = _epfl_list_samples(golden_dir)
    assert len(samples) >= 118
    errs = {}
    for k, (dataset, ti, S, CalM, Rt0) in enumerate(samples):
        meths = ["LinearTFTPoseEstimation", "LinearFPoseEstimation"]
        if k % 6 == 0 and S.shape[1] >= 12:
            meths += ["ResslTFTPoseEstimation", "OptimFPoseEstimation", "PiPoseEstimation"]
        for m in meths:
            errs.setdefault((dataset, m), []).append(_gt_err(Rt0, getattr(O, m)(S.copy(), CalM)))
    for (dataset, m), v in errs.items():
        a = np.array(v)
        assert np.median(a[:, 0]) < (0.3 if dataset == "fountain" else 0.8), (dataset, m, np.median(a[:, 0]))
        assert np.median(a[:, 1]) < (0.9 if dataset == "fountain" else 1.5), (dataset, m, np.median(a[:, 1]))
        assert a[:, 0].max() < 3.0 and a[:, 1].max() < 7.0, (dataset, m, a.max(axis=0))


def test_mp_callback_is_an_independent_restatement_of_ressl():
    """oracle/gh_mp_oracle.py restates Ressl's Gauss-Helmert callback index by index (no Kronecker products); converted to
    fp64 it must equal the numpy oracle's f, A, B, and one Gauss-Helmert run in 50-digit arithmetic must land within the
    reference's own fp64 noise (~1e-5 at N = 12) of the LAPACK evaluation."""
    from oracle import gh_mp_oracle as G
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, _, _ = generate_scene_batch(1, 9, noise=1.0, seed=3)
    Cb = C[0].T.copy()
    x, x_est, p0, Ind, normals = G.ressl_start(Cb, CalM)
    f, g_, A, B, Cc, _ = O._ressl_constraintsGH(x_est, p0, Ind)
    S, e21, e31, mn, T, Ind2 = G._ressl_unpack(G.to_mp(p0), Ind)
    D = G._ressl_D(S, e21, e31, mn, Ind2)
    xi = G.to_mp(x_est)
    for i in range(9):
        fi, Ap, Bi = G._blocks(xi[6 * i:6 * i + 6], T)
        assert np.abs(G.to_float(fi) - f[4 * i:4 * i + 4]).max() < 1e-14
        assert np.abs(G.to_float(Ap.dot(D)) - A[4 * i:4 * i + 4]).max() < 1e-13 * np.abs(A).max()
        assert np.abs(G.to_float(Bi) - B[4 * i:4 * i + 4, 6 * i:6 * i + 6]).max() < 1e-13 * np.abs(B).max()
    p_mp, _, it, reason = G.gauss_helmert_ressl_mp(x, x_est, p0, Ind)
    func = lambda a, b, c: O._ressl_constraintsGH(a, b, Ind)
    _, p_np, _, it_np, _ = O.Gauss_Helmert(func, x_est, p0, np.zeros(0), x, None, True)
    assert abs(it - it_np) <= 3 and np.abs(p_mp - p_np).max() < 1e-3


def test_mp_callback_is_an_independent_restatement_of_nordberg():
    """The same for Nordberg's callback (NordbergTFTPoseEstimation.m:128-222: three Rodrigues rotations and their derivatives, ten
    sparse tensor entries): g, C, A = Ap J of the 50-digit restatement equal the numpy oracle's at fp64 inputs, and one 50-digit
    Gauss-Helmert run lands within the LAPACK evaluation's own noise of it."""
    from oracle import gh_mp_oracle as G
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, _, _ = generate_scene_batch(1, 9, noise=1.0, seed=4)
    Cb = C[0].T.copy()
    x, x_est, p0, normals = G.nordberg_start(Cb, CalM)
    f, g_, A, B, Cc, _ = O._nordberg_constrGH(x_est, p0)
    T, J, gm, Cm = G.nordberg_model(G.to_mp(p0))
    assert abs(float(gm[0]) - g_[0]) < 1e-15 and np.abs(G.to_float(Cm) - Cc).max() < 1e-15
    xi = G.to_mp(x_est)
    for i in range(9):
        fi, Ap, Bi = G._blocks(xi[6 * i:6 * i + 6], T)
        assert np.abs(G.to_float(fi) - f[4 * i:4 * i + 4]).max() < 1e-14
        assert np.abs(G.to_float(Ap.dot(J)) - A[4 * i:4 * i + 4]).max() < 1e-13 * np.abs(A).max()
        assert np.abs(G.to_float(Bi) - B[4 * i:4 * i + 4, 6 * i:6 * i + 6]).max() < 1e-13 * np.abs(B).max()
    p_mp, _, it, reason = G.gauss_helmert_mp(x, x_est, p0, G.nordberg_model, 19, 1)
    func = lambda a, b, c: O._nordberg_constrGH(a, b)
    _, p_np, _, it_np, _ = O.Gauss_Helmert(func, x_est, p0, np.zeros(0), x, None, True)
    assert abs(it - it_np) <= 3 and np.abs(p_mp - p_np).max() < 1e-3


def test_mp_callback_is_an_independent_restatement_of_faugpapa():
    """... and for Faugeras-Papadopoulo's callback (FaugPapaTFTPoseEstimation.m:87-159: determinants and signed cofactors of 3 x 3
    stacks of tensor entries): g and C of the 50-digit restatement equal the numpy oracle's at fp64 inputs."""
    from oracle import gh_mp_oracle as G
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, _, _ = generate_scene_batch(1, 9, noise=1.0, seed=6)
    Cb = C[0].T.copy()
    x, x_est, p0, normals = G.faugpapa_start(Cb, CalM)
    f, g_, A, B, Cc, _ = O._faugpapa_constrGH(x_est, p0)
    T, D, gm, Cm = G.faugpapa_model(G.to_mp(p0))
    assert np.abs(G.to_float(gm) - g_).max() < 1e-15 * max(1.0, np.abs(g_).max())
    assert np.abs(G.to_float(Cm) - Cc).max() < 1e-14 * np.abs(Cc).max()
    assert np.abs(G.to_float(D) - np.eye(27)).max() == 0.0
    xi = G.to_mp(x_est)
    for i in range(9):
        fi, Ap, Bi = G._blocks(xi[6 * i:6 * i + 6], T)
        assert np.abs(G.to_float(Ap) - A[4 * i:4 * i + 4]).max() < 1e-13 * np.abs(A).max()


def test_mp_callback_is_an_independent_restatement_of_pi():
    """... and for the Ponce-Hebert Pi-matrix callback (PiPoseEstimation.m:109-182), which has per-correspondence blocks of its own
    (three epipolar equations and one trilinearity): f, A, B, g, C of the 50-digit restatement equal the numpy oracle's at fp64 inputs."""
    from oracle import gh_mp_oracle as G
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, _, _ = generate_scene_batch(1, 9, noise=1.0, seed=8)
    Cb = C[0].T.copy()
    x, x_est, p0, normals = G.pi_start(Cb, CalM)
    f, g_, A, B, Cc, _ = O._pi_constraintsGH(x_est, p0)
    point_fn, gm, Cm = G.pi_model(G.to_mp(p0))
    assert np.abs(G.to_float(gm) - g_).max() < 1e-15 and np.abs(G.to_float(Cm) - Cc).max() < 1e-15
    xi = G.to_mp(x_est)
    for i in range(9):
        fi, Ai, Bi = point_fn(xi[6 * i:6 * i + 6])
        assert np.abs(G.to_float(fi) - f[4 * i:4 * i + 4]).max() < 1e-14
        assert np.abs(G.to_float(Ai) - A[4 * i:4 * i + 4]).max() < 1e-13 * np.abs(A).max()
        assert np.abs(G.to_float(Bi) - B[4 * i:4 * i + 4, 6 * i:6 * i + 6]).max() < 1e-13 * np.abs(B).max()
