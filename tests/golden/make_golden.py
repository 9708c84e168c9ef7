"""
Generates the committed golden fixtures under tests/golden/.

The reference (MATLAB) cannot run in the build container and ships no tests or
golden vectors, so the expected outputs here come from oracle/tft_oracle.py --
the literal numpy/LAPACK restatement of the .m files -- run on

  * synthetic scenes with the geometry of generateSyntheticScene.m (own Philox
    RNG, see tft_vs_fund_amd/scenes.py), N in {7, 8, 12, 100, 200, 1000},
    sigma in {0, 1, 3};
  * EPFL fountain-P11 / Herz-Jesu-P8 triplets: the correspondences and the
    ground-truth cameras are DATA read from /root/reference/Data (the .mat and
    .camera files the reference's experiments_real.m consumes), filtered with
    its 1-px inlier rule (experiments_real.m:93-99) and sub-sampled to 100.

Run from the repo root (needs /root/reference for the EPFL part only):
    python tests/golden/make_golden.py
Outputs: synthetic_linear.npz, synthetic_gh.npz, epfl.npz, optimf.npz, pi.npz, ba.npz  (inputs +
expected outputs; no reference source text).  optimf.npz (OptimFPoseEstimation)
covers synthetic scenes and the 100-correspondence EPFL samples of epfl.npz; pi.npz holds
PiPoseEstimation on the same kinds of input and PiColPoseEstimation on scenes with collinear
camera centres (angle = 180 in generateSyntheticScene.m's terms), LAPACK sign conventions.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tft_oracle as O                      # noqa: E402
from tft_vs_fund_amd.scenes import generate_scene_batch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/Data"

LINEAR_CASES = [  # (N, sigma, seed, B)
    (7, 1.0, 11, 4), (8, 1.0, 12, 4), (8, 0.0, 13, 2), (12, 1.0, 14, 4), (12, 3.0, 15, 3),
    (100, 0.0, 16, 2), (100, 1.0, 17, 3), (200, 1.0, 18, 4), (200, 3.0, 19, 2), (1000, 1.0, 20, 2),
]
GH_CASES = [(12, 1.0, 31, 3), (12, 0.25, 32, 2), (50, 1.0, 33, 2), (100, 3.0, 34, 2), (200, 1.0, 35, 2)]


def _linear_outputs(C, CalM):
    out = {}
    B = C.shape[0]
    keys = ("Rt2", "Rt3", "T", "Rec")
    for meth, fn in (("tft", O.LinearTFTPoseEstimation), ("f", O.LinearFPoseEstimation)):
        acc = {k: [] for k in keys}
        for b in range(B):
            if meth == "f" and C.shape[1] < 8:
                continue
            R2, R3, Rec, T, it = fn(C[b].T.copy(), CalM)
            acc["Rt2"].append(R2); acc["Rt3"].append(R3); acc["T"].append(T); acc["Rec"].append(Rec)
        for k in keys:
            if acc[k]:
                out["%s_%s" % (meth, k)] = np.stack(acc[k])
    # intermediates of the TFT path: unconstrained/constrained tensors, epipoles, votes, lambda
    dbg = {k: [] for k in ("lin_T", "lin_P2", "lin_P3", "votes2", "votes3", "lam", "rankE")}
    for b in range(B):
        Cb = C[b].T.copy()
        x1, N1 = O.Normalize2Ddata(Cb[0:2]); x2, N2 = O.Normalize2Ddata(Cb[2:4]); x3, N3 = O.Normalize2Ddata(Cb[4:6])
        T, P1, P2, P3, d = O.linearTFT(x1, x2, x3, return_debug=True)
        Tn = O.transform_TFT(T, N1, N2, N3, 1)
        _, _, dd = O.R_t_from_TFT(Tn, CalM, Cb, return_debug=True)
        dbg["lin_T"].append(T); dbg["lin_P2"].append(P2); dbg["lin_P3"].append(P3)
        dbg["votes2"].append(dd["votes2"]); dbg["votes3"].append(dd["votes3"]); dbg["lam"].append(dd["lam"])
        dbg["rankE"].append(d["rankE"])
    for k, v in dbg.items():
        out["dbg_" + k] = np.array(v)
    return out


def make_synthetic_linear():
    data = {}
    for ci, (N, sigma, seed, B) in enumerate(LINEAR_CASES):
        C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=sigma, seed=seed)
        pre = "c%d_" % ci
        data[pre + "Corresp"] = C
        data[pre + "CalM"] = CalM
        data[pre + "meta"] = np.array([N, sigma, seed, B], dtype=np.float64)
        data[pre + "Rt0"] = np.stack(Rt0)
        for k, v in _linear_outputs(C, CalM).items():
            data[pre + k] = v
        print("linear case", ci, N, sigma, "done", flush=True)
    np.savez_compressed(os.path.join(HERE, "synthetic_linear.npz"), **data)


def make_synthetic_gh():
    data = {}
    for ci, (N, sigma, seed, B) in enumerate(GH_CASES):
        C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=sigma, seed=seed)
        pre = "c%d_" % ci
        data[pre + "Corresp"] = C
        data[pre + "CalM"] = CalM
        data[pre + "meta"] = np.array([N, sigma, seed, B], dtype=np.float64)
        for meth, fn in (("ressl", O.ResslTFTPoseEstimation), ("nordberg", O.NordbergTFTPoseEstimation),
                         ("faugpapa", O.FaugPapaTFTPoseEstimation)):
            acc = {k: [] for k in ("Rt2", "Rt3", "T", "Rec", "iter", "reason")}
            for b in range(B):
                R2, R3, Rec, T, it, d = fn(C[b].T.copy(), CalM, return_debug=True)
                acc["Rt2"].append(R2); acc["Rt3"].append(R3); acc["T"].append(T); acc["Rec"].append(Rec)
                acc["iter"].append(it); acc["reason"].append(d["reason"])
            for k in ("Rt2", "Rt3", "T", "Rec"):
                data[pre + meth + "_" + k] = np.stack(acc[k])
            data[pre + meth + "_iter"] = np.array(acc["iter"], dtype=np.int32)
            data[pre + meth + "_reason"] = np.array(acc["reason"])
        print("gh case", ci, N, sigma, "done", flush=True)
    np.savez_compressed(os.path.join(HERE, "synthetic_gh.npz"), **data)


OPTIMF_CASES = [(8, 1.0, 41, 3), (12, 1.0, 42, 3), (12, 3.0, 43, 2), (50, 0.0, 44, 2), (100, 1.0, 45, 3), (200, 1.0, 46, 3),
                (200, 3.0, 47, 2), (700, 1.0, 48, 2)]


def make_optimf():
    """OptimFPoseEstimation (F_methods/OptimFPoseEstimation.m) on synthetic scenes and on the EPFL samples of epfl.npz."""
    data = {}
    for ci, (N, sigma, seed, B) in enumerate(OPTIMF_CASES):
        C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=sigma, seed=seed)
        pre = "c%d_" % ci
        data[pre + "Corresp"] = C
        data[pre + "CalM"] = CalM
        data[pre + "meta"] = np.array([N, sigma, seed, B], dtype=np.float64)
        acc = {k: [] for k in ("Rt2", "Rt3", "T", "Rec", "iter")}
        for b in range(B):
            R2, R3, Rec, T, it = O.OptimFPoseEstimation(C[b].T.copy(), CalM)
            acc["Rt2"].append(R2); acc["Rt3"].append(R3); acc["T"].append(T); acc["Rec"].append(Rec); acc["iter"].append(it)
        for k in ("Rt2", "Rt3", "T", "Rec"):
            data[pre + "optimf_" + k] = np.stack(acc[k])
        data[pre + "optimf_iter"] = np.array(acc["iter"], dtype=np.int32)
        print("optimf case", ci, N, sigma, acc["iter"], flush=True)
    g = np.load(os.path.join(HERE, "epfl.npz"))
    for n in range(int(g["count"])):
        pre = "t%d_" % n
        R2, R3, Rec, T, it = O.OptimFPoseEstimation(g[pre + "sample"].copy(), g[pre + "CalM"])
        data[pre + "optimf_Rt2"] = R2; data[pre + "optimf_Rt3"] = R3; data[pre + "optimf_T"] = T; data[pre + "optimf_Rec"] = Rec
        data[pre + "optimf_iter"] = np.array(it)
        print("optimf epfl", n, it, flush=True)
    np.savez_compressed(os.path.join(HERE, "optimf.npz"), **data)


PI_CASES = [(12, 1.0, 51, 3, None), (12, 0.25, 52, 2, None), (50, 1.0, 53, 3, None), (100, 3.0, 54, 2, None), (200, 1.0, 55, 3, None),
            (40, 0.0, 56, 2, None)]
PICOL_CASES = [(40, 0.0, 61, 3, 180), (12, 1.0, 62, 3, 180), (50, 1.0, 63, 3, 180), (200, 1.0, 64, 2, 180)]


def make_pi():
    """PiPoseEstimation / PiColPoseEstimation (TFT_methods/Pi*.m) goldens."""
    data = {}
    for key, fn, cases in (("pi", O.PiPoseEstimation, PI_CASES), ("picol", O.PiColPoseEstimation, PICOL_CASES)):
        for ci, (N, sigma, seed, B, angle) in enumerate(cases):
            C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=sigma, seed=seed, angle=angle)
            pre = "%s%d_" % (key[0] if key == "pi" else "q", ci)
            data[pre + "Corresp"] = C
            data[pre + "CalM"] = CalM
            data[pre + "Rt0"] = np.stack(Rt0)
            data[pre + "meta"] = np.array([N, sigma, seed, B, -1 if angle is None else angle], dtype=np.float64)
            acc = {k: [] for k in ("Rt2", "Rt3", "T", "Rec", "iter", "reason")}
            for b in range(B):
                R2, R3, Rec, T, it, d = fn(C[b].T.copy(), CalM, True)
                acc["Rt2"].append(R2); acc["Rt3"].append(R3); acc["T"].append(T); acc["Rec"].append(Rec)
                acc["iter"].append(it); acc["reason"].append(d["reason"])
            for k in ("Rt2", "Rt3", "T", "Rec"):
                data[pre + key + "_" + k] = np.stack(acc[k])
            data[pre + key + "_iter"] = np.array(acc["iter"], dtype=np.int32)
            data[pre + key + "_reason"] = np.array(acc["reason"])
            print(key, "case", ci, N, sigma, acc["iter"], acc["reason"], flush=True)
    g = np.load(os.path.join(HERE, "epfl.npz"))
    for n in range(int(g["count"])):
        pre = "t%d_" % n
        R2, R3, Rec, T, it = O.PiPoseEstimation(g[pre + "sample"].copy(), g[pre + "CalM"])
        data[pre + "pi_Rt2"] = R2; data[pre + "pi_Rt3"] = R3; data[pre + "pi_T"] = T; data[pre + "pi_Rec"] = Rec
        data[pre + "pi_iter"] = np.array(it)
        print("pi epfl", n, it, flush=True)
    np.savez_compressed(os.path.join(HERE, "pi.npz"), **data)


BA_CASES = [(12, 1.0, 71, 3), (12, 3.0, 72, 2), (50, 1.0, 73, 3), (100, 1.0, 74, 2), (200, 1.0, 75, 2), (30, 0.0, 76, 2)]


def make_ba():
    """BundleAdjustment (oracle/ba_oracle.py) started from LinearTFTPoseEstimation's output, with and without Reconst0;
    plus the EPFL samples (first 50 correspondences of each 100-sample, as experiments_real.m refines on a subset)."""
    from oracle import ba_oracle as BA
    data = {}
    for ci, (N, sigma, seed, B) in enumerate(BA_CASES):
        C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=sigma, seed=seed)
        pre = "c%d_" % ci
        data[pre + "Corresp"] = C
        data[pre + "CalM"] = CalM
        data[pre + "meta"] = np.array([N, sigma, seed, B], dtype=np.float64)
        acc = {k: [] for k in ("Rt2_in", "Rt3_in", "Rec_in", "Rt2", "Rt3", "Rec", "iter", "err", "Rt2_tri", "Rt3_tri", "iter_tri", "err_tri")}
        for b in range(B):
            R2, R3, Rec, T, _ = O.LinearTFTPoseEstimation(C[b].T.copy(), CalM)
            R_t_0 = np.vstack([np.eye(3, 4), R2, R3])
            Rt, Recn, it, err = BA.BundleAdjustment(CalM, R_t_0, C[b].T.copy(), Rec)
            Rtt, _, itt, errt = BA.BundleAdjustment(CalM, R_t_0, C[b].T.copy(), None)
            for k, v in zip(acc, (R2, R3, Rec, Rt[3:6], Rt[6:9], Recn, it, err, Rtt[3:6], Rtt[6:9], itt, errt)):
                acc[k].append(v)
        for k, v in acc.items():
            data[pre + k] = np.array(v) if k.startswith(("iter", "err")) else np.stack(v)
        print("ba case", ci, N, sigma, acc["iter"], acc["iter_tri"], flush=True)
    g = np.load(os.path.join(HERE, "epfl.npz"))
    for n in range(int(g["count"])):
        pre = "t%d_" % n
        Cs = g[pre + "sample"][:, :50].copy()
        R_t_0 = np.vstack([np.eye(3, 4), g[pre + "tft_Rt2"], g[pre + "tft_Rt3"]])
        Rt, Recn, it, err = BA.BundleAdjustment(g[pre + "CalM"], R_t_0, Cs, None)
        data[pre + "ba_Rt2"] = Rt[3:6]; data[pre + "ba_Rt3"] = Rt[6:9]; data[pre + "ba_iter"] = np.array(it); data[pre + "ba_err"] = np.array(err)
        print("ba epfl", n, it, err, flush=True)
    np.savez_compressed(os.path.join(HERE, "ba.npz"), **data)


def _read_camera(path):
    """Data/readCalibrationOrientation_EPFL.m: K (3 rows), skip, R' (3 rows), C, size."""
    with open(path) as f:
        rows = [[float(v) for v in line.split()] for line in f.read().strip().splitlines()]
    K = np.array(rows[0:3])
    R = np.array(rows[4:7]).T
    t = -R @ np.array(rows[7])
    return K, R, t


EPFL_TRIPLETS = {
    "fountain-P11": [(5, 6, 7), (6, 7, 8), (3, 4, 5), (4, 6, 9)],
    "Herz-Jesu-P8": [(6, 7, 8), (5, 6, 7), (3, 4, 5), (2, 6, 8)],
}


def make_epfl():
    import scipy.io
    data = {}
    n = 0
    for ds, trips in EPFL_TRIPLETS.items():
        m = scipy.io.loadmat(os.path.join(REF, ds, "Corresp_triplets.mat"))
        names = [str(x[0]) for x in m["im_names"].ravel()]
        for (i1, i2, i3) in trips:
            Corresp = np.ascontiguousarray(m["Corresp"][i1 - 1, i2 - 1, i3 - 1].T)        # 6 x N (experiments_real.m:80)
            cams = [_read_camera(os.path.join(REF, ds, names[i - 1] + ".camera")) for i in (i1, i2, i3)]
            (K1, R1, t1), (K2, R2, t2), (K3, R3, t3) = cams
            CalM = np.vstack([K1, K2, K3])
            Rt0 = [np.hstack([R2 @ R1.T, (t2 - R2 @ R1.T @ t1).reshape(3, 1)]),
                   np.hstack([R3 @ R1.T, (t3 - R3 @ R1.T @ t1).reshape(3, 1)])]                # :90-91
            Ps = [K1 @ np.eye(3, 4), K2 @ Rt0[0], K3 @ Rt0[1]]
            Rec0 = O.triangulation3D(Ps, Corresp)
            Rec0 = Rec0[0:3] / Rec0[3:4]
            resid = O.project3Dpoints(Rec0, Ps) - Corresp
            inl = np.sum(np.abs(resid) > 1.0, axis=0) == 0                                     # :98
            Ci = Corresp[:, inl]
            rng = np.random.Generator(np.random.Philox(key=1000 + n))
            sel = np.sort(rng.choice(Ci.shape[1], size=min(100, Ci.shape[1]), replace=False))
            Cs = Ci[:, sel]
            pre = "t%d_" % n
            data[pre + "name"] = np.array("%s (%d,%d,%d)" % (ds, i1, i2, i3))
            data[pre + "Corresp_all"] = Corresp
            data[pre + "CalM"] = CalM
            data[pre + "Rt0"] = np.stack(Rt0)
            data[pre + "n_inliers"] = np.array(int(inl.sum()))
            data[pre + "sample"] = Cs
            data[pre + "repr_gt_inliers"] = np.array(O.ReprError(Ps, Ci))
            for meth, fn in (("tft", O.LinearTFTPoseEstimation), ("f", O.LinearFPoseEstimation),
                             ("ressl", O.ResslTFTPoseEstimation)):
                out = fn(Cs.copy(), CalM)
                R2e, R3e, Rec, T, it = out[:5]
                data[pre + meth + "_Rt2"] = R2e; data[pre + meth + "_Rt3"] = R3e
                data[pre + meth + "_T"] = T; data[pre + meth + "_Rec"] = Rec
                data[pre + meth + "_iter"] = np.array(it)
                data[pre + meth + "_repr_all"] = np.array(O.ReprError(
                    [CalM[0:3] @ np.eye(3, 4), CalM[3:6] @ R2e, CalM[6:9] @ R3e], Ci))         # :130-131
            print(ds, (i1, i2, i3), "N", Corresp.shape[1], "inliers", int(inl.sum()), flush=True)
            n += 1
    data["count"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "epfl.npz"), **data)


if __name__ == "__main__":
    what = sys.argv[1:] or ["linear", "gh", "epfl", "optimf", "pi", "ba"]
    if "linear" in what:
        make_synthetic_linear()
    if "gh" in what:
        make_synthetic_gh()
    if "epfl" in what:
        make_epfl()
    if "optimf" in what:
        make_optimf()
    if "pi" in what:
        make_pi()
    if "ba" in what:
        make_ba()
