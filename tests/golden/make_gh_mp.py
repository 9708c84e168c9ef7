"""Generates tests/golden/gh_mp.npz (Ressl) / gh_mp_nordberg.npz: the method with its Gauss-Helmert loop evaluated in 50-digit
arithmetic (oracle/gh_mp_oracle.py) on seeded synthetic scenes, N in {12, 60, 200}.  Build-container script (needs mpmath; ~20 min
on 8 cores for Ressl's 64 scenes); the fixtures hold inputs and expected outputs only.
PiCol (`picol`): its start depends on sign / basis choices the reference leaves to MATLAB's svd (PiColPoseEstimation.m:93-94 are not
covariant), so the fixture holds the 50-digit result from the start under each of the four sign choices of linearTFT's cameras P2, P3,
with the null vectors built as tests/helpers.py::kernel_null_convention states (generalised cross products) -- a convention written
down there, not an output of the kernel; scenes with collinear centres (angle 180).
Usage: python tests/golden/make_gh_mp.py [ressl|nordberg|faugpapa|pi|picol]"""
import os, sys, time
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import tft_oracle as O
from oracle import gh_mp_oracle as G
from tft_vs_fund_amd.scenes import generate_scene_batch

METHOD = sys.argv[1] if len(sys.argv) > 1 else "ressl"
CASES = {"ressl": [(12, 32, 1.0), (60, 20, 1.0), (200, 12, 1.0)],               # N, scenes, pixel noise
         "nordberg": [(12, 24, 1.0), (60, 12, 1.0), (200, 6, 1.0)],
         "faugpapa": [(12, 24, 1.0), (60, 16, 1.0), (200, 8, 1.0)],
         "pi": [(12, 24, 1.0), (60, 16, 1.0), (200, 8, 1.0)],
         "picol": [(12, 12, 1.0), (60, 6, 1.0), (200, 3, 1.0)]}[METHOD]
SEED0 = {"ressl": 4000, "nordberg": 5000, "faugpapa": 6000, "pi": 7000, "picol": 8000}[METHOD]
ANGLE = 180 if METHOD == "picol" else None
PICOL_SIGNS = [(1.0, 1.0), (1.0, -1.0), (-1.0, 1.0), (-1.0, -1.0)]
MP_FN = {"ressl": G.ResslTFTPoseEstimation_mp, "nordberg": G.NordbergTFTPoseEstimation_mp, "faugpapa": G.FaugPapaTFTPoseEstimation_mp,
         "pi": G.PiPoseEstimation_mp, "picol": G.PiColPoseEstimation_mp}[METHOD]
NP_FN = {"ressl": O.ResslTFTPoseEstimation, "nordberg": O.NordbergTFTPoseEstimation, "faugpapa": O.FaugPapaTFTPoseEstimation,
         "pi": O.PiPoseEstimation, "picol": O.PiColPoseEstimation}[METHOD]
OUT = {"ressl": "gh_mp.npz", "nordberg": "gh_mp_nordberg.npz", "faugpapa": "gh_mp_faugpapa.npz", "pi": "gh_mp_pi.npz", "picol": "gh_mp_picol.npz"}[METHOD]


def _nan_result():
    return np.full((3, 4), np.nan), np.full((3, 4), np.nan), np.full((3, 3, 3), np.nan), -1, "failed"


def one(args):
    Cb, CalM = args
    t0 = time.time()
    if METHOD == "picol":                            # convention 0 first; a convention whose start does not exist (:88) is stored as NaN
        from helpers import kernel_null_convention
        res = []
        for sg in PICOL_SIGNS:
            try:
                res.append(MP_FN(Cb, CalM, null=kernel_null_convention, cam_signs=sg))
            except ValueError:
                res.append(_nan_result())
        try:
            o2, o3, _, oT, oit, dbg = NP_FN(Cb, CalM, True)
        except ValueError:
            o2, o3, oT, oit, dbg = np.full((3, 4), np.nan), np.full((3, 4), np.nan), np.full((3, 3, 3), np.nan), -1, dict(reason="failed")
        R2, R3, T, it, reason = res[0]
        return R2, R3, T, it, reason, o2, o3, oT, oit, dbg["reason"], time.time() - t0, res[1:]
    R2, R3, T, it, reason = MP_FN(Cb, CalM)
    o2, o3, _, oT, oit, dbg = NP_FN(Cb, CalM, True)
    alt = []
    if METHOD == "nordberg":                         # the other seven sign conventions of linearTFT's three V(:,end) (tests/helpers.py)
        for sg in [(a, b, c) for c in (1, -1) for a in (1, -1) for b in (1, -1)][1:]:
            O.set_epipole_signs(sg)
            try:
                alt.append(MP_FN(Cb, CalM))
            finally:
                O.set_epipole_signs(None)
    return R2, R3, T, it, reason, o2, o3, oT, oit, dbg["reason"], time.time() - t0, alt


if __name__ == "__main__":
    out = {}
    with Pool(int(os.environ.get("MP_WORKERS", "8"))) as pool:
        for ci, (N, B, noise) in enumerate(CASES):
            C, CalM, _, _ = generate_scene_batch(B, N, noise=noise, seed=SEED0 + N, angle=ANGLE)
            res = pool.map(one, [(C[b].T.copy(), CalM) for b in range(B)], chunksize=1)
            pre = "c%d_" % ci
            out[pre + "meta"] = np.array([N, B, noise])
            out[pre + "Corresp"] = C
            out[pre + "CalM"] = CalM
            out[pre + "mp_Rt2"] = np.stack([r[0] for r in res]); out[pre + "mp_Rt3"] = np.stack([r[1] for r in res])
            out[pre + "mp_T"] = np.stack([r[2] for r in res]); out[pre + "mp_iter"] = np.array([r[3] for r in res])
            out[pre + "mp_reason"] = np.array([r[4] for r in res])
            out[pre + "np_Rt2"] = np.stack([r[5] for r in res]); out[pre + "np_Rt3"] = np.stack([r[6] for r in res])
            out[pre + "np_T"] = np.stack([r[7] for r in res]); out[pre + "np_iter"] = np.array([r[8] for r in res])
            out[pre + "np_reason"] = np.array([r[9] for r in res])
            if METHOD in ("nordberg", "picol"):      # [scene, convention]: Nordberg: convention 0 = numpy's LAPACK as is (the mp_* arrays above);
                                                     # PiCol: the four camera-sign choices under the stated null-vector convention
                out[pre + "mp4_Rt2"] = np.stack([np.stack([r[0]] + [a[0] for a in r[11]]) for r in res])
                out[pre + "mp4_Rt3"] = np.stack([np.stack([r[1]] + [a[1] for a in r[11]]) for r in res])
                out[pre + "mp4_T"] = np.stack([np.stack([r[2]] + [a[2] for a in r[11]]) for r in res])
                out[pre + "mp4_iter"] = np.array([[r[3]] + [a[3] for a in r[11]] for r in res])
            print("N=%d: %d scenes, %.0f s of mp arithmetic; iterations mp %s / numpy %s" % (N, B, sum(r[10] for r in res), out[pre + "mp_iter"].tolist(), out[pre + "np_iter"].tolist()), flush=True)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), OUT), **out)
