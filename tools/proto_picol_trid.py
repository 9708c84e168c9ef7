"""Would FaugPapa's tridiagonal + twisted-RQI pseudo-inverse (csrc/wave_trid.h, numpy twin tools/proto_trid_pinv.py) serve PiCol's 38 x 38 KKT systems?  Captures the
matrices the oracle hands to pinv (Gauss_Helmert.m:67) on generic and collinear scenes and runs the twin on them: the kept eigenvalues of these matrices sit
1e-10 .. 1e-15 |M| apart, independent Rayleigh-quotient iterations do not return orthogonal vectors there, the solve misses by 1e-7 .. 0.2.  (Build-container diagnostic.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, warnings
from oracle import tft_oracle as O
from tft_vs_fund_amd.scenes import generate_scene_batch
import proto_trid_pinv as PT
caps = []
orig = O.pinv
def cap_pinv(A, *a, **k):
    if A.shape[0] == 38: caps.append(A.copy())
    return orig(A, *a, **k)
O.pinv = cap_pinv
for angle, seed in ((None, 1), (None, 2), (180, 3), (175, 4), (160, 5), (178, 6)):
    C, CalM, _, _ = generate_scene_batch(2, 60, noise=1.0, seed=seed, angle=angle)
    for b in range(2):
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out = O.PiColPoseEstimation(C[b].T.copy(), CalM)
            print("angle", angle, "iter", out[4])
        except Exception as ex:
            print("angle", angle, "failed:", ex)
print(len(caps), "KKT matrices")
rng = np.random.default_rng(0)
worst = 0
for M in caps:
    lam, V = np.linalg.eigh(M)
    tol = 38 * np.spacing(np.abs(lam).max())
    keep = np.abs(lam) > tol
    b = rng.standard_normal(38)
    xr = V[:, keep] @ ((V[:, keep].T @ b) / lam[keep])
    st = []
    x, kept = PT.pinv_solve_sym(M, b, tol, st)
    err = np.linalg.norm(x - xr) / np.linalg.norm(xr)
    kl = np.sort(np.abs(lam[keep]))
    gaps = np.diff(np.sort(lam[keep])) / np.abs(lam).max()
    worst = max(worst, err)
    print("kept %d/%d (ref %d) err %.1e  |lam| kept min %.2e max %.2e  truncated max %.2e tol %.1e  min rel gap %.1e  stats %s" % (kept, 38, keep.sum(), err, kl[0], kl[-1], np.abs(lam[~keep]).max() if (~keep).any() else 0, tol, gaps.min(), st))
print("worst", worst)
