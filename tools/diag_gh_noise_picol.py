"""PiColPoseEstimation: deviation of (i) the LAPACK-backed numpy oracle and (ii) the HIP kernel from the 50-digit evaluation of the reference's
formulas (tests/golden/gh_mp_picol.npz: four sign conventions of linearTFT's cameras under the null-vector convention of
tests/helpers.py::kernel_null_convention), best convention per scene, and the iteration-count differences.  --no-gpu: oracle column only."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import rel_err_T, rel_err, golden_cases, kernel_null_convention
from oracle import tft_oracle as O

SIGNS = [(1.0, 1.0), (1.0, -1.0), (-1.0, 1.0), (-1.0, -1.0)]
g = np.load(os.path.join(ROOT, "tests", "golden", "gh_mp_picol.npz"))
ctx = None
if "--no-gpu" not in sys.argv:
    from tft_vs_fund_amd import api
    ctx = api.Context(0)
    ctx.set_rows(int(os.environ.get("TFF_ROWS", "1")))       # 0: one-triplet-per-wavefront linear stage (A/B)
fmt = lambda d: "p50 %.1e  p90 %.1e  max %.1e" % (np.quantile(d, 0.5), np.quantile(d, 0.9), np.max(d))
for ci, pre in golden_cases(g):
    N, B, _ = g[pre + "meta"]; N, B = int(N), int(B)
    C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
    T4, R24, R34, it4 = g[pre + "mp4_T"], g[pre + "mp4_Rt2"], g[pre + "mp4_Rt3"], g[pre + "mp4_iter"]
    do, dio = [], []
    for b in range(B):
        best = (np.inf, 0)
        for c, sg in enumerate(SIGNS):
            if it4[b, c] < 0:
                continue
            try:
                o2, o3, _, oT, oit = O.PiColPoseEstimation(C[b].T.copy(), CalM, null=kernel_null_convention, cam_signs=sg)
            except ValueError:
                continue
            d = max(rel_err_T(oT, T4[b, c]), rel_err(o2, R24[b, c]), rel_err(o3, R34[b, c]))
            best = min(best, (d, oit - int(it4[b, c])))
        do.append(best[0]); dio.append(abs(best[1]))
    print("N=%-4d scenes %-3d LAPACK oracle (worst... best convention per scene): %s  iter diff %s" % (N, B, fmt(np.array(do)), np.bincount(dio).tolist()))
    if ctx is not None:
        out = ctx.pose_batch("PiColPoseEstimation", C, CalM, reconst=False)
        dk, dik, st = [], [], np.asarray(out["status"])
        for b in range(B):
            cand = [(max(rel_err_T(out["T"][b], T4[b, c]), rel_err(out["R_t_2"][b], R24[b, c]), rel_err(out["R_t_3"][b], R34[b, c])), abs(int(out["iter"][b]) - int(it4[b, c])))
                    for c in range(4) if it4[b, c] >= 0]
            d, di = min(cand)
            dk.append(d); dik.append(di)
        print("                   HIP kernel (best convention per scene)                  : %s  iter diff %s  (status != 0: %d)" % (fmt(np.array(dk)), np.bincount(dik).tolist(), int((st != 0).sum())))
        print("                   per scene: " + " ".join("%.1e" % v for v in dk) + "   iterations " + str(np.asarray(out["iter"]).tolist()))
