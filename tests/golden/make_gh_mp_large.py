"""Generates tests/golden/gh_mp_large.npz: Ressl / Nordberg / Pi with the Gauss-Helmert loop in 50-digit arithmetic at N = 1000
correspondences (the upper end of the target range; MATLAB's pinv tolerance 4 N eps(|W|) truncates the strong directions there),
two scenes each; Nordberg under the eight sign conventions.  Build-container script (~1 CPU-hour).  Usage: python tests/golden/make_gh_mp_large.py"""
import os, sys, time
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import tft_oracle as O
from oracle import gh_mp_oracle as G
from tft_vs_fund_amd.scenes import generate_scene_batch

HERE = os.path.dirname(os.path.abspath(__file__))
CONV = [(a, b, c) for c in (1, -1) for a in (1, -1) for b in (1, -1)]
FN = {"ressl": G.ResslTFTPoseEstimation_mp, "nordberg": G.NordbergTFTPoseEstimation_mp, "pi": G.PiPoseEstimation_mp}
N, B = 1000, 2


def one(task):
    t, method, conv, Cb, CalM = task
    O.set_epipole_signs(None if conv == (1, 1, 1) else conv)
    try:
        R2, R3, T, it, reason = FN[method](Cb, CalM)
    finally:
        O.set_epipole_signs(None)
    return t, method, conv, R2, R3, T, it


if __name__ == "__main__":
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=9000)
    tasks = []
    for t in range(B):
        Cb = C[t].T.copy()
        for method in ("ressl", "pi"):
            tasks.append((t, method, (1, 1, 1), Cb, CalM))
        for conv in CONV:
            tasks.append((t, "nordberg", conv, Cb, CalM))
    t0 = time.time()
    with Pool(8) as pool:
        res = pool.map(one, tasks, chunksize=1)
    out = {"n_triplets": np.array(B), "CalM": CalM}
    for t in range(B):
        out["t%d_Corresp" % t] = C[t].T.copy(); out["t%d_CalM" % t] = CalM
        for method in ("ressl", "pi"):
            r = [x for x in res if x[0] == t and x[1] == method][0]
            out["t%d_%s_Rt2" % (t, method)] = r[3]; out["t%d_%s_Rt3" % (t, method)] = r[4]; out["t%d_%s_T" % (t, method)] = r[5]; out["t%d_%s_iter" % (t, method)] = np.array(r[6])
        rs = [[x for x in res if x[0] == t and x[1] == "nordberg" and x[2] == c][0] for c in CONV]
        out["t%d_nordberg_Rt2" % t] = np.stack([r[3] for r in rs]); out["t%d_nordberg_Rt3" % t] = np.stack([r[4] for r in rs])
        out["t%d_nordberg_T" % t] = np.stack([r[5] for r in rs]); out["t%d_nordberg_iter" % t] = np.array([r[6] for r in rs])
    np.savez_compressed(os.path.join(HERE, "gh_mp_large.npz"), **out)
    print("%d tasks, %.0f s wall" % (len(tasks), time.time() - t0))
