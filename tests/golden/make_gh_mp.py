"""Generates tests/golden/gh_mp.npz: ResslTFTPoseEstimation with the Gauss-Helmert loop evaluated in 50-digit arithmetic
(oracle/gh_mp_oracle.py) on seeded synthetic scenes, N in {12, 60, 200}.  Build-container script (needs mpmath; ~20 min on
8 cores); the fixture holds inputs and expected outputs only.  Usage: python tests/golden/make_gh_mp.py"""
import os, sys, time
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import tft_oracle as O
from oracle import gh_mp_oracle as G
from tft_vs_fund_amd.scenes import generate_scene_batch

CASES = [(12, 32, 1.0), (60, 20, 1.0), (200, 12, 1.0)]            # N, scenes, pixel noise


def one(args):
    Cb, CalM = args
    t0 = time.time()
    R2, R3, T, it, reason = G.ResslTFTPoseEstimation_mp(Cb, CalM)
    o2, o3, _, oT, oit, dbg = O.ResslTFTPoseEstimation(Cb, CalM, True)
    return R2, R3, T, it, reason, o2, o3, oT, oit, dbg["reason"], time.time() - t0


if __name__ == "__main__":
    out = {}
    with Pool(8) as pool:
        for ci, (N, B, noise) in enumerate(CASES):
            C, CalM, _, _ = generate_scene_batch(B, N, noise=noise, seed=4000 + N)
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
            print("N=%d: %d scenes, %.0f s of mp arithmetic; iterations mp %s / numpy %s" % (N, B, sum(r[10] for r in res), out[pre + "mp_iter"].tolist(), out[pre + "np_iter"].tolist()), flush=True)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gh_mp.npz"), **out)
