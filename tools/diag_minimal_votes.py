"""Minimal samples where the HIP result differs from the oracle: compare the cheirality votes of the four candidates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import tft_oracle as O
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
from helpers import rel_err_T, rel_err
ctx = api.Context(0)
N, noise, B = 7, 2.0, 300
C, CalM, _, _ = generate_scene_batch(B, N, noise=noise, seed=1000 + 7 * N + int(10 * noise))
out = ctx.pose_batch("LinearTFTPoseEstimation", C, CalM, reconst=True, debug=True)
nbad = 0
for b in range(B):
    R2, R3, Rec, T, _ = O.LinearTFTPoseEstimation(C[b].T.copy(), CalM)
    eT = rel_err_T(out["T"][b], T)
    e = max(eT, rel_err(out["R_t_2"][b], R2), rel_err(out["R_t_3"][b], R3))
    if e > 1e-6:
        nbad += 1
        dbg = R_t = O.R_t_from_TFT(T, CalM, C[b].T.copy(), return_debug=True)
        print("triplet", b, "errT %.2e  err %.2e" % (eT, e))
        print("  kernel votes", out["debug"][b, 60:68])
        print("  oracle debug", [x for x in dbg[2:]] if isinstance(dbg, tuple) else dbg)
print("bad", nbad)
