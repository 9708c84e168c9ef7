"""Which methods accept which N (LDS workspace limits)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
ctx = api.Context(0)
for N in (400, 700, 1000, 1500, 3000):
    C, CalM, Rt0, _ = generate_scene_batch(64, N, noise=1.0, seed=7)
    for meth in api.POSE_METHODS:
        try:
            out = ctx.pose_batch(meth, C, CalM, reconst=True)
            R = out["R_t_3"][:, :, :3]
            c = (np.einsum("ij,bij->b", Rt0[1][:, :3], R) - 1) / 2
            print("N %4d %-26s ok: status %s mean iter %.2f mean rot err %.4f deg" % (N, meth, dict(zip(*np.unique(out["status"], return_counts=True))), out["iter"].mean(),
                                                                                 np.degrees(np.arccos(np.clip(c, -1, 1))).mean()))
        except Exception as e:
            print("N %4d %-26s ERROR %s" % (N, meth, str(e)[:90]))
