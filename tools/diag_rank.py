"""How often do the Gauss-Helmert kernels report TFF_ST_RANK (KKT system numerically singular)?  Collinear and generic scenes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
ctx = api.Context(0)
for angle in (None, 170, 178, 180):
    for N in (7, 12, 100):
        C, CalM, Rt0, _ = generate_scene_batch(2000, N, noise=1.0, seed=7, angle=angle)
        for meth in ("ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "FaugPapaTFTPoseEstimation", "PiPoseEstimation"):
            out = ctx.pose_batch(meth, C, CalM, reconst=False)
            st = out["status"]
            print("angle %s N %3d %-26s status counts %s  mean iter %.2f" % (angle, N, meth, dict(zip(*np.unique(st, return_counts=True))), out["iter"].mean()))
