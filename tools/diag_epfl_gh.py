"""GPU vs golden on the EPFL samples for the iterative methods: iterations, status, pose errors (diagnostic)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tft_vs_fund_amd import api
from tft_vs_fund_amd.metrics import AngError
g = np.load("tests/golden/epfl.npz"); gp = np.load("tests/golden/pi.npz"); go = np.load("tests/golden/optimf.npz")
ctx = api.Context(0)
for n in range(int(g["count"])):
    pre = "t%d_" % n
    C = np.ascontiguousarray(g[pre + "sample"].T)[None]
    for meth, key, src in (("ResslTFTPoseEstimation", "ressl", g), ("PiPoseEstimation", "pi", gp), ("OptimFPoseEstimation", "optimf", go),
                           ("NordbergTFTPoseEstimation", None, None), ("FaugPapaTFTPoseEstimation", None, None)):
        out = ctx.pose_batch(meth, C, g[pre + "CalM"], reconst=True)
        r3, t3 = AngError(g[pre + "Rt0"][1], out["R_t_3"][0])
        line = "%d %-26s st %d it %2d  rot3 %.4f t3 %.4f" % (n, meth, out["status"][0], out["iter"][0], r3, t3)
        if key:
            o3, ot3 = AngError(g[pre + "Rt0"][1], src[pre + key + "_Rt3"])
            line += "   | oracle it %2d rot3 %.4f t3 %.4f" % (int(src[pre + key + "_iter"]), o3, ot3)
        print(line)
