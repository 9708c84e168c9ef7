import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
ctx = api.Context(0)
for N in (200, 500, 700, 1000):
    C, CalM, _, _ = generate_scene_batch(2000, N, noise=1.0, seed=1)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    row = []
    for m in ("ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "PiPoseEstimation"):
        out = ctx.pose_batch(m, d, calm, reconst=False)
        it = out["iter"].cpu().numpy()
        row.append("%s it mean %.2f max %d bad %d" % (m[:6], it.mean(), it.max(), int((out["status"].cpu().numpy() != 0).sum())))
    print(N, " | ".join(row))
