import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
ctx = api.Context(0)
for N, noise in ((7, 1.0), (7, 3.0), (8, 2.0)):
    C, CalM, _, _ = generate_scene_batch(60000, N, noise=noise, seed=5)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    out = ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=False, debug=True)
    st = out["status"].cpu().numpy(); dbg = out["debug"].cpu().numpy()
    it1 = dbg[:, 69]
    print(N, noise, "status counts", {int(k): int((st == k).sum()) for k in np.unique(st)}, "retried (Jacobi) triplets:", int((it1 >= 1000).sum()))
