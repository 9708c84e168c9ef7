import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
ctx = api.Context(0)
for seed, N in ((1, 200), (1000, 200), (5, 12), (6, 7)):
    C, CalM, _, _ = generate_scene_batch(10000, N, noise=1.0, seed=seed)
    out = ctx.pose_batch("LinearTFTPoseEstimation", torch.from_numpy(C).cuda(), torch.from_numpy(CalM).cuda(), reconst=False, debug=True)
    dbg = out["debug"].cpu().numpy()
    i27, i15 = dbg[:, 69], dbg[:, 70]
    print("seed", seed, "N", N, "its27 hist", np.unique(i27, return_counts=True), "its15 hist", np.unique(i15, return_counts=True))
