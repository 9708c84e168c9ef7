import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
ctx = api.Context(0)
np.set_printoptions(linewidth=250)
for seed, N in ((1, 200), (5, 12), (6, 7)):
    C, CalM, _, _ = generate_scene_batch(10000, N, noise=1.0, seed=seed)
    out = ctx.pose_batch("LinearTFTPoseEstimation", torch.from_numpy(C).cuda(), torch.from_numpy(CalM).cuda(), reconst=False, debug=True)
    dbg = out["debug"].cpu().numpy()
    for name, col in (("its27", 69), ("its15", 70)):
        v = dbg[:, col]
        jac = v >= 1000
        q = np.percentile(v[~jac], [50, 90, 99, 99.9]) if (~jac).any() else []
        print("seed", seed, "N", N, name, "jacobi fix-ups: %d  | inverse-iteration counts: median/p90/p99/p99.9 =" % jac.sum(), q, " mean %.2f max %d" % (v[~jac].mean(), v[~jac].max()))
    tot = dbg[:, 93] - dbg[:, 80]
    print("   cycles/wave mean %.0f  p99 %.0f  max %.0f" % (tot.mean(), np.percentile(tot, 99), tot.max()))
