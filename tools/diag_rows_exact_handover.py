"""How many minimal samples does the four-per-wavefront exact kernel hand on to the one-triplet exact kernel, and why?  (debug entry point:
dbg[69] / dbg[70] carry 20000 + iterations from k_linear_tft_pose_rows_exact, 10000 + ... from k_linear_tft_pose<true>)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
ctx = api.Context(0)
for tag, (B, N, noise) in (("generic scenes", (40000, 7, 3.0)), ("generic scenes", (40000, 7, 0.5)), ("generic scenes", (20000, 9, 1.0))):
    C, CalM, _, _ = generate_scene_batch(B, N, noise=noise, seed=5)
    out = ctx.pose_batch("LinearTFTPoseEstimation", torch.from_numpy(C).cuda(), torch.from_numpy(CalM).cuda(), reconst=False, debug=True)
    dbg = out["debug"].cpu().numpy()
    it1, it2 = dbg[:, 69], dbg[:, 70]
    rows = it1 >= 20000
    print("%s B=%d N=%d noise=%.1f: finished by the rows kernel %.2f %%; handed on %.2f %%; of the finished: its27 mean %.1f max %d, its15 mean %.1f max %d" % (
        tag, B, N, noise, 100 * rows.mean(), 100 * (1 - rows.mean()), (it1[rows] - 20000).mean(), (it1[rows] - 20000).max(), (it2[rows] - 20000).mean(), (it2[rows] - 20000).max()))
    ho = ~rows
    if ho.any():
        print("    handed on: wave kernel stamps its27 %s its15 %s (>= 11000: one-sided Jacobi fall-back)" % (np.unique((it1[ho] // 1000).astype(int), return_counts=True), np.unique((it2[ho] // 1000).astype(int), return_counts=True)))
# config 4's situation: seven-point samples of ONE scene with 25 % gross outliers
Ns, H = 400, 100000
Cs, CalM, _, _ = generate_scene_batch(1, Ns, noise=0.5, seed=7)
scene = Cs[0].copy()
rng = np.random.default_rng(1)
bad = rng.choice(Ns, Ns // 4, replace=False)
scene[bad, 2:6] += rng.uniform(20, 80, size=(bad.size, 4))
idx = np.argsort(rng.random((H, Ns)), axis=1)[:, :7]
C = np.ascontiguousarray(scene[idx])
out = ctx.pose_batch("LinearTFTPoseEstimation", torch.from_numpy(C).cuda(), torch.from_numpy(CalM).cuda(), reconst=False, debug=True)
dbg = out["debug"].cpu().numpy()
it1, it2 = dbg[:, 69], dbg[:, 70]
rows = it1 >= 20000
n_out = np.isin(idx, bad).sum(axis=1)
print("config-4 scene, %d seven-point samples: finished by the rows kernel %.2f %%, handed on %.2f %%" % (H, 100 * rows.mean(), 100 * (1 - rows.mean())))
for k in range(0, 5):
    m = n_out == k
    if m.any():
        print("   samples with %d outliers: %6d, handed on %.2f %%; its27 of the finished: mean %.1f p99 %d" % (k, m.sum(), 100 * (1 - rows[m].mean()), (it1[m & rows] - 20000).mean(), np.quantile(it1[m & rows] - 20000, 0.99)))
