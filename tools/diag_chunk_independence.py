import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
Ns, H, n = 200, 3000, 14
Cs, CalM, _, _ = generate_scene_batch(1, Ns, noise=0.5, seed=21)
scene = Cs[0].copy()
rng = np.random.default_rng(3)
bad = rng.choice(Ns, Ns // 5, replace=False)
scene[bad, 2:6] += rng.uniform(20, 80, size=(bad.size, 4))
d_scene = torch.from_numpy(scene).cuda(); calm = torch.from_numpy(CalM).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(9)
idx = torch.rand((H, Ns), device="cuda", generator=g).argsort(dim=1)[:, :n].to(torch.int32).contiguous()
ctx = api.Context(0)
whole = ctx.pose_sampled("LinearTFTPoseEstimation", d_scene, calm, idx)
for chunk in (37, 4, 1500):
    parts = [ctx.pose_sampled("LinearTFTPoseEstimation", d_scene, calm, idx[s:s + chunk].contiguous()) for s in range(0, H, chunk)]
    for key in ("R_t_2", "R_t_3", "T"):
        got = torch.cat([p[key] for p in parts]).cpu().numpy().reshape(H, -1); ref = whole[key].cpu().numpy().reshape(H, -1)
        d = np.abs(got - ref).max(axis=1)
        w = np.nonzero(d > 0)[0]
        print(chunk, key, "differ:", w.size, "max", d.max() if w.size else 0, "first", w[:10], "pos in chunk", (w[:10] % chunk))
# solver=1 (exact kernel for all) comparison: which of the differing ones were retried?
ctx2 = api.Context(0); ctx2.set_solver("exact")
ex = ctx2.pose_sampled("LinearTFTPoseEstimation", d_scene, calm, idx)
got = torch.cat([p["R_t_3"] for p in parts]).cpu().numpy().reshape(H, -1)
ref = whole["R_t_3"].cpu().numpy().reshape(H, -1); exr = ex["R_t_3"].cpu().numpy().reshape(H, -1)
w = np.nonzero(np.abs(got - ref).max(axis=1) > 0)[0]
print("differing:", w.size, " of which whole==exact bitwise:", int((np.abs(ref[w] - exr[w]).max(axis=1) == 0).sum()), " chunk==exact:", int((np.abs(got[w] - exr[w]).max(axis=1) == 0).sum()))
