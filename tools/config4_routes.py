"""config 4 (1 M seven-point TFT hypotheses of one scene): the whole-batch exact route (TFF_OPT_EXACT_BELOW = 12, default) against
the flag-and-redo route (EXACT_BELOW = 0: four-hypotheses-per-wavefront fast kernel, exact kernel over what it flags)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
H = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
Ns = 400
ctx = api.Context(0)
C, CalM, _, _ = generate_scene_batch(1, Ns, noise=0.5, seed=7)
scene = C[0].copy()
rng = np.random.default_rng(1)
bad = rng.choice(Ns, Ns // 4, replace=False)
scene[bad, 2:6] += rng.uniform(20, 80, size=(bad.size, 4))
d_scene = torch.from_numpy(scene).cuda(); d_calm = torch.from_numpy(CalM).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(1234)
idx = torch.rand((H, Ns), device="cuda", generator=g).argsort(dim=1)[:, :7].to(torch.int32).contiguous()
res = {}
for eb in (12, 0):
    ctx.set_exact_below(eb)
    for rep in range(3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        hyp = ctx.pose_sampled("LinearTFTPoseEstimation", d_scene, d_calm, idx)
        e[1].record()
        torch.cuda.synchronize()
    cnt = ctx.inlier_count(d_scene, d_calm, hyp["R_t_2"], hyp["R_t_3"], 1.0)
    res[eb] = (hyp, cnt)
    print("exact_below=%2d: pose %.2f ms for %d hypotheses = %.2f M/s; status!=0: %d; best count %d" % (
        eb, e[0].elapsed_time(e[1]), H, H / e[0].elapsed_time(e[1]) / 1e3, int((hyp["status"] != 0).sum()), int(cnt.max())))
a, b = res[12], res[0]
ok = (a[0]["status"] == 0) & (b[0]["status"] == 0)
print("status equal: %s; inlier counts differ in %d of %d" % (bool((a[0]["status"] == b[0]["status"]).all()), int((a[1] != b[1]).sum()), H))
d3 = (a[0]["R_t_3"] - b[0]["R_t_3"]).abs().flatten(1).max(dim=1).values / a[0]["R_t_3"].abs().flatten(1).max(dim=1).values.clamp(min=1.0)
d3 = d3[ok]
print("R_t_3 deviation between the routes: p50 %.2e p99 %.2e max %.2e; > 1e-6: %d" % (d3.quantile(0.5), d3.quantile(0.99), d3.max(), int((d3 > 1e-6).sum())))
