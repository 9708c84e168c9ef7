"""Inlier counts of 1 M hypotheses against one 400-correspondence scene: one hypothesis per wavefront (k_inlier_count_staged) against four
(k_inlier_count_rows, TFF_OPT_COUNT_ROWS).  Same counts?  ms per launch.   python tools/ab_count_rows.py [H]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
H = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
Ns = 400
Cs, CalM, _, _ = generate_scene_batch(1, Ns, noise=0.5, seed=77)
scene = Cs[0].copy()
rng = np.random.default_rng(5)
bad = rng.choice(Ns, Ns // 4, replace=False)
scene[bad, 2:6] += rng.uniform(20, 80, size=(bad.size, 4))
dev = torch.device("cuda:0")
d_scene = torch.from_numpy(scene).to(dev); calm = torch.from_numpy(CalM).to(dev)
ctx = api.Context(0)
for method, n in (("LinearTFTPoseEstimation", 7), ("LinearFPoseEstimation", 8)):
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    idx = torch.rand((H, Ns), device=dev, generator=gen).argsort(dim=1)[:, :n].to(torch.int32).contiguous()
    hyp = ctx.pose_sampled(method, d_scene, calm, idx)
    res = {}
    for rows in (0, 1, 0, 1):
        ctx.set_count_rows(rows)
        cnt = ctx.inlier_count(d_scene, calm, hyp["R_t_2"], hyp["R_t_3"], 1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            cnt = ctx.inlier_count(d_scene, calm, hyp["R_t_2"], hyp["R_t_3"], 1.0)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(rows, []).append(e0.elapsed_time(e1) / 3)
        res[("cnt", rows)] = cnt.clone()
    print("%s hypotheses: one per wavefront %.3f ms, four per wavefront %.3f ms; counts differ in %d of %d (best %d)" % (
        method, min(res[0]), min(res[1]), int((res[("cnt", 0)] != res[("cnt", 1)]).sum()), H, int(res[("cnt", 1)].max())))
