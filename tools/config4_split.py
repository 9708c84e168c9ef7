"""config 4 on one GPU, the two launches timed separately (HIP events): minimal-sample pose hypotheses, then inlier counts."""
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
for method, n in (("LinearTFTPoseEstimation", 7), ("LinearFPoseEstimation", 8)):
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    idx = torch.rand((H, Ns), device="cuda", generator=g).argsort(dim=1)[:, :n].to(torch.int32).contiguous()
    for rep in range(3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        hyp = ctx.pose_sampled(method, d_scene, d_calm, idx)
        e[1].record()
        cnt = ctx.inlier_count(d_scene, d_calm, hyp["R_t_2"], hyp["R_t_3"], 1.0)
        e[2].record()
        torch.cuda.synchronize()
    cnt_full, _ = ctx.inlier_count(d_scene, d_calm, hyp["R_t_2"], hyp["R_t_3"], 1.0, with_error=True)
    print("   count-only vs full-triangulation counts differ in %d of %d hypotheses" % (int((cnt_full != cnt).sum()), H))
    print("%-26s pose %.2f ms  inlier count %.2f ms  (%d hypotheses, best %d)" % (method, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), H, int(cnt.max())))
