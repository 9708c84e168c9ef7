"""Time k_linear_tft_pose for several builds of the library (tft_vs_fund_amd/variants/*.so) and LDS-staging modes."""
import sys, os, glob, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
B, N = 10000, 200
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
ref = None
libs = sorted(glob.glob(os.path.join(os.path.dirname(api.__file__), "variants", "*.so"))) or [None]
for lib in libs:
    for stage in (1, 0):
        ctx = api.Context(0, stage_lds=stage, lib_path=lib)
        for _ in range(3):
            out = ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 20
        e0.record()
        for _ in range(K):
            out = ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=False)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        T = out["T"].cpu().numpy()
        if ref is None: ref = T
        dev = np.abs(np.abs(T) - np.abs(ref)).max()
        print("%-28s stage_lds=%d  %.3f ms/batch  %.3e triplets/s  bad=%d  dev_vs_first=%.1e" % (
            os.path.basename(lib) if lib else "default", stage, ms, B / ms * 1e3, int((out["status"] != 0).sum()), dev), flush=True)
