"""Workgroup kernels (TFF_OPT_KERNEL = 2) against the fused single-wavefront kernels (1) of the iterative TFT methods over N: where the launcher's
crossover belongs.  python tools/ab_wg_fused.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = api.Context(0)
for method in ("ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "PiPoseEstimation", "PiColPoseEstimation"):
    for N in (12, 25, 40, 60, 80, 100, 128, 160):
        C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=N)
        d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
        line = "%-26s N=%4d:" % (method, N)
        for variant in (2, 1):
            ctx.set_kernel_variant(variant)
            for _ in range(2):
                ctx.pose_batch(method, d, calm, reconst=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                out = ctx.pose_batch(method, d, calm, reconst=False)
            torch.cuda.synchronize()
            line += "  %s %.3f ms" % ("workgroup" if variant == 2 else "fused", (time.perf_counter() - t0) / 5 * 1e3)
        print(line, flush=True)
