"""Gauss-Helmert methods at small N (experiments.m's default N = 12): workgroup path vs fused single-wavefront kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
ctx = api.Context(0)
for N in [int(x) for x in os.environ.get("NS", "12,40,64,100,130").split(",")]:
    C, CalM, _, _ = generate_scene_batch(20000, N, noise=1.0, seed=1)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    for meth in os.environ.get("METHODS", "ResslTFTPoseEstimation,NordbergTFTPoseEstimation,PiPoseEstimation,FaugPapaTFTPoseEstimation,PiColPoseEstimation").split(","):
        row = []
        for variant in (2, 1, 0):
            ctx.set_kernel_variant(variant)
            for _ in range(2):
                ctx.pose_batch(meth, d, calm, reconst=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ctx.pose_batch(meth, d, calm, reconst=False)
            e1.record(); torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 5)
        print("N %3d %-26s workgroup %8.3f ms   fused %8.3f ms   automatic %8.3f ms   (20000 triplets)" % (N, meth, row[0], row[1], row[2]), flush=True)
ctx.set_kernel_variant(0)
