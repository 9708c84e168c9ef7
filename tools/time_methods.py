"""ms per 10 000 x N batch of the iterative methods (HIP events around K calls).  python tools/time_methods.py [N] [method ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
methods = sys.argv[2:] or ["ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "FaugPapaTFTPoseEstimation", "PiPoseEstimation", "PiColPoseEstimation", "OptimFPoseEstimation"]
B = 10000
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
ctx = api.Context(0, stage_lds=int(os.environ.get("TFF_STAGE", "-1")))
ctx.set_kernel_variant(int(os.environ.get("TFF_VARIANT", "0")))      # 1: fused single-wavefront kernels (A/B)
ctx.set_spill_only_if_needed(int(os.environ.get("TFF_SPILL", "0")))  # 1: state stays in LDS whenever it fits
ctx.set_rows(int(os.environ.get("TFF_ROWS", "1")))                   # 0: one triplet per wavefront in the linear stage (A/B)
for m in methods:
    for _ in range(2):
        out = ctx.pose_batch(m, d, calm, reconst=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 5
    e0.record()
    for _ in range(K):
        out = ctx.pose_batch(m, d, calm, reconst=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print("%-28s N=%d  %.3f ms per batch  %.2f M triplets/s  (bad %d, mean iter %.2f)" % (m, N, ms, B / ms / 1e3, int((out["status"] != 0).sum()), float(out["iter"].double().mean())))
