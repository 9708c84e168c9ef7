"""Time per launch of the two LinearTFT fast kernels over a range of batch sizes (wave-slot quantisation: 2048 resident wavefronts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
METHOD = os.environ.get("METHOD", "LinearTFTPoseEstimation")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
Bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2048, 4096, 8192, 10000, 12288, 16384, 24576, 32768, 65536]
C, CalM, _, _ = generate_scene_batch(max(Bs), N, noise=1.0, seed=1)
dall = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
ctx = api.Context(0)
for B in Bs:
    d = dall[:B].contiguous()
    line = "%s B=%6d N=%d:" % (METHOD[:8], B, N)
    for rows in (1, 0):
        ctx.set_rows(rows)
        for _ in range(3):
            out = ctx.pose_batch(METHOD, d, calm, reconst=False)
        torch.cuda.synchronize()
        K = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            out = ctx.pose_batch(METHOD, d, calm, reconst=False)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        line += "  rows=%d %.3f ms %.2f M/s" % (rows, ms, B / ms / 1e3)
    print(line)
