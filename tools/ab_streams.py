"""Throughput of K back-to-back LinearTFT batches issued on one stream against round-robin on two / three streams (one context per stream):
the next batch's wavefronts fill the wave slots that the tail of the previous one leaves idle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
for S in (1, 2, 3):
    ctxs = [api.Context(0) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [None] * S
    def step(k):
        with torch.cuda.stream(streams[k % S]):
            outs[k % S] = ctxs[k % S].pose_batch("LinearTFTPoseEstimation", d, calm, reconst=False)
    for k in range(6):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bad = sum(int((o["status"] != 0).sum()) for o in outs)
    print("%d stream(s): %.3f ms per batch of %d x %d = %.2f M triplets/s (failed %d)" % (S, 1e3 * dt / K, B, N, K * B / dt / 1e6, bad))
