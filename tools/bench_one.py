"""Run one pose method K times on the 10k x 200 batch (for rocprofv3 --kernel-trace --stats)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
method = sys.argv[1]; K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 10000; N = int(sys.argv[4]) if len(sys.argv) > 4 else 200
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
ctx = api.Context(0)
if os.environ.get("TFF_SPILL_ONLY_IF_NEEDED"):
    ctx.set_spill_only_if_needed(True)     # per-correspondence state stays in LDS whenever it fits (A/B)
import time
for _ in range(3):
    out = ctx.pose_batch(method, d, calm, reconst=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    out = ctx.pose_batch(method, d, calm, reconst=False)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / K * 1e3
print("%.3f ms per batch, %.2f M triplets/s" % (ms, B / ms / 1e3))
print(method, "bad", int((out["status"] != 0).sum()), "mean iter", float(out["iter"].double().mean()))
