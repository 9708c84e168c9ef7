"""Throughput of the pose methods over the north-star range of correspondences per triplet (100 ... 1000), device-resident batch."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
methods = sys.argv[2].split(",") if len(sys.argv) > 2 else ["LinearTFTPoseEstimation", "LinearFPoseEstimation", "ResslTFTPoseEstimation", "OptimFPoseEstimation"]
stage = int(os.environ.get("STAGE", "-1"))
ctx = api.Context(0, stage_lds=stage); lib = ctx.lib
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
print("| N | " + " | ".join(m.replace("PoseEstimation", "") + " M/s (GB/s)" for m in methods) + " |")
print("|---|" + "---|" * len(methods))
for N in [int(x) for x in os.environ.get("NS", "100,200,500,1000").split(",")]:
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(np.ascontiguousarray(CalM.T).reshape(27)).cuda()
    Rt2 = torch.empty(B * 12, dtype=torch.float64, device="cuda"); Rt3 = torch.empty_like(Rt2)
    T = torch.empty(B * 27, dtype=torch.float64, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
    row = []
    for m in methods:
        fn = getattr(lib, api.POSE_METHODS[m] + "_dev")
        args = (ctx.handle, p(d), p(calm), 0, B, N, p(Rt2), p(Rt3), p(T), None, p(it), p(st))
        for _ in range(3):
            assert fn(*args) == 0, lib.tff_last_error()
        torch.cuda.synchronize()
        K = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            fn(*args)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        rate = B / ms * 1e3
        row.append("%.2f (%.0f)" % (rate / 1e6, rate * (48 * N + 624) / 1e9))
        assert int((st != 0).sum()) == 0
    print("| %d | " % N + " | ".join(row) + " |", flush=True)
