"""Kernel-level timing of every pose method on the configs[1]/[2] batch (device pointers, preallocated outputs)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(np.ascontiguousarray(CalM.T).reshape(27)).cuda()
ctx = api.Context(0); lib = ctx.lib
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
Rt2 = torch.empty(B * 12, dtype=torch.float64, device="cuda"); Rt3 = torch.empty_like(Rt2)
T = torch.empty(B * 27, dtype=torch.float64, device="cuda"); rec = torch.empty(B * 3 * N, dtype=torch.float64, device="cuda")
it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
variants = [(n, st, 0) for n, st in api.POSE_METHODS.items()] + [("LinearTFT (paired kernel, A/B)", "tff_linear_tft_pose_batch", 1)]
for name, stem, variant in variants:
    ctx.set_kernel_variant(variant)
    fn = getattr(lib, stem + "_dev")
    for with_rec in (False, True):
        args = (ctx.handle, p(d), p(calm), 0, B, N, p(Rt2), p(Rt3), p(T), p(rec) if with_rec else None, p(it), p(st))
        for _ in range(5):
            assert fn(*args) == 0, lib.tff_last_error()
        torch.cuda.synchronize()
        K = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            fn(*args)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        print("%-26s reconst=%d  %8.3f ms/batch  %.3e triplets/s  bad=%d  mean iter %.2f" % (
            name, with_rec, ms, B / ms * 1e3, int((st != 0).sum()), float(it.double().mean())), flush=True)
# BundleAdjustment from the linear TFT poses (N = 100 as experiments_real.m's samples, and the configs[1] batch)
for Bb, Nb in ((10000, 100), (B, N)):
    Cb, CalMb, _, _ = generate_scene_batch(Bb, Nb, noise=1.0, seed=2)
    db = torch.from_numpy(Cb).cuda(); cb = torch.from_numpy(CalMb).cuda()
    ctx.set_kernel_variant(0)
    lin = ctx.pose_batch("LinearTFTPoseEstimation", db, cb, reconst=True)
    r2, r3, x0 = lin["R_t_2"].contiguous(), lin["R_t_3"].contiguous(), lin["Reconst"].contiguous()
    for _ in range(3):
        out = ctx.bundle_adjust(cb, r2, r3, db, x0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = ctx.bundle_adjust(cb, r2, r3, db, x0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-26s N=%-4d    %8.3f ms/batch  %.3e triplets/s  bad=%d  mean iter %.2f" % ("BundleAdjustment", Nb, ms, Bb / ms * 1e3, int((out["status"] != 0).sum()),
                                                                                    float(out["iter"].double().mean())), flush=True)
