"""A/B timing of LinearTFT kernel variants: library builds (tft_vs_fund_amd/variants/*.so) x kernel (paired / single) x LDS staging."""
import sys, os, glob, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
B, N = 10000, 200
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(np.ascontiguousarray(CalM.T).reshape(27)).cuda()
Rt2 = torch.empty(B * 12, dtype=torch.float64, device="cuda"); Rt3 = torch.empty_like(Rt2)
T = torch.empty(B * 27, dtype=torch.float64, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
libs = sorted(glob.glob(os.path.join(os.path.dirname(api.__file__), "variants", "*.so"))) or [None]
for lib in libs:
    for kern in (0, 1):
        for stage in (1, 0):
            ctx = api.Context(0, stage_lds=stage, lib_path=lib)
            ctx.set_kernel_variant(kern)
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            fn = ctx.lib.tff_linear_tft_pose_batch_dev
            args = (ctx.handle, p(d), p(calm), 0, B, N, p(Rt2), p(Rt3), p(T), None, None, p(st))
            for _ in range(5):
                assert fn(*args) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                fn(*args)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 30
            print("%-12s kernel=%s stage_lds=%d  %.3f ms  %.3e triplets/s  bad=%d" % (os.path.basename(lib) if lib else "default", "pair" if kern == 0 else "single", stage, ms, B / ms * 1e3, int((st != 0).sum())), flush=True)
