"""Writes the kernel's outputs on tests/golden/gh_mp_nordberg.npz (all cases) to gpurun_out/nordberg_gpu_outputs.npz (diagnostics)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tft_vs_fund_amd import api
ctx = api.Context(0)
g = np.load(os.path.join(ROOT, "tests", "golden", "gh_mp_nordberg.npz"))
out = {}
for ci in range(3):
    pre = "c%d_" % ci
    for method, variant in (("NordbergTFTPoseEstimation", 0), ("LinearTFTPoseEstimation", 0), ("NordbergTFTPoseEstimation", 1)):
        ctx.set_kernel_variant(variant)
        o = ctx.pose_batch(method, g[pre + "Corresp"], g[pre + "CalM"], reconst=False)
        ctx.set_kernel_variant(0)
        for k in ("T", "R_t_2", "R_t_3", "iter", "status"):
            out[pre + method[:4] + ("_v1" if variant else "") + "_" + k] = np.asarray(o[k])
    # the same scenes one at a time (B = 1 launches)
    Ts = []
    for b in range(g[pre + "Corresp"].shape[0]):
        o = ctx.pose_batch("NordbergTFTPoseEstimation", g[pre + "Corresp"][b:b + 1], g[pre + "CalM"], reconst=False)
        Ts.append(np.asarray(o["T"])[0])
    out[pre + "Nord_single_T"] = np.stack(Ts)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "nordberg_gpu_outputs.npz"), **out)
print("written")
