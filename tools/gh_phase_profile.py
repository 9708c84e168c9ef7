"""Shader-clock shares of the first Gauss-Helmert iteration (debug entry point): k_gh_block<Model> (default) or, with a third
argument 1, the fused single-wavefront k_gh_tft_pose<Model>.   python tools/gh_phase_profile.py [B] [N] [variant] [method]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
names = ["model.eval (T, D, C, g)", "W = BB', max eigenvalue", "W+ (Jacobi 4x4), w", "10 sweeps: Ghat, ghat", "Ghat -> Y = Ghat D -> M", "KKT solve",
         "v = -B'W+(A dt - w), obj"]
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
method = sys.argv[4] if len(sys.argv) > 4 else "ResslTFTPoseEstimation"
ctx = api.Context(0)
ctx.set_kernel_variant(variant)
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
out = ctx.pose_batch(method, d, calm, reconst=False, debug=True)
torch.cuda.synchronize()
dbg = out["debug"].cpu().numpy()
st = dbg[:, 120:128]
dt = np.diff(st, axis=1)
tot = st[:, 7] - st[:, 0]
print("%s, variant %d, N = %d: first GH iteration %.0f cycles (mean iterations %.2f)" % (method, variant, N, tot.mean(), out["iter"].double().mean().item()))
if variant == 0:
    s4 = dbg[:, 116:120]
    print("  k_gh_block: model.init %.0f, reprojection %.0f, whole iteration loop %.0f cycles" % ((s4[:, 1] - s4[:, 0]).mean(), (s4[:, 2] - s4[:, 1]).mean(), (s4[:, 3] - s4[:, 2]).mean()))
for k, nme in enumerate(names):
    print("  %-28s %9.0f  %5.1f%%" % (nme, dt[:, k].mean(), 100 * dt[:, k].mean() / tot.mean()))
if variant == 0:
    s3 = dbg[:, 110:113]
    if s3[:, 2].max() > 0:
        print("  inside '10 sweeps': strong-direction terms per correspondence %.0f, their sums %.0f cycles" % ((s3[:, 1] - s3[:, 0]).mean(), (s3[:, 2] - s3[:, 1]).mean()))
