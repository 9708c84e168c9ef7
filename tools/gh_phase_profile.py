"""Shader-clock shares of the first Gauss-Helmert iteration of k_gh_tft_pose<ResslModel> (debug entry point)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
names = ["model.eval (T, D, C, g)", "W = BB', max eigenvalue", "W+ (Jacobi 4x4), w", "10 sweeps: Ghat, ghat", "Ghat -> Y = Ghat D -> M", "KKT solve",
         "v = -B'W+(A dt - w), obj"]
ctx = api.Context(0)
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
out = ctx.pose_batch("ResslTFTPoseEstimation", d, calm, reconst=False, debug=True)
torch.cuda.synchronize()
dbg = out["debug"].cpu().numpy()
st = dbg[:, 120:128]
dt = np.diff(st, axis=1)
tot = st[:, 7] - st[:, 0]
lin = dbg[:, 80 + 13] - dbg[:, 80]
print("N = %d: first GH iteration %.0f cycles/wave (mean iterations %.2f); stamps 0..13 of the linear stage span %.0f" % (N, tot.mean(), out["iter"].double().mean().item(), lin.mean()))
for k, nme in enumerate(names):
    print("  %-28s %9.0f  %5.1f%%" % (nme, dt[:, k].mean(), 100 * dt[:, k].mean() / tot.mean()))
