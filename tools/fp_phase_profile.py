"""Shader-clock shares of the first Gauss-Helmert iteration of k_fp_block (FaugPapa, gh_fp_kernel.h) through the debug entry point.
python tools/fp_phase_profile.py [B] [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
names = ["constraints g, C (owner)", "W = BB', finite, tolerance", "deflated pinv, n'w, sq", "R and Hs: 19 butterflies", "block_any + combine", "basis Q (owner)",
         "rotated Gram: 13 butterflies, RQ", "Q'(RQ), combine", "assemble M'", "norm, Cholesky, solves (owner)", "Schur complement",
         "tridiagonal reduction (owner)", "counts + isolation (owner)", "twisted RQI (owner)", "x^, back-transform (owner)", "z1, dt = Qz (owner)", "v, obj"]
slots = [16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 32, 33, 34]
ctx = api.Context(0)
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
out = ctx.pose_batch("FaugPapaTFTPoseEstimation", d, calm, reconst=False, debug=True)
torch.cuda.synchronize()
dbg = out["debug"].cpu().numpy()
st = dbg[:, [80 + k for k in slots]]
dt = np.diff(st, axis=1)
tot = st[:, -1] - st[:, 0]
print("FaugPapa k_fp_block, N = %d: first GH iteration %.0f cycles of the shader clock (mean iterations %.2f, RQI rounds %.1f); whole loop %.0f" % (
    N, tot.mean(), out["iter"].double().mean().item(), dbg[:, 79].mean(), (dbg[:, 80 + 35] - st[:, 0]).mean()))
for k, nme in enumerate(names):
    print("  %-36s %9.0f  %5.1f%%" % (nme, dt[:, k].mean(), 100 * dt[:, k].mean() / tot.mean()))
