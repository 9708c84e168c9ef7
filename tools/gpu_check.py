"""Quick on-GPU sanity run: LinearTFTPoseEstimation vs the oracle + a rough timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
from oracle import tft_oracle as O

ctx = api.Context(0)
for solver in ("invit", "jacobi"):
    ctx.set_solver(solver)
    for N, noise in ((200, 1.0), (12, 1.0), (7, 1.0), (100, 0.0), (1000, 3.0)):
        B = 6
        C, CalM, Rt0, X = generate_scene_batch(B, N, noise=noise, seed=7)
        out = ctx.pose_batch("LinearTFTPoseEstimation", C, CalM, reconst=True)
        worst = 0
        for b in range(B):
            o2, o3, orec, oT, _ = O.LinearTFTPoseEstimation(C[b].T.copy(), CalM)
            s = np.sign(np.sum(out["T"][b] * oT))
            e = max(np.abs(s * out["T"][b] - oT).max(), np.abs(out["R_t_2"][b] - o2).max(), np.abs(out["R_t_3"][b] - o3).max(),
                    np.abs(out["Reconst"][b] - orec).max() / np.abs(orec).max())
            worst = max(worst, e)
        print(solver, N, noise, "status", out["status"], "worst err %.2e" % worst, flush=True)
ctx.set_solver("invit")
B, N = 10000, 200
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
d = torch.from_numpy(C).cuda()
calm = torch.from_numpy(CalM).cuda()
for rec in (False, True):
    for _ in range(2):
        out = ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=rec)
    torch.cuda.synchronize()
    t0 = time.time()
    K = 5
    for _ in range(K):
        out = ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=rec)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / K
    print("reconst", rec, "B=%d N=%d: %.3f ms/batch  %.3e triplets/s  status!=0: %d" % (B, N, dt * 1e3, B / dt, int((out["status"] != 0).sum())), flush=True)
