"""A/B of the two LinearTFT (or, 4th argument, LinearF) fast kernels on one GPU: four triplets per wavefront (TFF_OPT_ROWS = 1, default) against one triplet per
wavefront (0).  Prints per-launch time (HIP events around K launches), the agreement of the two routes and, for a few triplets,
the deviation from the oracle (test infrastructure).  Usage: python tools/ab_rows.py [B] [N] [K] [method]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
K = int(sys.argv[3]) if len(sys.argv) > 3 else 30
METHOD = sys.argv[4] if len(sys.argv) > 4 else "LinearTFTPoseEstimation"
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
ctx = api.Context(0)
res = {}
for reconst in (False, True):
    for rows in (1, 0):
        ctx.set_rows(rows)
        for _ in range(3):
            out = ctx.pose_batch(METHOD, d, calm, reconst=reconst)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            out = ctx.pose_batch(METHOD, d, calm, reconst=reconst)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        res[(reconst, rows)] = {k: (v.cpu().numpy() if v is not None else None) for k, v in out.items() if k != "_raw"}
        print("reconst=%d rows=%d: %.3f ms per %d x %d (incl. python call overhead) = %.2f M triplets/s, status!=0: %d" % (
            reconst, rows, ms, B, N, B / ms / 1e3, int((out["status"] != 0).sum())))
    a, b = res[(reconst, 1)], res[(reconst, 0)]
    sg = np.sign(np.sum(a["T"] * b["T"], axis=(1, 2, 3)))[:, None, None, None]
    print("  rows vs wave: T %.2e  R_t_2 %.2e  R_t_3 %.2e  status equal %s" % (
        np.abs(a["T"] * sg - b["T"]).max(), np.abs(a["R_t_2"] - b["R_t_2"]).max(),
        np.abs(a["R_t_3"] - b["R_t_3"]).max() / max(1.0, np.abs(b["R_t_3"]).max()), np.array_equal(a["status"], b["status"])))
    if reconst:
        print("  Reconst %.2e (relative)" % (np.abs(a["Reconst"] - b["Reconst"]).max() / np.abs(b["Reconst"]).max()))
try:
    from oracle import tft_oracle as O
    a = res[(True, 1)]
    worst = 0.0
    for b in list(range(6)) + [B - 1, B - 2, B - 3]:
        o2, o3, orec, oT, _ = getattr(O, METHOD)(C[b].T.copy(), CalM)
        s = np.sign(np.sum(a["T"][b] * oT))
        worst = max(worst, np.abs(s * a["T"][b] - oT).max(), np.abs(a["R_t_2"][b] - o2).max(), np.abs(a["R_t_3"][b] - o3).max(),
                    np.abs(a["Reconst"][b] - orec).max() / np.abs(orec).max())
    print("rows kernel vs oracle (9 triplets): %.2e" % worst)
except Exception as e:  # the oracle is test infrastructure; absent -> skip
    print("oracle check skipped:", e)
