"""Minimal samples where the HIP result differs from the oracle: compare the cheirality votes of the four candidates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import tft_oracle as O
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
from helpers import rel_err_T, rel_err
ctx = api.Context(0)
import torch
meth = sys.argv[1] if len(sys.argv) > 1 else "LinearTFTPoseEstimation"
N, noise, B = (int(sys.argv[2]) if len(sys.argv) > 2 else 7), 2.0, 300
C, CalM, _, _ = generate_scene_batch(B, N, noise=noise, seed=1000 + 7 * N + int(10 * noise))
out = ctx.pose_batch(meth, torch.from_numpy(C).cuda(), torch.from_numpy(CalM).cuda(), reconst=True, debug=True)
out = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in out.items()}
nbad = 0
for b in range(B):
    R2, R3, Rec, T, _ = getattr(O, meth)(C[b].T.copy(), CalM)
    eT = rel_err_T(out["T"][b], T)
    e = max(eT, rel_err(out["R_t_2"][b], R2), rel_err(out["R_t_3"][b], R3))
    if e > 1e-6:
        nbad += 1
        print("triplet", b, "errT %.2e  errR2 %.2e errR3 %.2e" % (eT, rel_err(out["R_t_2"][b], R2), rel_err(out["R_t_3"][b], R3)))
        print("  kernel votes", out["debug"][b, 60:68])
        if meth.startswith("LinearTFT"):
            dbg = O.R_t_from_TFT(T, CalM, C[b].T.copy(), return_debug=True)
            print("  oracle votes", dbg[2]["votes2"], dbg[2]["votes3"])
        else:
            K1, K2, K3 = CalM[0:3], CalM[3:6], CalM[6:9]
            x = C[b].T.copy()
            n1, N1 = O.Normalize2Ddata(x[0:2]); n2, N2 = O.Normalize2Ddata(x[2:4]); n3, N3 = O.Normalize2Ddata(x[4:6])
            F21 = N2.T @ O.linearF(n1, n2) @ N1; F31 = N3.T @ O.linearF(n1, n3) @ N1
            P1 = np.hstack([K1, np.zeros((3, 1))])
            print("  oracle votes", O._recover_R_t_core(K2.T @ F21 @ K1, P1, K2, x[0:2], x[2:4], True)[2], O._recover_R_t_core(K3.T @ F31 @ K1, P1, K3, x[0:2], x[4:6], True)[2])
print("bad", nbad)
