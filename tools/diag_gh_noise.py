"""Deviation of the Gauss-Helmert kernels from the dense LAPACK oracle, Cholesky path vs eigen-decomposition path (diagnostic)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
from oracle import tft_oracle as O
from helpers import rel_err_T, rel_err
ctx = api.Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for N in (12, 60, 200):
    C, CalM, Rt0, _ = generate_scene_batch(B, N, noise=1.0, seed=4242 + N)
    for meth in ("ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "PiPoseEstimation"):
        ref = [getattr(O, meth)(C[b].T.copy(), CalM)[:5] for b in range(B)]
        for exact in (True, False):
            ctx.set_gh_exact(exact)
            out = ctx.pose_batch(meth, C, CalM, reconst=True)
            dev = np.array([max(rel_err_T(out["T"][b], ref[b][3]), rel_err(out["R_t_3"][b], ref[b][1])) for b in range(B)])
            dit = np.array([int(out["iter"][b]) - ref[b][4] for b in range(B)])
            same = dit == 0
            q = lambda a: "-" if a.size == 0 else "med %.1e p90 %.1e max %.1e" % (np.median(a), np.quantile(a, 0.9), a.max())
            print("N %3d %-26s %-8s same-iter %2d/%d: %s | other: %s | dit range [%d,%d]" % (
                N, meth, "eig" if exact else "chol", same.sum(), B, q(dev[same]), q(dev[~same]), dit.min(), dit.max()), flush=True)
ctx.set_gh_exact(False)
