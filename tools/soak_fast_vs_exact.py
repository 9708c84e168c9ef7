"""Soak (GPU box): the default route (fast tiers, exact kernel on demand) against the exact kernel (TFF_OPT_SOLVER = 1) for the three linear methods
over the geometry sweeps of experiments.m (angle between the camera centres 0 ... 180 degrees, focal length, noise, N).  Both sides run on the
GPU, so the batches can be large; a deviation beyond ~1e-9 means a fast tier delivered a result it should have handed over."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
fast, exact = api.Context(0), api.Context(0, solver="jacobi")
fast.set_rows(1)                         # the row kernels whatever the batch size (the default goes by batch size)
exact.set_rows(0)                        # the one-triplet exact kernel for all
B = int(os.environ.get("SOAK_B", "400"))
def dev(a, b):
    T = np.minimum(np.abs(a["T"] - b["T"]).reshape(B, -1).max(axis=1), np.abs(a["T"] + b["T"]).reshape(B, -1).max(axis=1)) / np.abs(b["T"]).reshape(B, -1).max(axis=1)
    R = np.maximum(np.abs(a["R_t_2"] - b["R_t_2"]).reshape(B, -1).max(axis=1), np.abs(a["R_t_3"] - b["R_t_3"]).reshape(B, -1).max(axis=1))
    return np.maximum(T, R)
t0 = time.time(); worst = {}
cells = [(N, noise, f, ang) for N in (12, 25, 100, 200) for noise in (0.0, 1.0, 3.0) for f, ang in ((50.0, None), (20.0, None), (300.0, None), (50.0, 60), (50.0, 120), (50.0, 160), (50.0, 175), (50.0, 180))]
for N, noise, f, ang in cells:
    C, CalM, _, _ = generate_scene_batch(B, N, noise=noise, seed=int(1000 * noise) + N + int(f) + (0 if ang is None else ang), focalL=f, angle=ang)
    for m in ("LinearTFTPoseEstimation", "LinearFPoseEstimation", "OptimFPoseEstimation"):
        a, e = fast.pose_batch(m, C, CalM, reconst=False), exact.pose_batch(m, C, CalM, reconst=False)
        okb = (a["status"] == 0) & (e["status"] == 0)
        d = dev(a, e)[okb]
        key = (m, "collinear-ish" if (ang is not None and ang >= 160) else "generic")
        w = worst.get(key, (0.0, None, 0))
        worst[key] = (max(w[0], d.max() if d.size else 0.0), (N, noise, f, ang) if (d.size and d.max() > w[0]) else w[1], w[2] + int((a["status"] != e["status"]).sum()))
print("B = %d per cell, %d cells, %.0f s" % (B, len(cells), time.time() - t0))
for k, (v, where, ns) in sorted(worst.items()):
    print("%-26s %-14s worst |default - exact| %.2e at (N, noise, focal, angle) = %s; status differences %d" % (k[0], k[1], v, where, ns))
