"""Shader-clock shares of the exact row kernels on config-4 samples (minimal samples of ONE scene with 25 % gross outliers, gathered into a batch so that
the debug entry point can stamp them): python tools/config4_phase_profile.py [H]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
H = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
Ns = 400
Cs, CalM, _, _ = generate_scene_batch(1, Ns, noise=0.5, seed=77)
scene = Cs[0].copy()
rng = np.random.default_rng(5)
bad = rng.choice(Ns, Ns // 4, replace=False)
scene[bad, 2:6] += rng.uniform(20, 80, size=(bad.size, 4))
dev = torch.device("cuda:0")
d_scene = torch.from_numpy(scene).to(dev); calm = torch.from_numpy(CalM).to(dev)
ctx = api.Context(0)
ctx.set_rows(1)
names = ["centroids + distances", "middle (QR, inverse iterations)", "prepare (transform / SVDs / candidates)", "fast votes (all four)", "exact votes",
         "t3 scale pass", "T / stores"]
slots = [0, 1, 2, 10, 3, 11, 12, 13]
for method, n in (("LinearTFTPoseEstimation", 7), ("LinearFPoseEstimation", 8)):
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    idx = torch.rand((H, Ns), device=dev, generator=gen).argsort(dim=1)[:, :n]
    batch = d_scene[idx].contiguous()                                         # (H, n, 6)
    out = ctx.pose_batch(method, batch, calm, reconst=False, debug=True)
    torch.cuda.synchronize()
    dbg = out["debug"].cpu().numpy()
    st = dbg[:, [80 + s for s in slots]]
    good = (st[:, -1] > st[:, 0]) & (st[:, 0] > 0)
    st = st[good]
    dt = np.diff(st, axis=1)
    tot = st[:, -1] - st[:, 0]
    print("%s, %d-point samples, %d stamped hypotheses: %.0f cycles per wavefront pass (failed %d)" % (method, n, good.sum(), tot.mean(), int((out["status"] != 0).sum())))
    its = dbg[:, 69:71] % 10000
    H4 = (its.shape[0] // 4) * 4
    per_wave = its[:H4].reshape(-1, 4, 2).max(axis=1)                        # a wavefront iterates until its slowest row is done
    print("  inverse iterations (first / second solve): mean %.1f / %.1f per hypothesis, %.1f / %.1f per wavefront (max of its four rows); p99 %d / %d"
          % (its[:, 0].mean(), its[:, 1].mean(), per_wave[:, 0].mean(), per_wave[:, 1].mean(), np.percentile(its[:, 0], 99), np.percentile(its[:, 1], 99)))
    for k, nme in enumerate(names):
        print("  %-42s %9.0f  %5.1f%%" % (nme, dt[:, k].mean(), 100 * dt[:, k].mean() / tot.mean()))
    if n == 8:                                                                # inside the eight-point kernel's middle
        sm = dbg[good][:, [81, 84, 85, 86, 82]]
        for nme, v in zip(["linearF's own normalisation", "N x 9 system + QR + inverse iteration, pair (1,2)", "the same, pair (1,3)", "de-normalisation, rank 2, E (two positions)"], np.diff(sm, axis=1).T):
            print("      %-50s %9.0f  %5.1f%%" % (nme, v.mean(), 100 * v.mean() / tot.mean()))
    if n == 7:                                                                # inside the seven-point kernel's middle
        sm = dbg[good][:, [81, 84, 85, 86, 87, 82]]
        for nme, v in zip(["4N x 27 system + Householder QR", "inverse iteration, 27 columns", "epipoles + frames", "R Up + QR, 15 columns", "inverse iteration, 15 columns + t"], np.diff(sm, axis=1).T):
            print("      %-38s %9.0f  %5.1f%%" % (nme, v.mean(), 100 * v.mean() / tot.mean()))
