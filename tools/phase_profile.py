"""Per-phase shader-clock shares of k_linear_tft_pose (debug entry point), B x N batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
names = ["stage+calm->normalise", "moments", "gram27", "eig27", "epipoles", "Gp build", "eig15+t", "P/misc",
         "transform x2+epi2+E", "svd3+cands", "votes (+ scale sums)", "t3 scale pass", "reconst/finish"]
ctx = api.Context(0)
ctx.set_rows(1)
if os.environ.get("TFF_ADAPTIVE", "1") != "0":
    ctx.set_debug_adaptive(True)           # production vote logic (TFF_ADAPTIVE=0: all four candidates, what the debug record otherwise holds)
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
for rec in (False, True):
    out = ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=rec, debug=True)
    torch.cuda.synchronize()
    dbg = out["debug"].cpu().numpy()
    st = dbg[:, 80:94]
    dt = np.diff(st, axis=1)
    tot = st[:, 13] - st[:, 0]
    print("reconst", rec, "mean cycles/wave %.0f  (its27 mean %.2f, its15 mean %.2f)" % (tot.mean(), dbg[:, 69].mean(), dbg[:, 70].mean()))
    print("   fall-backs to one-sided Jacobi: 27-column solve %d, 15-column solve %d of %d" % (((dbg[:, 69] % 10000) >= 1000).sum(), ((dbg[:, 70] % 10000) >= 1000).sum(), B))
    for k, nme in enumerate(names):
        print("  %-24s %9.0f  %5.1f%%" % (nme, dt[:, k].mean(), 100 * dt[:, k].mean() / tot.mean()))
