"""GPU side of the Nordberg-on-real-data study: re-create the (triplet, trial) problems of
results/real_fountain_trials.json (tft_vs_fund_amd.experiments.real_trials: same sampling and noise keys), run Nordberg and Ressl on them,
and save the trials whose Nordberg ReprError (all inliers, experiments_real.m:130-131) exceeds 50 px -- inputs and kernel outputs -- for
tools/nordberg_divergence_check.py (LAPACK oracle + 50-digit iteration, CPU).   python tools/nordberg_divergence_extract.py [n_trials] [keep]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd import experiments as E

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
keep = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ctx = api.Context(0)
trips = E.load_epfl_all(os.path.join(ROOT, "tests", "golden", "epfl_all.npz"), "fountain", 70)
sigma, seed, nsample = 0.5, 1, 100
rows = []
tot = 0
for ti, tr in enumerate(trips):
    Ci = E.epfl_inliers(ctx, tr)
    Ni = Ci.shape[1]
    n = min(nsample, Ni)
    if n < 7:
        continue
    S = np.empty((n_trials, n, 6))
    for k in range(n_trials):
        rng = np.random.Generator(np.random.Philox(key=[seed, 1000003 * ti + k]))
        sel = np.sort(rng.choice(Ni, size=n, replace=False))
        S[k] = Ci[:, sel].T + sigma * rng.standard_normal((n, 6))
    C = torch.from_numpy(S).cuda()
    CalB = torch.from_numpy(np.broadcast_to(tr["CalM"], (n_trials, 9, 3)).copy()).cuda()
    res = {}
    for m in ("NordbergTFTPoseEstimation", "ResslTFTPoseEstimation", "LinearTFTPoseEstimation"):
        out = ctx.pose_batch(m, C, CalB, reconst=False)
        st = out["status"].cpu().numpy(); R2 = out["R_t_2"].cpu().numpy(); R3 = out["R_t_3"].cpu().numpy()
        err = np.full(n_trials, np.inf)
        ok = st == 0
        P = E._cameras(tr["CalM"], R2[ok], R3[ok])
        err[ok] = E._np(ctx.repr_error(P, np.ascontiguousarray(Ci.T)))
        res[m] = (err, out["iter"].cpu().numpy(), R2, R3, out["T"].cpu().numpy(), st)
    tot += n_trials
    en = res["NordbergTFTPoseEstimation"][0]
    for k in np.nonzero(en > 50.0)[0]:
        rows.append(dict(triplet=ti, trial=int(k), name=tr["name"], S=S[k], CalM=tr["CalM"], Ci=Ci, n=n,
                         **{key + "_" + suf: res[m][j][k] for m, key in (("NordbergTFTPoseEstimation", "nord"), ("ResslTFTPoseEstimation", "ressl"), ("LinearTFTPoseEstimation", "lin"))
                            for j, suf in enumerate(("repr", "iter", "Rt2", "Rt3", "T", "status"))}))
print("%d of %d (triplet, trial) problems have Nordberg ReprError > 50 px" % (len(rows), tot))
by_trip = {}
for r in rows:
    by_trip.setdefault(r["triplet"], []).append(r)
print("   spread over %d triplets; per triplet: %s" % (len(by_trip), {t: len(v) for t, v in sorted(by_trip.items())}))
# keep a spread: the worst of as many different triplets as possible, sample size 100 only (one fixture shape)
pick = []
for t, v in sorted(by_trip.items(), key=lambda kv: -len(kv[1])):
    v = [r for r in v if r["n"] == 100]
    if v:
        pick.append(max(v, key=lambda r: r["nord_repr"] if np.isfinite(r["nord_repr"]) else 1e300))
pick = pick[:keep]
out = dict(n_total=np.array(tot), n_divergent=np.array(len(rows)), sigma=np.array(sigma), names=np.array([r["name"] for r in pick]),
           triplet=np.array([r["triplet"] for r in pick]), trial=np.array([r["trial"] for r in pick]),
           Corresp=np.stack([r["S"] for r in pick]), CalM=np.stack([r["CalM"] for r in pick]))
for key in ("nord", "ressl", "lin"):
    for suf in ("repr", "iter", "Rt2", "Rt3", "T", "status"):
        out["gpu_%s_%s" % (key, suf)] = np.stack([np.asarray(r[key + "_" + suf]) for r in pick])
# the inlier sets the ReprError is taken over (ragged): concatenated + offsets
out["inliers"] = np.concatenate([r["Ci"].T for r in pick]); out["inlier_offsets"] = np.cumsum([0] + [r["Ci"].shape[1] for r in pick])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "nordberg_divergent.npz"), **out)
for r in pick:
    print("  %-22s trial %3d: Nordberg repr %.3g px (iter %d)  Ressl %.3g px (iter %d)  Linear %.3g px" % (r["name"], r["trial"], r["nord_repr"], r["nord_iter"], r["ressl_repr"], r["ressl_iter"], r["lin_repr"]))
