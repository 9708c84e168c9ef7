"""
Batched reproduction of the reference's two experiment drivers:

  * experiments.m:28-141   -- synthetic sweeps over noise / focal length / number of points / collinearity
                              angle, `n_sim` scenes per interval value, every pose method;
  * experiments_real.m:25-148 -- EPFL triplets: 1-px inlier filter with the ground-truth cameras, a
                              100-correspondence sample per triplet, every (non-collinear) method,
                              ReprError on ALL inliers after re-triangulation.

The reference loops `for it=1:n_sim, for m=methods_to_test` calling one method on one triplet; here all the
triplets of an interval value go through one batched C-ABI call per method.  Recorded per (interval value,
method): mean ReprError (with the returned Reconst, experiments.m:112-114), mean rotation / translation
AngError over both poses (:117-120), mean iterations, time per triplet (GPU time of the batch / batch size
-- the reference records cputime per call).  Output: a JSON-serialisable dict.

Differences, stated: MATLAB's rng stream cannot be reproduced, scenes come from the Philox generator of
scenes.py (the reference draws N+100 points and keeps a random N: the same distribution as drawing N);
the bundle-adjustment columns (experiments.m:127-141, index 2 of the reference's result arrays) are
produced by `Context.bundle_adjust` under the keys `*_ba` (repr_err_ba is BundleAdjustment.m:105's residual norm in
normalised units, as the reference records it; its optimiser is MATLAB's closed-source lsqnonlin, so iteration counts
are those of this repo's Levenberg-Marquardt loop).

  python -m tft_vs_fund_amd.experiments --option noise --n-sim 20 --out noise.json
  python -m tft_vs_fund_amd.experiments --real tests/golden/epfl.npz --out real.json
"""
import argparse
import json
import os
import time

import numpy as np

from .metrics import AngError_batch
from .scenes import generate_scene_batch

# experiments.m:51-59 (same order, 1-based index m in the reference)
METHODS = ["LinearTFTPoseEstimation", "ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "FaugPapaTFTPoseEstimation",
           "PiPoseEstimation", "PiColPoseEstimation", "LinearFPoseEstimation", "OptimFPoseEstimation"]

# experiments.m:37-47
INTERVALS = {
    "noise": [0.25 * k for k in range(13)],
    "focal": list(range(20, 301, 20)),
    "points": [7, 8, 9, 10, 15, 20, 25],
    "angle": [166, 168, 170, 172, 174, 175, 176, 177, 178, 179, 179.5, 180],
}


def methods_to_test(option):
    """experiments.m:61-65: all eight for the collinearity sweep, otherwise everything but PiCol."""
    return list(range(8)) if option == "angle" else [0, 1, 2, 3, 4, 6, 7]


def _np(t):
    return t if isinstance(t, np.ndarray) else t.detach().cpu().numpy()


def _run_method(ctx, method, C, CalM, timer=None):
    """One batched call -> numpy outputs + seconds per triplet."""
    B = C.shape[0]
    if timer is None:
        t0 = time.perf_counter()
        out = ctx.pose_batch(method, C, CalM, reconst=True)
        dt = time.perf_counter() - t0
    else:
        out, dt = timer(lambda: ctx.pose_batch(method, C, CalM, reconst=True))
    return {k: (None if out[k] is None else _np(out[k])) for k in ("R_t_2", "R_t_3", "Reconst", "T", "iter", "status")}, dt / max(B, 1)


def _cuda_timer(torch):
    def run(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()                                            # warm-up (LDS attribute set-up, allocations)
        torch.cuda.synchronize()
        a.record()
        out = fn()
        b.record()
        torch.cuda.synchronize()
        return out, a.elapsed_time(b) * 1e-3
    return run


def _cameras(CalM, R2, R3):
    B = R2.shape[0]
    P = np.empty((B, 3, 3, 4))
    P[:, 0] = CalM[0:3] @ np.eye(3, 4)
    P[:, 1] = np.einsum("ij,bjk->bik", CalM[3:6], R2)
    P[:, 2] = np.einsum("ij,bjk->bik", CalM[6:9], R3)
    return P


def _scatter(ok, values):
    out = np.full(ok.shape[0], np.inf)
    out[ok] = values
    return out


def evaluate_batch(ctx, method, C, CalM, R_t0, device=True, with_ba=False):
    """All scenes of one interval value through one method.  Returns the per-triplet metrics."""
    B, N, _ = C.shape
    timer = None
    Cin, Kin = C, CalM
    if device:
        import torch
        Cin, Kin = torch.from_numpy(C).cuda(), torch.from_numpy(CalM).cuda()
        timer = _cuda_timer(torch)
    out, sec = _run_method(ctx, method, Cin, Kin, timer)
    ok = out["status"] == 0
    R2, R3 = out["R_t_2"], out["R_t_3"]
    repr_err = np.full(B, np.inf)
    if ok.any():
        P = _cameras(CalM, R2[ok], R3[ok])
        repr_err[ok] = _np(ctx.repr_error(P, C[ok], out["Reconst"][ok]))                        # experiments.m:112-114
    dummy = np.hstack([np.eye(3), np.ones((3, 1))])                                              # placeholder for triplets without a pose
    r2, t2 = AngError_batch(R_t0[0], np.where(ok[:, None, None], R2, dummy))                     # :117-118
    r3, t3 = AngError_batch(R_t0[1], np.where(ok[:, None, None], R3, dummy))
    rot = np.where(ok, (r2 + r3) / 2, np.inf)
    tr = np.where(ok, (t2 + t3) / 2, np.inf)
    res = dict(repr_err=repr_err, rot_err=rot, t_err=tr, iter=out["iter"].astype(np.float64), ok=ok, seconds_per_triplet=sec)
    if with_ba and ok.any() and hasattr(ctx, "bundle_adjust"):
        # BundleAdjustment(CalM, [eye(3,4); R_t_2; R_t_3], Corresp, Reconst)   (experiments.m:127-129)
        call = lambda: ctx.bundle_adjust(CalM, R2[ok], R3[ok], C[ok], out["Reconst"][ok])
        if device:
            import torch
            ba, ba_sec = _cuda_timer(torch)(call)
        else:
            t0 = time.perf_counter(); ba = call(); ba_sec = time.perf_counter() - t0
        bR2, bR3 = _np(ba["R_t_2"]), _np(ba["R_t_3"])
        a2, b2 = AngError_batch(R_t0[0], bR2); a3, b3 = AngError_batch(R_t0[1], bR3)                # :134-137
        full = lambda v: _scatter(ok, v)
        res.update(repr_err_ba=full(_np(ba["repr_err"])), rot_err_ba=full((a2 + a3) / 2), t_err_ba=full((b2 + b3) / 2),
                   iter_ba=full(_np(ba["iter"]).astype(np.float64)), seconds_per_triplet_ba=ba_sec / max(int(ok.sum()), 1))
    return res


def synthetic_sweep(ctx, option="noise", n_sim=20, N=12, noise=1.0, f=50.0, angle=0.0, interval=None, methods=None,
                    seed0=1, device=True, with_ba=True):
    """experiments.m:28-125 for one `option`.  Returns a dict of (len(interval) x 8) lists (means over the simulations
    that returned a pose; `failed` counts the others), inf where the reference records inf (too few points)."""
    interval = list(INTERVALS[option] if interval is None else interval)
    mt = methods_to_test(option) if methods is None else list(methods)
    keys = ("repr_err", "rot_err", "t_err", "iter", "time")
    if with_ba:
        keys = keys + tuple(k + "_ba" for k in keys)
    res = {k: np.zeros((len(interval), len(METHODS))) for k in keys}
    failed = np.zeros((len(interval), len(METHODS)), dtype=np.int64)
    for i, val in enumerate(interval):
        Ni, noisei, fi, anglei = N, noise, f, angle
        if option == "noise":
            noisei = float(val)
        elif option == "focal":
            fi = float(val)
        elif option == "points":
            Ni = int(val)
        elif option == "angle":
            anglei = float(val)
        # one scene per simulation `it` (experiments.m:93-96); scenes.py: angle < 70 means no collinearity, as the reference
        C, CalM, R_t0, _ = generate_scene_batch(n_sim, Ni, noise=noisei, seed=seed0 + 1000 * i, focalL=fi, angle=anglei)
        for m in mt:
            if (m > 5 and Ni < 8) or Ni < 7:                                                    # experiments.m:99-104
                for k in keys:
                    res[k][i, m] = np.inf
                continue
            ev = evaluate_batch(ctx, METHODS[m], C, CalM, R_t0, device, with_ba)
            ok = ev["ok"]
            failed[i, m] = int((~ok).sum())
            for k in ("repr_err", "rot_err", "t_err", "iter"):
                res[k][i, m] = float(np.mean(ev[k][ok])) if ok.any() else np.inf
                if with_ba:
                    res[k + "_ba"][i, m] = float(np.mean(ev[k + "_ba"][ok])) if (ok.any() and k + "_ba" in ev) else np.inf
            res["time"][i, m] = ev["seconds_per_triplet"]
            if with_ba:
                res["time_ba"][i, m] = ev.get("seconds_per_triplet_ba", np.inf)
    out = dict(option=option, interval=[float(v) for v in interval], methods=METHODS, methods_tested=[METHODS[m] for m in mt],
               n_sim=n_sim, N=N, noise=noise, focal=f, angle=angle, failed=failed.tolist(),
               note="means over the simulations that returned a pose; time = seconds per triplet of the batched call; "
                    "*_ba: after BundleAdjustment (repr_err_ba in normalised units, as experiments.m:131 records it)")
    out.update({k: v.tolist() for k, v in res.items()})
    return out


# ---- experiments_real.m ------------------------------------------------------------------------------------------------
def load_epfl_fixture(npz_path):
    """The committed EPFL fixture (tests/golden/epfl.npz): eight triplets with all their correspondences."""
    g = np.load(npz_path)
    trips = []
    for n in range(int(g["count"])):
        pre = "t%d_" % n
        trips.append(dict(name=str(g[pre + "name"]), Corresp=np.asarray(g[pre + "Corresp_all"]), CalM=np.asarray(g[pre + "CalM"]),
                          R_t0=[np.asarray(g[pre + "Rt0"][0]), np.asarray(g[pre + "Rt0"][1])]))
    return trips


def load_epfl_dataset(path_to_data, n_triplets):
    """Data/<dataset>/Corresp_triplets.mat + *.camera files, as experiments_real.m:39-91 reads them (needs scipy and the
    user's copy of the reference's Data directory)."""
    import scipy.io
    m = scipy.io.loadmat(os.path.join(path_to_data, "Corresp_triplets.mat"))
    names = [str(x[0]) for x in m["im_names"].ravel()]
    order = np.asarray(m["indexes_sorted"])[:n_triplets, 0:3].astype(int)

    def cam(name):                                      # Data/readCalibrationOrientation_EPFL.m
        with open(os.path.join(path_to_data, name + ".camera")) as fh:
            rows = [[float(v) for v in line.split()] for line in fh.read().strip().splitlines()]
        K = np.array(rows[0:3]); R = np.array(rows[4:7]).T
        return K, R, -R @ np.array(rows[7])
    trips = []
    for (i1, i2, i3) in order:
        Corresp = np.ascontiguousarray(m["Corresp"][i1 - 1, i2 - 1, i3 - 1].T)
        (K1, R1, t1), (K2, R2, t2), (K3, R3, t3) = cam(names[i1 - 1]), cam(names[i2 - 1]), cam(names[i3 - 1])
        trips.append(dict(name="(%d,%d,%d)" % (i1, i2, i3), Corresp=Corresp, CalM=np.vstack([K1, K2, K3]),
                          R_t0=[np.hstack([R2 @ R1.T, (t2 - R2 @ R1.T @ t1).reshape(3, 1)]),
                                np.hstack([R3 @ R1.T, (t3 - R3 @ R1.T @ t1).reshape(3, 1)])]))
    return trips


def real_sweep(ctx, triplets, initial_sample_size=100, repr_err_th=1.0, methods=None, seed0=1, bundle_adj_size=50):
    """experiments_real.m:76-138 over `triplets` (dicts with Corresp 6xN, CalM, R_t0).  Per triplet and method:
    ReprError over all inliers (re-triangulated, :130-131), AngError, iterations, time."""
    mt = [0, 1, 2, 3, 4, 6, 7] if methods is None else list(methods)                               # :62
    T = len(triplets)
    keys = ("repr_err", "rot_err", "t_err", "iter", "time")
    keys = keys + tuple(k + "_ba" for k in keys)
    res = {k: np.zeros((T, len(METHODS))) for k in keys}
    info = []
    for it, tr in enumerate(triplets):
        Corresp, CalM, R_t0 = tr["Corresp"], tr["CalM"], tr["R_t0"]
        P0 = _cameras(CalM, R_t0[0][None], R_t0[1][None])[0]
        # inliers of the ground-truth cameras: triangulate, reproject, |residual| <= 1 px in every coordinate   (:93-99)
        X = _np(ctx.triangulate(P0, np.ascontiguousarray(Corresp.T)[None]))[0]
        X = X[0:3] / X[3:4]
        proj = np.concatenate([(P @ np.vstack([X, np.ones(X.shape[1])]))[0:2] / (P @ np.vstack([X, np.ones(X.shape[1])]))[2:3] for P in P0])
        inl = np.sum(np.abs(proj - Corresp) > repr_err_th, axis=0) == 0
        Ci = Corresp[:, inl]
        N = Ci.shape[1]
        rng = np.random.Generator(np.random.Philox(key=seed0 + it))                                 # :104-105 (rng(it); randsample)
        sel = np.sort(rng.choice(N, size=min(initial_sample_size, N), replace=False))
        Cs = np.ascontiguousarray(Ci[:, sel].T)[None]
        Call = np.ascontiguousarray(Ci.T)[None]
        ref = np.sort(rng.choice(sel, size=min(bundle_adj_size, sel.size), replace=False))          # :106-108 (ref_sample out of init_sample)
        Cref = np.ascontiguousarray(Ci[:, ref].T)[None]
        info.append(dict(name=tr.get("name", str(it)), matches=int(Corresp.shape[1]), inliers=int(N), sample=int(sel.size),
                         repr_err_gt=float(_np(ctx.repr_error(P0[None], Call))[0])))
        for m in mt:
            if (m > 5 and N < 8) or N < 7:                                                       # :113-118
                for k in keys:
                    res[k][it, m] = np.inf
                continue
            import torch
            out, sec = _run_method(ctx, METHODS[m], torch.from_numpy(Cs).cuda(), torch.from_numpy(CalM).cuda(), _cuda_timer(torch))
            if int(out["status"][0]) != 0:
                for k in keys:
                    res[k][it, m] = np.inf
                continue
            P = _cameras(CalM, out["R_t_2"], out["R_t_3"])
            res["repr_err"][it, m] = float(_np(ctx.repr_error(P, Call))[0])                      # :130-131 (re-triangulates)
            r2, t2 = AngError_batch(R_t0[0], out["R_t_2"]); r3, t3 = AngError_batch(R_t0[1], out["R_t_3"])
            res["rot_err"][it, m] = float((r2[0] + r3[0]) / 2); res["t_err"][it, m] = float((t2[0] + t3[0]) / 2)
            res["iter"][it, m] = float(out["iter"][0]); res["time"][it, m] = sec
            # BundleAdjustment(CalM, [eye(3,4); R_t_2; R_t_3], Corresp_ref) -- no initial points   (:140-142)
            ba, ba_sec = _cuda_timer(torch)(lambda: ctx.bundle_adjust(CalM, out["R_t_2"], out["R_t_3"], Cref, None))
            if int(_np(ba["status"])[0]) != 0:
                for k in keys[5:]:
                    res[k][it, m] = np.inf
                continue
            bR2, bR3 = _np(ba["R_t_2"]), _np(ba["R_t_3"])
            res["repr_err_ba"][it, m] = float(_np(ctx.repr_error(_cameras(CalM, bR2, bR3), Call))[0])    # :145-147
            r2, t2 = AngError_batch(R_t0[0], bR2); r3, t3 = AngError_batch(R_t0[1], bR3)
            res["rot_err_ba"][it, m] = float((r2[0] + r3[0]) / 2); res["t_err_ba"][it, m] = float((t2[0] + t3[0]) / 2)
            res["iter_ba"][it, m] = float(_np(ba["iter"])[0]); res["time_ba"][it, m] = ba_sec
    out = dict(methods=METHODS, methods_tested=[METHODS[m] for m in mt], triplets=info,
               note="one row per triplet; time = seconds of the single-triplet call; *_ba: after BundleAdjustment on a 50-correspondence "
                    "subset of the sample, ReprError again over all inliers (experiments_real.m:140-153)")
    out.update({k: v.tolist() for k, v in res.items()})
    return out


def load_epfl_all(npz_path, dataset, n_triplets=None):
    """The all-triplets EPFL fixture (tests/golden/epfl_all.npz, made by tests/golden/make_epfl_all.py from the reference's Data/
    directory): `dataset` in {"fountain", "herzjesu"}, the first `n_triplets` of indexes_sorted (experiments_real.m:31-35,78:
    70 for fountain-P11, 50 for Herz-Jesu-P8; None = all)."""
    g = np.load(npz_path)
    trips = np.asarray(g[dataset + "_triplets"]); off = np.asarray(g[dataset + "_offsets"]); C = np.asarray(g[dataset + "_corresp"])
    K, R, t = np.asarray(g[dataset + "_K"]), np.asarray(g[dataset + "_R"]), np.asarray(g[dataset + "_t"])
    out = []
    for n, (i1, i2, i3, cnt) in enumerate(trips[:n_triplets]):
        a, b, c = i1 - 1, i2 - 1, i3 - 1
        Rt0 = [np.hstack([R[b] @ R[a].T, (t[b] - R[b] @ R[a].T @ t[a]).reshape(3, 1)]),
               np.hstack([R[c] @ R[a].T, (t[c] - R[c] @ R[a].T @ t[a]).reshape(3, 1)])]                # experiments_real.m:90-91
        out.append(dict(name="%s (%d,%d,%d)" % (dataset, i1, i2, i3), Corresp=np.ascontiguousarray(C[off[n]:off[n + 1]].T),
                        CalM=np.vstack([K[a], K[b], K[c]]), R_t0=Rt0))
    return out


def epfl_inliers(ctx, tr, repr_err_th=1.0):
    """experiments_real.m:93-99: triangulate with the ground-truth cameras, keep the correspondences whose six reprojection
    residuals are all <= 1 px.  Returns the inlier correspondences (6 x Ni)."""
    Corresp, CalM, R_t0 = tr["Corresp"], tr["CalM"], tr["R_t0"]
    P0 = _cameras(CalM, R_t0[0][None], R_t0[1][None])[0]
    X = _np(ctx.triangulate(P0, np.ascontiguousarray(Corresp.T)[None]))[0]
    X = X[0:3] / X[3:4]
    Xh = np.vstack([X, np.ones(X.shape[1])])
    proj = np.concatenate([(P @ Xh)[0:2] / (P @ Xh)[2:3] for P in P0])
    return Corresp[:, np.sum(np.abs(proj - Corresp) > repr_err_th, axis=0) == 0]


def real_trials(ctx, triplets, n_trials=100, sigma=0.5, initial_sample_size=100, methods=None, seed=1):
    """BASELINE.json configs[4]: experiments_real.m's evaluation repeated over `n_trials` noise trials per triplet, all
    (triplet, trial) problems of equal sample size batched into ONE C-ABI call per method.
    Trial k of triplet t: a fresh min(100, Ni)-correspondence sample of the inliers (Philox keyed by (seed, t, k); the reference
    takes one sample per triplet, experiments_real.m:104-105) with N(0, sigma^2) noise added to the sampled coordinates (the
    reference adds none to real data: sigma = 0 reproduces it).  Per trial and method: ReprError over ALL inliers of the triplet
    after re-triangulation with the estimated cameras (:130-131), AngError of both poses against the `.camera` ground truth
    (:133-136), iterations.  Returns per-method means over all (triplet, trial) pairs and over triplets, and the throughput."""
    import torch
    mt = [0, 1, 2, 3, 4, 6, 7] if methods is None else list(methods)                               # :62
    info, groups = [], {}
    for ti, tr in enumerate(triplets):
        Ci = epfl_inliers(ctx, tr)
        Ni = Ci.shape[1]
        info.append(dict(name=tr.get("name", str(ti)), matches=int(tr["Corresp"].shape[1]), inliers=int(Ni)))
        n = min(initial_sample_size, Ni)
        if n < 7:
            continue
        S = np.empty((n_trials, n, 6))
        for k in range(n_trials):
            rng = np.random.Generator(np.random.Philox(key=[seed, 1000003 * ti + k]))
            sel = np.sort(rng.choice(Ni, size=n, replace=False))
            S[k] = Ci[:, sel].T + sigma * rng.standard_normal((n, 6))
        groups.setdefault(n, []).append((ti, S, Ci))
    T = len(triplets)
    keys = ("repr_err", "rot_err", "t_err", "iter")
    res = {k: np.full((len(METHODS), T, n_trials), np.inf) for k in keys}
    gpu_seconds = np.zeros(len(METHODS))
    problems = np.zeros(len(METHODS))
    for n, members in groups.items():
        C = torch.from_numpy(np.concatenate([m[1] for m in members])).cuda()                     # (len(members) * n_trials, n, 6)
        CalB = torch.from_numpy(np.concatenate([np.broadcast_to(triplets[m[0]]["CalM"], (n_trials, 9, 3)) for m in members]).copy()).cuda()
        for mi in mt:
            if mi > 5 and n < 8:
                continue
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ctx.pose_batch(METHODS[mi], C, CalB, reconst=False)                                  # warm-up (workspace growth)
            torch.cuda.synchronize()
            a.record()
            out = ctx.pose_batch(METHODS[mi], C, CalB, reconst=False)
            b.record()
            torch.cuda.synchronize()
            gpu_seconds[mi] += a.elapsed_time(b) * 1e-3
            problems[mi] += C.shape[0]
            st = _np(out["status"]); R2 = _np(out["R_t_2"]); R3 = _np(out["R_t_3"]); its = _np(out["iter"])
            for j, (ti, _, Ci) in enumerate(members):
                sl = slice(j * n_trials, (j + 1) * n_trials)
                ok = st[sl] == 0
                if not ok.any():
                    continue
                tr = triplets[ti]
                P = _cameras(tr["CalM"], R2[sl][ok], R3[sl][ok])
                res["repr_err"][mi, ti, ok] = _np(ctx.repr_error(P, np.ascontiguousarray(Ci.T)))   # all inliers, shared by the trials
                r2, t2 = AngError_batch(tr["R_t0"][0], R2[sl][ok]); r3, t3 = AngError_batch(tr["R_t0"][1], R3[sl][ok])
                res["rot_err"][mi, ti, ok] = (r2 + r3) / 2
                res["t_err"][mi, ti, ok] = (t2 + t3) / 2
                res["iter"][mi, ti, ok] = its[sl][ok]
    summary = {}
    for mi in mt:
        fin = np.isfinite(res["rot_err"][mi])
        summary[METHODS[mi]] = dict(
            problems=int(problems[mi]), solved=int(fin.sum()),
            mean_repr_err=float(np.mean(res["repr_err"][mi][fin])) if fin.any() else None,
            median_repr_err=float(np.median(res["repr_err"][mi][fin])) if fin.any() else None,
            mean_rot_err_deg=float(np.nanmean(res["rot_err"][mi][fin])) if fin.any() else None,
            mean_t_err_deg=float(np.nanmean(res["t_err"][mi][fin])) if fin.any() else None,
            median_rot_err_deg=float(np.nanmedian(res["rot_err"][mi][fin])) if fin.any() else None,
            mean_iter=float(np.mean(res["iter"][mi][fin])) if fin.any() else None,
            gpu_seconds=float(gpu_seconds[mi]), problems_per_s=float(problems[mi] / gpu_seconds[mi]) if gpu_seconds[mi] > 0 else None)
    return dict(n_trials=n_trials, sigma=sigma, sample=initial_sample_size, triplets=info, methods_tested=[METHODS[m] for m in mt], summary=summary,
                per_triplet_mean_rot_err={METHODS[mi]: [float(np.nanmean(np.where(np.isfinite(res["rot_err"][mi, t]), res["rot_err"][mi, t], np.nan)))
                                                        if np.isfinite(res["rot_err"][mi, t]).any() else None for t in range(T)] for mi in mt},
                note="(triplet, trial) problems of one sample size go through ONE batched call per method; AngError's acos is not clamped "
                     "(AngError.m:21-28): a NaN means ~0 degrees and is skipped in the means")


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--option", choices=sorted(INTERVALS), default="noise")
    ap.add_argument("--n-sim", type=int, default=20)
    ap.add_argument("--points", type=int, default=12)
    ap.add_argument("--noise", type=float, default=1.0)
    ap.add_argument("--focal", type=float, default=50.0)
    ap.add_argument("--angle", type=float, default=0.0)
    ap.add_argument("--real", default=None, help="tests/golden/epfl.npz, or a Data/<dataset> directory of the reference")
    ap.add_argument("--n-triplets", type=int, default=None, help="first n of indexes_sorted (default 70 fountain / 50 herzjesu, experiments_real.m:31-35)")
    ap.add_argument("--dataset", choices=["fountain", "herzjesu"], default=None, help="with --real tests/golden/epfl_all.npz")
    ap.add_argument("--noise-trials", type=int, default=0, help="configs[4]: K noise trials per triplet, batched (real_trials)")
    ap.add_argument("--sigma", type=float, default=0.5, help="pixel noise of the trials")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from . import api
    ctx = api.Context(0)
    if args.real:
        if args.dataset:
            trips = load_epfl_all(args.real, args.dataset, args.n_triplets or (70 if args.dataset == "fountain" else 50))
        else:
            trips = load_epfl_fixture(args.real) if args.real.endswith(".npz") else load_epfl_dataset(args.real, args.n_triplets or 70)
        res = real_trials(ctx, trips, args.noise_trials, args.sigma, seed=args.seed) if args.noise_trials else real_sweep(ctx, trips)
    else:
        res = synthetic_sweep(ctx, args.option, n_sim=args.n_sim, N=args.points, noise=args.noise, f=args.focal, angle=args.angle)
    txt = json.dumps(res)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(txt)
    else:
        print(txt)


if __name__ == "__main__":
    main()
