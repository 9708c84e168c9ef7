"""The batched experiment harness (tft_vs_fund_amd/experiments.py: experiments.m / experiments_real.m).
CPU: the sweep logic against a stand-in context that answers with the oracle.  GPU: the same sweeps through the C ABI,
metrics compared with the oracle's on the same scenes and with the EPFL goldens."""
import json
import os

import numpy as np
import pytest

from tft_vs_fund_amd import experiments as X
from tft_vs_fund_amd.scenes import generate_scene_batch


class OracleContext:
    """Stand-in for api.Context in the CPU test: answers every batched call with the oracle (test infrastructure)."""

    def pose_batch(self, method, C, CalM, reconst=True):
        from oracle import tft_oracle as O
        B, N, _ = C.shape
        out = dict(R_t_2=np.zeros((B, 3, 4)), R_t_3=np.zeros((B, 3, 4)), Reconst=np.zeros((B, 3, N)), T=np.zeros((B, 3, 3, 3)),
                   iter=np.zeros(B, dtype=np.int32), status=np.zeros(B, dtype=np.int32))
        for b in range(B):
            R2, R3, Rec, T, it = getattr(O, method)(C[b].T.copy(), CalM)[:5]
            out["R_t_2"][b], out["R_t_3"][b], out["Reconst"][b], out["T"][b], out["iter"][b] = R2, R3, Rec, T, it
        return out

    def repr_error(self, cams, corresp, pts3d=None):
        from oracle import tft_oracle as O
        return np.array([O.ReprError(list(cams[b]), corresp[b].T.copy(), None if pts3d is None else pts3d[b]) for b in range(cams.shape[0])])


def test_synthetic_sweep_logic_with_oracle_context():
    res = X.synthetic_sweep(OracleContext(), "points", n_sim=3, interval=[7, 9], methods=[0, 6, 7], device=False)
    json.dumps(res)                                                          # serialisable
    assert res["methods"][0] == "LinearTFTPoseEstimation" and res["methods"][5] == "PiColPoseEstimation" and len(res["methods"]) == 8
    r = np.array(res["rot_err"]); it = np.array(res["iter"])
    assert np.isfinite(r[0, 0]) and np.isinf(r[0, 6]) and np.isinf(r[0, 7])       # N = 7: F methods get inf (experiments.m:99-104)
    assert np.all(np.isfinite(r[1, [0, 6, 7]])) and it[1, 0] == 0 and it[1, 6] == 0 and it[1, 7] >= 2
    assert r[1, 1] == 0                                                       # not tested -> stays 0 as in the reference's arrays
    assert X.methods_to_test("angle") == list(range(8)) and 5 not in X.methods_to_test("noise")
    assert X.INTERVALS["noise"][-1] == 3.0 and X.INTERVALS["angle"][-2:] == [179.5, 180] and X.INTERVALS["points"] == [7, 8, 9, 10, 15, 20, 25]


@pytest.mark.gpu
def test_synthetic_sweep_matches_oracle_metrics(gpu_ctx):
    from oracle import tft_oracle as O
    n_sim, interval = 6, [9, 20]
    res = X.synthetic_sweep(gpu_ctx, "points", n_sim=n_sim, interval=interval, seed0=1)
    assert np.array(res["failed"]).sum() == 0
    for i, N in enumerate(interval):
        C, CalM, Rt0, _ = generate_scene_batch(n_sim, N, noise=1.0, seed=1 + 1000 * i, focalL=50.0, angle=0.0)
        for m, name in enumerate(X.METHODS):
            if name == "PiColPoseEstimation":
                assert res["rot_err"][i][m] == 0                              # not run outside the angle sweep
                continue
            rep, rot, tr, its = [], [], [], []
            for b in range(n_sim):
                R2, R3, Rec, T, it = getattr(O, name)(C[b].T.copy(), CalM)[:5]
                P = [CalM[0:3] @ np.eye(3, 4), CalM[3:6] @ R2, CalM[6:9] @ R3]
                rep.append(O.ReprError(P, C[b].T.copy(), Rec))
                a2, b2 = O.AngError(Rt0[0], R2); a3, b3 = O.AngError(Rt0[1], R3)
                rot.append((a2 + a3) / 2); tr.append((b2 + b3) / 2); its.append(it)
            linear = name.startswith("Linear") or name == "OptimFPoseEstimation"
            tol = 1e-6 if linear else 0.08                                    # Gauss-Helmert: statistical parity (test_gpu_parity.py)
            assert abs(res["repr_err"][i][m] - np.mean(rep)) <= tol * np.mean(rep), (N, name)
            assert abs(res["rot_err"][i][m] - np.mean(rot)) <= tol * np.mean(rot) + (0 if linear else 0.02), (N, name)
            assert abs(res["t_err"][i][m] - np.mean(tr)) <= tol * np.mean(tr) + (0 if linear else 0.02), (N, name)
            assert abs(res["iter"][i][m] - np.mean(its)) <= (0 if linear else 1.5), (N, name)
            assert 0 < res["time"][i][m] < 1.0
            # second column of the reference's arrays: after BundleAdjustment (experiments.m:127-141)
            assert np.isfinite(res["rot_err_ba"][i][m]) and res["iter_ba"][i][m] >= 1 and 0 < res["time_ba"][i][m] < 1.0
            if name == "LinearTFTPoseEstimation":
                from oracle import ba_oracle as BA
                rot_ba, rep_ba = [], []
                for b in range(n_sim):
                    R2, R3, Rec, T, it = O.LinearTFTPoseEstimation(C[b].T.copy(), CalM)
                    Rt, _, _, err = BA.BundleAdjustment(CalM, np.vstack([np.eye(3, 4), R2, R3]), C[b].T.copy(), Rec)
                    a2, _ = O.AngError(Rt0[0], Rt[3:6]); a3, _ = O.AngError(Rt0[1], Rt[6:9])
                    rot_ba.append((a2 + a3) / 2); rep_ba.append(err)
                assert abs(res["rot_err_ba"][i][m] - np.mean(rot_ba)) <= 1e-6 * np.mean(rot_ba)
                assert abs(res["repr_err_ba"][i][m] - np.mean(rep_ba)) <= 1e-6 * np.mean(rep_ba)


@pytest.mark.gpu
def test_angle_sweep_runs_all_eight_methods(gpu_ctx):
    res = X.synthetic_sweep(gpu_ctx, "angle", n_sim=8, N=30, interval=[170, 180])
    r = np.array(res["rot_err"]); failed = np.array(res["failed"])
    assert r.shape == (2, 8) and np.all(np.isfinite(r)) and np.all(r[:, [0, 6]] < 5)      # the iterative methods may diverge on some collinear scenes, as in the reference
    assert failed[:, [0, 6]].sum() == 0


@pytest.mark.gpu
def test_real_sweep_on_epfl_fixture(gpu_ctx, golden_dir):
    """experiments_real.m on the eight committed EPFL triplets: inlier counts and the linear methods' ReprError over all
    inliers reproduce the goldens (same 100-correspondence samples: seed0 = 1000 as in make_golden.py)."""
    g = np.load(os.path.join(golden_dir, "epfl.npz"))
    trips = X.load_epfl_fixture(os.path.join(golden_dir, "epfl.npz"))
    res = X.real_sweep(gpu_ctx, trips, seed0=1000)
    json.dumps(res)
    ressl_dev = []
    for n, info in enumerate(res["triplets"]):
        pre = "t%d_" % n
        assert info["inliers"] == int(g[pre + "n_inliers"]) and info["sample"] == min(100, info["inliers"])
        assert abs(info["repr_err_gt"] - float(g[pre + "repr_gt_inliers"])) < 1e-9 * float(g[pre + "repr_gt_inliers"]) + 1e-12
        assert abs(res["repr_err"][n][0] - float(g[pre + "tft_repr_all"])) < 1e-7 * float(g[pre + "tft_repr_all"])
        assert abs(res["repr_err"][n][6] - float(g[pre + "f_repr_all"])) < 1e-7 * float(g[pre + "f_repr_all"])
        ressl_dev.append(abs(res["repr_err"][n][1] - float(g[pre + "ressl_repr_all"])) / float(g[pre + "ressl_repr_all"]))
        assert res["repr_err"][n][5] == 0                                     # PiCol is not run on real data (experiments_real.m:62)
        assert all(np.isfinite(res["rot_err"][n][m]) and res["rot_err"][n][m] < 5 for m in (0, 1, 2, 3, 4, 6, 7))
        assert all(np.isfinite(res["rot_err_ba"][n][m]) and res["rot_err_ba"][n][m] < 5 and res["iter_ba"][n][m] >= 1 for m in (0, 1, 2, 3, 4, 6, 7))
    # Gauss-Helmert on real matches: statistical parity.  Where the KKT matrix is ill-conditioned (fountain triplet 4: cond 1e6) the
    # 1e12-weighted rounding noise of A'Ww dominates the weak directions of the first step in ANY implementation,
    # and the "objective rose" exit then fires at a different iteration: most triplets agree to a few %, a minority does not.
    assert np.median(ressl_dev) < 0.03 and sum(d > 0.3 for d in ressl_dev) <= 2, ressl_dev


@pytest.mark.gpu
def test_config5_noise_trials_on_all_epfl_triplets(gpu_ctx, golden_dir):
    """BASELINE.json configs[4] on one GPU: the reference's real-data evaluation (experiments_real.m:75-138) over its own triplet
    lists -- the first 70 of fountain-P11 and the first 50 of Herz-Jesu-P8 in `indexes_sorted` order -- x 100 noise trials x the
    seven methods it runs on real data, every (triplet, trial) problem of a sample size in one batched C-ABI call per method.
    Known answers: the eight inlier counts of SURVEY.md section 4 (deterministic, from the reference's data and its 1-px rule)."""
    path = os.path.join(golden_dir, "epfl_all.npz")
    known = {"fountain (5,6,7)": (1400, 1360), "fountain (6,7,8)": (1306, 1253), "fountain (3,4,5)": (1302, 1250), "fountain (4,6,9)": (95, 85),
             "herzjesu (6,7,8)": (1482, 1222), "herzjesu (5,6,7)": (1267, 1037), "herzjesu (3,4,5)": (1117, 920), "herzjesu (2,6,8)": (97, 39)}
    seen = {}
    for dataset, n_trip in (("fountain", 70), ("herzjesu", 50)):
        trips = X.load_epfl_all(path, dataset, n_trip)
        assert len(trips) == n_trip and trips[0]["Corresp"].shape[1] >= trips[-1]["Corresp"].shape[1]      # sorted by match count
        res = X.real_trials(gpu_ctx, trips, n_trials=100, sigma=0.5)
        json.dumps(res)
        for info in res["triplets"]:
            if info["name"] in known:
                seen[info["name"]] = (info["matches"], info["inliers"])
        assert res["methods_tested"] == [X.METHODS[m] for m in (0, 1, 2, 3, 4, 6, 7)]                      # experiments_real.m:62
        n_ok = sum(1 for i in res["triplets"] if min(100, i["inliers"]) >= 8)
        for m, s in res["summary"].items():
            assert s["problems"] >= 100 * (n_ok - 2), (dataset, m, s["problems"])
            assert s["solved"] >= 0.97 * s["problems"], (dataset, m, s)
            # real matches + 0.5 px of added noise, 100-correspondence samples: poses within a few degrees of the .camera ground truth
            # (means are reported too; Nordberg's is dominated by the few trials in which its axis-angle parameterisation diverges)
            assert s["median_rot_err_deg"] < 1.0 and s["median_repr_err"] < 10.0, (dataset, m, s)
        lin, opt = res["summary"]["LinearFPoseEstimation"], res["summary"]["OptimFPoseEstimation"]
        assert opt["mean_iter"] > 2 and lin["mean_iter"] == 0
    # (2,6,8) of Herz-Jesu and (4,6,9) of fountain are far down the sorted lists: look them up in the full fixture
    for dataset in ("fountain", "herzjesu"):
        for tr in X.load_epfl_all(path, dataset, None):
            if tr["name"] in known and tr["name"] not in seen:
                seen[tr["name"]] = (tr["Corresp"].shape[1], X.epfl_inliers(gpu_ctx, tr).shape[1])
    assert seen == known, seen


@pytest.mark.gpu
def test_config5_at_one_thousand_trials(gpu_ctx, golden_dir):
    """BASELINE.json configs[4] at its stated trial count on one dataset: the first 50 Herz-Jesu-P8 triplets of the reference's list x 1000 noise
    trials x the seven methods of experiments_real.m:62 (50 000 problems per method, one batched call each)."""
    trips = X.load_epfl_all(os.path.join(golden_dir, "epfl_all.npz"), "herzjesu", 50)
    res = X.real_trials(gpu_ctx, trips, n_trials=1000, sigma=0.5)
    n_ok = sum(1 for i in res["triplets"] if min(100, i["inliers"]) >= 8)
    assert len(res["summary"]) == 7
    for m, s in res["summary"].items():
        assert s["problems"] >= 1000 * (n_ok - 2), (m, s["problems"])
        assert s["solved"] >= 0.97 * s["problems"], (m, s)
        assert s["median_rot_err_deg"] < 1.0 and s["median_repr_err"] < 10.0, (m, s)
