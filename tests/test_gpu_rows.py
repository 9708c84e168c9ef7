"""
The four-triplets-per-wavefront kernels (csrc/tft_rows_kernel.h, f_rows_kernel.h, gh_rows_kernel.h; TFF_OPT_ROWS = 1; the default, 2, takes them for
batches of 1024 triplets and more) on the MI355X
against the one-triplet-per-wavefront kernels (TFF_OPT_ROWS = 0) and the oracle: same arithmetic per matrix entry and correspondence, sums taken in
a different order, so the two routes must agree to rounding on every batch shape -- full and ragged last wavefronts, one to many trips
per data pass, well-posed and outlier-ridden data (adaptive cheirality votes: second sweep), sampled hypotheses (config 4).
The oracle comparisons of tests/test_gpu_parity.py run through both routes as well (tests/conftest.py::ROUTES).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from helpers import rel_err_T, rel_err   # noqa: E402

TOL = 1e-9


@pytest.fixture(scope="module")
def gpu_ctx():
    """(not the route-parametrised context of conftest.py: every test here switches between the routes itself and leaves TFF_OPT_ROWS = 1 behind)"""
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.build import build_library
    build_library()
    ctx = api.Context(0)
    ctx.set_rows(1)
    return ctx


def _both_routes(ctx, *args, **kw):
    out = {}
    for rows in (1, 0):
        ctx.set_rows(rows)
        try:
            o = ctx.pose_batch(*args, **kw)
        finally:
            ctx.set_rows(1)
        out[rows] = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in o.items() if k != "_raw" and v is not None}
    return out[1], out[0]


@pytest.mark.parametrize("method", ["LinearTFTPoseEstimation", "LinearFPoseEstimation"])
@pytest.mark.parametrize("B,N,sigma", [(10000, 200, 1.0), (3001, 33, 0.5), (1000, 500, 1.0), (2047, 16, 2.0), (5, 100, 1.0), (1, 64, 1.0)])
def test_rows_and_wave_routes_agree(gpu_ctx, method, B, N, sigma):
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, _, _ = generate_scene_batch(B, N, noise=sigma, seed=7 * N + B)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    r, w = _both_routes(gpu_ctx, method, d, calm, reconst=True)
    assert np.array_equal(r["status"], w["status"]) and np.all(r["status"] == 0) and np.all(r["iter"] == 0)
    sg = np.sign(np.sum(r["T"] * w["T"], axis=(1, 2, 3)))[:, None, None, None]
    assert np.abs(r["T"] * sg - w["T"]).max() < TOL
    assert np.abs(r["R_t_2"] - w["R_t_2"]).max() < TOL
    assert np.abs(r["R_t_3"] - w["R_t_3"]).max() < TOL * max(1.0, np.abs(w["R_t_3"]).max())
    assert np.abs(r["Reconst"] - w["Reconst"]).max() < 1e-8 * np.abs(w["Reconst"]).max()
    # and against the oracle, first and last triplets (the last wavefront is ragged unless B is a multiple of four)
    from oracle import tft_oracle as O
    for b in sorted(set([0, B // 2, max(B - 2, 0), B - 1])):
        R2, R3, Rec, T, _ = getattr(O, method)(C[b].T.copy(), CalM)
        assert rel_err_T(r["T"][b], T) < TOL and rel_err(r["R_t_2"][b], R2) < TOL and rel_err(r["R_t_3"][b], R3) < TOL
        assert rel_err(r["Reconst"][b], Rec) < TOL


@pytest.mark.parametrize("method", ["ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "FaugPapaTFTPoseEstimation", "PiPoseEstimation", "OptimFPoseEstimation"])
def test_iterative_methods_do_not_depend_on_the_layout_of_their_linear_stage(gpu_ctx, method):
    """k_gh_linear_rows (four triplets per wavefront) against k_gh_linear<false>: the start of the Gauss-Helmert iteration agrees to rounding,
    so do the results -- same iteration counts, 1e-7 (the iteration amplifies the start's last bits; the 50-digit gates of
    tests/test_gpu_gh_noise.py run on the rows route, the default).  OptimF: the three stages of csrc/optimf_rows_kernel.h against the fused
    one-triplet kernel k_f_pose<false, 1>."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B, N = 1001, 200
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=606)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    r, w = _both_routes(gpu_ctx, method, d, calm, reconst=False)
    assert np.array_equal(r["status"], w["status"]) and np.all(r["status"] == 0)
    assert (r["iter"] != w["iter"]).mean() < 0.01
    same = r["iter"] == w["iter"]
    sg = np.sign(np.sum(r["T"] * w["T"], axis=(1, 2, 3)))[:, None, None, None]
    assert np.abs(r["T"] * sg - w["T"])[same].max() < 1e-7 and np.abs(r["R_t_3"] - w["R_t_3"])[same].max() < 1e-7 * max(1.0, np.abs(w["R_t_3"]).max())


def test_rows_result_does_not_depend_on_the_row_or_the_neighbours(gpu_ctx):
    """A triplet gives bit-identical results in every row of a wavefront, whatever the other rows hold."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, _, _ = generate_scene_batch(9, 200, noise=1.0, seed=99)
    calm = torch.from_numpy(CalM).cuda()
    base = gpu_ctx.pose_batch("LinearTFTPoseEstimation", torch.from_numpy(C[:1].copy()).cuda(), calm, reconst=True)
    for pos in range(4):
        idx = [1, 2, 3, 4, 5, 6, 7, 8]
        idx.insert(pos, 0)
        out = gpu_ctx.pose_batch("LinearTFTPoseEstimation", torch.from_numpy(np.ascontiguousarray(C[idx])).cuda(), calm, reconst=True)
        for k in ("T", "R_t_2", "R_t_3", "Reconst"):
            assert torch.equal(out[k][pos], base[k][0]), (pos, k)


def test_rows_adaptive_votes_on_outlier_data(gpu_ctx):
    """A fifth of the correspondences replaced by random image points, noise 2 px, N = 30: cheirality scores fall short of 2 N, votes go
    uncertified, triplets are handed to the exact kernel -- on both routes alike."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B, N = 8000, 30
    C, CalM, _, _ = generate_scene_batch(B, N, noise=2.0, seed=31)
    rng = np.random.default_rng(8)
    C = C.copy()
    for b in range(B):
        bad = rng.choice(N, 6, replace=False)
        C[b, bad, 2:6] = rng.uniform([0, 0, 0, 0], [1800, 1200, 1800, 1200], size=(6, 4))
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    r, w = _both_routes(gpu_ctx, "LinearTFTPoseEstimation", d, calm, reconst=False)
    assert np.array_equal(r["status"], w["status"])
    ok = r["status"] == 0
    assert ok.mean() > 0.5
    e2 = np.abs(r["R_t_2"] - w["R_t_2"]).reshape(B, -1).max(axis=1)[ok]
    e3 = np.abs(r["R_t_3"] - w["R_t_3"]).reshape(B, -1).max(axis=1)[ok]
    # rounding-level cheirality ties between the two rotations may fall differently (the reference leaves them open: tests/helpers.py)
    assert (e2 > 1e-7).mean() < 2e-3 and (e3 > 1e-6 * np.abs(w["R_t_3"]).reshape(B, -1).max(axis=1)[ok].clip(1.0)).mean() < 2e-3


def test_rows_sampled_hypotheses(gpu_ctx):
    """Config-4 style index gathering through the rows kernel (TFF_OPT_EXACT_BELOW = 0, 12-point samples): same poses as the
    one-triplet kernel's LDS gather, and an index outside the scene is reported for its own hypothesis only."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    Ns, H, n = 300, 4001, 12
    Cs, CalM, _, _ = generate_scene_batch(1, Ns, noise=0.5, seed=11)
    scene = torch.from_numpy(Cs[0].copy()).cuda(); calm = torch.from_numpy(CalM).cuda()
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    idx = torch.rand((H, Ns), device="cuda", generator=g).argsort(dim=1)[:, :n].to(torch.int32).contiguous()
    idx[17, 3] = Ns            # outside the scene
    idx[18, 0] = -1
    res = {}
    gpu_ctx.set_exact_below(0)
    try:
        for rows in (1, 0):
            gpu_ctx.set_rows(rows)
            o = gpu_ctx.pose_sampled("LinearTFTPoseEstimation", scene, calm, idx)
            res[rows] = {k: v.cpu().numpy() for k, v in o.items() if k != "_raw"}
    finally:
        gpu_ctx.set_rows(1)
        gpu_ctx.set_exact_below(12)
    r, w = res[1], res[0]
    assert np.array_equal(r["status"], w["status"])
    assert r["status"][17] == 1 and r["status"][18] == 1 and np.all(np.isnan(r["T"][17])) and (r["status"] == 1).sum() == 2
    ok = r["status"] == 0
    assert ok.sum() >= H - 2 - 40
    e3 = np.abs(r["R_t_3"] - w["R_t_3"]).reshape(H, -1).max(axis=1)[ok]
    assert (e3 > 1e-6).mean() < 5e-3


@pytest.mark.parametrize("method,n", [("LinearTFTPoseEstimation", 7), ("LinearFPoseEstimation", 8)])
def test_exact_rows_kernel_against_the_one_triplet_exact_kernel(gpu_ctx, method, n):
    """40 000 seven- / eight-point samples at 3 px noise through the exact tiers: four triplets per wavefront (tft_rows_exact_kernel.h, the default for
    N < TFF_OPT_EXACT_BELOW) against one per wavefront (TFF_OPT_ROWS = 0).  Same statuses; the same poses wherever the reference's answer is
    unique (cheirality ties between the two rotations, ~0.2 % of minimal samples, may fall differently: tests/helpers.py)."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B = 40000
    C, CalM, _, _ = generate_scene_batch(B, n, noise=3.0, seed=5)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    r, w = _both_routes(gpu_ctx, method, d, calm, reconst=True)
    assert np.array_equal(r["status"], w["status"])
    ok = r["status"] == 0
    assert ok.mean() > 0.99
    sg = np.sign(np.sum(r["T"] * w["T"], axis=(1, 2, 3)))[:, None, None, None]
    eT = np.abs(r["T"] * sg - w["T"]).reshape(B, -1).max(axis=1)[ok]
    e3 = (np.abs(r["R_t_3"] - w["R_t_3"]).reshape(B, -1).max(axis=1) / np.abs(w["R_t_3"]).reshape(B, -1).max(axis=1).clip(1.0))[ok]
    assert np.quantile(eT, 0.999) < 1e-7 and (e3 > 1e-6).mean() < 3e-3, (np.quantile(eT, 0.999), (e3 > 1e-6).mean())


@pytest.mark.parametrize("method", ["LinearTFTPoseEstimation", "LinearFPoseEstimation"])
def test_default_route_is_the_row_kernels_at_any_batch_size(method):
    """TFF_OPT_ROWS = 2 (the default of a new context) is the row kernels whatever the batch size since the end of round 5 (include/tftfund.h): bit-identical
    to TFF_OPT_ROWS = 1 from one triplet to 6 000, minimal samples included, and NOT to the one-triplet kernels (the two routes differ in the last bits,
    which is why the route must not depend on the batch)."""
    import torch
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    for B, N in [(1, 100), (7, 9), (300, 100), (1023, 200), (1024, 200), (2049, 300), (6000, 100)]:
        C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=B)
        d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
        auto = api.Context(0)
        forced = api.Context(0)
        forced.set_rows(1)
        other = api.Context(0)
        other.set_rows(0)
        a = auto.pose_batch(method, d, calm, reconst=False)
        f = forced.pose_batch(method, d, calm, reconst=False)
        o = other.pose_batch(method, d, calm, reconst=False)
        for k in ("T", "R_t_2", "R_t_3", "iter", "status"):
            assert torch.equal(a[k], f[k]), (B, N, k)
        if B >= 300:
            assert not torch.equal(a["T"], o["T"])                           # (so the check above says which route ran)
        with pytest.raises(Exception):
            auto.set_rows(3)


@pytest.mark.parametrize("method", ["ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "FaugPapaTFTPoseEstimation", "PiPoseEstimation", "PiColPoseEstimation",
                                    "OptimFPoseEstimation", "LinearTFTPoseEstimation", "LinearFPoseEstimation"])
def test_iterative_methods_do_not_depend_on_the_batch_a_triplet_arrives_in(method):
    """(All eight methods since the end of round 5: the linear ones no longer pick their kernel by batch size either.)  The iterative methods amplify a last-bit difference of their start (the exit test of Gauss_Helmert.m:71-82 can flip on it), so the library's
    DEFAULT route for them must not depend on the batch size (capi.hip::rows_for_iterative): the same 40 triplets alone, at the head of a batch
    of 1023 and scattered through a batch of 1024 / 2500 give bit-identical T, R_t_2, R_t_3 and the same `iter` and status."""
    import torch
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    N = 60
    C, CalM, _, _ = generate_scene_batch(2500, N, noise=1.0, seed=4242)
    calm = torch.from_numpy(CalM).cuda()
    ctx = api.Context(0)                                                     # library defaults
    pick = np.arange(0, 2500, 63)[:40]
    small = ctx.pose_batch(method, torch.from_numpy(np.ascontiguousarray(C[pick])).cuda(), calm, reconst=False)
    for B in (1023, 1024, 2500):
        idx = np.arange(B)
        pos = pick[pick < B] if B == 2500 else np.arange(len(pick)) * (B // len(pick))
        Cb = C[:B].copy()
        Cb[pos] = C[pick[:len(pos)]]
        big = ctx.pose_batch(method, torch.from_numpy(Cb).cuda(), calm, reconst=False)
        for k in ("T", "R_t_2", "R_t_3", "iter", "status"):
            assert torch.equal(big[k][torch.from_numpy(pos).cuda()] if big[k].is_cuda else big[k][pos], small[k][:len(pos)]), (method, B, k)


@pytest.mark.parametrize("method", ["ResslTFTPoseEstimation", "PiPoseEstimation", "LinearTFTPoseEstimation", "LinearFPoseEstimation", "OptimFPoseEstimation"])
def test_a_failed_triplet_has_all_nan_outputs_and_its_neighbours_do_not_notice(gpu_ctx, method):
    """Rows route with Reconst requested: a triplet with a NaN coordinate (status != 0) gets NaN in EVERY output, Reconst included -- its row still
    walks through the pose tail on a dummy tensor and must not store -- and the three triplets that share its wavefront are bit-identical to a
    run in which the failed one is replaced by a healthy triplet (rows_pose_tail: per-row Reconst store)."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B, N = 8, 24
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=515)
    calm = torch.from_numpy(CalM).cuda()
    ref = gpu_ctx.pose_batch(method, torch.from_numpy(C).cuda(), calm, reconst=True)
    Cb = C.copy()
    Cb[5, 7, 2] = np.nan
    out = gpu_ctx.pose_batch(method, torch.from_numpy(Cb).cuda(), calm, reconst=True)
    st = out["status"].cpu().numpy() if hasattr(out["status"], "cpu") else np.asarray(out["status"])
    assert st[5] != 0
    for k in ("T", "R_t_2", "R_t_3", "Reconst"):
        v = out[k].cpu().numpy() if hasattr(out[k], "cpu") else np.asarray(out[k])
        r = ref[k].cpu().numpy() if hasattr(ref[k], "cpu") else np.asarray(ref[k])
        assert np.all(np.isnan(v[5])), (method, k, v[5].ravel()[:6])
        for b in (0, 1, 2, 3, 4, 6, 7):
            assert np.array_equal(v[b], r[b]), (method, k, b)


@pytest.mark.parametrize("B,N", [(4099, 200), (1025, 60), (300, 500)])
def test_moments_pre_kernel_agrees_with_the_fused_passes(B, N):
    """TFF_OPT_PRE = 1 (k_tft_moments + k_linear_tft_pose_rows<true>, csrc/tft_moments_kernel.h; N = 500: the variant that re-reads global memory)
    against the default (the two passes inside the row kernel): same arithmetic per correspondence, sums in a different order -- 1e-12 -- and
    against the oracle at 1e-9; Ressl through k_gh_linear_rows<true>: same iteration counts."""
    import torch
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    from oracle import tft_oracle as O
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=B + N)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    ctx = api.Context(0)
    ctx.set_rows(1)
    ref = ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=True)
    ctx.set_pre(1)
    out = ctx.pose_batch("LinearTFTPoseEstimation", d, calm, reconst=True)
    assert torch.equal(out["status"], ref["status"]) and int((out["status"] != 0).sum()) == 0
    r, o = {k: v.cpu().numpy() for k, v in ref.items() if k in ("T", "R_t_2", "R_t_3", "Reconst")}, {k: v.cpu().numpy() for k, v in out.items() if k in ("T", "R_t_2", "R_t_3", "Reconst")}
    sg = np.sign(np.sum(r["T"] * o["T"], axis=(1, 2, 3)))[:, None, None, None]
    assert np.abs(o["T"] * sg - r["T"]).max() < 1e-12 and np.abs(o["R_t_2"] - r["R_t_2"]).max() < 1e-12
    assert np.abs(o["R_t_3"] - r["R_t_3"]).max() < 1e-12 * max(1.0, np.abs(r["R_t_3"]).max())
    assert np.abs(o["Reconst"] - r["Reconst"]).max() < 1e-10 * np.abs(r["Reconst"]).max()
    for b in (0, B // 3, B - 1):
        R2, R3, Rec, T, _ = O.LinearTFTPoseEstimation(C[b].T.copy(), CalM)
        assert rel_err_T(o["T"][b], T) < TOL and rel_err(o["R_t_2"][b], R2) < TOL and rel_err(o["R_t_3"][b], R3) < TOL and rel_err(o["Reconst"][b], Rec) < TOL
    if N <= 200:
        g1 = ctx.pose_batch("ResslTFTPoseEstimation", d[:512].contiguous(), calm, reconst=False)
        ctx.set_pre(0)
        g0 = ctx.pose_batch("ResslTFTPoseEstimation", d[:512].contiguous(), calm, reconst=False)
        assert int((g1["status"] != 0).sum()) == 0 and float((g1["iter"] != g0["iter"]).double().mean()) < 0.01


@pytest.mark.parametrize("method", ["ResslTFTPoseEstimation", "LinearTFTPoseEstimation"])
def test_a_row_the_exact_tiers_redo_does_not_touch_its_neighbours(gpu_ctx, method):
    """Rows route with Reconst: one triplet with collinear camera centres (the fast null vectors report, its row is redone on the exact tiers --
    in k_gh_finish_rows the whole wavefront walks through the pose tail a second time and only that row may store) among generic ones: every
    other triplet, the three that share its wavefront included, is bit-identical to a run in which it is replaced by a generic triplet."""
    import torch
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B, N = 12, 60
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=808)
    Cc, _, _, _ = generate_scene_batch(1, N, noise=1.0, seed=360, angle=180)
    calm = torch.from_numpy(CalM).cuda()
    ref = gpu_ctx.pose_batch(method, torch.from_numpy(C).cuda(), calm, reconst=True)
    Cx = C.copy(); Cx[6] = Cc[0]
    out = gpu_ctx.pose_batch(method, torch.from_numpy(Cx).cuda(), calm, reconst=True)
    assert int((out["status"] != 0).sum()) == 0 and int((ref["status"] != 0).sum()) == 0
    keep = torch.tensor([b for b in range(B) if b != 6]).cuda()
    for k in ("T", "R_t_2", "R_t_3", "Reconst", "iter"):
        assert torch.equal(out[k][keep], ref[k][keep]), (method, k)
    assert not torch.equal(out["T"][6], ref["T"][6])


@pytest.mark.parametrize("method,n", [("LinearTFTPoseEstimation", 7), ("LinearFPoseEstimation", 8), ("LinearTFTPoseEstimation", 14)])
def test_sampled_hypotheses_do_not_depend_on_the_chunk_they_arrive_in(method, n):
    """A RANSAC loop may evaluate its hypotheses in chunks of any size: on a DEFAULT context (route by batch size for the linear methods) the *_sampled
    entry points always take the row kernels, so a hypothesis gets the same bits alone in a chunk of 37, of 1 500 or in one call of 3 000."""
    import torch
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    Ns, H = 200, 3000
    Cs, CalM, _, _ = generate_scene_batch(1, Ns, noise=0.5, seed=21)
    scene = Cs[0].copy()
    rng = np.random.default_rng(3)
    bad = rng.choice(Ns, Ns // 5, replace=False)
    scene[bad, 2:6] += rng.uniform(20, 80, size=(bad.size, 4))
    d_scene = torch.from_numpy(scene).cuda(); calm = torch.from_numpy(CalM).cuda()
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    idx = torch.rand((H, Ns), device="cuda", generator=g).argsort(dim=1)[:, :n].to(torch.int32).contiguous()
    ctx = api.Context(0)                                                      # default options
    whole = ctx.pose_sampled(method, d_scene, calm, idx)
    for chunk in (37, 1500):
        parts = [ctx.pose_sampled(method, d_scene, calm, idx[s:s + chunk].contiguous()) for s in range(0, H, chunk)]
        for key in ("R_t_2", "R_t_3", "T"):
            got = torch.cat([p[key] for p in parts]).cpu().numpy(); ref = whole[key].cpu().numpy()
            assert np.array_equal(got, ref, equal_nan=True), (key, chunk)
        assert torch.equal(torch.cat([p["status"] for p in parts]), whole["status"])


@pytest.mark.parametrize("N", [40, 200])
def test_picol_plain_solve_with_certificate_agrees_with_the_eigen_decomposition(N):
    """PiColPoseEstimation's 38 x 38 KKT system (Gauss_Helmert.m:67, `pinv(M + 1e-12 I) * b`): the block kernel solves it by elimination when Sturm counts certify
    that pinv's tolerance truncates nothing (pi_wg_kernel.h::pi_spectrum_clears_tolerance; nearly every scene at N = 40, about half at N = 200) and by the
    eigen-decomposition otherwise; the fused kernel (TFF_OPT_KERNEL = 1) always takes the eigen-decomposition.  Same iteration counts, same poses."""
    import torch
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    B = 600
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=909 + N)
    d = torch.from_numpy(C).cuda(); calm = torch.from_numpy(CalM).cuda()
    blk = api.Context(0)
    fused = api.Context(0); fused.set_kernel_variant(1)
    a = blk.pose_batch("PiColPoseEstimation", d, calm, reconst=False)
    f = fused.pose_batch("PiColPoseEstimation", d, calm, reconst=False)
    ok = (a["status"] == 0) & (f["status"] == 0)
    assert int(ok.sum()) >= B - 3
    same_it = (a["iter"] == f["iter"]) & ok
    assert int(same_it.sum()) >= int(0.99 * B)
    Ta = a["T"][same_it].reshape(-1, 27); Tf = f["T"][same_it].reshape(-1, 27)
    s = torch.sign((Ta * Tf).sum(dim=1, keepdim=True))
    assert float((s * Ta - Tf).abs().max()) < 1e-6
    assert float((a["R_t_3"][same_it] - f["R_t_3"][same_it]).abs().max()) < 1e-6
