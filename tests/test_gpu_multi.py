"""Multi-GPU behind the C ABI (include/tftfund.h, multi-GPU section): tff_multi_* over however many devices the
box has.  One device: the same code path with a clique of one (host threads, shard bounds, ncclCommInitAll, ncclAllGather).
Two or more: shards really run on different devices (skipped otherwise: the driver's boxes have one GPU -- UNMEASURED ON HARDWARE
with more than one until a multi-GPU node runs this file)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from helpers import rel_err_T, rel_err   # noqa: E402

pytestmark = pytest.mark.gpu


def _devices():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("method", ["LinearTFTPoseEstimation", "LinearFPoseEstimation", "ResslTFTPoseEstimation"])
def test_host_multi_equals_single_device(method):
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    G = _devices()
    C, CalM, _, _ = generate_scene_batch(101, 40, noise=1.0, seed=17)        # odd batch: uneven shards when G > 1
    ref = api.Context(0).pose_batch(method, C, CalM, reconst=True)
    m = api.MultiContext(list(range(G)))
    assert m.size == G
    bounds = [m.shard(101, g) for g in range(G)]
    assert bounds[0][0] == 0 and bounds[-1][1] == 101 and all(bounds[g][1] == bounds[g + 1][0] for g in range(G - 1))
    out = m.pose_batch(method, C, CalM, reconst=True)
    for k in ("T", "R_t_2", "R_t_3", "Reconst", "iter", "status"):
        assert np.array_equal(np.asarray(out[k]), np.asarray(ref[k]), equal_nan=True), (method, k)
    # per-triplet calibration (calm_stride 27) through the sharded path
    CalB = np.broadcast_to(CalM, (101, 9, 3)).copy()
    out2 = m.pose_batch(method, C, CalB, reconst=False)
    assert np.array_equal(out2["T"], ref["T"])
    m.close()


@pytest.mark.parametrize("B", [37, 1])                                         # 37: uneven last shard for G >= 2; 1: empty shards for G >= 2 (B < G)
def test_dev_multi_gathers_records_on_every_device(B):
    import torch
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    G = _devices()
    N = 30
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=23)
    ref = api.Context(0).pose_batch("LinearTFTPoseEstimation", C, CalM, reconst=False)
    m = api.MultiContext(list(range(G)))
    shards, calms = [], []
    for g in range(G):
        b0, b1 = m.shard(B, g)
        dev = torch.device("cuda", g)
        shards.append(torch.from_numpy(np.ascontiguousarray(C[b0:b1])).to(dev))
        calms.append(torch.from_numpy(CalM).to(dev))
    recs, sts, chunk = m.pose_batch_dev("LinearTFTPoseEstimation", shards, calms, B)
    for g in range(G):                                                        # every device holds every shard
        r = recs[g].cpu().numpy().reshape(G, chunk * 51)
        st = sts[g].cpu().numpy().reshape(G, chunk)
        for src in range(G):
            b0, b1 = m.shard(B, src)
            n = b1 - b0
            Rt2 = r[src, :12 * chunk].reshape(chunk, 12)[:n].reshape(n, 4, 3).transpose(0, 2, 1)
            T = r[src, 24 * chunk:51 * chunk].reshape(chunk, 27)[:n].reshape(n, 3, 3, 3).transpose(0, 3, 2, 1)
            assert np.array_equal(Rt2, ref["R_t_2"][b0:b1]) and np.array_equal(T, ref["T"][b0:b1]), (g, src)
            assert np.all(st[src, :n] == 0)
    m.close()


def test_dev_multi_gives_the_streams_back_and_leaves_the_device_alone():
    """pose_batch_dev lends every context torch's current stream for the call only: afterwards the contexts launch on their own streams
    again (a user stream destroyed after the call must not be touched), and the calling thread's current device is what it was."""
    import ctypes
    import torch
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    G = _devices()
    B, N = 9, 20
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=3)
    m = api.MultiContext(list(range(G)))
    own = [int(m.lib.tff_ctx_get_stream(m.lib.tff_multi_ctx(m.handle, g)) or 0) for g in range(G)]
    shards, calms = [], []
    for g in range(G):
        b0, b1 = m.shard(B, g)
        dev = torch.device("cuda", g)
        shards.append(torch.from_numpy(np.ascontiguousarray(C[b0:b1])).to(dev))
        calms.append(torch.from_numpy(CalM).to(dev))
    torch.cuda.set_device(0)
    user = torch.cuda.Stream(torch.device("cuda", 0))
    with torch.cuda.stream(user):
        recs, sts, chunk = m.pose_batch_dev("LinearTFTPoseEstimation", shards, calms, B)
    assert torch.cuda.current_device() == 0
    now = [int(m.lib.tff_ctx_get_stream(m.lib.tff_multi_ctx(m.handle, g)) or 0) for g in range(G)]
    assert now == own and user.cuda_stream not in now
    del user
    out = m.pose_batch("LinearTFTPoseEstimation", C, CalM, reconst=False)      # runs on the contexts' own streams
    ref = api.Context(0).pose_batch("LinearTFTPoseEstimation", C, CalM, reconst=False)
    assert np.array_equal(out["T"], ref["T"]) and torch.cuda.current_device() == 0
    m.close()


def test_multi_rejects_bad_arguments():
    from tft_vs_fund_amd import api
    with pytest.raises(api.TffError):
        api.MultiContext([0, 0])                                             # duplicate device
    with pytest.raises(api.TffError):
        api.MultiContext([_devices() + 3])                                   # no such device
    m = api.MultiContext([0])
    import ctypes
    rc = m.lib.tff_pose_batch_host_multi(m.handle, 99, None, None, 0, 0, 10, None, None, None, None, None, None)
    assert rc != 0 and b"method" in m.lib.tff_last_error()
    m.close()


@pytest.mark.skipif("_devices() < 2")
def test_two_devices_really_split_the_batch():
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.scenes import generate_scene_batch
    C, CalM, _, _ = generate_scene_batch(2000, 100, noise=1.0, seed=5)
    ref = api.Context(0).pose_batch("LinearTFTPoseEstimation", C, CalM, reconst=False)
    m = api.MultiContext([0, 1])
    out = m.pose_batch("LinearTFTPoseEstimation", C, CalM, reconst=False)
    assert np.array_equal(out["T"], ref["T"]) and m.shard(2000, 1) == (1000, 2000)
    m.close()
