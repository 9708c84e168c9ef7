"""World-size-2 test of the sharding + result gather on CPU tensors (gloo).
The compute itself has no CPU path; records here are synthetic."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from tft_vs_fund_amd import dist as tdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, B, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = tdist.init_from_env("cpu")
    lo, hi = tdist.shard_bounds(B, w, r)
    full = torch.arange(B * 51, dtype=torch.float64).reshape(B, 51)        # record b = 51 known numbers
    Rt2, Rt3, T = full[lo:hi, :12], full[lo:hi, 12:24], full[lo:hi, 24:]
    g, work = tdist.all_gather_records(tdist.pack_records(Rt2, Rt3, T), B, async_op=True)
    work.wait()
    a2, a3, aT = tdist.assemble(g, B)
    ok = torch.equal(a2, full[:, :12]) and torch.equal(a3, full[:, 12:24]) and torch.equal(aT, full[:, 24:])
    counts = torch.arange(lo, hi, dtype=torch.int32)
    allc = tdist.all_gather_counts(counts, B)
    ok = ok and torch.equal(allc, torch.arange(B, dtype=torch.int32))
    q.put((rank, bool(ok)))
    torch.distributed.destroy_process_group()


def test_shard_bounds_cover_batch():
    for B in (0, 1, 7, 10000, 1000003):
        for world in (1, 2, 3, 8):
            b = [tdist.shard_bounds(B, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == B
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_gather_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    for B in (10, 11, 1):                            # even and uneven shards, and an empty one (B < world)
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
        for p in ps: p.start()
        res = sorted(q.get(timeout=120) for _ in range(2))
        for p in ps: p.join(timeout=60)
        assert res == [(0, True), (1, True)]
