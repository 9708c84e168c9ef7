"""bench.py's double-buffered step / gather pipeline (tft_vs_fund_amd.dist.OverlappedGather) under gloo, world size 2, with the
compute stubbed: every step's records must arrive complete and unmixed on every rank although the gather of step k is still in
flight when step k + 1 computes, and a buffer must not be rewritten before its previous gather has completed."""
import os
import socket

import torch
import torch.multiprocessing as mp

from tft_vs_fund_amd import dist as tdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, steps, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = tdist.init_from_env("cpu")
    numel = tdist.RECORD_DOUBLES * 40
    seen = []

    def compute(rec, k):                              # stub of the HIP launch: the records of (rank, step)
        rec.copy_(torch.full((numel,), float(1000 * r + k), dtype=torch.float64) + torch.arange(numel, dtype=torch.float64) * 1e-6)

    pipe = tdist.OverlappedGather(w, numel, torch.device("cpu"), compute)
    ok = True
    for k in range(steps):
        buf = pipe.step(k)
        if k >= 1:                                    # the gather of step k - 1 may still be in flight: wait for it, then check it
            pb = (k - 1) % pipe.nbuf
            if pipe.pending[pb] is not None:
                pipe.pending[pb].wait(); pipe.pending[pb] = None
            g = pipe.gathered[pb]
            for src in range(w):
                exp = torch.full((numel,), float(1000 * src + k - 1), dtype=torch.float64) + torch.arange(numel, dtype=torch.float64) * 1e-6
                ok = ok and torch.equal(g[src], exp)
        seen.append(buf)
    pipe.drain()
    g = pipe.gathered[(steps - 1) % pipe.nbuf]
    for src in range(w):
        exp = torch.full((numel,), float(1000 * src + steps - 1), dtype=torch.float64) + torch.arange(numel, dtype=torch.float64) * 1e-6
        ok = ok and torch.equal(g[src], exp)
    ok = ok and seen == [k % 2 for k in range(steps)] and all(p is None for p in pipe.pending)
    q.put((rank, bool(ok)))
    torch.distributed.destroy_process_group()


def test_overlapped_gather_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in ps: p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_overlapped_gather_single_rank_has_no_collective():
    calls = []
    pipe = tdist.OverlappedGather(1, 8, torch.device("cpu"), lambda rec, k: calls.append(k) or rec.fill_(k))
    for k in range(5):
        pipe.step(k)
    pipe.drain()
    assert calls == [0, 1, 2, 3, 4] and pipe.gathered is None
