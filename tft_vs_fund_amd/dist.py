"""
Multi-GPU layer: one process per GPU, independent triplets sharded as contiguous
blocks of the batch index, no collective on the data path; a single all-gather
of fixed-size result records (RCCL over xGMI; `nccl` backend == RCCL on ROCm)
brings the results together.  The reference has no parallelism
at all (experiments.m:74-144 runs its independent trials serially).

The same code runs under `gloo` on CPU tensors for the world_size-2 tests.
"""
import os

import torch
import torch.distributed as dist

RECORD_DOUBLES = 51          # R_t_2 (12) + R_t_3 (12) + T (27)


def shard_bounds(B, world, rank):
    """Contiguous block [lo, hi) of the batch owned by `rank` (GPU g of G gets [g*B/G, (g+1)*B/G))."""
    return (rank * B) // world, ((rank + 1) * B) // world


def init_from_env(device_type=None, force_group=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (set by torch.distributed.run).
    Returns (rank, world, local_rank).  Single process: (0, 1, 0) without a process group -- unless force_group asks for a
    clique-of-one group (a test seam: the collective path then executes against RCCL on one GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if device_type is None:
            device_type = "cuda" if torch.cuda.is_available() else "cpu"
        if device_type == "cuda":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, local


def pack_records(Rt2, Rt3, T):
    """(b,12), (b,12), (b,27) -> one (51*b,) record block [all Rt2 | all Rt3 | all T]."""
    return torch.cat([Rt2.reshape(-1), Rt3.reshape(-1), T.reshape(-1)])


def unpack_records(block, b):
    Rt2 = block[: 12 * b].reshape(b, 12)
    Rt3 = block[12 * b: 24 * b].reshape(b, 12)
    T = block[24 * b: 51 * b].reshape(b, 27)
    return Rt2, Rt3, T


def all_gather_records(local_block, B, async_op=False, out=None):
    """All-gather the per-rank record blocks of a batch of B triplets sharded with
    shard_bounds.  Shards may differ by one triplet; blocks are padded to the
    largest shard.  Returns (gathered (world, 51*bmax) tensor, work handle or None)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    bmax = -(-B // world)
    n = RECORD_DOUBLES * bmax
    if local_block.numel() < n:
        local_block = torch.cat([local_block, local_block.new_zeros(n - local_block.numel())])
    if world == 1:
        return local_block.reshape(1, n), None
    if out is None:
        out = local_block.new_empty((world, n))
    work = dist.all_gather_into_tensor(out.reshape(-1), local_block.contiguous(), async_op=async_op)
    return out, work


def assemble(gathered, B):
    """(world, 51*bmax) gathered blocks -> Rt2 (B,12), Rt3 (B,12), T (B,27) in batch order."""
    world = gathered.shape[0]
    bmax = -(-B // world)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(B, world, r)
        b = hi - lo
        blk = gathered[r]
        # block layout uses the padded shard size only if the sender padded *after* packing its own b
        Rt2 = blk[: 12 * b].reshape(b, 12)
        Rt3 = blk[12 * b: 24 * b].reshape(b, 12)
        T = blk[24 * b: 51 * b].reshape(b, 27)
        parts.append((Rt2, Rt3, T))
    return (torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]), torch.cat([p[2] for p in parts]))


def all_gather_counts(local_counts, B):
    """int32 per-hypothesis counts (config 4: inlier counts) -> all ranks, batch order."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local_counts
    bmax = -(-B // world)
    pad = local_counts.new_zeros(bmax)
    pad[: local_counts.numel()] = local_counts
    out = local_counts.new_empty(world * bmax)
    dist.all_gather_into_tensor(out, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(B, world, r)
        parts.append(out[r * bmax: r * bmax + (hi - lo)])
    return torch.cat(parts)


class OverlappedGather:
    """The double-buffered step / gather pipeline of bench.py: step k computes into record buffer k % nbuf and then
    all-gathers it with async_op=True, so that the gather of step k overlaps the compute of step k + 1; a buffer is reused only
    after its previous gather has completed.  `compute(record_buffer, k)` enqueues the work of one step (the HIP launch in
    bench.py; a stub in the gloo test).  world == 1: no collective, no extra buffers.

    Several compute streams (bench.py --streams 2): call `step(k)` inside `with torch.cuda.stream(stream_of_step_k)` -- the requirement below then
    holds per step, and a step's gather is ordered against that step's stream only.

    STREAM REQUIREMENT: `compute` must enqueue its work on torch's CURRENT stream of `device` -- the stream the process group orders
    its collectives against (all_gather_into_tensor waits for the current stream's work, Work.wait() makes the current stream wait
    for the collective).  A tft_vs_fund_amd.api.Context launches on its own non-blocking stream by default: call
    `ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)` first (bench.py does), or the gather reads half-written records
    and the buffer reuse is unprotected.  The gloo CPU test cannot see a violation of this."""

    def __init__(self, world, numel, device, compute, nbuf=2, dtype=torch.float64, collective=None):
        self.world, self.nbuf, self.compute = world, nbuf, compute
        self.collective = (world > 1) if collective is None else bool(collective)      # (True at world 1: a clique-of-one group, test seam)
        self.recs = [torch.empty(numel, dtype=dtype, device=device) for _ in range(nbuf)]
        self.gathered = [torch.empty((world, numel), dtype=dtype, device=device) for _ in range(nbuf)] if self.collective else None
        self.pending = [None] * nbuf

    def step(self, k):
        buf = k % self.nbuf
        if self.pending[buf] is not None:              # the gather that last used this buffer
            self.pending[buf].wait()
            self.pending[buf] = None
        r = self.recs[buf]
        self.compute(r, k)
        if self.collective:
            self.pending[buf] = dist.all_gather_into_tensor(self.gathered[buf].reshape(-1), r, async_op=True)
        return buf

    def drain(self):
        for i in range(self.nbuf):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None
