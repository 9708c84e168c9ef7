"""Which way of issuing consecutive batches on two streams overlaps them?  (diagnostic for bench.py)"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch
B, N, K = 10000, 200, 100
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
dev = torch.device("cuda", 0)
d = torch.from_numpy(C).to(dev); calm = torch.from_numpy(np.ascontiguousarray(CalM.T).reshape(27)).to(dev)
p = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 8 * off)

def run(tag, S, mode):
    ctxs = [api.Context(0) for _ in range(S)]
    lib = ctxs[0].lib
    recs = [torch.empty(51 * B, dtype=torch.float64, device=dev) for _ in range(S)]
    sts = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(S)]
    if mode == "torch":
        side = [torch.cuda.Stream(dev) for _ in range(S)]
        for c, s in zip(ctxs, side):
            c.set_stream(s.cuda_stream)
    elif mode == "own":
        pass                                               # every context keeps the stream it created
    elif mode == "current":
        for c in ctxs:
            c.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    def step(k):
        j = k % S
        rc = lib.tff_linear_tft_pose_batch_dev(ctxs[j].handle, p(d), p(calm), 0, B, N, p(recs[j], 0), p(recs[j], 12 * B), p(recs[j], 24 * B), None, None,
                                               ctypes.c_void_p(sts[j].data_ptr()))
        assert rc == 0
    for k in range(6):
        step(k)
    torch.cuda.synchronize(); [c.synchronize() for c in ctxs]
    t0 = time.perf_counter()
    for k in range(K):
        step(k)
    [c.synchronize() for c in ctxs]; torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-28s S=%d: %.3f ms per batch = %.2f M triplets/s" % (tag, S, 1e3 * dt / K, K * B / dt / 1e6), flush=True)

run("current stream", 1, "current")
run("own streams", 1, "own")
run("own streams", 2, "own")
run("torch pool streams", 2, "torch")
run("own streams", 3, "own")
run("own streams", 4, "own")
