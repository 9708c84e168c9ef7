"""A/B of TFF_OPT_PRE on one GPU: the normalisations + moment sums of the trifocal row kernels inside k_linear_tft_pose_rows<false> (0) against
k_tft_moments + k_linear_tft_pose_rows<true> (1), for a sweep of N; one batch at a time (HIP events around K back-to-back calls on one stream)
and two batches in flight (two contexts on their own streams, wall clock).  Prints the agreement of the two routes.
Usage: python tools/ab_pre.py [B] [K] [method] [N ...]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
METHOD = sys.argv[3] if len(sys.argv) > 3 else "LinearTFTPoseEstimation"
NS = [int(x) for x in sys.argv[4:]] or [12, 32, 48, 64, 100, 200, 300, 500, 1000]
dev = torch.device("cuda", 0)
ctxs = [api.Context(0), api.Context(0)]
for c in ctxs:
    c.set_rows(1)
    c.use_own_stream()
lib = ctxs[0].lib
fn = getattr(lib, api.POSE_METHODS[METHOD] + "_dev")
p = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 8 * off)
for N in NS:
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=N)
    d = torch.from_numpy(C).to(dev)
    calm = torch.from_numpy(np.ascontiguousarray(CalM.T).reshape(27)).to(dev)
    recs = [torch.zeros(51 * B, dtype=torch.float64, device=dev) for _ in range(2)]
    sts = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
    its = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
    def call(j):
        rc = fn(ctxs[j].handle, p(d), p(calm), 0, B, N, p(recs[j], 0), p(recs[j], 12 * B), p(recs[j], 24 * B), None, ctypes.c_void_p(its[j].data_ptr()), ctypes.c_void_p(sts[j].data_ptr()))
        assert rc == 0, lib.tff_last_error()
    out = {}
    line = "N=%4d:" % N
    for pre in (0, 1):
        for c in ctxs:
            c.set_pre(pre)
        for _ in range(3):
            call(0); call(1)
        for c in ctxs: c.synchronize()
        s0 = torch.cuda.ExternalStream(ctxs[0].stream_ptr(), device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s0)
        for _ in range(K): call(0)
        e1.record(s0)
        ctxs[0].synchronize()
        ms1 = e0.elapsed_time(e1) / K
        t0 = time.perf_counter()
        for k in range(2 * K): call(k % 2)
        for c in ctxs: c.synchronize()
        ms2 = 1e3 * (time.perf_counter() - t0) / (2 * K)
        out[pre] = recs[0].clone()
        line += "  pre=%d one batch %.4f ms (%.1f M/s), two in flight %.4f ms (%.1f M/s), failed %d;" % (pre, ms1, B / ms1 / 1e3, ms2, B / ms2 / 1e3, int((sts[0] != 0).sum()))
    a, b = out[0].cpu().numpy(), out[1].cpu().numpy()
    T0, T1 = a[24 * B:].reshape(B, 27), b[24 * B:].reshape(B, 27)
    sg = np.sign(np.sum(T0 * T1, axis=1))[:, None]
    line += "  agree: T %.1e Rt %.1e" % (np.nanmax(np.abs(T0 * sg - T1)), np.nanmax(np.abs(a[:24 * B] - b[:24 * B]) / np.maximum(1.0, np.abs(a[:24 * B]))))
    print(line, flush=True)
