"""Same-box A/B of library builds: every libtftfund*.so given on the command line (default: tools/ab_libs/*.so and the in-tree library) times every
pose method on the configs[1] batch, one batch at a time (HIP events around K back-to-back calls), interleaved over R rounds so that clock /
temperature drift hits all builds alike; prints the median per build and method, and the agreement of T with the first build.
Usage: python tools/ab_libs.py [K] [R] [lib.so ...]"""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tft_vs_fund_amd import api
from tft_vs_fund_amd.scenes import generate_scene_batch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
libs = sys.argv[3:] or sorted(glob.glob(os.path.join(ROOT, "tools", "ab_libs", "*.so"))) + [os.path.join(ROOT, "tft_vs_fund_amd", "libtftfund.so")]
B, N = 10000, 200
C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1)
dev = torch.device("cuda", 0)
d = torch.from_numpy(C).to(dev); calm = torch.from_numpy(np.ascontiguousarray(CalM.T).reshape(27)).to(dev)
p = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 8 * off)
stream = torch.cuda.current_stream(dev)
handles = []
for path in libs:
    lib = ctypes.CDLL(path)
    lib.tff_ctx_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    lib.tff_ctx_set_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.tff_last_error.restype = ctypes.c_char_p
    h = ctypes.c_void_p()
    assert lib.tff_ctx_create(ctypes.byref(h), 0) == 0
    assert lib.tff_ctx_set_stream(h, ctypes.c_void_p(stream.cuda_stream)) == 0
    handles.append((os.path.basename(path), lib, h))
rec = torch.zeros(51 * B, dtype=torch.float64, device=dev)
it = torch.zeros(B, dtype=torch.int32, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
methods = ["ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "PiPoseEstimation", "PiColPoseEstimation", "FaugPapaTFTPoseEstimation", "OptimFPoseEstimation",
           "LinearTFTPoseEstimation", "LinearFPoseEstimation"]
times = {(n, m): [] for n, _, _ in handles for m in methods}
Tref = {}
for r in range(R):
    for m in methods:
        for name, lib, h in handles:
            fn = getattr(lib, api.POSE_METHODS[m] + "_dev")
            fn.argtypes = None
            call = lambda: fn(h, p(d), p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N), p(rec, 0), p(rec, 12 * B), p(rec, 24 * B), None,
                              ctypes.c_void_p(it.data_ptr()), ctypes.c_void_p(st.data_ptr()))
            for _ in range(2):
                assert call() == 0, lib.tff_last_error()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(K):
                call()
            e1.record(stream)
            torch.cuda.synchronize(dev)
            times[(name, m)].append(e0.elapsed_time(e1) / K)
            if r == 0:
                T = rec[24 * B:].clone()
                if m not in Tref:
                    Tref[m] = (T, it.clone())
                    agree = ""
                else:
                    sg = torch.sign((T.view(B, 27) * Tref[m][0].view(B, 27)).sum(1, keepdim=True))
                    agree = "  vs first build: max |dT| %.1e, iter differs in %d, failed %d" % (float((T.view(B, 27) * sg - Tref[m][0].view(B, 27)).abs().nan_to_num().max()),
                                                                                             int((it != Tref[m][1]).sum()), int((st != 0).sum()))
                print("%-28s %-22s %.4f ms%s" % (m, name, times[(name, m)][-1], agree), flush=True)
print("\nmedian of %d rounds x %d calls, ms per 10 000 x 200:" % (R, K))
print("%-28s" % "method" + "".join("%22s" % n for n, _, _ in handles))
for m in methods:
    print("%-28s" % m + "".join("%22.4f" % float(np.median(times[(n, m)])) for n, _, _ in handles))
