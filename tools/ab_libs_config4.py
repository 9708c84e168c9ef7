"""Same-box A/B of library builds on config 4 (1 M minimal-sample hypotheses of one 400-correspondence scene with 25 % gross outliers): the pose
launch of every libtftfund*.so (default: tools/ab_libs/*.so and the in-tree library), HIP events, interleaved rounds, median; agreement of the
statuses and poses with the first build.   python tools/ab_libs_config4.py [H] [R] [lib.so ...]"""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tft_vs_fund_amd.scenes import generate_scene_batch

H = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
libs = sys.argv[3:] or sorted(glob.glob(os.path.join(ROOT, "tools", "ab_libs", "*.so"))) + [os.path.join(ROOT, "tft_vs_fund_amd", "libtftfund.so")]
Ns = 400
dev = torch.device("cuda", 0)
C, CalM, _, _ = generate_scene_batch(1, Ns, noise=0.5, seed=7)
scene = C[0].copy()
rng = np.random.default_rng(1)
bad = rng.choice(Ns, Ns // 4, replace=False)
scene[bad, 2:6] += rng.uniform(20, 80, size=(bad.size, 4))
d_scene = torch.from_numpy(scene).to(dev); calm = torch.from_numpy(np.ascontiguousarray(CalM.T).reshape(27)).to(dev)
stream = torch.cuda.current_stream(dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
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
out = {}
for m, stem, n in (("LinearTFTPoseEstimation", "tff_linear_tft_pose_sampled_dev", 7), ("LinearFPoseEstimation", "tff_linear_f_pose_sampled_dev", 8)):
    g = torch.Generator(device=dev); g.manual_seed(1234)
    idx = torch.rand((H, Ns), device=dev, generator=g).argsort(dim=1)[:, :n].to(torch.int32).contiguous()
    Rt2 = torch.empty((H, 12), dtype=torch.float64, device=dev); Rt3 = torch.empty_like(Rt2); T = torch.empty((H, 27), dtype=torch.float64, device=dev)
    st = torch.zeros(H, dtype=torch.int32, device=dev)
    times = {nme: [] for nme, _, _ in handles}
    ref = None
    for r in range(R + 1):
        for nme, lib, h in handles:
            fn = getattr(lib, stem)
            call = lambda: fn(h, p(d_scene), ctypes.c_int(Ns), p(calm), p(idx), ctypes.c_long(H), ctypes.c_int(n), p(Rt2), p(Rt3), p(T), p(st))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record(stream)
            assert call() == 0, lib.tff_last_error()
            e1.record(stream)
            torch.cuda.synchronize(dev)
            if r > 0:
                times[nme].append(e0.elapsed_time(e1))
            elif ref is None:
                ref = (Rt3.clone(), st.clone())
            else:
                same = (st == ref[1])
                print("%s %s: status differs in %d hypotheses, failed %d; max |dR_t_3| on equal status %.1e" % (
                    m, nme, int((~same).sum()), int((st != 0).sum()), float((Rt3 - ref[0])[same & (st == 0)].abs().nan_to_num().max())))
    print("%-26s" % m + "".join("  %s %.2f ms" % (nme, float(np.median(times[nme]))) for nme, _, _ in handles), flush=True)
    del idx
