#!/usr/bin/env python3
"""
bench.py -- triplet-hypotheses/sec of the hot path on MI355X.

Workload (BASELINE.json configs[1]): a batch of 10 000 synthetic three-view
scenes (geometry of generateSyntheticScene.m, sigma = 1 px), 200
correspondences each; one "step" = linearTFT + R_t_from_TFT for the whole batch
(LinearTFTPoseEstimation without the Reconst output), inputs resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 is launched by the driver as `python -m torch.distributed.run ...`: one
process per GPU, every rank owns a batch of its own (weak scaling, independent
triplets, no collective on the data path) and the fixed-size result records are
all-gathered over RCCL, overlapped with the next step's compute.

Steps are independent batches, so they are issued round-robin on `--streams`
HIP streams (default 2, one library context per stream, result records
double-buffered): 10 000 triplets are 2 500 wavefronts of four triplets on
2 048 wavefront slots, and the next batch's wavefronts fill the slots the
previous batch's tail leaves idle.  Every step still runs whole inside the
timed region.  What the line reports, so that it closes on itself:
  * `value`, `ms_per_step`: K steps between barrier + synchronize, repeated
    `--reps` times, the MEDIAN repetition (all of them listed);
  * `in_flight`: batches that may be resident at once (= streams);
  * `roofline.kernel_ms` / `achieved` / `frac`: one call's launches with the
    GPU TO ITSELF (one stream, HIP events on that stream around every call,
    nothing else queued) -- the figure `rocprofv3 --kernel-trace --stats` of
    `bench.py --streams 1` reproduces (profiles/);
  * `overlap_factor` = kernel_ms / ms_per_step (<= in_flight): how much of a
    launch is hidden behind its neighbour; kernel_ms * steps / in_flight <=
    ms_per_step * steps can be checked from the line alone.

Rank 0 prints ONE JSON line (metric, value, roofline, cpu_baseline, ...).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_triplet(N, reconst=False):
    # in Corresp 48N + CalM 216; out T 216 + R_t_2,R_t_3 192 (+ Reconst 24N + iter,status 8)
    return 48 * N + 216 + 216 + 192 + ((24 * N + 8) if reconst else 0)


def cpu_baseline(C, CalM, sample):
    """The plain-C restatement (oracle/tft_oracle_c.c, 'port') on all host cores, bounded sample."""
    from oracle import c_oracle
    cores = os.cpu_count() or 1
    c_oracle.linear_tft_pose_batch(C[:max(2, min(cores, sample))], CalM, reconst=False, threads=cores)   # warm
    t0 = time.perf_counter()
    out = c_oracle.linear_tft_pose_batch(C[:sample], CalM, reconst=False, threads=cores)
    dt = time.perf_counter() - t0
    return dict(value=sample / dt, unit="triplet-hypotheses/s", cores=int(out["threads"]), kind="port",
                sample="first %d triplets of the same batch (N=%d), %.1f s wall, C restatement of the reference "
                       "algorithm (explicit 4Nx27 SVDs), OpenMP over triplets" % (sample, C.shape[1], dt))


def cpu_baseline_lapack(C, CalM, budget_s=8.0):
    """The LAPACK-backed numpy oracle (oracle/tft_oracle.py: full svd() calls as the MATLAB reference makes them) on ONE core,
    a bounded sample: the closer stand-in for MATLAB's own arithmetic (MATLAB cannot run here)."""
    from oracle import tft_oracle as O
    n, t0 = 0, time.perf_counter()
    while n < C.shape[0] and time.perf_counter() - t0 < budget_s:
        O.LinearTFTPoseEstimation(C[n].T.copy(), CalM)
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="triplet-hypotheses/s", cores=1, kind="port",
                sample="first %d triplets of the same batch (N=%d), %.1f s wall, numpy/LAPACK restatement (full SVDs, per-point "
                       "triangulation loops in Python), one core" % (n, C.shape[1], dt))


def cpu_baseline_reference(N):
    """The reference's OWN timing, when someone with MATLAB has produced tests/golden/reference_golden.mat (matlab/reference_pin/
    make_reference_golden.m: tic/toc around the reference's LinearTFTPoseEstimation on the committed fixture inputs).  Measured on that
    person's machine, not on this GPU box -- reported as such, beside the port's figure measured here."""
    path = os.path.join(ROOT, "tests", "golden", "reference_golden.mat")
    if not os.path.exists(path):
        return None
    try:
        from scipy.io import loadmat
        m = loadmat(path, squeeze_me=True, struct_as_record=False)
        rows = []
        for r in np.atleast_1d(m["results"]):
            o = getattr(r, "LinearTFTPoseEstimation", None)
            if o is not None and np.isfinite(float(o.seconds)):
                rows.append((abs(int(r.N) - N), int(r.N), float(o.seconds)))
        if not rows:
            return None
        d0 = min(rows)[0]
        secs = [s_ for d_, n_, s_ in rows if d_ == d0]
        n_used = [n_ for d_, n_, s_ in rows if d_ == d0][0]
        info = m.get("info")
        return dict(value=1.0 / float(np.median(secs)), unit="triplet-hypotheses/s", cores=float(getattr(info, "threads", float("nan"))), kind="reference",
                    sample="%d calls of the reference's LinearTFTPoseEstimation at N=%d (full wrapper incl. Reconst), best-of-%s tic/toc each, %s on %s -- "
                           "timed by matlab/reference_pin/make_reference_golden.m on the file author's machine, NOT on this GPU box"
                           % (len(secs), n_used, getattr(info, "repeats", "?"), getattr(info, "release", "?"), getattr(info, "computer", "?")))
    except Exception as ex:
        return {"error": repr(ex)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=10000)
    ap.add_argument("--ncorr", type=int, default=200)
    ap.add_argument("--cpu-sample", type=int, default=0, help="triplets timed on the CPU (default: 40 per host core, at least 1024, at most the batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary blocks (other methods, N sweep, config 4; 1 GPU, rank 0)")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the steps alternate between (1 = strictly one batch at a time)")
    ap.add_argument("--reps", type=int, default=7, help="repetitions of the timed K-step region; the median repetition is reported")
    ap.add_argument("--force-process-group", action="store_true",
                    help="TEST SEAM (tests/test_gpu_process_group.py): at WORLD_SIZE = 1 still create the nccl (= RCCL) process group, a clique of one, and "
                         "run the overlapped all-gather of every step through it; the line then carries `process_group_check`")
    ap.add_argument("--stub-compute", action="store_true",
                    help="TEST SEAM (tests/test_bench_flow_gloo.py): CPU tensors + gloo, the HIP launch replaced by a tagged fill -- everything around it "
                         "(rendezvous, rank-0 build + barrier, step / gather pipeline, weak-scaling accounting, the JSON line) runs as on the GPUs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tft_vs_fund_amd import api, dist as tdist
    from tft_vs_fund_amd.build import build_library
    from tft_vs_fund_amd.scenes import generate_scene_batch

    stub = args.stub_compute
    rank, world, local = tdist.init_from_env("cpu" if stub else "cuda", force_group=args.force_process_group)
    use_pg = world > 1 or args.force_process_group
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    if stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if rank == 0 and not stub:
        build_library()
    if world > 1:
        dist.barrier()
    S = max(1, args.streams)
    ctxs = [api.Context(local) for _ in range(S)] if not stub else []
    ctx = ctxs[0] if ctxs else None
    dsync = (lambda: None) if stub else (lambda: torch.cuda.synchronize(dev))

    B, N = args.batch, args.ncorr
    C, CalM, _, _ = generate_scene_batch(B, N, noise=1.0, seed=1000 + rank)
    d_C = torch.from_numpy(C).to(dev)
    d_calm = torch.from_numpy(np.ascontiguousarray(CalM.T).reshape(27)).to(dev)

    # result records, double-buffered so that the gather of step k overlaps the compute of step k+1 (tdist.OverlappedGather)
    statuses = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(S)]
    status = statuses[0]
    import ctypes
    lib = ctx.lib if ctx else None
    stream = torch.cuda.current_stream(dev) if not stub else None   # the stream the process group orders its gathers against
    if stub:
        side = []
    elif S > 1:
        # every context launches on the stream the library created for it (hipStreamNonBlocking); measured: two of torch's pool streams
        # do not overlap their kernels on this stack, the contexts' own streams do (tools/ab_streams2.py)
        for c in ctxs:
            c.use_own_stream()
        side = [torch.cuda.ExternalStream(c.stream_ptr(), device=dev) for c in ctxs]
    else:
        side = [stream]
        ctx.set_stream(stream.cuda_stream)
    p = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 8 * off)
    timing = {"events": None}
    stub_ms = []

    def compute_stub(r, k):                                # the records of (rank, step): what the gather check at the end looks for
        t_ = time.perf_counter()
        r.copy_(torch.full((r.numel(),), float(1000 * rank + k), dtype=torch.float64) + torch.arange(r.numel(), dtype=torch.float64) * 1e-9)
        if timing["events"] is not None:
            stub_ms.append(1e3 * (time.perf_counter() - t_))

    def compute(r, k):
        # step k runs on stream k % S (result buffer k % S: a stream only ever reuses its own buffer).  While a step is issued its stream is
        # torch's CURRENT stream (step() below), so the process group orders the step's gather against that stream only: Work.wait() of the
        # gather that last read this buffer blocks this stream, the new gather waits for this stream's launch -- the other stream is not involved
        j = k % S
        ev = timing["events"][k] if timing["events"] is not None else None
        if args.force_process_group:       # test seam: a gather that ran ahead of this step's launch would deliver zeros, not the previous step's (equal) records
            r.zero_()
        if ev is not None:
            ev[0].record(side[j])
        rc = lib.tff_linear_tft_pose_batch_dev(ctxs[j].handle, p(d_C), p(d_calm), 0, B, N, p(r, 0), p(r, 12 * B), p(r, 24 * B),
                                               None, None, ctypes.c_void_p(statuses[j].data_ptr()))
        if ev is not None:
            ev[1].record(side[j])
        if rc != 0:
            raise RuntimeError("tff_linear_tft_pose_batch_dev failed: %s" % lib.tff_last_error().decode())

    pipe = tdist.OverlappedGather(world, tdist.RECORD_DOUBLES * B, dev, compute_stub if stub else compute, nbuf=max(2, S), collective=use_pg)
    drain = pipe.drain
    if stub or S == 1:
        step = pipe.step
    else:
        def step(k):
            with torch.cuda.stream(side[k % S]):
                return pipe.step(k)

    def sync_all():
        for c in ctxs:
            c.synchronize()
        dsync()

    for k in range(args.warmup):
        step(k)
    drain()
    sync_all()

    def timed_region():
        """EXACTLY --steps steps between barrier + synchronize on both sides; returns the seconds (max over ranks)."""
        if use_pg:
            dist.barrier()
        dsync()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        drain()
        sync_all()
        if use_pg:
            dist.barrier()
        dsync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # the region is a few milliseconds of wall clock: it is repeated and the MEDIAN repetition is the one reported (every repetition listed)
    reps = max(1, args.reps)
    rep_s = [timed_region() for _ in range(reps)]
    elapsed = float(np.median(rep_s))
    # per-launch HIP events of one more, UNTIMED repetition of the same loop: how long a launch lasts while its neighbour shares the GPU
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if not stub else [None] * args.steps
    timing["events"] = events
    for k in range(args.steps):
        step(k)
    drain()
    timing["events"] = None
    sync_all()
    if world > 1:
        dist.barrier()
    dsync()
    n_bad = sum(int((st_ != 0).sum().item()) for st_ in statuses)
    pg_last, pg_check = None, None
    if use_pg and not stub and rank == 0:                   # rank 0's own records as the last step's all-gather delivered them
        pg_last = pipe.gathered[(args.steps - 1) % pipe.nbuf][rank].clone()
    for c in ctxs:                                          # the secondary blocks below run on the main stream, one call at a time
        c.set_stream(stream.cuda_stream)
    if stub:
        inflight_ms = np.array(stub_ms) if stub_ms else np.array([float("nan")])
    else:
        inflight_ms = np.array([a.elapsed_time(b) for a, b in events]) if events else np.array([float("nan")])
    gather_check = None
    if stub and world > 1:                                  # every rank's records of the last step arrived complete and in rank order
        kl = args.steps - 1
        g_ = pipe.gathered[kl % pipe.nbuf]
        n_ = g_.shape[1]
        gather_check = all(torch.equal(g_[src], torch.full((n_,), float(1000 * src + kl), dtype=torch.float64) + torch.arange(n_, dtype=torch.float64) * 1e-9)
                           for src in range(world))

    # secondary metrics of SURVEY 8(d) on the same resident batch (rank 0, outside the contract's timed region)
    secondary, n_sweep, config4 = {}, {}, {}

    def time_calls(call, reps):
        for _ in range(2):
            call()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            call()
        e1.record(stream)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps

    single, iso_ms = None, None
    if rank == 0 and not stub:
        # the same call with the GPU to itself: one stream, nothing else queued, HIP events on that stream around every call (both launches of a
        # call inside the pair)
        r0 = pipe.recs[0]
        call = lambda: lib.tff_linear_tft_pose_batch_dev(ctx.handle, p(d_C), p(d_calm), 0, B, N, p(r0, 0), p(r0, 12 * B), p(r0, 24 * B), None, None,
                                                         ctypes.c_void_p(status.data_ptr()))
        n_iso = max(20, args.steps)
        for _ in range(3):
            call()
        torch.cuda.synchronize(dev)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_iso)]
        for e0, e1 in evs:
            e0.record(stream)
            call()
            e1.record(stream)
            torch.cuda.synchronize(dev)                                      # the next call starts on an idle device
        iso_ms = np.array([e0.elapsed_time(e1) for e0, e1 in evs])
        # ... and back to back on that one stream (no host synchronisation between the calls): the one-batch-at-a-time throughput
        ms1 = time_calls(call, max(20, args.steps))
        single = {"ms_per_batch": ms1, "value": B / (ms1 * 1e-3), "unit": "triplet-hypotheses/s",
                  "achieved_GBs": algorithmic_bytes_per_triplet(N) * B / (ms1 * 1e-3) / 1e9}
        if pg_last is not None:          # --force-process-group: what the last step's overlapped gather delivered == this one-stream result, bit for bit
            torch.cuda.synchronize(dev)
            pg_check = {"forced": args.force_process_group, "backend": dist.get_backend(), "world": world, "streams": S,
                        "gathered_equals_single_stream": bool(torch.equal(pg_last, r0)), "records": int(r0.numel())}
    if rank == 0 and world == 1 and not args.no_secondary and not stub:
        it32 = torch.zeros(B, dtype=torch.int32, device=dev)
        r = pipe.recs[0]
        for name in ("LinearFPoseEstimation", "OptimFPoseEstimation", "ResslTFTPoseEstimation", "NordbergTFTPoseEstimation",
                     "FaugPapaTFTPoseEstimation", "PiPoseEstimation", "PiColPoseEstimation"):
            fn = getattr(lib, api.POSE_METHODS[name] + "_dev")
            call = lambda: fn(ctx.handle, p(d_C), p(d_calm), 0, B, N, p(r, 0), p(r, 12 * B), p(r, 24 * B), None,
                              ctypes.c_void_p(it32.data_ptr()), ctypes.c_void_p(status.data_ptr()))
            ms = time_calls(call, 10 if "Linear" in name else 4)
            secondary[name] = {"value": B / (ms * 1e-3), "unit": "triplet-hypotheses/s", "ms_per_batch": ms,
                               "failed_triplets": int((status != 0).sum().item()), "mean_iterations": float(it32.double().mean().item())}
        # north_star's range of correspondences per triplet: LinearTFT at N = 100, 500, 1000 (same batch size), HBM roofline fraction per N
        for Nn in (100, 500, 1000):
            Cn, _, _, _ = generate_scene_batch(B, Nn, noise=1.0, seed=2000 + Nn)
            d_Cn = torch.from_numpy(Cn).to(dev)
            call = lambda: lib.tff_linear_tft_pose_batch_dev(ctx.handle, p(d_Cn), p(d_calm), 0, B, Nn, p(r, 0), p(r, 12 * B), p(r, 24 * B), None, None,
                                                             ctypes.c_void_p(status.data_ptr()))
            ms = time_calls(call, 10)
            gbs = algorithmic_bytes_per_triplet(Nn) * B / (ms * 1e-3) / 1e9
            n_sweep["N=%d" % Nn] = {"value": B / (ms * 1e-3), "unit": "triplet-hypotheses/s", "ms_per_batch": ms, "failed_triplets": int((status != 0).sum().item()),
                                    "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}}
            del d_Cn
        # configs[3] on one GPU: 1 M minimal-sample hypotheses of ONE scene (25 % gross outliers) + int32 inlier counts (1-px rule)
        try:
            H, Ns = 1000000, 400
            Cs, _, _, _ = generate_scene_batch(1, Ns, noise=0.5, seed=77)
            scene = Cs[0].copy()
            rng = np.random.default_rng(5)
            bad = rng.choice(Ns, Ns // 4, replace=False)
            scene[bad, 2:6] += rng.uniform(20, 80, size=(bad.size, 4))
            d_scene = torch.from_numpy(scene).to(dev)
            for name, nmin in (("LinearTFTPoseEstimation", 7), ("LinearFPoseEstimation", 8)):
                gen = torch.Generator(device=dev); gen.manual_seed(1234)
                idx = torch.rand((H, Ns), device=dev, generator=gen).argsort(dim=1)[:, :nmin].to(torch.int32).contiguous()
                d_cm = torch.from_numpy(CalM).to(dev)
                for timed in (False, True):                                     # first pass: warm-up (workspace growth)
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    hyp = ctx.pose_sampled(name, d_scene, d_cm, idx)
                    cnt = ctx.inlier_count(d_scene, d_cm, hyp["R_t_2"], hyp["R_t_3"], 1.0)
                    torch.cuda.synchronize(dev)
                    dt = time.perf_counter() - t0
                del idx
                config4[name] = {"hypotheses": H, "sample_size": nmin, "scene_correspondences": Ns, "seconds": dt, "value": H / dt,
                                 "unit": "hypotheses/s (pose + inlier count, exact kernel for minimal samples)",
                                 "best_inlier_count": int(cnt.max().item()), "scene_inliers": int(Ns - bad.size),
                                 "failed": int((hyp["status"] != 0).sum().item())}
        except Exception as ex:                       # the block is informative only; never let it take the headline line down
            config4 = {"error": repr(ex)}

    if rank == 0:
        total = world * B * args.steps
        value = total / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        alg = algorithmic_bytes_per_triplet(N) * B
        kern_ms = float(iso_ms.mean()) if iso_ms is not None else float(inflight_ms.mean())
        achieved = alg / (kern_ms * 1e-3) / 1e9
        traffic, valu = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic = pj.get("hbm_bytes_per_launch")
                # the bound that actually binds: fp64 vector issue.  Counters from the committed rocprofv3 PMC passes;
                # the issue-limited rate = SIMDs x clock / (cycles per wave64 fp64 VALU instruction, profiles/r5_dpp_fmac.txt) / instructions per triplet.
                ipt = pj.get("valu_instructions_per_triplet")
                if ipt:
                    simds, clock_hz = 256 * 4, 2.4e9
                    valu = {"instructions_per_triplet": ipt, "busy_fraction": pj.get("valu_busy_fraction"),
                            "issue_limited_triplets_per_s": simds * clock_hz / 4.0 / ipt, "source": "profiles/pmc_latest.json (rocprofv3 --pmc, same command)"}
                    valu["frac_of_issue_limit"] = (value / world) / valu["issue_limited_triplets_per_s"]       # of the per-GPU rate actually delivered
                    if single:
                        valu["frac_of_issue_limit_one_batch_at_a_time"] = single["value"] / valu["issue_limited_triplets_per_s"]
                    # ... and against the MEASURED issue rate (tools/micro/fp64_issue.hip, profiles/r5_fp64_issue.txt: independent v_fma_f64 /
                    # v_fmac_f64_dpp streams at two wavefronts per SIMD; the shader clock settles near 2.0 - 2.35 GHz under fp64 load, so the
                    # nominal 2.4 GHz / 4 cycles above is 10 - 25 % optimistic)
                    mr = pj.get("measured_fp64_issue_inst_per_ns_per_simd_at_2_waves") or {}
                    lo, hi = mr.get("v_fma_f64"), mr.get("v_fmac_f64_dpp row_newbcast")
                    if lo and hi:
                        lim = [simds * r * 1e9 / ipt for r in (lo, hi)]
                        valu["measured_issue_rate_inst_per_ns_per_simd"] = [lo, hi]
                        valu["measured_issue_limited_triplets_per_s"] = lim
                        valu["frac_of_measured_issue_limit"] = [(value / world) / lim[1], (value / world) / lim[0]]
            except Exception:
                traffic, valu = None, None
        out = {
            "metric": "triplet-hypotheses/sec (linearTFT+R,t) at N=200 corresp.",
            "value": value, "unit": "triplet-hypotheses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            # batches that may be resident on a GPU at once (consecutive steps alternate between that many streams), and how much of a launch
            # is hidden behind its neighbour: kernel_ms (GPU to itself) / ms_per_step; kernel_ms * steps / in_flight <= ms_per_step * steps
            "in_flight": S, "overlap_factor": kern_ms / ms_per_step,
            "repetitions": {"n": reps, "reported": "median", "ms_per_step_each": [1e3 * t_ / args.steps for t_ in rep_s],
                            "timed_region_ms_each": [1e3 * t_ for t_ in rep_s]},
            "config": {"workload": "configs[1]: batch of %d synthetic triplets x %d correspondences, sigma=1px, "
                                   "linearTFT + R_t_from_TFT (LinearTFTPoseEstimation without Reconst), one batch per GPU" % (B, N),
                       "batch_per_gpu": B, "correspondences": N,
                       "streams": "%d (consecutive batches alternate between streams and may overlap)" % S if S > 1 else "1",
                       "gather": ("%s all_gather of 408-B result records, overlapped" % ("gloo (stub)" if stub else "RCCL")) if world > 1 else "none (1 GPU)",
                       "failed_triplets": n_bad},
            # ONE launch, ONE duration: the call with the GPU to itself (see the docstring); `in_flight_launch_ms` is the same call's HIP-event
            # duration while its neighbour on the other stream shares the GPU (longer: not the kernel's own time, reported for the record only)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_linear_tft_pose_rows", "kernel_ms": kern_ms,
                         "kernel_ms_how": "HIP events on the launch stream around every call (k_linear_tft_pose_rows + the exact kernel's status scan), "
                                          "one stream, device idle before every call, mean of %d calls" % (len(iso_ms) if iso_ms is not None else 0),
                         "kernel_ms_min": float(iso_ms.min()) if iso_ms is not None else None,
                         "kernel_ms_median": float(np.median(iso_ms)) if iso_ms is not None else None,
                         "in_flight_launch_ms": float(inflight_ms.mean()),
                         "algorithmic_bytes_per_launch": alg,
                         # the rate the device delivers with `in_flight` batches resident: algorithmic bytes of a step over ms_per_step
                         "delivered": {"achieved": alg / (ms_per_step * 1e-3) / 1e9, "frac": alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s per GPU"}},
        }
        if single:
            out["single_stream"] = single
        if valu:
            out["fp64_valu"] = valu
        if secondary:
            out["secondary"] = secondary
        if n_sweep:
            out["n_sweep"] = n_sweep
        if config4:
            out["config4"] = config4
        if pg_check:
            out["process_group_check"] = pg_check
        if stub:
            out["stub"] = {"compute": "tagged fill on CPU tensors (test seam, not a measurement)", "gather_check": gather_check}
        if world == 1 and not args.no_cpu_baseline and not stub:
            sample = args.cpu_sample or max(1024, 40 * (os.cpu_count() or 1))   # ~10 s of wall time on the box's host cores
            out["cpu_baseline"] = cpu_baseline(C, CalM, min(sample, B))
            out["cpu_baseline_lapack"] = cpu_baseline_lapack(C, CalM)
            ref_t = cpu_baseline_reference(N)
            if ref_t:
                out["cpu_baseline_reference"] = ref_t
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
