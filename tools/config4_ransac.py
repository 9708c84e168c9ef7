#!/usr/bin/env python3
"""
BASELINE.json config 4: 1M RANSAC-style minimal-sample hypotheses of ONE scene, sharded across the GPUs of a
node (contiguous hypothesis blocks per rank, no data-path collective), int32 inlier counts gathered over RCCL.

  python tools/config4_ransac.py [--hyp 1000000] [--method tft|f] [--scene 400]
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/config4_ransac.py

The reference has no RANSAC and no 7-point F solver: a hypothesis is linearTFT on 7 / linearF on
8 correspondences (the minimum the reference accepts, experiments.m:99 / linearF.m:35) and the inlier rule is
the 1-px per-coordinate residual test of experiments_real.m:94-98.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hyp", type=int, default=1000000)
    ap.add_argument("--method", default="tft")
    ap.add_argument("--scene", type=int, default=400)
    ap.add_argument("--outliers", type=float, default=0.25)
    ap.add_argument("--variant", type=int, default=0, help="TFF_OPT_KERNEL (1: paired kernel, A/B)")
    ap.add_argument("--stub", action="store_true", help="TEST SEAM (tests/test_bench_flow_gloo.py): CPU + gloo, the counts replaced by a function of "
                                                        "the global hypothesis index -- the sharding and the count gather run as on the GPUs")
    args = ap.parse_args()
    from tft_vs_fund_amd import api, dist as tdist
    from tft_vs_fund_amd.scenes import generate_scene_batch
    if args.stub:
        rank, world, _ = tdist.init_from_env("cpu")
        lo, hi = tdist.shard_bounds(args.hyp, world, rank)
        cnt = ((torch.arange(lo, hi, dtype=torch.int64) * 7919) % 401).to(torch.int32)      # this rank's block of "inlier counts"
        allc = tdist.all_gather_counts(cnt, args.hyp)
        exp = ((torch.arange(args.hyp, dtype=torch.int64) * 7919) % 401).to(torch.int32)
        if rank == 0:
            print(json.dumps({"config": "config4", "stub": True, "hypotheses": args.hyp, "n_gpus": world, "shard": [lo, hi],
                              "gathered": int(allc.numel()), "order_ok": bool(torch.equal(allc, exp)), "best_inliers": int(allc.max())}))
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    rank, world, local = tdist.init_from_env("cuda")
    torch.cuda.set_device(local)
    ctx = api.Context(local)
    ctx.set_kernel_variant(args.variant)
    C, CalM, Rt0, _ = generate_scene_batch(1, args.scene, noise=0.5, seed=7)
    scene = C[0].copy()
    rng = np.random.default_rng(1)
    bad = rng.choice(args.scene, int(args.outliers * args.scene), replace=False)
    scene[bad, 2:6] += rng.uniform(20, 80, size=(bad.size, 4))
    n = 7 if args.method == "tft" else 8
    method = "LinearTFTPoseEstimation" if args.method == "tft" else "LinearFPoseEstimation"
    lo, hi = tdist.shard_bounds(args.hyp, world, rank)
    g = torch.Generator(device="cuda"); g.manual_seed(1234 + rank)
    idx = torch.rand((hi - lo, args.scene), device="cuda", generator=g).argsort(dim=1)[:, :n].to(torch.int32).contiguous()
    d_scene = torch.from_numpy(scene).cuda(); d_calm = torch.from_numpy(CalM).cuda()
    for timed in (False, True):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hyp = ctx.pose_sampled(method, d_scene, d_calm, idx)
        cnt = ctx.inlier_count(d_scene, d_calm, hyp["R_t_2"], hyp["R_t_3"], 1.0)
        allc = tdist.all_gather_counts(cnt, args.hyp)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        dt = time.perf_counter() - t0
    if rank == 0:
        best = int(allc.argmax())
        print(json.dumps({"config": "config4", "method": method, "hypotheses": args.hyp, "n_gpus": world, "scene": args.scene,
                          "seconds": dt, "hypotheses_per_s": args.hyp / dt, "best_inliers": int(allc[best]),
                          "true_inliers": int(args.scene - bad.size), "failed": int((hyp["status"] != 0).sum())}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
