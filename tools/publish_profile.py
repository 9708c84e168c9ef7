"""Copy the judged evidence of a tools/gpu_profile.sh run from gpurun_out/ into profiles/ (tracked).

  python tools/publish_profile.py gpurun_out/prof_r1b r1_final

writes profiles/<tag>_summary.json, profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py)
and refreshes profiles/pmc_latest.json, which bench.py reads for roofline.traffic.
"""
import glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summ = json.load(open(os.path.join(src, "summary.txt")))
K = "void tff::k_linear_tft_pose<false>(tff::LinearTftArgs) | "
B, N = 10000, 200
get = lambda grp, name: summ[grp][K + name]["mean_per_dispatch"]
fetch, write = get("pmc_fetch", "FETCH_SIZE"), get("pmc_write", "WRITE_SIZE")
rd, wr = int(round(fetch * 1024 * 2)), int(round(write * 1024))
stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
shutil.copy(stats, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
json.dump(summ, open(os.path.join(ROOT, "profiles", tag + "_summary.json"), "w"), indent=1)
kern = [k for k in summ["kernel_stats"] if "<false>" in k["Name"]][0]
pmc = {
    "_comment": "HBM traffic of k_linear_tft_pose<false> per launch (B=%d, N=%d), rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                "separate passes (profiles/%s_summary.json). FETCH_SIZE is in KiB and, for 16-B-per-lane streaming loads on gfx950, "
                "reports half the bytes (MI355X_MICROARCH.md, HBM section): read = FETCH_SIZE KiB * 1024 * 2; write = WRITE_SIZE KiB * 1024." % (B, N, tag),
    "kernel": "k_linear_tft_pose<false>",
    "fetch_size_kib": fetch, "write_size_kib": write,
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "algorithmic_bytes_per_launch": (48 * N + 216 + 216 + 192) * B,
    "kernel_trace_average_ns": float(kern["AverageNs"]),
    "valu_instructions_per_triplet": get("pmc_sq", "SQ_INSTS_VALU") / B,
    "salu_instructions_per_triplet": get("pmc_sq", "SQ_INSTS_SALU") / B,
    # SQ_ACTIVE_INST_VALU counts cycles a wave has a VALU instruction in flight; two waves share a SIMD (occupancy 2)
    "valu_busy_fraction": get("pmc_sq", "SQ_ACTIVE_INST_VALU") / (get("pmc_sq", "SQ_WAVE_CYCLES") / 2.0),
}
json.dump(pmc, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
print(json.dumps(pmc, indent=1))
