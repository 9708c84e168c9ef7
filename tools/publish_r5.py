"""Copy the round-4 rocprofv3 evidence (tools/gpu_profile_r5.sh -> gpurun_out/r5/) into profiles/ and refresh profiles/pmc_latest.json, which
bench.py reads for roofline.traffic and the fp64_valu block.   python tools/publish_r5.py"""
import glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "r5")
for f in sorted(glob.glob(os.path.join(src, "r5_*"))):
    shutil.copy(f, os.path.join(ROOT, "profiles", os.path.basename(f)))
s = json.load(open(os.path.join(src, "r5_headline_summary.json")))
s1 = json.load(open(os.path.join(src, "r5_headline1_summary.json")))
c = s["counters_per_launch"]
B, N = 10000, 200
pmc = {
    "_comment": "k_linear_tft_pose_rows per launch (B=%d, N=%d), rocprofv3 --kernel-trace --stats and separate --pmc passes of `python3 bench.py --no-cpu-baseline "
                "--no-secondary --steps 20 --warmup 3` (profiles/r5_headline_summary.json, tools/gpu_profile_r5.sh; r5_headline1_*: the same with --streams 1). "
                "FETCH_SIZE is in KiB and, for 16-B-per-lane streaming loads on gfx950, reports half the bytes (MI355X_MICROARCH.md, HBM section): "
                "read = FETCH_SIZE KiB * 1024 * 2; write = WRITE_SIZE KiB * 1024.  The kernel makes three passes over the correspondences (they are not "
                "staged in LDS): the bytes counted at the L2 boundary are ~2.7x the algorithmic ones, most of them MALL hits (the 96 MB batch fits the "
                "256 MB infinity cache)." % (B, N),
    "kernel": "k_linear_tft_pose_rows",
    "fetch_size_kib": c["FETCH_SIZE"], "write_size_kib": c["WRITE_SIZE"],
    "hbm_read_bytes_per_launch": int(round(s["hbm_read_bytes"])), "hbm_write_bytes_per_launch": int(round(s["hbm_write_bytes"])),
    "hbm_bytes_per_launch": int(round(s["hbm_read_bytes"] + s["hbm_write_bytes"])),
    "algorithmic_bytes_per_launch": (48 * N + 216 + 216 + 192) * B,
    "kernel_trace_average_ns": s["average_ns"], "kernel_trace_average_ns_one_stream": s1["average_ns"],
    "valu_instructions_per_triplet": s["valu_instructions_per_unit"], "salu_instructions_per_triplet": s["salu_instructions_per_unit"],
    "valu_busy_fraction": s["valu_busy_fraction"], "lane_utilisation": s["lane_utilisation"], "waiting_fraction": s["waiting_fraction"],
}
# the measured fp64 issue rate (tools/micro/fp64_issue.hip -> profiles/r5_fp64_issue.txt): instructions per ns and SIMD at two wavefronts per SIMD
rates = {}
fi = os.path.join(ROOT, "profiles", "r5_fp64_issue.txt")
if os.path.exists(fi):
    import re
    for line in open(fi):
        m = re.match(r"(\S+(?: \S+)*?)\s+2 wave\(s\)/SIMD:\s+[\d.]+ ms, ([\d.]+) inst/ns/SIMD", line)
        if m:
            rates[m.group(1)] = float(m.group(2))
pmc["measured_fp64_issue_inst_per_ns_per_simd_at_2_waves"] = rates
json.dump(pmc, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
print(json.dumps(pmc, indent=1))
