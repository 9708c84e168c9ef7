"""One compact JSON per profiled method from a tools/gpu_profile_r3.sh pass directory: kernel-trace average of the dominant kernel and
the derived counter figures (VALU instructions per unit, busy / waiting fractions, lane utilisation, HBM bytes with the gfx950
FETCH_SIZE correction) -- the numbers DESIGN.md 4 quotes.   python tools/summarize_r3.py <dir> <kernel substring> <units per launch> <algorithmic bytes per launch> <waves per SIMD>"""
import csv, glob, json, os, sys
root, kern, units, alg_bytes, occ = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
out = {"kernel": kern, "units_per_launch": units, "algorithmic_bytes_per_launch": alg_bytes}
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "tff::" in r["Name"]]
    out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage")} for r in rows]
    for r in rows:
        if kern in r["Name"]:
            out["average_ns"] = float(r["AverageNs"])
c = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    acc = {}
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kern in row.get("Kernel_Name", ""):
                a = acc.setdefault(row["Counter_Name"], [0.0, 0]); a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, v in acc.items():
        c[k] = v[0] / v[1]
out["counters_per_launch"] = c
g = c.get
if g("SQ_INSTS_VALU"):
    out["valu_instructions_per_unit"] = g("SQ_INSTS_VALU") / units
    out["salu_instructions_per_unit"] = g("SQ_INSTS_SALU", 0) / units
    out["lds_instructions_per_unit"] = g("SQ_INSTS_LDS", 0) / units
if g("SQ_ACTIVE_INST_VALU") and g("SQ_WAVE_CYCLES"):
    out["valu_busy_fraction"] = g("SQ_ACTIVE_INST_VALU") / (g("SQ_WAVE_CYCLES") / occ)     # per SIMD slot at `occ` waves per SIMD
if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
    out["waiting_fraction"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    out["lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
if g("SQ_LDS_BANK_CONFLICT") and g("SQ_ACTIVE_INST_LDS"):
    out["lds_bank_conflict_fraction"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS")
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    rd, wr = g("FETCH_SIZE") * 1024 * 2, g("WRITE_SIZE") * 1024                              # KiB; x2: gfx950 wide-load correction (MI355X_MICROARCH.md)
    out["hbm_read_bytes"] = rd; out["hbm_write_bytes"] = wr; out["hbm_bytes_over_algorithmic"] = (rd + wr) / alg_bytes
if out.get("average_ns") and out.get("valu_instructions_per_unit"):
    rate = units / (out["average_ns"] * 1e-9)
    out["units_per_second_in_kernel"] = rate
    out["fp64_issue_limit_units_per_second"] = 1024 * 2.4e9 / 4.0 / out["valu_instructions_per_unit"]
    out["fraction_of_issue_limit"] = rate / out["fp64_issue_limit_units_per_second"]
print(json.dumps(out, indent=1))
