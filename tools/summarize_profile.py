"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text/JSON summary for profiles/."""
import csv, glob, json, os, sys
root = sys.argv[1]
KERN = os.environ.get("PROFILE_KERNEL", "k_linear_tft_pose")
out = {}
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if KERN in row["Name"]:
            out.setdefault("kernel_stats", []).append({k: row[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
def pmc(dirname):
    acc = {}
    for f in glob.glob(os.path.join(root, dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if KERN not in row.get("Kernel_Name", ""):
                continue
            key = (row["Kernel_Name"][:60], row["Counter_Name"])
            a = acc.setdefault(key, [0.0, 0])
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return {"%s | %s" % k: {"mean_per_dispatch": v[0] / v[1], "dispatches": v[1]} for k, v in acc.items()}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    out[d] = pmc(d)
print(json.dumps(out, indent=1))
