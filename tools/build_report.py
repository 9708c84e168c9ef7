"""Build libtftfund.so with -Rpass-analysis=kernel-resource-usage and print one line per kernel (VGPRs, scratch, occupancy, spills)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "tft_vs_fund_amd", "csrc")
out = os.path.join(ROOT, "tft_vs_fund_amd", "libtftfund.so")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + csrc, "-shared", "-fPIC", "-o", out, "capi.hip",
                    "-Rpass-analysis=kernel-resource-usage"], cwd=csrc, capture_output=True, text=True)
txt = r.stderr
if r.returncode:
    print("\n".join(l for l in txt.splitlines() if "error" in l or "note" in l)[:4000]); sys.exit(1)
cur = None; rows = {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}
    for key in ("VGPRs:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:", "VGPRs Spill:"):
        m = re.search(re.escape(key) + r" (\d+)", line)
        if m and cur: rows[cur][key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for (k, v), name in zip(rows.items(), names):
    print("%-78s vgpr %3d scratch %4d occ %d vspill %3d" % (name[:78], v.get("VGPRs:", -1), v.get("ScratchSize [bytes/lane]:", -1),
                                                            v.get("Occupancy [waves/SIMD]:", -1), v.get("VGPRs Spill:", -1)))
for l in txt.splitlines():
    if "warning" in l and "unroll" not in l: print(l)
