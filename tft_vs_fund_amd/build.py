"""Build libtftfund.so (hipcc, gfx950 only) in-tree.  `python -m tft_vs_fund_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtftfund.so")
SOURCES = ["capi.hip"]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, "..", "include", "tftfund.h"))
    return any(os.path.getmtime(p) > t for p in deps if os.path.isfile(p))


def build_library(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + CSRC, "-shared", "-fPIC", "-o", OUT] + SOURCES
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return OUT


if __name__ == "__main__":
    build_library(force=True, verbose="-v" in sys.argv)
    print(OUT)
