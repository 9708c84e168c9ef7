"""Builds tests/emu/_build/libtff_emu.so: the HIP kernels compiled by g++ against
the lane emulator (test infrastructure only; optional ASan/UBSan build)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT_DIR = os.path.join(HERE, "_build")


def build(sanitize=False):
    """sanitize=True: AddressSanitizer + UBSan build of the linear trifocal kernels (k_linear_tft_pose<false|true>, k_linear_tft_pose_rows) only.
    It must be loaded into a process that has the sanitizer runtimes preloaded (sanitizer_env())."""
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, "libtff_emu_san.so" if sanitize else "libtff_emu.so")
    src = os.path.join(HERE, "emu_lib.cpp")
    deps = [src, os.path.join(HERE, "hip_emu.h")]
    csrc = os.path.join(ROOT, "tft_vs_fund_amd", "csrc")
    deps += [os.path.join(csrc, f) for f in os.listdir(csrc)]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = ["g++", "-std=c++20", "-O1", "-g", "-pthread", "-shared", "-fPIC", "-I" + HERE, "-I" + csrc, "-o", out, src]   # tests/emu first: <wave_target.h>
    if sanitize:
        cmd[3:3] = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-DTFF_EMU_LINEAR_TFT_ONLY"]
    subprocess.run(cmd, check=True)
    return out


def sanitizer_env():
    """Environment for a child process that loads the sanitize=True library: the ASan / UBSan runtimes must come first in its library list."""
    env = dict(os.environ)
    libs = [subprocess.run(["gcc", "-print-file-name=" + n], check=True, capture_output=True, text=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    env["LD_PRELOAD"] = " ".join(libs)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1"       # (CPython itself leaks by design; an error must fail the child loudly)
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    return env


def load():
    lib = ctypes.CDLL(build())
    return lib
