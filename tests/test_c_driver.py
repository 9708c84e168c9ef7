"""tests/c_driver.c: a plain-C program on the C ABI (what a C/C++ host or the MEX shim binds).  CPU: it compiles against
include/tftfund.h and links against libtftfund.so.  GPU: it runs and recovers the ground-truth poses."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from tft_vs_fund_amd.build import build_library
    build_library()
    exe = str(tmp_path / "c_driver")
    cmd = ["gcc", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "c_driver.c"), "-I" + os.path.join(ROOT, "include"),
           "-L" + os.path.join(ROOT, "tft_vs_fund_amd"), "-ltftfund", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_driver_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_driver_runs(tmp_path):
    exe = _build(tmp_path)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "tft_vs_fund_amd"), "/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "c_driver ok" in r.stdout, r.stdout + r.stderr
