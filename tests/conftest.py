import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def gpu_ctx():
    """A tff_ctx on cuda:0.  No CPU fallback: fails if the library or the device is missing.
    TFF_OPT_ROWS is forced to 1: the parity tests use small batches, which the default (2: by batch size) would send to the one-triplet kernels;
    the four-triplets-per-wavefront kernels are the ones the large batches of bench.py and BASELINE's configs run on, so they are the ones the oracle
    comparisons go through (tests/test_gpu_rows.py compares the two routes with each other and checks what the default picks)."""
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.build import build_library
    build_library()
    ctx = api.Context(0)
    ctx.set_rows(1)
    return ctx
