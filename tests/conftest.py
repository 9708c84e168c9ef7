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


# TFF_OPT_ROWS of the parity suite's contexts: 1 = four triplets per wavefront (what the large batches of bench.py and BASELINE's configs run on),
# 0 = one triplet per wavefront -- exactly what the library's DEFAULT (2: by batch size) picks for a batch under 1024 triplets, i.e. for the MEX
# drop-in's call pattern (one triplet, or a few hundred).  Every oracle / golden / 50-digit comparison runs on both.
ROUTES = [pytest.param(1, id="rows"), pytest.param(0, id="wave")]


@pytest.fixture(scope="session", params=ROUTES)
def gpu_ctx(request):
    """A tff_ctx on cuda:0, once per kernel route.  No CPU fallback: fails if the library or the device is missing."""
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.build import build_library
    build_library()
    ctx = api.Context(0)
    ctx.set_rows(request.param)
    ctx.route = request.param
    return ctx
