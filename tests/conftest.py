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
    """A tff_ctx on cuda:0.  No CPU fallback: fails if the library or the device is missing."""
    from tft_vs_fund_amd import api
    from tft_vs_fund_amd.build import build_library
    build_library()
    return api.Context(0)
