import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def sdlib():
    from sonicdiffusionbayeslab_amd import _lib
    return _lib.load()


def pytest_report_header(config):
    """The SD_* switches select kernels, split factors and rounding points: a parity run's environment is part of
    its result, so it is printed with it (none set = the product defaults)."""
    env = {k: v for k, v in sorted(os.environ.items()) if k.startswith("SD_")}
    return f"SD_* environment: {env if env else 'none set (product defaults)'}"
