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


def pytest_sessionstart(session):
    """Build the HIP library (hipcc cross-compiles without a GPU, ~35 s) if a fresh checkout has not done so yet: the
    CPU-side tests check its exported symbols and use its host-side packers."""
    so = os.path.join(ROOT, "q-palette_amd", "libqpal_hip.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "q-palette_amd", "csrc"), "-j", "8"],
                              stdout=subprocess.DEVNULL)
