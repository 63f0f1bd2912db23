import importlib
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package.  Its directory name has a hyphen, so it is imported by name."""
    return importlib.import_module("mca-paper_amd")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _reset_measurement_knobs():
    """The A/B knobs of the library (include/mca_hip_debug.h) are process-global: whatever a test set, and however it
    ended, they are all back to 0 before the next test."""
    yield
    hip = sys.modules.get("mca-paper_amd.hip")
    if hip is not None and getattr(hip, "_lib", None) is not None:
        hip._lib.mca_debug_reset()
