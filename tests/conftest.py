import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    """CPU oracle (test infrastructure)."""
    return entry.load_oracle()


@pytest.fixture(scope="session")
def vsl():
    return entry.load_package()


@pytest.fixture(scope="session")
def synth(vsl):
    import importlib
    return importlib.import_module("visual_slam_amd.synth")


@pytest.fixture(scope="session")
def ctx(vsl):
    """A context on GPU 0.  GPU tests FAIL (not skip) when the HIP library or device is unusable."""
    c = vsl.Context(0)
    yield c
    c.close()


GOLDEN = ROOT / "tests" / "golden"
