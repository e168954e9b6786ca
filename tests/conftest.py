import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the first GPU call: see mellow_amd/__init__.py

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def synth_sd():
    """The seeded synthetic checkpoint (real state_dict layout) shared by oracle and engine tests."""
    from mellow_amd import synth
    return synth.make_state_dict(0)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
