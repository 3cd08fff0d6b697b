import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "gpu_long: needs a real MI355X, tens of seconds each; not part of -m gpu (run by hand: -m gpu_long)")


@pytest.fixture(scope="session")
def hip():
    """The product package (ctypes over libcosnarks_hip.so). Missing library = hard error, never a skip."""
    import cosnarks_amd
    cosnarks_amd.lib()
    return cosnarks_amd


@pytest.fixture(scope="session")
def gpu(hip):
    """For -m gpu tests: a device must be present on the GPU box; in the CPU container the test is skipped."""
    if not hip.have_device():
        if os.environ.get("GRAFT_REPO_ROOT"):
            raise RuntimeError("GPU box without a visible HIP device")
        pytest.skip("no HIP device in this container")
    return hip
