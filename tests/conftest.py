import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    import dl3_amd  # noqa: F401
    from dl3_amd import capi
    return capi.lib()


@pytest.fixture(autouse=True)
def _release_device_tensors(request):
    yield
    if request.node.get_closest_marker("gpu"):
        from tests import gpu_util
        gpu_util.release()
