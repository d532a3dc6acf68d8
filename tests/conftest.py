import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """A HipLd context on cuda:0 -- fails loudly (no CPU fallback) when there is no device."""
    import tomahawk_amd as T
    if T.device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests cannot fall back to the CPU")
    eng = T.HipLd(0)
    yield eng
    eng.close()
