import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load the HIP library; GPU tests must run the native path."""
    from diffews_amd import build, _lib
    if not os.path.isfile(_lib.LIB_PATH):
        build.build()
    return _lib.lib()
