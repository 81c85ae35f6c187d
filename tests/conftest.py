import os
import sys

import pytest

# The ORACLE side of the GPU parity tests runs torch's own convolutions (MIOpen) at ~60 distinct SD-2.1 shapes x 3 dtypes on a box that
# has never seen them: MIOpen's default find mode benchmarks every solver per new shape, which was 5 of the suite's 18 minutes.  FAST =
# find-db hit or the immediate-mode heuristic pick.  It changes which (exact fp32 / rounded 16-bit) library algorithm the checker runs,
# nothing in the product, which never calls MIOpen.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

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
