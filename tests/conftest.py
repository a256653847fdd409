import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _reproducible_streams(request):
    """Every test starts from the same process-wide seeds (the product seeds itself from os.urandom when none is given): the
    statistical checks see the same rays on every run."""
    import zlib
    import numpy
    key = zlib.crc32(request.node.nodeid.encode())
    numpy.random.seed(key % (2 ** 31))
    try:
        from tracer_amd import rng
        rng.seed(0x5EED0000 + key)
    except Exception:           # the package itself is under test for importability elsewhere
        pass
    yield
