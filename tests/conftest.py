import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'coupled-vae_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _no_persistent_kernel_timeout(request):
    """After every GPU test: no bounded spin of the whole-sequence recurrent kernels ran out."""
    yield
    if request.node.get_closest_marker('gpu') is None:
        return
    from cvae_amd import _lib
    assert _lib.query('cvae_persist_status') == 0, 'a persistent recurrent kernel timed out'
