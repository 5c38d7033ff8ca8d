import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'top-k-rec_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
os.environ.setdefault('TKR_NO_CACHE', '1')      # no .npy copies beside the committed fixtures; cache tests re-enable it


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
