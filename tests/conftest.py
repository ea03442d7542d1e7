import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    return torch.device('cuda:0')


def pytest_sessionstart(session):
    """FEMASR_TEST_SO=<path>: run the suite against an experiment build of the library (tools/build_debug.sh)."""
    so = os.environ.get('FEMASR_TEST_SO')
    if so:
        from femasr_amd import _lib
        _lib.SO_PATH = so
