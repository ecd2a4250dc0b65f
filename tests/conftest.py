import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (gfx950); run with `-m gpu` on the GPU box')


@pytest.fixture(scope='session')
def gpu():
    import torch
    from dualvar_amd import _lib
    _lib.require_device()          # fails loudly (no silent CPU fallback)
    return torch.device('cuda:0')
