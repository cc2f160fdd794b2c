import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def S():
    import sc2bench_amd
    return sc2bench_amd


@pytest.fixture(scope='session')
def R():
    from oracle import cpu_ref
    return cpu_ref


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    import sc2bench_amd
    assert sc2bench_amd.hip.lib().sc2_device_count() > 0
    return torch.device('cuda:0')
