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


@pytest.fixture(autouse=True)
def _sc2_dispatch_policy(monkeypatch):
    """The package reads no SC2_* dispatch variable (round 5: one policy struct through the C-ABI + hip.host_policy).  Tests select
    kernel variants the way the tools' A/B scripts do -- `monkeypatch.setenv('SC2_CONV_FORCE_BIG', '1')` -- so this fixture routes
    such a call through tools/env_policy.py (the one place that maps the variables onto the policy) as it is made, and puts the
    defaults back when the test ends."""
    from tools import env_policy
    if not any(k in os.environ for k in env_policy.ENV) and not os.path.exists(os.path.join(ROOT, 'sc2-benchmark_amd', 'libsc2amd.so')):
        yield
        return
    import ctypes
    import sc2bench_amd
    hip = sc2bench_amd.hip
    touched = []
    orig_set, orig_del = monkeypatch.setenv, monkeypatch.delenv

    def defaults():
        d = hip.Policy()
        hip.lib().sc2_policy_default(ctypes.byref(d))
        out = {n: getattr(d, n) for n in hip.POLICY_FIELDS if n != 'struct_bytes'}
        out.update({k: v for k, v in vars(hip.HostPolicy).items() if not k.startswith('_')})
        return out

    def setenv(name, value, *a, **k):
        orig_set(name, value, *a, **k)
        if name in env_policy.ENV:
            field, conv = env_policy.ENV[name]
            hip.configure(**{field: conv(str(value))})
            touched.append(field)

    def delenv(name, *a, **k):
        orig_del(name, *a, **k)
        if name in env_policy.ENV:
            field = env_policy.ENV[name][0]
            hip.configure(**{field: defaults()[field]})

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    if touched:
        d = defaults()
        hip.configure(**{f: d[f] for f in set(touched)})
