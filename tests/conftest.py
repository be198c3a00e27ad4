import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def rel_err(a, b):
    """max |a-b| / max |b| (the reference's own CPU<->GPU metric,
    tests/test_cpu_gpu_deposition.py:96-98)."""
    a = np.asarray(a)
    b = np.asarray(b)
    den = np.abs(b).max()
    if den == 0:
        return np.abs(a).max()
    return np.abs(a - b).max() / den


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc
