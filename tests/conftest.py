import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy
    d = os.path.join(ROOT, 'tests', 'golden')
    return {name: numpy.load(os.path.join(d, name + '.npz')) for name in ('window', 'decompose', 'cycle16', 'whitenoise')}


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle (test infrastructure); built on demand."""
    from oracle import oracle as O
    O.lib('oracle')
    return O


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(params=['double', pytest.param('hip', marks=pytest.mark.gpu)])
def be(request):
    """The kernel backend under test.

    'hip'    (-m gpu)      : the product — libpmesh_amd.so on the GPU.
    'double' (-m "not gpu"): the host layer driven against the CPU oracle
                             (tests/oracle_backend.py), no GPU needed.
    """
    from pmesh_amd import backend
    if request.param == 'hip':
        backend.reset()
        b = backend.get()      # raises if the library or the GPU is missing
        assert b.name == 'hip'
    else:
        from tests import oracle_backend
        b = oracle_backend.install()
    yield b
    backend.reset()
