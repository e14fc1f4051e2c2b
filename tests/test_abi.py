"""The C-ABI boundary: libpmesh_amd.so loads (no GPU needed) and exports every
symbol that include/pmesh_amd.h declares; the ctypes table declares every one of
them; the product backend refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'pmesh_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pmx_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound():
    from pmesh_amd import backend, _abi
    names = declared_symbols()
    assert len(names) >= 25
    path = backend.library_path()
    assert os.path.exists(path), 'build the library first: python -c "import __graft_entry__ as g; g.build()"'
    lib = ctypes.CDLL(path)
    for n in names:
        assert hasattr(lib, n), 'libpmesh_amd.so does not export %s' % n
        short = n[len('pmx_'):]
        assert short in _abi.PROTOTYPES or short in _abi.DEVICE_ONLY, 'no ctypes prototype for %s' % n
    # and nothing is bound that the header does not declare
    for short in list(_abi.PROTOTYPES) + list(_abi.DEVICE_ONLY):
        assert 'pmx_' + short in names, 'pmx_%s is bound but not declared in the header' % short
    backend.load_library()          # declares every prototype; raises on a missing symbol
    assert lib.pmx_version() >= 100


def test_oracle_exports_the_same_signatures():
    from pmesh_amd import _abi
    from oracle import oracle as O
    lib, prefix = O.lib('oracle')
    for short in _abi.PROTOTYPES:
        assert hasattr(lib, prefix + short)


def test_no_cpu_fallback():
    """Without a GPU the product backend raises; it never computes on the host."""
    import torch
    from pmesh_amd import backend
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible here')
    backend.reset()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        backend.get()
    from pmesh_amd.window import CIC
    import numpy
    with pytest.raises(RuntimeError):
        CIC.paint(numpy.zeros((4, 4)), [[1., 1.]])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'pmesh_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, fn)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, fn
                assert 'liboracle.so' not in text.replace('(oracle/liboracle.so)', ''), fn
