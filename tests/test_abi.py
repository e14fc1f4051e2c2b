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


def test_cython_shim_is_the_header():
    """pmesh_amd/_pmx (the Cython shim the product binds the library with) is generated from include/pmesh_amd.h: the
    committed .pyx is what the generator writes today, it wraps exactly the header's entry points, it resolves all of
    them in the built library, and it turns the argument forms the host code passes into the right addresses."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'pmesh_amd', 'csrc'))
    try:
        import gen_pyx
    finally:
        sys.path.pop(0)
    protos = gen_pyx.prototypes(open(os.path.join(ROOT, 'include', 'pmesh_amd.h')).read())
    assert sorted(p[1] for p in protos) == declared_symbols()
    assert open(os.path.join(ROOT, 'pmesh_amd', '_pmx.pyx')).read() == gen_pyx.emit(protos), \
        'pmesh_amd/_pmx.pyx is stale: run `make -C pmesh_amd/csrc`'
    from pmesh_amd import backend, _abi, _pmx
    assert sorted(_pmx.NAMES) == declared_symbols()
    lib = backend.load_library()
    assert lib is _pmx and _pmx.bound() == backend.library_path()
    assert lib.pmx_version() >= 100 and isinstance(lib.pmx_build_flags(), bytes)
    # argument forms: None, int, c_void_p, byref(struct), struct, array, POINTER, Struct
    p = _abi.Painter()
    arr = (ctypes.c_int64 * 3)(1, 2, 3)
    assert _pmx.address(None) == 0 and _pmx.address(12345) == 12345
    assert _pmx.address(ctypes.c_void_p(77)) == 77 and _pmx.address(ctypes.c_void_p()) == 0
    assert _pmx.address(ctypes.byref(p)) == ctypes.addressof(p) == _pmx.address(p)
    assert _pmx.address(arr) == ctypes.addressof(arr)
    assert _pmx.address(ctypes.cast(arr, ctypes.POINTER(ctypes.c_int64))) == ctypes.addressof(arr)
    st = _pmx.Struct(_abi.Painter)
    st.view.kind = 5
    assert _pmx.address(st) == st.addr == ctypes.addressof(st.view) and st.addr % 16 == 0
    # a call with a struct: the same answer as through ctypes
    clib = backend.load_library(binding='ctypes')
    p.kind, p.ndim, p.canvas_elsize = 5, 3, 8
    for d in range(3):
        p.period[d] = p.size[d] = 64
        p.strides[d] = 8 * 64 ** (2 - d)
    for n in (1000, 10 ** 8):
        assert lib.pmx_binplan_supported(p, n) == clib.pmx_binplan_supported(ctypes.byref(p), n)
    with pytest.raises((OverflowError, TypeError)):
        lib.pmx_colfft_supported('512', 8)


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
