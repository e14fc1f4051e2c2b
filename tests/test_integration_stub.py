"""Executes the reference-side binding documented in INTEGRATION.md in its two forms — Cython, the language of the
reference's own native layer (tests/integration_stub_cy.pyx), and ctypes (tests/integration_stub.py):
paint / readout / get_fwindow with the argument lists of pmesh/_window.pyx:128-205, on device arrays,
against the CPU oracle and the golden fwindow values."""
import numpy
import pytest
import torch
from numpy.testing import assert_allclose, assert_array_equal

import importlib.util
import os
import subprocess
import sys
import sysconfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CY = {}


def cython_stub():
    """tests/integration_stub_cy.pyx compiled (cython, the host compiler) and linked against libpmesh_amd.so, in a
    build directory of its own under tests/ (git-ignored); built once per session"""
    if 'mod' in _CY:
        return _CY['mod']
    src = os.path.join(ROOT, 'tests', 'integration_stub_cy.pyx')
    out = os.path.join(ROOT, 'tests', '_build_stub')
    os.makedirs(out, exist_ok=True)
    ext = os.path.join(out, 'integration_stub_cy' + sysconfig.get_config_var('EXT_SUFFIX'))
    lib = os.path.join(ROOT, 'pmesh_amd')
    stale = (not os.path.exists(ext) or os.path.getmtime(ext) < os.path.getmtime(src)
             or os.path.getmtime(ext) < os.path.getmtime(os.path.join(lib, 'libpmesh_amd.so')))
    if stale:
        c = os.path.join(out, 'integration_stub_cy.c')
        subprocess.run([sys.executable, '-m', 'cython', '-3', src, '-o', c], check=True, capture_output=True)
        subprocess.run(['cc', '-O2', '-fPIC', '-shared', '-fno-strict-aliasing', '-I' + sysconfig.get_paths()['include'],
                        '-I' + os.path.join(ROOT, 'include'), c, '-o', ext, '-L' + lib, '-lpmesh_amd',
                        '-Wl,-rpath,' + lib, '-Wl,-rpath,/opt/rocm/lib'], check=True, capture_output=True)
    spec = importlib.util.spec_from_file_location('integration_stub_cy', ext)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _CY['mod'] = mod
    return mod


def test_cython_binding_builds_and_loads():
    """(no GPU needed) the Cython form of the reference-side binding compiles against include/pmesh_amd.h, links against
    the library and answers a call that touches no device"""
    stub = cython_stub()
    W = stub.ResampleWindow('tunedcic')
    assert (W.support, W.nativesupport, W.kind) == (2, 2, 'tunedcic')
    assert stub.ResampleWindow('tunedpcs').support == 4
    import numpy as np
    assert abs(W.get_fwindow(np.array([0.0, 1.0]))[0] - 1.0) < 1e-15


@pytest.mark.gpu
@pytest.mark.parametrize('binding', ['ctypes', 'cython'])
@pytest.mark.parametrize('kind', ['tunedcic', 'tunedtsc', 'tunedpcs', 'cubic'])
def test_reference_shaped_binding(oracle, golden, kind, binding):
    if binding == 'cython':
        stub = cython_stub()
    else:
        from tests import integration_stub as stub
    dev = torch.device('cuda', 0)
    W = stub.ResampleWindow(kind)
    OW = oracle.Window(kind)
    assert (W.support, W.nativesupport) == (OW.support, OW.nativesupport)
    rs = numpy.random.RandomState(31)
    shape, period = (12, 10, 16), (24, 10, 16)               # a slab-local block of a periodic mesh
    scale, translate = (1.0, 0.5, 2.0), (-4.0, 0.25, 0.0)
    pos_h = rs.uniform(-3, 30, size=(5000, 3))
    mass_h = rs.uniform(0.5, 1.5, size=5000)
    aff = oracle.Affine(3, scale=scale, translate=translate, period=period)
    for order in ([0, 0, 0], [0, 1, 0]):
        diffdir = order.index(1) if 1 in order else None
        real = torch.zeros(shape, dtype=torch.float64, device=dev)
        pos = torch.from_numpy(pos_h).to(dev)
        W.paint(real, pos, None, torch.from_numpy(mass_h).to(dev), order, scale, translate, period)
        want = numpy.zeros(shape)
        OW.paint(want, pos_h, mass=mass_h, transform=aff, diffdir=diffdir)
        assert_allclose(real.cpu().numpy(), want, rtol=0, atol=1e-12 * max(1.0, abs(want).max()))
        ones = torch.ones(1, dtype=torch.float64, device=dev)           # window.py:146: mass broadcast
        real1 = torch.zeros(shape, dtype=torch.float64, device=dev)
        W.paint(real1, pos, None, ones, order, scale, translate, period)
        want1 = numpy.zeros(shape)
        OW.paint(want1, pos_h, transform=aff, diffdir=diffdir)
        assert_allclose(real1.cpu().numpy(), want1, rtol=0, atol=1e-12 * max(1.0, abs(want1).max()))
        field_h = rs.normal(size=shape)
        out = torch.empty(len(pos_h), dtype=torch.float64, device=dev)
        W.readout(torch.from_numpy(field_h).to(dev), pos, None, out, order, scale, translate, period)
        assert_array_equal(out.cpu().numpy(), OW.readout(field_h, pos_h, transform=aff, diffdir=diffdir))
    name = {'tunedcic': 'cic', 'tunedtsc': 'tsc', 'tunedpcs': 'pcs', 'cubic': 'cubic'}[kind]
    assert_allclose(W.get_fwindow(golden['window']['W/w']), golden['window']['W/%s/fwindow' % name], rtol=1e-15)
