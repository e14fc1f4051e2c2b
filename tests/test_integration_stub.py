"""Executes the reference-side ctypes binding documented in INTEGRATION.md (tests/integration_stub.py):
paint / readout / get_fwindow with the argument lists of pmesh/_window.pyx:128-205, on device arrays,
against the CPU oracle and the golden fwindow values."""
import numpy
import pytest
import torch
from numpy.testing import assert_allclose, assert_array_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('kind', ['tunedcic', 'tunedtsc', 'tunedpcs', 'cubic'])
def test_reference_shaped_binding(oracle, golden, kind):
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
