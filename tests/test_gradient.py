"""Finite-difference checks of the VJP/JVP compositions (pmesh/tests/test_gradient.py
pattern, dx = 1e-6..1e-7, rtol 1e-4..1e-5): readout_vjp / readout_jvp, paint_vjp /
paint_jvp and c2r_vjp are pure compositions of the gradient-readout / gradient-paint
kernels (order[d] = 1) already on the path."""
import numpy
import pytest
from numpy.testing import assert_allclose

from pmesh_amd.pm import ParticleMesh, RealField


def _setup(resampler='cic'):
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4, 4], dtype='f8', resampler=resampler)
    rs = numpy.random.RandomState(1234)
    real = pm.create('real', value=rs.normal(size=(4, 4, 4)))
    pos = numpy.array(numpy.indices(real.shape), dtype='f8').reshape(3, -1).T
    pos += 0.5                      # avoid the mesh points: the cic gradient jumps there
    pos += rs.uniform(-0.3, 0.3, size=pos.shape)
    pos *= pm.BoxSize / pm.Nmesh
    return pm, real, pos


@pytest.mark.parametrize('resampler', ['cic', 'tsc'])
def test_readout_gradients(be, resampler):          # test_gradient.py:104-170
    pm, real, pos = _setup(resampler)
    layout = pm.decompose(pos)

    def objective(real, pos):
        value = real.readout(pos, layout=layout)
        return (value ** 2).sum()

    value = real.readout(pos, layout=layout)
    grad_real, grad_pos = real.readout_vjp(pos, v=value * 2, layout=layout)
    obj = objective(real, pos)
    dx = 1e-6
    # d obj / d pos
    for ind in [(0, 0), (5, 1), (17, 2), (63, 0)]:
        p1 = pos.copy()
        p1[ind] += dx
        ng = (objective(real, p1) - obj) / dx
        assert_allclose(ng, grad_pos[ind], rtol=1e-4, atol=1e-6)
        # forward mode agrees
        v_pos = numpy.zeros_like(pos)
        v_pos[ind] = 1.0
        fg = (real.readout_jvp(pos, v_pos=v_pos, layout=layout) * value * 2).sum()
        assert_allclose(fg, grad_pos[ind], rtol=1e-9, atol=1e-12)
    # d obj / d real
    g = numpy.asarray(grad_real)
    for ind in [(0, 0, 0), (1, 2, 3), (3, 3, 1)]:
        r1 = real.copy()
        v = numpy.asarray(r1)
        v[ind] += dx
        r1[...] = v
        ng = (objective(r1, pos) - obj) / dx
        assert_allclose(ng, g[ind], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('resampler', ['cic', 'tsc'])
def test_paint_gradients(be, resampler):            # test_gradient.py:172-250
    pm, field, pos = _setup(resampler)
    rs = numpy.random.RandomState(7)
    mass = rs.uniform(0.5, 1.5, size=len(pos))
    layout = pm.decompose(pos)

    def objective(pos, mass):
        real = pm.paint(pos, mass=mass, layout=layout)
        return real.cdot(real)

    real = pm.paint(pos, mass=mass, layout=layout)
    v = real * 2
    grad_pos, grad_mass = pm.paint_vjp(v, pos, mass=mass, layout=layout)
    obj = objective(pos, mass)
    dx = 1e-6
    for ind in [(0, 0), (9, 1), (33, 2)]:
        p1 = pos.copy()
        p1[ind] += dx
        ng = (objective(p1, mass) - obj) / dx
        assert_allclose(ng, grad_pos[ind], rtol=1e-4, atol=1e-5)
        v_pos = numpy.zeros_like(pos)
        v_pos[ind] = 1.0
        fg = pm.paint_jvp(pos, mass=mass, v_pos=v_pos, layout=layout).cdot(v)
        assert_allclose(fg, grad_pos[ind], rtol=1e-9, atol=1e-12)
    for i in [0, 20, 63]:
        m1 = mass.copy()
        m1[i] += dx
        ng = (objective(pos, m1) - obj) / dx
        assert_allclose(ng, grad_mass[i], rtol=1e-4, atol=1e-5)


def test_c2r_r2c_vjp(be):                           # test_gradient.py:71-102
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4], dtype='f8')
    rs = numpy.random.RandomState(3)
    real = pm.create('real', value=rs.normal(size=(4, 4)) + 1.0)
    comp = real.r2c()

    def objective(comp):
        r = comp.c2r()
        return (numpy.asarray(r) ** 2).sum()

    grad_real = RealField(pm)
    grad_real[...] = real[...] * 2
    grad_comp = grad_real.c2r_vjp(grad_real)
    grad_comp.decompress_vjp(grad_comp)
    g = numpy.asarray(grad_comp)
    obj = objective(comp)
    dx = 1e-7
    for ind in [(0, 0), (1, 1), (2, 2), (3, 1)]:
        for part in (0, 1):
            c1 = comp.copy()
            v = numpy.asarray(c1)
            # perturb one stored mode and its hermitian partner consistently (csetitem semantics)
            v[ind] += dx if part == 0 else 1j * dx
            i0, i1 = ind
            if i1 in (0, 2):                    # self-conjugate column of the compressed axis
                j0 = (-i0) % 4
                if (j0, i1) != ind:
                    v[j0, i1] += dx if part == 0 else -1j * dx
                elif part == 1:
                    continue                    # purely real mode: no imaginary perturbation
            c1[...] = v
            ng = (objective(c1) - obj) / dx
            ag = g[ind].real if part == 0 else g[ind].imag
            assert_allclose(ng, ag, rtol=1e-4, atol=1e-5)
