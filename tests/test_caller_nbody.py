"""A time-stepping caller's force evaluation on *numpy* inputs, through pmesh_amd's drop-in surface (B4 of SURVEY.md
8b): the sequence of calls the reference's N-body example makes per step (examples/nbody.py:199-218 — create,
decompose with the window's support as smoothing, paint, normalise to the mean density, r2c, one `apply` of a Python
transfer callable + c2r + readout per direction), written here as this repository's own caller of that API; the
finite-difference force kernel is the one the fixture was generated with (examples/nbody.py:162-171).  Expected
values: tests/golden/caller_nbody.npz, computed from the reference's own window kernels + numpy.fft
(tests/golden/make_caller_fixture.py).

`-m gpu`: the HIP library; `-m "not gpu"`: the same host code over the CPU oracle double.
"""
import os

import numpy
import pytest
from numpy.testing import assert_allclose

from pmesh_amd.pm import ParticleMesh

HERE = os.path.dirname(os.path.abspath(__file__))
OMEGA_M = 0.31                             # the fixture's cosmology (only the matter density enters the force)


class LongRangeForce(object):
    """`apply` callable: -grad of the inverse Laplacian along one axis, the gradient as the four-point finite
    difference D(w) = (8 sin w - sin 2w) / 6 per cell of that axis."""
    def __init__(self, axis):
        self.axis = axis

    def __call__(self, k, v):
        ksq = k[0] ** 2
        for kd in k[1:]:
            ksq = ksq + kd ** 2
        ksq[ksq == 0] = 1.0                # (the mean mode: its numerator is zero anyway)
        cell = (v.BoxSize / v.Nmesh)[self.axis]
        phase = k[self.axis] * cell
        stencil = (8 * numpy.sin(phase) - numpy.sin(2 * phase)) * (1.0 / cell * 1 / 6.0)
        return 1j * stencil / ksq * v


def pm_force(pm, lattice, displacement):
    """the particle-mesh force on particles at lattice + displacement, (n, ndim) numpy array"""
    positions = displacement + lattice
    routes = pm.decompose(positions, smoothing=1.0 * pm.resampler.support)
    density = pm.create('real')
    density.paint(positions, layout=routes, hold=False)
    density[...] *= 1.0 * pm.Nmesh.prod() / pm.comm.allreduce(len(positions))     # mean density 1
    spectrum = density.r2c()
    force = numpy.empty_like(lattice)
    for axis in range(pm.ndim):
        component = spectrum.apply(LongRangeForce(axis)).c2r()
        force[..., axis] = component.readout(positions, layout=routes)
    return 1.5 * OMEGA_M * force


@pytest.fixture(scope='module')
def fixture():
    return numpy.load(os.path.join(HERE, 'golden', 'caller_nbody.npz'))


@pytest.mark.parametrize('tag,resampler', [('n16_cic', 'cic'), ('n64_cic', 'cic'), ('n16_tsc', 'tsc')])
def test_force_of_nbody_example_on_numpy_inputs(be, fixture, tag, resampler):
    N, BoxSize, Om0 = fixture[tag + '_meta']
    N = int(N)
    Q, S, want = fixture[tag + '_Q'], fixture[tag + '_S'], fixture[tag + '_F']
    pm = ParticleMesh(BoxSize=BoxSize, Nmesh=[N, N, N], dtype='f8', resampler=resampler)
    F = pm_force(pm, Q, S)
    assert isinstance(F, numpy.ndarray) and F.dtype == numpy.dtype('f8') and F.shape == Q.shape
    scale = abs(want).max()
    assert_allclose(F, want, rtol=0, atol=1e-11 * scale)
