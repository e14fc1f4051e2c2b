"""The canonical caller of the PM cycle, examples/nbody.py:162-171 and 199-218 of the reference, restated
line for line on *numpy* inputs against pmesh_amd's drop-in surface (B4 of SURVEY.md 8b): the only
edit a user makes is the import.  Expected values: tests/golden/caller_nbody.npz, computed from the
reference's own window kernels + numpy.fft (tests/golden/make_caller_fixture.py).

`-m gpu`: the HIP library; `-m "not gpu"`: the same host code over the CPU oracle double.
"""
import os

import numpy
import pytest
from numpy.testing import assert_allclose

from pmesh_amd.pm import ParticleMesh     # reference: from pmesh.pm import ParticleMesh

HERE = os.path.dirname(os.path.abspath(__file__))


class pt:                                  # the cosmology object of examples/nbody.py (only Om0 is used here)
    Om0 = 0.31


def force_transfer(direction):             # examples/nbody.py:162-171
    def filter(k, v):
        k2 = sum(ki ** 2 for ki in k)
        k2[k2 == 0] = 1.0
        C = (v.BoxSize / v.Nmesh)[direction]
        w = k[direction] * C
        kfinite = 1.0 / C * 1 / 6.0 * (8 * numpy.sin(w) - numpy.sin(2 * w))
        return 1j * kfinite / k2 * v
    return filter


def force(pm, Q, S):                       # examples/nbody.py:199-218
    rho1 = pm.create('real')
    X = S + Q
    layout = pm.decompose(X, smoothing=1.0 * pm.resampler.support)
    rho1.paint(X, layout=layout, hold=False)

    N = pm.comm.allreduce(len(X))
    fac = 1.0 * pm.Nmesh.prod() / N
    rho1[...] *= fac
    rhok1 = rho1.r2c()

    rhok = rhok1

    F = numpy.empty_like(Q)
    for d in range(pm.ndim):
        F[..., d] = rhok.apply(force_transfer(d)) \
                  .c2r().readout(X, layout=layout)
    return 1.5 * pt.Om0 * F


@pytest.fixture(scope='module')
def fixture():
    return numpy.load(os.path.join(HERE, 'golden', 'caller_nbody.npz'))


@pytest.mark.parametrize('tag,resampler', [('n16_cic', 'cic'), ('n64_cic', 'cic'), ('n16_tsc', 'tsc')])
def test_force_of_nbody_example_on_numpy_inputs(be, fixture, tag, resampler):
    N, BoxSize, Om0 = fixture[tag + '_meta']
    N = int(N)
    Q, S, want = fixture[tag + '_Q'], fixture[tag + '_S'], fixture[tag + '_F']
    pm = ParticleMesh(BoxSize=BoxSize, Nmesh=[N, N, N], dtype='f8', resampler=resampler)
    F = force(pm, Q, S)
    assert isinstance(F, numpy.ndarray) and F.dtype == numpy.dtype('f8') and F.shape == Q.shape
    scale = abs(want).max()
    assert_allclose(F, want, rtol=0, atol=1e-11 * scale)
