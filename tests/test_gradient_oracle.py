"""The VJP / JVP operators of the PM cycle (pm.py:793-859 readout_vjp / readout_jvp, 1872-1935 paint_jvp /
paint_vjp, 865-870 c2r_vjp, 1021-1045 r2c_vjp / decompress_vjp) against compositions of the CPU ORACLE's
golden-pinned kernels — gradient readout x v, gradient paint of v * mass, numpy.fft under the reference's
normalisation — on 16^3 and 32^3 meshes with CIC / TSC / PCS.  fp64 tolerance 1e-12 of the largest value.
(tests/test_gradient.py keeps the finite-difference checks.)  Also get_compensation against golden vectors
produced by the reference's own window.py (tests/golden/make_compensation.py).
"""
import os

import numpy
import pytest
from numpy.testing import assert_allclose

from pmesh_amd.pm import ParticleMesh, RealField
from pmesh_amd.window import windows

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(16, 'cic'), (16, 'tsc'), (16, 'pcs'), (32, 'tsc'), (32, 'pcs')]


def _setup(N, name, seed=5):
    L = 10.0
    pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler=name)
    rs = numpy.random.RandomState(seed + N)
    n = 3000
    pos = rs.uniform(-0.2 * L, 1.2 * L, size=(n, 3))
    mass = rs.uniform(0.5, 1.5, size=n)
    field = rs.normal(size=(N, N, N))
    return pm, L, pos, mass, field, rs


def _close(a, b, tol=1e-12):
    a = numpy.asarray(a)
    b = numpy.asarray(b)
    assert_allclose(a, b, rtol=0, atol=tol * max(1.0, abs(b).max()))


@pytest.mark.parametrize('N,name', CASES)
def test_readout_vjp_jvp_equal_oracle_compositions(be, oracle, N, name):
    pm, L, pos, mass, field, rs = _setup(N, name)
    W = oracle.Window('tuned' + name)
    aff = oracle.Affine(3, scale=N / L, translate=0, period=N)
    real = pm.create('real', value=field)
    v = rs.normal(size=len(pos))
    layout = pm.decompose(pos)
    grad_real, grad_pos = real.readout_vjp(pos, v=v, layout=layout)
    want_pos = numpy.stack([W.readout(field, pos, transform=aff, diffdir=d) * v for d in range(3)], axis=1)
    want_real = numpy.zeros((N, N, N))
    W.paint(want_real, pos, mass=v, transform=aff)
    _close(grad_pos, want_pos)
    _close(grad_real, want_real)
    # forward mode: f_i = W_qi A_q
    v_pos = rs.normal(size=pos.shape)
    v_self_h = rs.normal(size=(N, N, N))
    v_self = pm.create('real', value=v_self_h)
    jvp = real.readout_jvp(pos, v_self=v_self, v_pos=v_pos, layout=layout)
    want = sum(W.readout(field, pos, transform=aff, diffdir=d) * v_pos[:, d] for d in range(3))
    want = want + W.readout(v_self_h, pos, transform=aff)
    _close(jvp, want)


@pytest.mark.parametrize('N,name', CASES)
def test_paint_vjp_jvp_equal_oracle_compositions(be, oracle, N, name):
    pm, L, pos, mass, field, rs = _setup(N, name, seed=9)
    W = oracle.Window('tuned' + name)
    aff = oracle.Affine(3, scale=N / L, translate=0, period=N)
    v = pm.create('real', value=field)
    layout = pm.decompose(pos)
    grad_pos, grad_mass = pm.paint_vjp(v, pos, mass=mass, layout=layout)
    want_pos = numpy.stack([W.readout(field, pos, transform=aff, diffdir=d) * mass for d in range(3)], axis=1)
    _close(grad_pos, want_pos)
    _close(grad_mass, W.readout(field, pos, transform=aff))
    v_pos = rs.normal(size=pos.shape)
    v_mass = rs.normal(size=len(pos))
    jvp = pm.paint_jvp(pos, mass=mass, v_pos=v_pos, v_mass=v_mass, layout=layout)
    want = numpy.zeros((N, N, N))
    for d in range(3):
        W.paint(want, pos, mass=v_pos[:, d] * mass, transform=aff, diffdir=d)
    W.paint(want, pos, mass=v_mass, transform=aff)
    _close(jvp, want)


@pytest.mark.parametrize('N', [16, 32])
def test_fft_vjps_equal_numpy_compositions(be, N):
    pm = ParticleMesh(BoxSize=10.0, Nmesh=[N, N, N], dtype='f8')
    rs = numpy.random.RandomState(N)
    field = rs.normal(size=(N, N, N))
    v = pm.create('real', value=field)
    # c2r_vjp(v) = r2c(v) * prod(Nmesh)  =  rfftn(v)
    g = v.c2r_vjp(v)
    _close(g, numpy.fft.rfftn(field), tol=1e-13 * N ** 1.5)
    # decompress_vjp: modes that are not their own conjugates count twice
    want = numpy.fft.rfftn(field)
    i0, i1, i2 = numpy.meshgrid(numpy.arange(N), numpy.arange(N), numpy.arange(N // 2 + 1), indexing='ij')
    selfconj = ((N - i0) % N == i0) & ((N - i1) % N == i1) & ((N - i2) % N == i2)
    want = numpy.where(selfconj, want, 2 * want)
    _close(g.decompress_vjp(g), want, tol=1e-13 * N ** 1.5)
    # r2c_vjp(c) = c2r(c) / prod(Nmesh)  =  irfftn(c)
    ck_h = numpy.fft.rfftn(rs.normal(size=(N, N, N)))
    ck = pm.create('complex', value=ck_h)
    _close(ck.r2c_vjp(ck), numpy.fft.irfftn(ck_h, s=(N, N, N), axes=(0, 1, 2)), tol=1e-13)


def test_compensation_equals_reference(be):
    """get_compensation() (window.py:65-80) on a block of circular frequencies against the output of the
    reference's own window.py (golden; tests/golden/make_compensation.py)."""
    g = numpy.load(os.path.join(HERE, 'golden', 'compensation.npz'))
    w = [g['w0'], g['w1'], g['w2']]
    v = g['v']
    for name in ['nnb', 'cic', 'tsc', 'pcs', 'nearest', 'linear', 'quadratic', 'cubic']:
        W = windows[name]
        assert_allclose(W.get_compensation()(w, v), g['%s/native' % name], rtol=1e-15, atol=0)
        assert_allclose(W.resize(6).get_compensation()(w, v), g['%s/resize6' % name], rtol=1e-15, atol=0)
