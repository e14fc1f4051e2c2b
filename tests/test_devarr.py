"""Callables of Field.apply on device arrays (pmesh_amd/_devarr.py; reference: pm.py:617-648, the transfer functions
of examples/nbody.py:151-197 and of fastpm-python's kernels, restated below as a caller writes them).

The reference hands `func(k, v)` numpy arrays.  Here `k` and `v` are numpy-flavoured handles on device tensors: the
same source must give the same numbers as on numpy arrays — including the spellings whose torch meaning differs
(`nonzero()` masks) — without the field leaving the device, and a callable that cannot be expressed must fall back to
the host evaluation without having damaged its input.
"""
import numpy
import pytest
import torch
from numpy.testing import assert_allclose

from pmesh_amd._devarr import DevArr
from pmesh_amd.pm import ParticleMesh
import pmesh_amd.pm as pmod


# ---- transfer functions as callers write them -----------------------------------------------------------------------
def laplace(k, v):                                   # fastpm: kernels.laplace
    kk = sum(ki ** 2 for ki in k)
    mask = (kk == 0).nonzero()
    kk[mask] = 1
    b = v / kk
    b[mask] = 0
    return b


def gradient_finite(d, C):                           # examples/nbody.py:162-171, fastpm: kernels.gradient
    def kernel(k, v):
        w = k[d] * C
        a = 1 / (6.0 * C) * (8 * numpy.sin(w) - numpy.sin(2 * w))
        return 1j * a * v
    return kernel


def longrange(r_split):                              # fastpm: kernels.longrange
    def kernel(k, v):
        kk = sum(ki ** 2 for ki in k)
        return v * numpy.exp(-kk * r_split ** 2)
    return kernel


def lowpass_sharp(kcut):
    def kernel(k, v):
        kk = k.normp(p=2) ** 0.5
        return numpy.where(kk > kcut, 0.0, v)
    return kernel


def cic_compensation(k, v):                          # nbodykit: CompensateCIC
    for i in range(3):
        wi = k[i]
        tmp = (1 - 2. / 3 * numpy.sin(0.5 * wi) ** 2) ** 0.5
        v = v / tmp
    return v


def tsc_sinc(k, v):
    out = v.copy()
    for i in range(3):
        out *= numpy.sinc(k[i] / (2 * numpy.pi)) ** -3
    return out


def phases(k, v):
    amp = numpy.abs(v)
    amp[amp == 0] = 1.0
    u = v / amp
    u.imag[...] = -u.imag
    return numpy.conj(u) * numpy.sqrt(amp) + numpy.real(v).clip(-0.5, 0.5) + numpy.maximum(numpy.imag(v), 0.0)


KERNELS = [laplace, gradient_finite(1, 0.37), longrange(1.25), lowpass_sharp(2.0), cic_compensation, tsc_sinc, phases]


def _coords(shape, rs):
    class xs(list):
        def normp(self, p=2, zeromode=None):
            kk = sum(abs(ki) ** p for ki in self)
            if zeromode is not None:
                kk[kk == 0] = zeromode
            return kk
    k = []
    for d, n in enumerate(shape):
        x = numpy.fft.fftfreq(n) * 2 * numpy.pi if d < 2 else numpy.arange(n) * numpy.pi / max(n - 1, 1)
        sh = [1, 1, 1]
        sh[d] = n
        k.append(x.reshape(sh))
    v = rs.normal(size=shape) + 1j * rs.normal(size=shape)
    return xs(k), v, xs


@pytest.mark.parametrize('kernel', KERNELS, ids=lambda f: getattr(f, '__name__', 'kernel'))
@pytest.mark.parametrize('ctype', ['c16', 'c8'])
def test_same_source_same_numbers(kernel, ctype):
    rs = numpy.random.RandomState(7)
    k, v, xs = _coords((8, 6, 5), rs)
    v = v.astype(ctype)
    rtype = 'f8' if ctype == 'c16' else 'f4'
    k = xs([x.astype(rtype) for x in k])
    want = kernel(xs([x.copy() for x in k]), v.copy())
    dk = xs([DevArr(torch.from_numpy(x.copy())) for x in k])
    got = kernel(dk, DevArr(torch.from_numpy(v.copy())))
    assert isinstance(got, DevArr)
    assert got.dtype == numpy.asarray(want).dtype
    assert_allclose(got.t.numpy(), want, rtol=1e-12 if ctype == 'c16' else 2e-5, atol=1e-13 if ctype == 'c16' else 1e-6)


def test_numpy_conventions():
    t = torch.arange(24, dtype=torch.float64).reshape(2, 3, 4)
    a = DevArr(t.clone())
    n = t.numpy().copy()
    m = (a % 5 == 0).nonzero()
    assert isinstance(m, tuple) and len(m) == 3                     # numpy's tuple of index arrays, not torch's matrix
    a[m] = -1
    n[(n % 5 == 0).nonzero()] = -1
    assert_allclose(a.t.numpy(), n)
    assert a.shape == (2, 3, 4) and a.ndim == 3 and a.size == 24 and a.dtype == numpy.dtype('f8')
    assert float(a.sum()) == n.sum() and float(numpy.sum(a)) == n.sum()
    assert_allclose(numpy.sum(a, axis=1).t.numpy(), n.sum(axis=1))
    assert_allclose((a > 3).any(axis=2).t.numpy(), (n > 3).any(axis=2))
    assert_allclose(numpy.add.reduce(a, axis=0).t.numpy(), numpy.add.reduce(n, axis=0))
    assert_allclose((1j * a).t.numpy(), 1j * n)                      # a python complex scalar promotes as in numpy
    assert (a * numpy.float32(2)).dtype == numpy.dtype('f8')
    assert (DevArr(t.float()) * 2.5).dtype == numpy.dtype('f4')
    assert_allclose((a ** 2).t.numpy(), n ** 2)
    assert_allclose((2.0 ** DevArr(t / 8)).t.numpy(), 2.0 ** (t.numpy() / 8))
    assert_allclose((n + a).t.numpy(), 2 * n)                        # an ndarray on the left defers to the device array
    assert_allclose(numpy.arctan2(a, 2.0).t.numpy(), numpy.arctan2(n, 2.0))
    out = DevArr(torch.empty_like(t))
    numpy.multiply(a, 3.0, out=out)
    assert_allclose(out.t.numpy(), 3 * n)
    with pytest.raises(TypeError):
        numpy.fft.fft(a)                                             # not mapped: the caller falls back to the host
    with pytest.raises(TypeError):
        numpy.asarray(a)


@pytest.mark.parametrize('kernel', KERNELS[:6], ids=lambda f: getattr(f, '__name__', 'kernel'))
def test_apply_on_device_arrays_equals_host_evaluation(be, kernel, monkeypatch):
    """ComplexField.apply(callable): the device-array evaluation and the reference's slab loop on numpy arrays
    (Field._apply_host, pm.py:633-647) give the same field, and the first never calls the second"""
    pm = ParticleMesh(BoxSize=[100.0, 80.0, 120.0], Nmesh=[8, 6, 10], dtype='f8')
    rs = numpy.random.RandomState(5)
    ck = pm.create(type='real', value=rs.normal(size=(8, 6, 10))).r2c()
    host = ck.copy()
    type(ck)._apply_host(ck, kernel, 'wavenumber', host.value)
    called = []
    orig = type(ck)._apply_host
    monkeypatch.setattr(type(ck), '_apply_host', lambda self, *a: (called.append(1), orig(self, *a))[1])
    dev = ck.apply(kernel)
    assert not called, 'the callable was sent to the host'
    assert_allclose(numpy.asarray(dev), numpy.asarray(host), rtol=1e-12, atol=1e-14)


def test_a_callable_that_needs_numpy_still_runs_and_never_on_damaged_input(be):
    pm = ParticleMesh(BoxSize=10.0, Nmesh=[8, 8, 8], dtype='f8')
    rs = numpy.random.RandomState(2)
    ck = pm.create(type='real', value=rs.normal(size=(8, 8, 8))).r2c()
    before = numpy.asarray(ck).copy()

    def needs_numpy(k, v):
        return numpy.fft.ifft(numpy.fft.fft(v, axis=-1), axis=-1) * 2.0      # no device form: host slab loop
    got = ck.apply(needs_numpy)
    assert_allclose(numpy.asarray(got), 2 * before, rtol=1e-12, atol=1e-14)
    assert_allclose(numpy.asarray(ck), before)

    def spoils_then_fails(k, v):
        v[...] = 0
        return numpy.fft.fft(v)
    with pytest.raises(RuntimeError, match='modified its input in place'):
        ck.apply(spoils_then_fails)


def test_slab_loops_in_numpy_terms(be):
    """`for k, slab in zip(field.slabs.x, field.slabs)` (pm.py:87-153; how nbodykit's algorithms walk a field): the
    slabs are device arrays too — numpy functions apply, an in-place update writes through to the field"""
    pm = ParticleMesh(BoxSize=[8.0, 6.0, 10.0], Nmesh=[8, 6, 10], dtype='f8')
    rs = numpy.random.RandomState(11)
    ck = pm.create(type='real', value=rs.normal(size=(8, 6, 10))).r2c()
    host = numpy.asarray(ck).copy()
    k = [numpy.asarray(x.cpu()) for x in ck.x]
    k2full = sum(ki ** 2 for ki in k)
    want_total = float(numpy.sum(numpy.abs(host) ** 2 * numpy.exp(-k2full)))
    total = 0.0
    nslabs = 0
    for kk, slab in zip(ck.slabs.x, ck.slabs):
        k2 = sum(ki ** 2 for ki in kk)
        total += float(numpy.sum(numpy.abs(slab) ** 2 * numpy.exp(-k2)))
        slab[...] *= numpy.exp(-0.5 * k2)
        assert slab.BoxSize is not None and len(slab.x) == 3
        nslabs += 1
    assert nslabs == ck.shape[0] or nslabs == max(ck.shape)
    assert abs(total - want_total) <= 1e-12 * abs(want_total)
    assert_allclose(numpy.asarray(ck), host * numpy.exp(-0.5 * k2full), rtol=1e-13, atol=1e-15)


def test_numpy_scalars_reductions_and_casts_follow_numpy():
    """numpy SCALARS are strong in numpy's promotion (NEP 50: `x * numpy.float64(c)` on a float32 array computes in
    float64; python floats are weak); `dtype=` of the reductions is the accumulator's type; a complex result does not
    go into a real array (numpy raises — Field.apply then evaluates the callable on the host)."""
    h = numpy.linspace(0.1, 3.0, 64, dtype='f4').reshape(4, 16)
    a = DevArr(torch.from_numpy(h.copy()))
    C = (numpy.array([1000.0]) / numpy.array([512]))[0]            # a numpy.float64, as (pm.BoxSize / pm.Nmesh)[d] is
    assert isinstance(C, numpy.float64)
    got, want = a * C, h * C
    assert got.dtype == want.dtype == numpy.float64
    assert_allclose(got.t.numpy(), want, rtol=0, atol=0)
    assert (a * 2.5).dtype == (h * 2.5).dtype == numpy.float32      # python scalars stay weak
    assert (a * numpy.float32(2.5)).dtype == numpy.float32
    big = DevArr(torch.full((1 << 16,), 1.0e-3, dtype=torch.float32))
    assert big.sum(dtype='f8').dtype == numpy.float64
    assert float(big.sum(dtype='f8')) == float(numpy.full(1 << 16, 1.0e-3, dtype='f4').sum(dtype='f8'))
    assert numpy.add.reduce(big, dtype='f8').dtype == numpy.float64
    assert big.mean(dtype='f8').dtype == numpy.float64
    r = DevArr(torch.zeros(8, dtype=torch.float64))
    with pytest.raises(TypeError):
        r[...] = DevArr(torch.ones(8, dtype=torch.complex128))
    with pytest.raises(TypeError):
        numpy.multiply(r, 1j, out=r)
    with pytest.raises(TypeError):
        r *= 1j
    c = DevArr(torch.ones(8, dtype=torch.complex128))
    c *= 1j                                                           # complex into complex is fine
    assert_allclose(c.t.numpy(), numpy.full(8, 1j))
