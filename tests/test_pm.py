"""ParticleMesh / RealField / ComplexField through the public API of pmesh_amd.pm.

Restates the reference's pmesh/tests/test_pm.py expectations for the hot path
(cited per test) plus the golden 16^3 PM cycle generated from the compiled
reference kernels and numpy.fft (tests/golden/cycle16.npz).  Two modes as in
test_window.py: `-m gpu` runs the HIP kernels + rocFFT, `-m "not gpu"` drives the
same host code against the CPU oracle.
"""
import numpy
import pytest
import torch
from numpy.testing import assert_array_equal, assert_allclose, assert_almost_equal

from pmesh_amd.pm import ParticleMesh, RealField, ComplexField, UntransposedComplexField
from pmesh_amd.transfer import Transfer
from pmesh_amd import window

# FFT tolerances: f8 rel-L2 <= 1e-13, f4 <= 5e-6; full cycle f8 <= 1e-11
def rel_l2(a, b):
    a = numpy.asarray(a); b = numpy.asarray(b)
    return numpy.sqrt((abs(a - b) ** 2).sum() / max((abs(b) ** 2).sum(), 1e-300))


def test_shapes_and_attributes(be):           # test_pm.py:30-42
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8], dtype='f8')
    real = RealField(pm)
    assert tuple(real.cshape) == (8, 8) and real.csize == 64
    comp = ComplexField(pm)
    assert tuple(comp.cshape) == (8, 5) and comp.csize == 40
    assert pm.ndim == 2 and tuple(pm.Nmesh) == (8, 8) and tuple(pm.BoxSize) == (8.0, 8.0)
    assert real.dtype == numpy.dtype('f8') and comp.dtype == numpy.dtype('c16')
    assert real.slices == (slice(0, 8), slice(0, 8))


def test_negnyquist(be):                      # test_pm.py:46-53
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8], dtype='f8')
    c = pm.create(type='complex')
    last = c.x[-1].cpu().numpy()
    assert (last[0][-1] < 0).all()
    assert (last[0][:-1] >= 0).all()


def test_indices(be):                         # test_pm.py:268-275
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4], dtype='f8')
    comp = pm.create(type='complex')
    real = pm.create(type='real')
    assert_almost_equal(comp.x[0].cpu(), [[0], [0.785], [-1.571], [-0.785]], decimal=3)
    assert_almost_equal(comp.x[1].cpu(), [[0, 0.785, -1.571]], decimal=3)
    assert_almost_equal(real.x[0].cpu(), [[0], [2], [-4], [-2]], decimal=3)
    assert_almost_equal(real.x[1].cpu(), [[0, 2, -4, -2]], decimal=3)


def test_fft(be):                             # test_pm.py:128-141
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4], dtype='f4')
    real = pm.create(type='real', value=0)
    real[...] = 2
    real[::2, ::2] = -2
    real3 = real.copy()
    complex = real.r2c()
    assert_almost_equal(numpy.asarray(real), numpy.asarray(real3), decimal=7)   # input preserved
    real2 = complex.c2r()
    assert_almost_equal(numpy.asarray(real), numpy.asarray(real2), decimal=6)


@pytest.mark.parametrize('dtype,tol', [('f8', 1e-13), ('f4', 5e-6)])
@pytest.mark.parametrize('Nmesh', [[8, 12, 10], [16, 16], [32]])
def test_fft_contract_vs_numpy(be, dtype, tol, Nmesh):
    """r2c == rfftn / prod(N); c2r == irfftn * prod(N) (pm.py:692; test_pm.py:422-428)."""
    pm = ParticleMesh(BoxSize=1.0, Nmesh=Nmesh, dtype=dtype)
    rs = numpy.random.RandomState(7)
    data = rs.normal(size=Nmesh).astype(dtype)
    real = pm.create(type='real', value=data)
    assert_array_equal(numpy.asarray(real), data)
    ck = real.r2c()
    ref = numpy.fft.rfftn(data.astype('f8')) / numpy.prod(Nmesh)
    assert rel_l2(ck, ref) < tol
    assert_array_equal(numpy.asarray(real), data)            # PRESERVE_INPUT (pm.py:1335)
    back = ck.c2r()
    assert rel_l2(back, data) < 4 * tol
    assert rel_l2(ck, ref) < tol                             # c2r preserves its input too
    # untransposed flavour gives the same modes
    cu = real.r2c(out=UntransposedComplexField(pm))
    assert rel_l2(cu, ref) < tol
    assert rel_l2(cu.c2r(), data) < 4 * tol


def test_inplace_fft(be):                     # the case of test_pm.py:167-192
    """out=Ellipsis transforms over the field's own buffer and gives what the out-of-place transform gives"""
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8], dtype='f8')
    ndim = len(pm.Nmesh)
    pos = (1.0 * numpy.arange(100 * ndim).reshape(-1, ndim) * (7, 7)) % (pm.Nmesh + 1)
    density = pm.paint(pm.decompose(pos).exchange(pos))
    spectrum = density.r2c()
    spectrum_ip = density.r2c(out=Ellipsis)
    assert density._base in spectrum_ip._base
    assert_almost_equal(numpy.asarray(spectrum), numpy.asarray(spectrum_ip), decimal=7)
    back = spectrum_ip.c2r()
    back_ip = spectrum_ip.c2r(out=Ellipsis)
    assert back_ip._base in spectrum_ip._base
    assert_almost_equal(numpy.asarray(back), numpy.asarray(back_ip), decimal=7)


def test_decompose_paint_equals_serial(be):   # test_pm.py:230-264
    pm = ParticleMesh(BoxSize=4.0, Nmesh=[4, 4, 4], dtype='f8')
    pos = pm.generate_uniform_particle_grid(shift=0.5)
    all_pos = pos.cpu().numpy()
    for resampler in ['cic', 'tsc', 'pcs']:
        truth = numpy.zeros(pm.Nmesh, dtype='f8')
        window.FindResampler(resampler).paint(truth, all_pos, transform=window.Affine(ndim=3, period=4))
        layout = pm.decompose(pos, smoothing=resampler)
        npos = layout.exchange(pos)
        real = pm.paint(npos, resampler=resampler)
        full = numpy.zeros(pm.Nmesh, dtype='f8')
        full[real.slices] = numpy.asarray(real)
        assert_almost_equal(full, truth)


def test_real_apply(be):                      # test_pm.py:329-341
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8], dtype='f8')
    real = RealField(pm)

    def filter(x, v):
        xnormp = x.normp()
        assert_allclose(xnormp.cpu() if hasattr(xnormp, 'cpu') else xnormp,
                        sum(xi ** 2 for xi in x).cpu() if hasattr(xnormp, 'cpu') else sum(xi ** 2 for xi in x))
        return x[0] * 10 + x[1]
    real.apply(filter, out=Ellipsis)
    for i, x, slab in zip(real.slabs.i, real.slabs.x, real.slabs):
        assert_array_equal(slab.cpu(), (x[0] * 10 + x[1]).cpu())


def test_complex_apply(be):                   # test_pm.py:343-355
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8], dtype='f8')
    complex = ComplexField(pm)

    def filter(k, v):
        return k[0] + k[1] * 1j
    complex.apply(filter, out=Ellipsis)
    for i, x, slab in zip(complex.slabs.i, complex.slabs.x, complex.slabs):
        assert_array_equal(slab.cpu(), (x[0] + x[1] * 1j).cpu())


def test_apply_numpy_only_callable_falls_back_to_host(be):
    """nbody.py's force_transfer calls numpy.sin on the wavenumbers: evaluated on the
    host slab by slab, as the reference does (pm.py:633-647)."""
    pm = ParticleMesh(BoxSize=100.0, Nmesh=[8, 8, 8], dtype='f8')
    rs = numpy.random.RandomState(3)
    ck = pm.create(type='real', value=rs.normal(size=(8, 8, 8))).r2c()

    def force_transfer(direction):            # examples/nbody.py:162-171
        def filter(k, v):
            k2 = sum(ki ** 2 for ki in k)
            k2[k2 == 0] = 1.0
            C = (v.BoxSize / v.Nmesh)[direction]
            w = k[direction] * C
            kfinite = 1.0 / C * 1 / 6.0 * (8 * numpy.sin(w) - numpy.sin(2 * w))
            return 1j * kfinite / k2 * v
        return filter
    for d in range(3):
        a = ck.apply(force_transfer(d))
        b = ck.apply(Transfer.force(d))
        assert rel_l2(a, b) < 1e-14


@pytest.mark.parametrize('dtype,tol', [('f8', 1e-14), ('f4', 1e-6)])
def test_fused_transfers_match_reference_filters(be, dtype, tol):
    """Transfer.{dx1,force,potential,lowpass,compensation} == the numpy filters of
    examples/nbody.py:154-181 and window.get_compensation (window.py:65-80)."""
    pm = ParticleMesh(BoxSize=[100.0, 80.0, 120.0], Nmesh=[8, 6, 10], dtype=dtype)
    rs = numpy.random.RandomState(5)
    ck = pm.create(type='real', value=rs.normal(size=(8, 6, 10))).r2c()
    v = numpy.asarray(ck).astype('c16')
    k = [x.cpu().numpy().astype('f8') for x in ck.x]
    # recompute k in f8 from the indices: the reference casts them to the pm dtype
    k = [2 * numpy.pi / L * (numpy.where(i.cpu().numpy() >= N // 2, i.cpu().numpy() - N, i.cpu().numpy()))
         for i, L, N in zip(ck.i, pm.BoxSize, pm.Nmesh)]
    k2 = sum(ki ** 2.0 for ki in k)
    k2[k2 == 0] = 1.0
    for d in range(3):
        assert rel_l2(ck.apply(Transfer.dx1(d)), 1j * k[d] / k2 * v) < tol
        C = pm.BoxSize[d] / pm.Nmesh[d]
        w = k[d] * C
        kf = 1.0 / C * 1 / 6.0 * (8 * numpy.sin(w) - numpy.sin(2 * w))
        assert rel_l2(ck.apply(Transfer.force(d)), 1j * kf / k2 * v) < tol
    assert rel_l2(ck.apply(Transfer.potential()), -1. / k2 * v) < tol
    kk = sum(ki ** 2.0 for ki in k)
    assert rel_l2(ck.apply(Transfer.lowpass(3.0)), numpy.exp(-0.5 * kk * 3.0 ** 2) * v) < tol
    comp = window.TSC.get_compensation()
    wlist = [ki * L / N for ki, L, N in zip(k, pm.BoxSize, pm.Nmesh)]
    assert rel_l2(ck.apply(Transfer.compensation('tsc'), kind='circular'), comp(wlist, v)) < tol
    # in place
    c2 = ck.copy()
    r = c2.apply(Transfer.potential(), out=Ellipsis)
    assert r is c2 and rel_l2(c2, -1. / k2 * v) < tol


def test_grid(be):                            # test_pm.py:828-847
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4, 4], dtype='f8')
    grid = pm.generate_uniform_particle_grid(shift=0.5)
    assert grid.shape[0] == pm.Nmesh.prod()
    real = pm.paint(grid)
    assert_array_equal(numpy.asarray(real), 1.0)
    grid, id = pm.generate_uniform_particle_grid(shift=0.5, return_id=True)
    allid = id.cpu().numpy()
    assert len(numpy.unique(allid)) == len(allid) and allid.max() == len(allid) - 1 and allid.min() == 0


def test_grid_shifted(be):                    # test_pm.py:850-867
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4, 4], dtype='f8')
    grid = pm.generate_uniform_particle_grid(shift=0.5)
    grid = grid + 4.0
    layout = pm.decompose(grid)
    real = pm.paint(grid, layout=layout)
    assert_allclose(numpy.asarray(real), 1.0)
    grid = grid - 6.1
    layout = pm.decompose(grid)
    real = pm.paint(grid, layout=layout)
    assert_allclose(numpy.asarray(real), 1.0)


def test_field_arithmetic_stays_a_field(be):  # Field.__array_ufunc__, pm.py:169-208
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4, 4], dtype='f8')
    a = pm.create('real', value=numpy.arange(64.).reshape(4, 4, 4))
    b = a * 2 + 1
    assert isinstance(b, RealField)
    assert_array_equal(numpy.asarray(b), numpy.arange(64.).reshape(4, 4, 4) * 2 + 1)
    a[...] *= 0.5                              # examples/nbody.py:207
    assert_array_equal(numpy.asarray(a), numpy.arange(64.).reshape(4, 4, 4) * 0.5)
    assert abs(a.csum() - 0.5 * 63 * 64 / 2) < 1e-9
    assert abs(a.cmean() - 0.5 * 63 / 2) < 1e-9
    assert abs(a.cnorm() - (numpy.asarray(a) ** 2).sum()) < 1e-9
    c = a.r2c()
    full = numpy.fft.fftn(numpy.asarray(a)) / 64
    assert abs(c.cnorm() - (abs(full) ** 2).sum()) < 1e-9           # test_pm.py:681-700
    assert abs(c.cdot(c).real - (abs(full) ** 2).sum()) < 1e-9


@pytest.mark.parametrize('name', ['cic', 'tsc', 'pcs', 'nnb'])
def test_golden_cycle16(be, golden, name):
    """paint -> r2c -> transfer -> c2r -> readout against the fixture made with the
    compiled reference kernels + numpy.fft."""
    g = golden['cycle16']
    N, L = int(g['N'][0]), float(g['L'][0])
    pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler=name)
    pos = torch.from_numpy(g['pos']).to(be.device)
    rho = pm.paint(pos)
    want = g['%s/paint' % name]
    assert abs(numpy.asarray(rho) - want).max() <= 1e-12 * max(1.0, abs(want).max())
    rhok = rho.r2c()
    assert rel_l2(rhok, g['%s/r2c' % name]) < 1e-13
    for tname, T in (('dx1_0', Transfer.dx1(0)), ('force_2', Transfer.force(2)),
                     ('pot', Transfer.potential())):
        back = rhok.apply(T).c2r()
        assert rel_l2(back, g['%s/%s/c2r' % (name, tname)]) < 1e-11
        out = back.readout(pos)
        assert rel_l2(out.cpu().numpy(), g['%s/%s/readout' % (name, tname)]) < 1e-11
        out0 = back.readout(pos, gradient=0)
        assert rel_l2(out0.cpu().numpy(), g['%s/%s/readout_g0' % (name, tname)]) < 1e-11
    # the in-place chain the benchmark uses gives the same numbers
    f = pm.paint(pos).r2c(out=Ellipsis).apply(Transfer.dx1(0), out=Ellipsis).c2r(out=Ellipsis).readout(pos)
    assert rel_l2(f.cpu().numpy(), g['%s/dx1_0/readout' % name]) < 1e-11


def test_vjp_compositions(be):                # test_gradient.py: paint/readout adjointness
    pm = ParticleMesh(BoxSize=4.0, Nmesh=[4, 4, 4], dtype='f8')
    rs = numpy.random.RandomState(9)
    pos = rs.uniform(0, 4, size=(50, 3))
    mass = rs.uniform(0.5, 1.5, size=50)
    field = pm.create('real', value=rs.normal(size=(4, 4, 4)))
    # <paint(pos, mass), field> == <mass, readout(field, pos)>
    lhs = pm.paint(pos, mass=mass).cdot(field)
    rhs = (mass * field.readout(pos)).sum()
    assert abs(lhs - rhs) < 1e-12 * abs(rhs)
    out_pos, out_mass = pm.paint_vjp(field, pos, mass=mass)
    assert_allclose(out_mass, field.readout(pos))
    for d in range(3):
        assert_allclose(out_pos[:, d], field.readout(pos, gradient=d) * mass)
    out_self, out_pos2 = field.readout_vjp(pos, v=mass)
    assert_allclose(numpy.asarray(out_self), numpy.asarray(pm.paint(pos, mass=mass)))


@pytest.mark.parametrize('dtype,tol', [('f8', 1e-13), ('f4', 5e-6)])
@pytest.mark.parametrize('Nmesh', [[64, 64, 64], [128, 64, 20], [64, 256, 34], [64, 128, 256], [64, 64, 1024],
                                   [192, 64, 384], [64, 384, 384], [64, 192, 128], [64, 320, 640]])
def test_fft_column_path_vs_numpy(be, dtype, tol, Nmesh):
    """The hybrid 3-d transform (rocFFT along the contiguous axis + LDS-resident column FFTs,
    csrc/pmx_colfft.hip) obeys the same contract and equals the all-rocFFT path."""
    from pmesh_amd import fft as _fft
    rs = numpy.random.RandomState(21)
    data = rs.normal(size=Nmesh).astype(dtype)
    ref = numpy.fft.rfftn(data.astype('f8')) / numpy.prod(Nmesh)
    res = {}
    for mode in ('auto', 'never'):
        _fft.COLFFT = mode
        try:
            pm = ParticleMesh(BoxSize=1.0, Nmesh=Nmesh, dtype=dtype)
            real = pm.create(type='real', value=data)
            ck = real.r2c()
            assert rel_l2(ck, ref) < tol
            assert_array_equal(numpy.asarray(real), data)
            assert rel_l2(ck.c2r(), data) < 4 * tol
            ck2 = real.r2c(out=Ellipsis)
            assert rel_l2(ck2, ref) < tol
            assert rel_l2(ck2.c2r(out=Ellipsis), data) < 4 * tol
            res[mode] = numpy.asarray(ck)
        finally:
            _fft.COLFFT = 'auto'
    assert rel_l2(res['auto'], res['never']) < 2 * tol


@pytest.mark.parametrize('Nmesh', [[8, 6, 10], [16, 16, 16], [64, 64, 40], [64, 64, 128], [12, 10], [64, 192, 384]])
def test_c2r_of_a_non_hermitian_spectrum(be, Nmesh):
    """c2r == irfftn * prod(N) also for spectra that are not exactly Hermitian (what
    i k_d / k^2 leaves on the Nyquist planes): like FFTW's c2r behind PFFT, numpy ignores the
    imaginary parts of the self-conjugate modes of the last axis; rocFFT and the LDS kernels
    must agree with that, in place and out of place."""
    pm = ParticleMesh(BoxSize=1.0, Nmesh=Nmesh, dtype='f8')
    rs = numpy.random.RandomState(12)
    ck = pm.create(type='complex')
    shape = tuple(ck.shape)
    val = rs.normal(size=shape) + 1j * rs.normal(size=shape)
    ck[...] = val
    want = numpy.fft.irfftn(val, s=Nmesh, axes=list(range(len(Nmesh)))) * numpy.prod(Nmesh)
    assert rel_l2(ck.c2r(), want) < 1e-13
    assert rel_l2(ck.c2r(out=Ellipsis), want) < 1e-13


@pytest.mark.parametrize('dtype,tol', [('f8', 1e-12), ('f4', 2e-5)])
def test_cycle_on_the_padded_plane_layout(be, oracle, dtype, tol):
    """One rank, 3-d, lengths of the LDS FFT kernels: complex rows are padded to 128 bytes and
    the plane stride by one more line (fft.Partition).  Every consumer must be stride
    agnostic: paint, r2c, apply (stand-alone and fused), c2r, readout, reductions — the whole
    cycle against the oracle, in place and out of place, and the same numbers as the dense
    layout."""
    from pmesh_amd import fft as _fft
    from oracle import oracle as O
    Nmesh, L = [64, 64, 128], 10.0
    rs = numpy.random.RandomState(3)
    pos = rs.uniform(0, L, size=(4000, 3))
    res = {}
    for pad in (True, False):
        _fft.PLANE_PAD = pad
        try:
            pm = ParticleMesh(BoxSize=L, Nmesh=Nmesh, dtype=dtype, resampler='tsc')
            part = pm.plans['forwardT'].partition
            assert (part.plane_c is not None) == pad
            if pad:
                assert part.plane_c > Nmesh[1] * part.pitch_c and part.o_strides[0] == part.plane_c
            rho = pm.paint(pos)
            assert abs(rho.csum() - len(pos)) < 1e-4 * len(pos)
            ck = rho.r2c()                                   # out of place
            ref = numpy.fft.rfftn(numpy.asarray(rho).astype('f8')) / numpy.prod(Nmesh)
            assert rel_l2(ck, ref) < tol
            T = Transfer.dx1(1)
            a = numpy.asarray(ck.apply(T).c2r())             # stand-alone transfer kernel
            b = numpy.asarray(ck.c2r(transfer=T))            # fused
            assert rel_l2(b, a) < 10 * tol
            f = rho.r2c(out=Ellipsis).c2r(out=Ellipsis, transfer=T).readout(pos)   # all in place
            res[pad] = (numpy.asarray(ck), a, numpy.asarray(f))
        finally:
            _fft.PLANE_PAD = True
    t = O.make_transfer(laplace_pow=-1, grad_dir=1, grad_kind=0)
    aff = O.Affine(3, scale=[n / L for n in Nmesh], period=Nmesh)
    real = numpy.zeros(Nmesh)
    O.Window('tunedtsc').paint(real, pos, transform=aff)
    ck = O.apply_transfer(t, O.r2c(real), (0, 0, 0), Nmesh, (L, L, L))
    want = O.Window('tunedtsc').readout(O.c2r(ck, Nmesh), pos, transform=aff)
    for pad in (True, False):
        assert rel_l2(res[pad][2], want) < 50 * tol
    for x, y in zip(res[True], res[False]):
        assert rel_l2(x, y) < 10 * tol


@pytest.mark.parametrize('dtype,tol', [('f8', 1e-13), ('f4', 5e-6)])
def test_c2r_with_fused_transfer(be, dtype, tol):
    """c2r(transfer=T) == apply(T).c2r(): fused into the first column pass where possible,
    composed otherwise; `self` is preserved unless the transform is in place."""
    Nmesh = [64, 64, 40]
    pm = ParticleMesh(BoxSize=[100.0, 80.0, 120.0], Nmesh=Nmesh, dtype=dtype)
    rs = numpy.random.RandomState(8)
    ck = pm.create(type='real', value=rs.normal(size=Nmesh).astype(dtype)).r2c()
    before = numpy.asarray(ck).copy()
    for T in (Transfer.dx1(0), Transfer.dx1(2), Transfer.potential(), Transfer(amplitude=2.5),
              Transfer.force(0), Transfer.force(1), Transfer.force(2), Transfer.lowpass(4.0)):
        want = numpy.asarray(ck.apply(T).c2r())
        got = numpy.asarray(ck.c2r(transfer=T))
        assert rel_l2(got, want) < 10 * tol
        assert_array_equal(numpy.asarray(ck), before)
        c2 = ck.copy()
        got2 = numpy.asarray(c2.c2r(out=Ellipsis, transfer=T))
        assert rel_l2(got2, want) < 10 * tol


def test_real_resample(be):                   # the case of test_pm.py:458-470
    fine = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8], dtype='f8')
    coarse = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4], dtype='f8')
    checker = coarse.create(type='real')
    checker.apply(lambda i, v: (i[0] % 2) * (i[1] % 2), kind='index', out=Ellipsis)
    total = checker.csum()
    for resampler in ['nearest', 'cic', 'tsc', 'cubic']:
        up = fine.upsample(checker, resampler=resampler, keep_mean=False)
        assert_almost_equal(total, up.csum())
        assert_almost_equal(total, coarse.downsample(up, resampler=resampler).csum())
    reall, pmh = checker, fine
    # nearest up then down is the identity on the coarse mesh
    realh = pmh.upsample(reall, resampler='nearest', keep_mean=True)
    assert_allclose(numpy.asarray(realh)[::2, ::2], numpy.asarray(reall))


def test_transpose(be):                       # test_pm.py:754-776
    pm = ParticleMesh(BoxSize=[8.0, 16.0, 32.0], Nmesh=[4, 6, 8], dtype='f8')
    rs = numpy.random.RandomState(1234)
    comp1 = pm.create('real', value=rs.normal(size=(4, 6, 8)))
    comp1t = comp1.ctranspose([0, 1, 2])
    assert_array_equal(comp1t.Nmesh, comp1.Nmesh)
    assert_array_equal(comp1t.BoxSize, comp1.BoxSize)
    assert_allclose(comp1t.cnorm(), comp1.cnorm())
    comp1t = comp1.ctranspose([1, 2, 0])
    assert_array_equal(comp1t.Nmesh, comp1.Nmesh[[1, 2, 0]])
    assert_array_equal(comp1t.BoxSize, comp1.BoxSize[[1, 2, 0]])
    assert_allclose(numpy.asarray(comp1t), numpy.asarray(comp1).transpose(1, 2, 0))
    comp1ttt = comp1t.ctranspose([1, 2, 0]).ctranspose([1, 2, 0])
    assert_allclose(numpy.asarray(comp1ttt), numpy.asarray(comp1))


# ---- C-order redistribution, Fourier resampling, collective item access, preview ----------

def test_sort(be):                            # test_pm.py:394-412
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 6], dtype='f8')
    real = RealField(pm)
    truth = numpy.arange(8 * 6)
    real[...] = truth.reshape(8, 6)[real.slices]
    unsorted = real.copy()
    with pytest.warns(DeprecationWarning):
        real.sort(out=Ellipsis)
    assert_array_equal(numpy.asarray(real).ravel(), truth)
    real.unravel(numpy.asarray(real))
    assert_array_equal(numpy.asarray(real), numpy.asarray(unsorted))
    cplx = ComplexField(pm)
    truth = numpy.arange(8 * 4)
    cplx[...] = truth.reshape(8, 4)[cplx.slices]
    cplx.ravel(out=Ellipsis)
    assert_array_equal(numpy.asarray(cplx).ravel(), truth)
    # ravel into a host array / a new device tensor; pm.unravel builds a field
    host = numpy.empty(8 * 4, dtype='c16')
    cplx.ravel(out=host)
    assert_array_equal(host, truth)
    flat = cplx.ravel()
    assert_array_equal(flat.cpu().numpy(), truth)
    again = pm.unravel(ComplexField, flat)
    assert_array_equal(numpy.asarray(again), numpy.asarray(cplx))


def _fill_by_csetitem(field, truth, zero=lambda ind: False, remap=lambda ind: ind):
    for ind in numpy.ndindex(*field.cshape):
        field.csetitem(ind, 0 if zero(ind) else truth[remap(ind)])


def test_fdownsample(be):                     # test_pm.py:416-454
    pm1 = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8], dtype='f8')
    pm2 = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4], dtype='f8')
    numpy.random.seed(3333)
    truth = numpy.fft.rfftn(numpy.random.normal(size=(8, 8)))
    complex1 = ComplexField(pm1)
    _fill_by_csetitem(complex1, truth)
    assert_almost_equal(numpy.asarray(complex1), numpy.asarray(complex1.c2r().r2c()))
    complex2 = ComplexField(pm2)
    _fill_by_csetitem(complex2, truth, zero=lambda ind: any(i == 2 for i in ind),
                      remap=lambda ind: tuple([i if i <= 2 else 8 - (4 - i) for i in ind]))
    tmpr = RealField(pm2)
    tmp = ComplexField(pm2)
    complex1.resample(tmp)
    assert_almost_equal(numpy.asarray(complex2), numpy.asarray(tmp), decimal=5)
    complex1.c2r().resample(tmp)
    assert_almost_equal(numpy.asarray(complex2), numpy.asarray(tmp), decimal=5)
    complex1.resample(tmpr)
    assert_almost_equal(numpy.asarray(tmpr.r2c()), numpy.asarray(tmp))
    complex1.c2r().resample(tmpr)
    assert_almost_equal(numpy.asarray(tmpr.r2c()), numpy.asarray(tmp))


def test_fupsample(be):                       # test_pm.py:493-537
    pm1 = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8], dtype='f8')
    pm2 = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4], dtype='f8')
    numpy.random.seed(3333)
    truth = numpy.fft.rfftn(numpy.random.normal(size=(8, 8)))
    complex1 = ComplexField(pm1)
    _fill_by_csetitem(complex1, truth, zero=lambda ind: any(i == 4 for i in ind) or any(2 <= i < 7 for i in ind))
    assert_almost_equal(numpy.asarray(complex1), numpy.asarray(complex1.c2r().r2c()))
    complex2 = ComplexField(pm2)
    _fill_by_csetitem(complex2, truth, zero=lambda ind: any(i == 2 for i in ind),
                      remap=lambda ind: tuple([i if i <= 2 else 8 - (4 - i) for i in ind]))
    tmpr = RealField(pm1)
    tmp = ComplexField(pm1)
    complex2.resample(tmp)
    assert_almost_equal(numpy.asarray(complex1), numpy.asarray(tmp), decimal=5)
    complex2.c2r().resample(tmp)
    assert_almost_equal(numpy.asarray(complex1), numpy.asarray(tmp), decimal=5)
    complex2.resample(tmpr)
    assert_almost_equal(numpy.asarray(tmpr.r2c()), numpy.asarray(tmp))
    complex2.c2r().resample(tmpr)
    assert_almost_equal(numpy.asarray(tmpr.r2c()), numpy.asarray(tmp))


def test_resample_3d_keeps_the_large_scales(be):
    """resampling white noise of a finer mesh down equals the white noise of the coarser mesh on
    the modes both have (the generator is scale invariant), Nyquist planes removed"""
    pm1 = ParticleMesh(BoxSize=8.0, Nmesh=[16, 16, 16], dtype='f8')
    pm0 = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8, 8], dtype='f8')
    c1 = pm1.generate_whitenoise(seed=8, unitary=True)
    down = ComplexField(pm0)
    c1.resample(down)
    c0 = numpy.asarray(pm0.generate_whitenoise(seed=8, unitary=True)).copy()
    c0[4, :, :] = 0
    c0[:, 4, :] = 0
    c0[:, :, 4] = 0
    assert_allclose(numpy.asarray(down), c0, rtol=0, atol=1e-14)


def test_ctol_cgetitem_csetitem(be):          # test_pm.py:553-630
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4], dtype='f8')
    value, local = ComplexField(pm)._ctol((3, 3))
    assert local is None
    for i in numpy.ndindex((4, 4)):
        real = RealField(pm)
        real[...] = 0
        v2 = real.csetitem(i, 100.)
        assert v2 == 100. and real.cgetitem(i) == v2
    once = {(0, 0), (0, 2), (2, 0), (2, 2)}           # self-conjugate modes keep the real part only
    twice = {(1, 0), (3, 0), (3, 2), (1, 2)}          # modes whose conjugate is stored too
    for i in numpy.ndindex((4, 3)):
        cplx = ComplexField(pm)
        cplx[...] = 0
        v2 = cplx.csetitem(i, 100. + 10j)
        cplx.c2r(out=Ellipsis).r2c(out=Ellipsis)
        v1 = cplx.cgetitem(i)
        total = complex(numpy.asarray(cplx).sum())
        if i in once:
            assert v2 == 100. and abs(total - 100.) < 1e-12
        elif i in twice:
            assert v2 == 100 + 10j and abs(total - 200.) < 1e-12
        else:
            assert v2 == 100. + 10j and abs(total - (100. + 10j)) < 1e-12
        assert abs(v1 - v2) < 1e-12
    for i in numpy.ndindex((4, 3, 2)):
        cplx = ComplexField(pm)
        cplx[...] = 0
        v2 = cplx.csetitem(i, 100.)
        cplx.c2r(out=Ellipsis).r2c(out=Ellipsis)
        v1 = cplx.cgetitem(i)
        if i[:2] in once and i[2] == 1:
            assert v2 == 0.
        else:
            assert v2 == 100.
        assert abs(v1 - v2) < 1e-12


def test_preview(be):                         # test_pm.py:780-814
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4, 4], dtype='f8')
    comp1 = pm.generate_whitenoise(1234, type='real')
    preview = comp1.preview(axes=(0, 1, 2))
    preview = comp1.preview(Nmesh=4, axes=(0, 1, 2))
    for ind1 in numpy.ndindex(*(list(comp1.cshape))):
        assert_allclose(preview[ind1], comp1.cgetitem(ind1))
    assert_allclose(comp1.preview(Nmesh=4, axes=(0, 1)), preview.sum(axis=2))
    assert_allclose(comp1.preview(Nmesh=4, axes=(1, 2)), preview.sum(axis=0))
    assert_allclose(comp1.preview(Nmesh=4, axes=(0, 2)), preview.sum(axis=1))
    assert_allclose(comp1.preview(Nmesh=4, axes=(2, 0)), preview.sum(axis=1).T)
    assert_allclose(comp1.preview(Nmesh=4, axes=(0,)), preview.sum(axis=(1, 2)))
    p8 = comp1.preview(Nmesh=8, axes=(0,))
    assert p8.shape == (8,)
    p2 = comp1.preview(Nmesh=2)
    assert p2.shape == (2, 2, 2) and abs(p2.mean() - preview.mean()) < 1e-12


def test_tile_order(be):
    """pm.tile_order: a permutation; rows that follow each other afterwards share a tile or
    sit in neighbouring ones; painting the reordered rows gives the same field"""
    pm = ParticleMesh(BoxSize=64.0, Nmesh=[32, 32, 64], dtype='f8')
    rs = numpy.random.RandomState(2)
    pos = rs.uniform(-10, 80, size=(5000, 3))
    o = pm.tile_order(pos)
    o = o.cpu().numpy()
    assert sorted(o.tolist()) == list(range(5000))
    cell = numpy.floor(pos[o] * (pm.Nmesh / pm.BoxSize)).astype('i8') % pm.Nmesh
    tile = cell // numpy.array([8, 16, 32])
    tid = (tile[:, 0] * 2 + tile[:, 1]) * 2 + tile[:, 2]
    assert (numpy.diff(tid) >= 0).all()                     # tile-major
    a = numpy.asarray(pm.paint(pos))
    b = numpy.asarray(pm.paint(pos[o]))
    assert_allclose(a, b, rtol=0, atol=1e-12 * abs(a).max())


@pytest.mark.gpu
@pytest.mark.parametrize('resampler', ['cic', 'tsc', 'pcs'])
def test_tile_order_from_the_bin_plan(resampler):
    """[r6] on the GPU the permutation is read off the bin plan of the positions (pmx_binplan_order): a permutation,
    tile-major in the plan's tiles (the tile of a particle's FIRST stencil cell), painting / reading the reordered rows
    gives the same numbers, and the plan that was built for the order serves the paint that follows"""
    from pmesh_amd import backend, window
    backend.reset()
    be = backend.get()
    old = window.BINNED
    try:
        window.BINNED = 'always'
        window.clear_bin_cache()
        pm = ParticleMesh(BoxSize=64.0, Nmesh=[32, 64, 64], dtype='f8', resampler=resampler)
        rs = numpy.random.RandomState(4)
        pos_h = rs.uniform(-10, 80, size=(60000, 3))
        pos = torch.from_numpy(pos_h).to(be.device)
        o = pm.tile_order(pos)
        assert o.dtype == torch.int64 and o.device == pos.device
        oh = o.cpu().numpy()
        assert numpy.array_equal(numpy.sort(oh), numpy.arange(len(pos_h)))
        S = pm.resampler.support
        first = numpy.floor(pos_h[oh] * (pm.Nmesh / pm.BoxSize) + (0.5 if S % 2 else 0.0)).astype('i8') - (S - 1) // 2
        tile = (first % pm.Nmesh) // numpy.array([8, 16, 32])
        tid = (tile[:, 0] * 4 + tile[:, 1]) * 2 + tile[:, 2]
        assert (numpy.diff(tid) >= 0).all()
        builds = window.bin_cache()
        a = pm.paint(pos)
        b = pm.paint(pos[o].contiguous())
        assert_allclose(numpy.asarray(a), numpy.asarray(b), rtol=0, atol=1e-12 * abs(numpy.asarray(a)).max())
        ra, rb = a.readout(pos), a.readout(pos[o].contiguous())
        assert_allclose(numpy.asarray(ra.cpu())[oh], numpy.asarray(rb.cpu()), rtol=0, atol=1e-12)
    finally:
        window.BINNED = old
        window.clear_bin_cache()
        backend.reset()


# ---- complex-to-complex meshes (ParticleMesh(dtype='c16' | 'c8')) ----------------------------

def test_c2c(be):                             # test_pm.py:196-226
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8], dtype='complex128')
    Npar = 100
    pos = 1.0 * (numpy.arange(Npar * len(pm.Nmesh))).reshape(-1, len(pm.Nmesh)) * (7, 7)
    pos %= (pm.Nmesh + 1)
    layout = pm.decompose(pos)
    npos = layout.exchange(pos)
    real = pm.paint(npos)
    cplx = real.r2c()
    real2 = cplx.c2r()
    assert numpy.iscomplexobj(numpy.asarray(real))
    assert numpy.iscomplexobj(numpy.asarray(real2))
    assert numpy.iscomplexobj(numpy.asarray(cplx))
    assert_array_equal(cplx.cshape, pm.Nmesh)
    assert_array_equal(real2.cshape, pm.Nmesh)
    assert_array_equal(real.cshape, pm.Nmesh)
    assert not cplx.compressed
    real.readout(npos)
    assert_almost_equal(numpy.asarray(real), numpy.asarray(real2), decimal=7)


@pytest.mark.parametrize('Nmesh,dtype,tol', [([8, 6, 10], 'c16', 1e-13), ([5, 7, 9], 'c16', 1e-13), ([12, 10], 'c16', 1e-13),
                                             ([16], 'c16', 1e-13), ([64, 64, 128], 'c16', 1e-13), ([8, 8, 8], 'c8', 5e-6)])
def test_c2c_vs_numpy(be, Nmesh, dtype, tol):
    """r2c == fftn / prod(N), c2r == ifftn * prod(N) on the full spectrum; complex values in
    configuration space are transformed as they are; in place and out of place"""
    pm = ParticleMesh(BoxSize=2.0, Nmesh=Nmesh, dtype=dtype)
    rs = numpy.random.RandomState(6)
    data = (rs.normal(size=Nmesh) + 1j * rs.normal(size=Nmesh)).astype(dtype)
    real = pm.create(type='real', value=data)
    ref = numpy.fft.fftn(data.astype('c16')) / numpy.prod(Nmesh)
    ck = real.r2c()
    assert tuple(ck.shape) == tuple(Nmesh)
    assert rel_l2(ck, ref) < tol
    assert_array_equal(numpy.asarray(real), data)
    assert rel_l2(ck.c2r(), data) < 4 * tol
    ck2 = real.r2c(out=Ellipsis)
    assert rel_l2(ck2, ref) < tol
    assert rel_l2(ck2.c2r(out=Ellipsis), data) < 4 * tol
    # Parseval on the uncompressed spectrum: every mode counts once
    assert abs(ck.cnorm() - (abs(ref) ** 2).sum()) < 100 * tol * (abs(ref) ** 2).sum()


def test_c2c_r2c_edges(be):                   # test_pm.py:816-826
    pm1 = ParticleMesh(BoxSize=8.0, Nmesh=[5, 7, 9], dtype='c16')
    pm2 = ParticleMesh(BoxSize=8.0, Nmesh=[5, 7, 9], dtype='f8')
    real1 = pm1.create(type='real')
    real2 = pm2.create(type='real')
    for d in range(3):
        assert_allclose(real1.x[d].cpu().numpy(), real2.x[d].cpu().numpy())


def test_c2c_paint_cycle(be, oracle):
    """paint writes the real part of a complex canvas (window.py:161-162); the cycle through the
    complex-to-complex transforms equals the real one"""
    N, L = 16, 10.0
    rs = numpy.random.RandomState(4)
    pos = rs.uniform(0, L, size=(500, 3))
    res = {}
    for dtype in ('f8', 'c16'):
        pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype=dtype, resampler='tsc')
        rho = pm.paint(pos)
        if dtype == 'c16':
            assert float(rho.value.imag.abs().max()) == 0
        back = rho.r2c().apply(Transfer.potential()).c2r()
        res[dtype] = (numpy.asarray(rho), numpy.asarray(back), numpy.asarray(back.readout(pos)))
    assert_allclose(res['c16'][0].real, res['f8'][0], rtol=0, atol=1e-14)
    assert_allclose(res['c16'][1].real, res['f8'][1], rtol=0, atol=1e-10 * abs(res['f8'][1]).max())
    assert abs(res['c16'][1].imag).max() < 1e-10 * abs(res['f8'][1]).max()
    assert_allclose(res['c16'][2], res['f8'][2], rtol=0, atol=1e-10 * abs(res['f8'][2]).max())


def test_default_process_mesh_and_plan_cache(be):
    """np=None follows pm.py:1317-1325 (3-d: pfft.split_size_2d, 2-d: a slab); a second ParticleMesh of
    the same mesh / communicator / dtype shares the plans of the living one (pm.py:1362-1404)."""
    from pmesh_amd.fft import split_size_2d
    assert split_size_2d(8) == (2, 4) and split_size_2d(4) == (2, 2) and split_size_2d(6) == (3, 2)
    assert split_size_2d(1) == (1, 1) and split_size_2d(7) == (1, 7)
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8, 8], dtype='f8')
    assert pm.np == [1]                                   # one rank: [1, 1] collapses to the slab [1]
    assert ParticleMesh(BoxSize=8.0, Nmesh=[8, 8], dtype='f8').np == [1]
    assert ParticleMesh(BoxSize=8.0, Nmesh=[8], dtype='f8').np == []
    twin = ParticleMesh(BoxSize=4.0, Nmesh=[8, 8, 8], dtype='f8', resampler='tsc')
    assert twin.plans is pm.plans and twin.procmesh is pm.procmesh
    assert twin.resampler is not pm.resampler and float(twin.BoxSize[0]) == 4.0
    other = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8, 8], dtype='f4')
    assert other.plans is not pm.plans
    # the shared plans serve both objects
    r = pm.create('real', value=1.0)
    t = twin.create('real', value=2.0)
    assert abs(float(r.r2c().c2r().value.mean()) - 1.0) < 1e-12
    assert abs(float(t.r2c().c2r().value.mean()) - 2.0) < 1e-12


@pytest.mark.parametrize('Nmesh,dtype', [([64, 64, 128], 'f8'), ([128, 64, 128], 'f4'), ([192, 64, 128], 'f8')])
def test_deferred_last_pass_is_invisible(be, Nmesh, dtype):
    """fft.DEFER_LAST_PASS: on one rank r2c leaves its last (axis-0) pass for an in-place c2r that follows at once —
    one kernel then does forward pass, scale, transfer and inverse pass with the column in LDS — and whoever looks
    at the spectrum first (value, apply with a callable, an out-of-place c2r, a second r2c into the buffer, paint)
    makes the deferred pass happen.  Same bits as the eager transforms in every case."""
    from pmesh_amd import fft as F
    pm = ParticleMesh(BoxSize=[3.0, 2.0, 5.0], Nmesh=Nmesh, dtype=dtype)
    data = numpy.random.RandomState(11).normal(size=Nmesh).astype(dtype)
    T = Transfer.dx1(0)
    saved = F.DEFER_LAST_PASS

    def run(defer, scenario):
        F.DEFER_LAST_PASS = defer
        real = pm.create('real', value=data)
        ck = real.r2c(out=Ellipsis)
        if scenario == 'fused':
            return numpy.asarray(ck.c2r(out=Ellipsis, transfer=T))
        if scenario == 'plain':
            return numpy.asarray(ck.c2r(out=Ellipsis))
        if scenario == 'peek':
            spec = numpy.asarray(ck).copy()
            return spec, numpy.asarray(ck.c2r(out=Ellipsis, transfer=T))
        if scenario == 'callable':
            ck.apply(lambda k, v: v * 2.0, out=Ellipsis)
            return numpy.asarray(ck.c2r(out=Ellipsis))
        if scenario == 'outofplace':
            back = ck.c2r(transfer=T)
            return numpy.asarray(back), numpy.asarray(ck).copy()
        if scenario == 'abandon':
            # the spectrum is never looked at: the buffer is painted over / transformed again
            real2 = pm.create('real', base=ck._base, value=data * 2)
            return numpy.asarray(real2.r2c(out=Ellipsis))
        if scenario == 'heldview':
            # a view of the caller's own complex field taken BEFORE r2c(out=field) holds the finished spectrum
            # afterwards, as a plain array attribute would (pm.py:234-242): nothing stays deferred on such a field
            target = pm.create('complex')
            held = target.value
            real3 = pm.create('real', value=data)
            real3.r2c(out=target)
            return held.cpu().numpy().copy(), numpy.asarray(target).copy()
        if scenario == 'dropped':
            # a field dropped with its last pass still deferred is freed without the cyclic collector
            import gc
            import weakref
            gc.disable()
            try:
                real4 = pm.create('real', value=data)
                ck4 = real4.r2c(out=Ellipsis)
                w = weakref.ref(ck4._base.storage)
                del real4, ck4
                alive = w() is not None
            finally:
                gc.enable()
            return numpy.array([alive])
    try:
        held, fresh = run(True, 'heldview')
        assert_array_equal(held, fresh)
        assert_array_equal(run(True, 'dropped'), [False])
        for scenario in ('fused', 'plain', 'peek', 'callable', 'outofplace', 'abandon', 'heldview'):
            a, b = run(False, scenario), run(True, scenario)
            if isinstance(a, tuple):
                for x, y in zip(a, b):
                    assert_array_equal(x, y, err_msg=scenario)
            else:
                assert_array_equal(a, b, err_msg=scenario)
        ref = numpy.fft.irfftn(numpy.fft.rfftn(data.astype("f8")), s=Nmesh, axes=(0, 1, 2))
        tol = 1e-12 if dtype == 'f8' else 5e-5
        assert abs(run(True, 'plain') - ref).max() < tol * abs(ref).max()
    finally:
        F.DEFER_LAST_PASS = saved


def test_staged_host_arrays(be):
    """ParticleMesh.stage: a numpy array registered once is uploaded once; paint / readout / decompose take
    the handle wherever the array would go, return numpy like for the array itself, and share one device
    copy (one bin plan, one layout memo); refresh() after an update in place"""
    from pmesh_amd._arrays import Staged
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8, 8], dtype='f8', resampler='cic')
    rs = numpy.random.RandomState(3)
    X = rs.uniform(0, 8, size=(500, 3))
    h = pm.stage(X)
    assert isinstance(h, Staged) and pm.stage(h) is h and len(h) == 500 and h.shape == (500, 3)
    a = pm.paint(X)
    b = pm.paint(h)
    assert_allclose(numpy.asarray(a), numpy.asarray(b), rtol=0, atol=1e-13)     # (atomics: the order of the adds may differ)
    layout = pm.decompose(h)
    c = pm.paint(h, layout=layout)
    assert_allclose(numpy.asarray(c), numpy.asarray(a), rtol=0, atol=1e-13)
    f = a.readout(h)
    assert isinstance(f, numpy.ndarray)
    assert_array_equal(f, a.readout(X))
    t0 = h.tensor
    X += 0.25
    assert_array_equal(a.readout(h.refresh()), a.readout(X))
    assert h.tensor is t0                                   # the same device storage, a new version
    assert_allclose(numpy.asarray(pm.paint(h)), numpy.asarray(pm.paint(X)), rtol=0, atol=1e-13)


def test_pack_arrays():
    from pmesh_amd.domain import pack_arrays          # domain.py:59-80
    a = numpy.arange(12.0).reshape(4, 3)
    b = numpy.arange(4, dtype='i4')
    s = pack_arrays([a, b])
    assert s.shape == (4,) and s.dtype[0].shape == (3,) and s.dtype[1] == numpy.dtype('i4')
    assert_array_equal(s[s.dtype.names[0]], a)
    assert_array_equal(s[s.dtype.names[1]], b)
    with pytest.raises(ValueError):
        pack_arrays([a, b[:3]])


def test_reshape(be):                         # test_pm.py:372-389
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8, 8], dtype='f8', np=[1, 1])
    pm2d = pm.reshape(Nmesh=[8, 8])
    assert pm2d.ndim == 2
    with pytest.raises(ValueError):
        pm.reshape(Nmesh=[8])                  # a 2-d process mesh cannot decompose a 1-d mesh
    pm4d = pm.reshape(Nmesh=[8, 8, 8, 8], BoxSize=8.0)
    assert pm4d.ndim == 4 and tuple(pm4d.Nmesh) == (8, 8, 8, 8)
    with pytest.raises(ValueError):
        pm.reshape(Nmesh=[8, 8, 8, 8])         # BoxSize of another dimension
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8, 8], dtype='f8', np=[1, ])
    assert pm.reshape(Nmesh=[8, 8]).ndim == 2


def _host(x):
    """a field value / result as a numpy array whichever backend made it"""
    return x.cpu().numpy() if hasattr(x, 'cpu') else numpy.asarray(x)


@pytest.mark.parametrize('Nmesh,dtype,tol', [([8, 6, 4, 8], 'f8', 1e-13), ([4, 6, 2, 4, 6], 'f8', 1e-13), ([8, 6, 4, 8], 'f4', 5e-6),
                                             ([4, 6, 5, 4], 'c16', 1e-13)])
def test_meshes_of_more_than_three_dimensions(be, oracle, Nmesh, dtype, tol):
    """The reference's ParticleMesh takes any number of dimensions (PFFT transforms them, the generic window kernels
    loop over them: _window_imp.h:50-60; pm.reshape(Nmesh=[8, 8, 8, 8]), test_pm.py:381).  Here: transforms against
    numpy.fft under the contract of pm.py:692, paint / readout of every kind of window (there is no tuned kernel
    beyond three dimensions in the reference either) against the oracle, which tests/test_oracle.py pins to the
    compiled reference; apply with coordinates; decompose on one rank."""
    nd = len(Nmesh)
    pm = ParticleMesh(BoxSize=[8.0, 6.0, 4.0, 8.0, 5.0][:nd], Nmesh=Nmesh, dtype=dtype)
    rs = numpy.random.RandomState(11)
    real = pm.create('real')
    v = rs.normal(size=real.shape)
    if dtype == 'c16':
        v = v + 1j * rs.normal(size=real.shape)
    real[...] = v
    cplx = real.r2c()
    want = (numpy.fft.fftn(v) if dtype == 'c16' else numpy.fft.rfftn(v)) / v.size
    assert_allclose(_host(cplx.value), want, rtol=0, atol=tol * abs(want).max() * 10)
    back = cplx.c2r()
    assert_allclose(_host(back.value), v, rtol=0, atol=tol * abs(v).max() * 10)
    real.r2c(out=Ellipsis).c2r(out=Ellipsis)           # in place
    assert_allclose(_host(real.value), v, rtol=0, atol=tol * abs(v).max() * 10)
    if dtype == 'c16':
        return
    pos = rs.uniform(-2.0, 10.0, size=(400, nd))
    mass = rs.uniform(0.5, 1.5, size=400)
    aff = oracle.Affine(nd, scale=pm.Nmesh / pm.BoxSize, period=pm.Nmesh)
    names = {'nnb': 'tunednnb', 'cic': 'tunedcic', 'tsc': 'tunedtsc', 'pcs': 'tunedpcs'}
    for res in ('nnb', 'cic', 'tsc', 'pcs', 'linear', 'quadratic', 'cubic'):
        f = pm.paint(pos, mass=mass, resampler=res)
        a = numpy.zeros(tuple(pm.Nmesh), dtype=dtype)
        oracle.Window(names.get(res, res)).paint(a, pos, mass=mass, transform=aff)
        assert_allclose(_host(f.value), a, rtol=0, atol=(1e-12 if dtype == 'f8' else 2e-6) * max(1.0, abs(a).max()))
        assert abs(f.csum() - mass.sum()) < (1e-9 if dtype == 'f8' else 1e-3) * mass.sum()
        for gradient in (None, nd - 1):
            r = f.readout(pos, resampler=res, gradient=gradient)
            rr = oracle.Window(names.get(res, res)).readout(_host(f.value), pos, diffdir=gradient, transform=aff)
            assert_allclose(_host(r), rr, rtol=0, atol=(1e-12 if dtype == 'f8' else 1e-5) * max(1.0, abs(rr).max()))
    layout = pm.decompose(pos)
    f2 = pm.paint(pos, mass=mass, layout=layout)
    assert_allclose(_host(f2.value), _host(pm.paint(pos, mass=mass).value), rtol=0, atol=1e-12 if dtype == 'f8' else 2e-6)
    smooth = real.r2c().apply(lambda k, v: v * numpy.exp(-0.5 * sum(ki ** 2 for ki in k)))
    kk = sum(numpy.meshgrid(*[2 * numpy.pi * numpy.fft.fftfreq(n, d=L / n) if i < nd - 1 else
                              2 * numpy.pi * numpy.fft.rfftfreq(n, d=L / n)
                              for i, (n, L) in enumerate(zip(pm.Nmesh, pm.BoxSize))], indexing='ij', sparse=True)[i] ** 2 for i in range(nd))
    assert_allclose(_host(smooth.value), numpy.fft.rfftn(v) / v.size * numpy.exp(-0.5 * kk), rtol=0,
                    atol=tol * 10 * abs(want).max())
