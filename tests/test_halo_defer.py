"""The halo merge of a tile-binned paint left to the forward transform (include/pmesh_amd.h:
pmx_paint_binned_defer / pmx_halo_merge / pmx_rowfft_halo; pmesh_amd/pm.py: HALO_DEFER).

`pm.paint(pos)` on one rank returns a field whose tile halos are still staged in the bin plan; `field.r2c()` adds
them inside its row pass, every other reader runs the merge first.  Whatever happens to the field in between, its
values are those of the eager paint (reference: pm.py:1795-1869 followed by pm.py:655-694; the order of the
additions into a cell is the only difference, as between two runs of the eager paint itself):

  * r2c of a deferred field == r2c of an eager one, every window x canvas type, cubic and non-cubic meshes, with and
    without the Infinity-Cache blocking of the row pass (which hands the gather a plane offset);
  * bit for bit on dyadic inputs (exact partial sums: pins which staged cell lands on which mesh cell);
  * `.value`, readout, a second paint with the same plan, a dropped field, a caller's `out` field, hold=True.
"""
import gc

import numpy
import pytest
import torch
from numpy.testing import assert_allclose, assert_array_equal

import pmesh_amd.pm as pmod
from pmesh_amd import fft as _fft
from pmesh_amd import window
from pmesh_amd.pm import ParticleMesh, RealField

pytestmark = pytest.mark.gpu


@pytest.fixture
def hip():
    from pmesh_amd import backend
    backend.reset()
    b = backend.get()
    old = (window.BINNED, pmod.HALO_DEFER, _fft.L3_BLOCK_BYTES)
    window.BINNED = 'always'
    yield b
    window.BINNED, pmod.HALO_DEFER, _fft.L3_BLOCK_BYTES = old
    window.clear_bin_cache()
    backend.reset()


def particles(pm, n, seed, dyadic=False, dev='cuda'):
    g = torch.Generator(device='cpu').manual_seed(seed)
    box = torch.as_tensor(numpy.asarray(pm.BoxSize, dtype='f8'))
    if dyadic:
        # positions on a grid of 1/8 cell, masses multiples of 2^-6: every partial sum is exact in double
        cells = torch.as_tensor(numpy.asarray(pm.Nmesh, dtype='f8'))
        pos = torch.floor(torch.rand(n, 3, generator=g, dtype=torch.float64) * cells * 8) / 8 * (box / cells)
        mass = torch.floor(torch.rand(n, generator=g, dtype=torch.float64) * 64 + 1) / 64
    else:
        pos = torch.rand(n, 3, generator=g, dtype=torch.float64) * box
        mass = torch.rand(n, generator=g, dtype=torch.float64) + 0.5
    return pos.to(dev), mass.to(dev)


def owes(field):
    return getattr(field._base.storage, '_pmx_halo', None) is not None


MESHES = [(128, 128, 128), (64, 192, 256), (128, 64, 512), (64, 64, 1024), (64, 64, 2048)]


@pytest.mark.parametrize('name', ['cic', 'tsc', 'pcs'])
@pytest.mark.parametrize('dtype', ['f8', 'f4'])
@pytest.mark.parametrize('nmesh', MESHES)
@pytest.mark.parametrize('blocked', [False, True])
def test_r2c_of_a_deferred_paint_equals_the_eager_one(hip, name, dtype, nmesh, blocked):
    if nmesh[2] == 2048 and dtype == 'f8':
        nmesh = (64, 64, 256)       # (rows of 2048 doubles keep the merge kernel: test_rows_the_gather_is_not_built_for)
    pm = ParticleMesh(Nmesh=nmesh, BoxSize=[100.0, 75.0, 130.0], dtype=dtype, resampler=name)
    n = int(numpy.prod(nmesh)) // 2
    pos, mass = particles(pm, n, 11)
    # blocks of a few planes: the gather is launched per block with the block's first plane
    _fft.L3_BLOCK_BYTES = (5 * nmesh[1] * (nmesh[2] + 16) * (8 if dtype == 'f8' else 4)) if blocked else 0
    pmod.HALO_DEFER = 'never'
    eager = pm.paint(pos, mass=mass)
    assert not owes(eager)
    ek = eager.r2c(out=Ellipsis).value.clone()
    pmod.HALO_DEFER = 'fresh'
    lazy = pm.paint(pos, mass=mass)
    assert owes(lazy), 'the paint did not leave its halo merge to the transform'
    lk = lazy.r2c(out=Ellipsis)
    assert not owes(lazy)
    lk = lk.value
    scale = float(ek.abs().max())
    tol = 1e-13 if dtype == 'f8' else 2e-6
    assert float((lk - ek).abs().max()) <= tol * scale
    # and the plan is free again: the next paint works and equals the eager one
    again = pm.paint(pos, mass=mass)
    assert_allclose(again.value.cpu().numpy(), _eager_value(pm, pos, mass), rtol=0, atol=tol * float(mass.max()) * 8)


@pytest.mark.parametrize('nmesh,dtype', [((64, 64, 2048), 'f8'), ((64, 64, 384), 'f8'), ((64, 128, 640), 'f4'),
                                         ((64, 64, 768), 'f4'), ((64, 64, 64), 'f8')])
def test_rows_the_gather_is_not_built_for(hip, nmesh, dtype):
    """pmx_rowfft_halo_supported: rows of 2048 doubles, the 3 * 2^k / 5 * 2^k lengths (measured: the gather is a loss
    there) and meshes the LDS row pass does not take at all — the paint merges its halos itself, nothing is deferred"""
    pm = ParticleMesh(Nmesh=nmesh, BoxSize=1.0, dtype=dtype, resampler='tsc')
    pos, mass = particles(pm, int(numpy.prod(nmesh)) // 2, 13)
    pmod.HALO_DEFER = 'fresh'
    f = pm.paint(pos, mass=mass)
    assert not owes(f)
    assert abs(float(f.csum()) - float(mass.sum())) <= (1e-11 if dtype == 'f8' else 2e-5) * float(mass.sum())
    k = f.r2c(out=Ellipsis)
    assert abs(float(k.value.flatten()[0].real) * 1.0 - float(mass.sum()) / float(numpy.prod(nmesh))) <= \
        (1e-11 if dtype == 'f8' else 2e-5) * float(mass.sum()) / float(numpy.prod(nmesh))


def _eager_value(pm, pos, mass):
    old = pmod.HALO_DEFER
    pmod.HALO_DEFER = 'never'
    try:
        return pm.paint(pos, mass=mass).value.cpu().numpy()
    finally:
        pmod.HALO_DEFER = old


@pytest.mark.parametrize('name', ['cic', 'tsc', 'pcs'])
@pytest.mark.parametrize('nmesh', MESHES[:4])
def test_dyadic_inputs_bit_for_bit(hip, name, nmesh):
    """exact partial sums: the spectrum of the deferred field equals the eager one bit for bit only if every staged
    halo cell was added to exactly the mesh cell the merge kernel adds it to (the row pass then sees equal rows)"""
    pm = ParticleMesh(Nmesh=nmesh, BoxSize=[float(x) for x in nmesh], dtype='f8', resampler=name)    # (scale 1: exact cells)
    n = int(numpy.prod(nmesh)) // 3
    pos, mass = particles(pm, n, 5, dyadic=True)
    _fft.L3_BLOCK_BYTES = 0
    pmod.HALO_DEFER = 'never'
    eager = pm.paint(pos, mass=mass)
    ev = eager.value.clone()
    ek = eager.r2c(out=Ellipsis).value.clone()
    pmod.HALO_DEFER = 'fresh'
    lazy = pm.paint(pos, mass=mass)
    assert owes(lazy)
    lk = lazy.r2c(out=Ellipsis).value
    if name == 'cic':       # CIC weights of 1/8-cell offsets are dyadic: sums exact, spectra identical
        assert_array_equal(lk.cpu().numpy(), ek.cpu().numpy())
    else:
        assert float((lk - ek).abs().max()) <= 1e-14 * float(ek.abs().max())
    # the merge kernel as the debt's other way out: .value
    lazy2 = pm.paint(pos, mass=mass)
    assert owes(lazy2)
    v = lazy2.value
    assert not owes(lazy2)
    if name == 'cic':
        assert_array_equal(v.cpu().numpy(), ev.cpu().numpy())
    else:
        assert_allclose(v.cpu().numpy(), ev.cpu().numpy(), rtol=0, atol=1e-13)


def test_every_other_reader_pays_the_debt_first(hip):
    pm = ParticleMesh(Nmesh=[128, 128, 128], BoxSize=1.0, dtype='f8', resampler='tsc')
    pos, mass = particles(pm, 1 << 20, 3)
    ref = _eager_value(pm, pos, mass)
    tol = dict(rtol=0, atol=1e-12)
    pmod.HALO_DEFER = 'fresh'
    # readout of the field
    f = pm.paint(pos, mass=mass)
    assert owes(f)
    r = f.readout(pos)
    assert not owes(f)
    pmod.HALO_DEFER = 'never'
    r0 = pm.paint(pos, mass=mass).readout(pos)
    pmod.HALO_DEFER = 'fresh'
    assert_allclose(r.cpu().numpy(), r0.cpu().numpy(), rtol=1e-12, atol=1e-12)
    # a second paint through the same plan while the first field still owes
    f1 = pm.paint(pos, mass=mass)
    f2 = pm.paint(pos, mass=mass)
    assert not owes(f1) and owes(f2)
    assert_allclose(f1.value.cpu().numpy(), ref, **tol)
    assert_allclose(f2.value.cpu().numpy(), ref, **tol)
    # a field dropped with its debt: the plan is released by whoever needs it next
    f3 = pm.paint(pos, mass=mass)
    assert owes(f3)
    del f3
    gc.collect()
    f4 = pm.paint(pos, mass=mass)
    assert_allclose(f4.value.cpu().numpy(), ref, **tol)
    # other positions (a rebuild of the plan) in between
    f5 = pm.paint(pos, mass=mass)
    pos2, mass2 = particles(pm, 1 << 20, 4)
    f6 = pm.paint(pos2, mass=mass2)
    assert_allclose(f5.value.cpu().numpy(), ref, **tol)
    assert_allclose(f6.value.cpu().numpy(), _eager_value(pm, pos2, mass2), **tol)
    # hold=True adds to a field that owes: the debt is paid, then the second batch added
    f7 = pm.paint(pos, mass=mass)
    pm.paint(pos2, mass=mass2, hold=True, out=f7)
    assert_allclose(f7.value.cpu().numpy(), ref + _eager_value(pm, pos2, mass2), rtol=0, atol=2e-12)
    # arithmetic on the field, a copy, csum
    f8 = pm.paint(pos, mass=mass)
    assert abs(float(f8.csum()) - float(mass.sum())) <= 1e-9 * float(mass.sum())
    f9 = pm.paint(pos, mass=mass)
    g = f9 * 2.0
    assert_allclose(g.value.cpu().numpy(), 2 * ref, rtol=0, atol=2e-12)
    # a non-in-place transform
    f10 = pm.paint(pos, mass=mass)
    k10 = f10.r2c()
    pmod.HALO_DEFER = 'never'
    k0 = pm.paint(pos, mass=mass).r2c()
    assert float((k10.value - k0.value).abs().max()) <= 1e-13 * float(k0.value.abs().max())
    assert_allclose(f10.value.cpu().numpy(), ref, **tol)


def test_a_callers_field_is_complete_when_paint_returns(hip):
    """in the reference `value` is a plain array (pm.py:234-242): a view taken before pm.paint(out=field) holds the
    finished mesh afterwards — nothing is deferred on a field the caller made, unless asked for ('always')"""
    pm = ParticleMesh(Nmesh=[128, 128, 128], BoxSize=1.0, dtype='f8', resampler='cic')
    pos, mass = particles(pm, 1 << 20, 9)
    ref = _eager_value(pm, pos, mass)
    pmod.HALO_DEFER = 'fresh'
    field = pm.create(type=RealField)
    view = field.value
    pm.paint(pos, mass=mass, out=field)
    assert not owes(field)
    assert_allclose(view.cpu().numpy(), ref, rtol=0, atol=1e-12)
    pmod.HALO_DEFER = 'always'
    pm.paint(pos, mass=mass, out=field)
    assert owes(field)
    k = field.r2c(out=Ellipsis)
    pmod.HALO_DEFER = 'never'
    k0 = pm.paint(pos, mass=mass).r2c(out=Ellipsis)
    assert float((k.value - k0.value).abs().max()) <= 1e-13 * float(k0.value.abs().max())


def test_full_cycle_with_the_deferred_merge(hip):
    """paint -> r2c -> c2r -> readout: forces of the cycle with the merge inside r2c == with the merge kernel"""
    from pmesh_amd.transfer import Transfer
    pm = ParticleMesh(Nmesh=[256, 256, 256], BoxSize=256.0, dtype='f8', resampler='cic')
    pos, mass = particles(pm, 1 << 23, 21)
    out = []
    for mode in ('never', 'fresh'):
        pmod.HALO_DEFER = mode
        window.clear_bin_cache()
        rho = pm.paint(pos, mass=mass)
        assert owes(rho) == (mode == 'fresh')
        rk = rho.r2c(out=Ellipsis)
        back = rk.c2r(out=Ellipsis)
        out.append(back.readout(pos).cpu().numpy())
    assert_allclose(out[1], out[0], rtol=0, atol=1e-11 * numpy.abs(out[0]).max())


@pytest.mark.parametrize('name', ['cic', 'pcs'])
@pytest.mark.parametrize('form', ['crowded', 'shuffled-sorted'])
def test_crowded_tiles_and_tile_ordered_copies(hip, name, form):
    """the pieces of crowded tiles are added to the mesh with global atomics by a kernel of their own (paint_heavy_kernel)
    — they commute with the halos the row pass adds later; a plan that streams its tile-ordered copy of the positions
    (rows in random order) stages the same halos"""
    pm = ParticleMesh(Nmesh=[128, 128, 128], BoxSize=128.0, dtype='f8', resampler=name)
    g = torch.Generator(device='cpu').manual_seed(17)
    n = 1 << 21
    pos = torch.rand(n, 3, generator=g, dtype=torch.float64) * 128.0
    if form == 'crowded':
        # half of the particles in a blob two cells wide: one tile holds 2^20 of them, 250 x its chunk
        pos[: n // 2] = 40.3 + 0.7 * torch.randn(n // 2, 3, generator=g, dtype=torch.float64)
        pos = pos % 128.0
    old_sorted = window.SORTED
    try:
        if form == 'shuffled-sorted':
            window.SORTED = 'always'
        pos = pos.cuda()
        mass = (torch.rand(n, generator=g, dtype=torch.float64) + 0.5).cuda()
        pmod.HALO_DEFER = 'never'
        window.clear_bin_cache()
        ek = pm.paint(pos, mass=mass).r2c(out=Ellipsis).value.clone()
        pmod.HALO_DEFER = 'fresh'
        window.clear_bin_cache()
        lazy = pm.paint(pos, mass=mass)
        assert owes(lazy)
        lk = lazy.r2c(out=Ellipsis).value
        assert float((lk - ek).abs().max()) <= 1e-13 * float(ek.abs().max())
    finally:
        window.SORTED = old_sorted
        window.clear_bin_cache()


@pytest.mark.parametrize('dtype', ['f8', 'f4'])
@pytest.mark.parametrize('blocked', [False, True])
def test_out_of_place_transforms_read_their_input_once_and_keep_it(hip, dtype, blocked):
    """r2c() / c2r() with out=None — the reference's default (pm.py:655-694, 987-1019) — on one rank: the first pass
    reads the input and writes the new field (pmx_rowfft_to / pmx_colfft_to, no copy in front of an in-place transform);
    same bits as the in-place transform of a copy, the input untouched; with a fused transfer; and on a field whose
    halo merge is still owed (the gather then writes elsewhere and the field keeps its debt)"""
    from pmesh_amd.transfer import Transfer
    nmesh = (128, 64, 256)
    pm = ParticleMesh(Nmesh=nmesh, BoxSize=[64.0, 32.0, 128.0], dtype=dtype, resampler='tsc')
    _fft.L3_BLOCK_BYTES = (9 * nmesh[1] * (nmesh[2] + 16) * (8 if dtype == 'f8' else 4)) if blocked else 0
    assert pm.plans['forwardT'].fills_output() and pm.plans['backwardT'].fills_output()
    pos, mass = particles(pm, int(numpy.prod(nmesh)) // 2, 23)
    pmod.HALO_DEFER = 'never'
    rho = pm.paint(pos, mass=mass)
    before = rho.value.clone()
    k_oop = rho.r2c()
    assert torch.equal(rho.value, before)
    k_ip = rho.copy().r2c(out=Ellipsis)
    assert torch.equal(k_oop.value, k_ip.value)
    kbefore = k_oop.value.clone()
    r_oop = k_oop.c2r()
    assert torch.equal(k_oop.value, kbefore)
    r_ip = k_oop.copy().c2r(out=Ellipsis)
    assert torch.equal(r_oop.value, r_ip.value)
    tol = 1e-12 if dtype == 'f8' else 2e-5
    assert float((r_oop.value - before).abs().max()) <= tol * float(before.abs().max())
    for T in (Transfer.dx1(0), Transfer.force(1), Transfer.potential()):
        a = k_oop.c2r(transfer=T)
        assert torch.equal(k_oop.value, kbefore)
        b = k_oop.apply(T).c2r(out=Ellipsis)
        assert float((a.value - b.value).abs().max()) <= tol * float(b.value.abs().max())
    # a field that still owes its halo merge, transformed out of place
    pmod.HALO_DEFER = 'fresh'
    lazy = pm.paint(pos, mass=mass)
    assert owes(lazy)
    k_lazy = lazy.r2c()
    assert owes(lazy), 'the input field lost its debt though its own values were never merged'
    assert float((k_lazy.value - k_ip.value).abs().max()) <= (1e-13 if dtype == 'f8' else 2e-6) * float(k_ip.value.abs().max())
    assert float((lazy.value - before).abs().max()) <= (1e-12 if dtype == 'f8' else 2e-6) * float(before.abs().max())
    assert not owes(lazy)
