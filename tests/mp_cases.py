"""Multi-rank cases, run as `python -m torch.distributed.run --nproc-per-node P tests/mp_cases.py`
(launched by tests/test_multirank.py).  One process per rank; on a CPU box the
backend is the oracle double + gloo, on a GPU box (PMESH_MP_BACKEND=hip) the HIP
library + RCCL.  Every assert runs on every rank; a failure exits non-zero.
"""
import os
import sys

# read by the HSA runtime when it initialises (the first torch.cuda call): before importing torch
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy
import torch
import torch.distributed as dist
from numpy.testing import assert_array_equal, assert_allclose, assert_almost_equal



def _pipeline_everything():
    """the cases below are about the CORRECTNESS of the pipelined transposes on meshes of a few dozen cells: no
    minimum chunk size (fft.OVERLAP_MIN_CHUNK_BYTES keeps production transforms from cutting chunks that cost
    more in launches than they hide of the wire)"""
    from pmesh_amd import fft as _F
    _F.OVERLAP_MIN_CHUNK_BYTES = 0


_pipeline_everything()

def setup():
    use_hip = os.environ.get('PMESH_MP_BACKEND', 'double') == 'hip'
    if use_hip:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
        import datetime
        os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '1')
        dist.init_process_group('nccl', timeout=datetime.timedelta(seconds=300))      # (a dead rank ends the job, see bench.py)
        from pmesh_amd import backend
        be = backend.get()
    else:
        import datetime
        dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=300))
        from tests import oracle_backend
        be = oracle_backend.install()
    from pmesh_amd.comm import TorchComm
    return be, TorchComm()


def gather_field(comm, field, full_shape):
    """assemble the global array from the local blocks (test_pm.py:257-259)"""
    full = numpy.zeros(full_shape, dtype=field.dtype)
    full[field.slices] = numpy.asarray(field)
    parts = comm.allgather(full)
    return sum(parts)


def case_exchange(be, comm):
    """pmesh/tests/test_domain.py:64-90, 243-266 known answers (first two ranks)"""
    from pmesh_amd import domain
    if comm.size != 2:
        return
    dcop = domain.GridND([[0, 1, 2], [0, 2]], comm=comm, periodic=True)
    if comm.rank == 0:
        pos = numpy.array(list(numpy.ndindex((2, 2))), dtype='f8')
        mass = numpy.array([0., 1, 2, 3])
    else:
        pos = numpy.empty((0, 2), dtype='f8')
        mass = numpy.array([], dtype='f8')
    layout = dcop.decompose(pos, smoothing=0)
    assert_array_equal(layout.get_exchange_cost(), [2, 0])
    npos = comm.allgather(layout.exchange(pos))
    assert_array_equal(npos[0], [[0, 0], [0, 1]])
    assert_array_equal(npos[1], [[1, 0], [1, 1]])
    nmass = layout.exchange(mass)
    mass2 = layout.gather(nmass)
    nmass = comm.allgather(nmass)
    assert_array_equal(nmass[0], [0, 1])
    assert_array_equal(nmass[1], [2, 3])
    assert_array_equal(mass2, mass)
    # smoothing 1: every particle is repeated once (test_domain.py:243-266)
    layout = dcop.decompose(pos, smoothing=1)
    npos = layout.exchange(pos)
    ones = numpy.ones(len(npos))
    assert_array_equal(layout.gather(ones, mode='sum'), 2 * numpy.ones(len(pos)))
    assert_array_equal(layout.gather(ones, mode='any'), numpy.ones(len(pos)))
    assert_array_equal(layout.gather(ones, mode=numpy.fmax), numpy.ones(len(pos)))
    assert_array_equal(layout.gather(npos, mode='local'), pos)
    allpos = comm.allgather(npos)
    assert_array_equal(allpos[0], [[0, 0], [0, 1], [1, 0], [1, 1]])
    assert_array_equal(allpos[1], [[0, 0], [0, 1], [1, 0], [1, 1]])


def case_period_empty_ranks(be, comm):
    """test_domain.py:196-216: degenerate (empty) domains receive nothing"""
    from pmesh_amd import domain
    if comm.size < 3:
        return
    dcop = domain.GridND([[0, 2, 4, 4], [0, 4]], comm=comm, periodic=True)
    pos = numpy.array([(0., 0.)])
    layout = dcop.decompose(pos, smoothing=1.5)
    p1 = layout.exchange(pos)
    if comm.rank == 2:
        assert len(p1) == 0
    if comm.rank in (0, 1):
        assert len(p1) == comm.size


def case_paint_distributed_equals_serial(be, comm):
    """test_pm.py:230-264: decompose + exchange + local paint, summed over ranks,
    equals painting all particles on one block; readout through the layout equals
    the serial readout."""
    from pmesh_amd.pm import ParticleMesh
    from oracle import oracle as O
    N, L = 12, 24.0
    rs = numpy.random.RandomState(100 + comm.rank)
    npart = 300 + 50 * comm.rank
    pos = rs.uniform(-L, 2 * L, size=(npart, 3))
    mass = rs.uniform(0.5, 1.5, size=npart)
    all_pos = numpy.concatenate(comm.allgather(pos), axis=0)
    all_mass = numpy.concatenate(comm.allgather(mass), axis=0)
    field = numpy.random.RandomState(5).normal(size=(N, N, N))
    for resampler, kind in (('cic', 'tunedcic'), ('tsc', 'tunedtsc'), ('pcs', 'tunedpcs')):
        pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], comm=comm, dtype='f8', resampler=resampler)
        aff = O.Affine(3, scale=1.0 * N / L, period=N)
        truth = numpy.zeros((N, N, N))
        O.Window(kind).paint(truth, all_pos, mass=all_mass, transform=aff)
        layout = pm.decompose(pos)
        real = pm.paint(pos, mass=mass, layout=layout)
        full = gather_field(comm, real, (N, N, N))
        assert_allclose(full, truth, rtol=0, atol=1e-12 * abs(truth).max())
        # readout with ghosts summed on the way back
        fld = pm.create('real', value=field[real.slices])
        got = fld.readout(pos, layout=layout)
        want = O.Window(kind).readout(field, pos, transform=aff)
        assert_allclose(got, want, rtol=0, atol=1e-12 * abs(want).max())
        g1 = fld.readout(pos, layout=layout, gradient=1)
        w1 = O.Window(kind).readout(field, pos, transform=aff, diffdir=1)
        assert_allclose(g1, w1, rtol=0, atol=1e-12 * abs(w1).max())


def case_halo_merge_left_to_the_slab_row_pass(be, comm):
    """pm.paint(pos, layout=...) on slab ranks returns a field of its own making whose tile halos are still staged
    (pm.HALO_DEFER, as on one rank: tests/test_halo_defer.py); the ghosts are added to it as it is, the row pass of
    r2c pays the debt (fft.Plan._slab_row_forward), every other reader merges first.  Spectrum and values equal the
    eager paint's (reference: pm.py:1795-1869 then pm.py:655-694; only the order of the additions differs)."""
    from pmesh_amd import pm as PM, window as W
    N, L = (128, 64, 128), 100.0
    rs = numpy.random.RandomState(31 + comm.rank)
    npart = 150000 + 1000 * comm.rank
    pos0 = rs.uniform(-0.2 * L, 1.2 * L, size=(npart, 3))
    mass0 = rs.uniform(0.5, 1.5, size=npart)
    old = (W.BINNED, PM.HALO_DEFER, W.BINNED_MIN_PARTICLES)
    try:
        # the rank's own particles through the tile kernels, the thin ghost band through the direct ones, as at full
        # size (a ghost batch that is itself binned needs the plan and pays the debt first: correct, nothing saved)
        comm.Barrier()
        W.BINNED, W.BINNED_MIN_PARTICLES = 'auto', 100000
        comm.Barrier()
        for resampler in ('cic', 'tsc', 'pcs'):
            for dtype in ('f8', 'f4'):
                pm = PM.ParticleMesh(BoxSize=L, Nmesh=N, comm=comm, dtype=dtype, resampler=resampler, np=[comm.size])     # slabs
                # every particle on the rank that holds its cell, as a time-stepping caller keeps them: what arrives
                # from other ranks in the paint is the thin band of ghosts
                home = pm.decompose(pos0, smoothing=0)
                pos, mass = home.exchange(pos0), home.exchange(mass0)
                layout = pm.decompose(pos)
                comm.Barrier()               # (thread ranks share the module's switches)
                PM.HALO_DEFER = 'never'
                comm.Barrier()
                eager = pm.paint(pos, mass=mass, layout=layout)
                assert getattr(eager._base.storage, '_pmx_halo', None) is None
                ev = numpy.array(numpy.asarray(eager))
                ek = numpy.array(numpy.asarray(eager.r2c(out=Ellipsis)))
                comm.Barrier()
                PM.HALO_DEFER = 'fresh'
                comm.Barrier()
                lazy = pm.paint(pos, mass=mass, layout=layout)
                owed = getattr(lazy._base.storage, '_pmx_halo', None) is not None
                if be.name == 'hip' and ev.shape[0] >= 16:     # (8 ranks: blocks of 16 planes, two layers of tiles and the offset)
                    assert owed, 'the paint did not leave its halo merge to the transform (%s %s, %d own rows, %d from other ranks)' % (
                        resampler, dtype, len(pos), layout.remote_recvlength)
                lk = numpy.array(numpy.asarray(lazy.r2c(out=Ellipsis)))
                assert getattr(lazy._base.storage, '_pmx_halo', None) is None
                tol = 1e-13 if dtype == 'f8' else 2e-6
                scale = comm.allreduce(float(abs(ek).max()) if ek.size else 0.0, op='max')
                assert float(abs(lk - ek).max()) <= tol * scale if ek.size else True
                # ... also when the ghost batch is itself large enough for the tile kernels (their plan lookup would
                # settle the debt: the ghosts of a field that owes its merge take the direct kernels instead)
                comm.Barrier()
                W.BINNED_MIN_PARTICLES = 64
                comm.Barrier()
                many = pm.paint(pos, mass=mass, layout=layout)
                if owed and layout.remote_recvlength >= 64:
                    assert getattr(many._base.storage, '_pmx_halo', None) is not None, \
                        'a ghost batch above BINNED_MIN_PARTICLES undid the deferral (%d ghosts)' % layout.remote_recvlength
                mk = numpy.array(numpy.asarray(many.r2c(out=Ellipsis)))
                assert float(abs(mk - ek).max()) <= tol * scale if ek.size else True
                comm.Barrier()
                W.BINNED_MIN_PARTICLES = 100000
                comm.Barrier()
                # a reader in between: the values are those of the eager paint
                lazy = pm.paint(pos, mass=mass, layout=layout)
                lv = numpy.array(numpy.asarray(lazy))
                assert getattr(lazy._base.storage, '_pmx_halo', None) is None
                assert float(abs(lv - ev).max()) <= tol * 8 * float(abs(ev).max())
                # a caller's field is complete when the call returns
                mine = pm.create('real')
                pm.paint(pos, mass=mass, layout=layout, out=mine)
                assert getattr(mine._base.storage, '_pmx_halo', None) is None
    finally:
        comm.Barrier()
        W.BINNED, PM.HALO_DEFER, W.BINNED_MIN_PARTICLES = old
        W.clear_bin_cache()


def case_ghosts_only_equals_literal(be, comm):
    """paint/readout with a layout: own particles in place + ghosts only (pm._ghosts_only) gives
    the same field / values as the reference's literal exchange -> local op -> gather
    (pm.py:1857-1868, 783-791); layouts that do not cover the window take the literal path."""
    from pmesh_amd import pm as PM
    N, L = 16, 8.0
    rs = numpy.random.RandomState(7 + comm.rank)
    npart = 257 + 31 * comm.rank
    pos = rs.uniform(-0.5 * L, 1.5 * L, size=(npart, 3))
    mass = rs.uniform(0.5, 1.5, size=npart)
    field = numpy.random.RandomState(5).normal(size=(N, N, N))
    for resampler in ('nnb', 'cic', 'tsc', 'pcs'):
        pm = PM.ParticleMesh(BoxSize=L, Nmesh=[N, N, N], comm=comm, dtype='f8', resampler=resampler)
        layout = pm.decompose(pos)
        assert PM._ghosts_only(layout, pm.resampler, pm.affine, None)
        fld = pm.create('real', value=field[pm.create('real').slices])
        res = {}
        for mode in ('auto', 'never'):
            comm.Barrier()              # thread ranks share the module: switch in step
            PM.GHOSTS_ONLY = mode
            comm.Barrier()
            try:
                a = pm.paint(pos, mass=mass, layout=layout)
                b = pm.paint(pos, mass=mass, layout=layout, gradient=2, hold=True, out=a.copy())
                r0 = fld.readout(pos, layout=layout)
                r1 = fld.readout(pos, layout=layout, gradient=0)
            finally:
                comm.Barrier()
                PM.GHOSTS_ONLY = 'auto'
            res[mode] = [numpy.asarray(x.value.cpu()) if hasattr(x, 'value') else numpy.asarray(x)
                         for x in (a, b, r0, r1)]
        for x, y in zip(res['auto'], res['never']):
            assert_allclose(x, y, rtol=0, atol=1e-12 * max(1.0, abs(y).max()))
    # a layout narrower than the window, or one built for another scaling, is not eligible
    pm = PM.ParticleMesh(BoxSize=L, Nmesh=[N, N, N], comm=comm, dtype='f8', resampler='pcs')
    narrow = pm.decompose(pos, smoothing=0.5)
    assert not PM._ghosts_only(narrow, pm.resampler, pm.affine, None)
    assert not PM._ghosts_only(pm.decompose(pos), pm.resampler, pm.affine, numpy.ones(npart))
    assert PM._ghosts_only(narrow, PM.FindResampler('nnb'), pm.affine, None)
    a = pm.paint(pos, layout=narrow)        # literal path: runs, conserves what was routed
    assert numpy.isfinite(a.csum())


def case_slab_fft(be, comm):
    """r2c / c2r on P ranks == numpy.fft on the gathered mesh, incl. uneven blocks"""
    from pmesh_amd.pm import ParticleMesh
    cases = [([8, 12, 10], 'f8', 1e-13), ([10, 6, 9], 'f8', 1e-13), ([9, 7], 'f8', 1e-13),
             ([16, 8, 8], 'f4', 5e-6),
             ([64, 64, 128], 'f8', 1e-13)]        # power-of-two: the LDS row/column kernels
    if be.name == 'hip':
        cases += [([64, 128, 128], 'f4', 5e-6), ([128, 64, 256], 'f8', 1e-13),
                  ([192, 192, 384], 'f8', 1e-13)]      # 3 * 2^k: radix-3 kernels, unfused pack
    for Nmesh, dtype, tol in cases:
        pm = ParticleMesh(BoxSize=1.0, Nmesh=Nmesh, comm=comm, dtype=dtype, np=[comm.size])
        if Nmesh == [64, 64, 128] and comm.size in (2, 4, 8):
            # the real side of this layout has its rows padded to 128 bytes (65 -> 72 complex)
            assert pm.plans['forwardT'].partition.pitch_i == 72
        data = numpy.random.RandomState(17).normal(size=Nmesh).astype(dtype)
        real = pm.create('real', value=data[pm.create('real').slices])
        ck = real.r2c()
        Nc = list(Nmesh[:-1]) + [Nmesh[-1] // 2 + 1]
        assert tuple(ck.cshape) == tuple(Nc)
        full = gather_field(comm, ck, Nc)
        ref = numpy.fft.rfftn(data.astype('f8')) / numpy.prod(Nmesh)
        err = numpy.sqrt((abs(full - ref) ** 2).sum() / (abs(ref) ** 2).sum())
        assert err < tol, (Nmesh, err)
        assert_array_equal(numpy.asarray(real), data[real.slices])       # input preserved
        back = ck.c2r()
        assert numpy.sqrt(((numpy.asarray(back) - data[back.slices]) ** 2).sum() /
                          max((data[back.slices] ** 2).sum(), 1e-300)) < 4 * tol
        # in place
        ck2 = real.r2c(out=Ellipsis)
        assert real._base in ck2._base
        full2 = gather_field(comm, ck2, Nc)
        assert numpy.sqrt((abs(full2 - ref) ** 2).sum() / (abs(ref) ** 2).sum()) < tol
        back2 = ck2.c2r(out=Ellipsis)
        assert numpy.sqrt(((numpy.asarray(back2) - data[back2.slices]) ** 2).sum() /
                          max((data[back2.slices] ** 2).sum(), 1e-300)) < 4 * tol
        # coordinates of the transposed field match its slices
        for d in range(len(Nmesh)):
            assert_array_equal(ck.i[d].cpu().numpy().ravel(),
                               numpy.arange(ck.slices[d].start, ck.slices[d].stop))


def case_pipelined_equals_single_exchange(be, comm):
    """the slab transform with its transposes pipelined over chunks of the last axis
    (fft.OVERLAP_CHUNKS) gives the same numbers as one exchange per transform"""
    from pmesh_amd import fft as _fft
    from pmesh_amd.pm import ParticleMesh
    from pmesh_amd.transfer import Transfer
    if comm.size not in (2, 4, 8):
        return
    Nmesh = [64, 64, 128]
    data = numpy.random.RandomState(31).normal(size=Nmesh)
    res = {}
    for chunks in (3, 2, 1):
        comm.Barrier()
        _fft.OVERLAP_CHUNKS = chunks
        comm.Barrier()
        try:
            pm = ParticleMesh(BoxSize=[3.0, 2.0, 5.0], Nmesh=Nmesh, comm=comm, dtype='f8', np=[comm.size])
            plan = pm.plans['forwardT']
            p = plan.partition
            got = plan._chunks(be, p, comm.size, 64, 64, 65, 64 // comm.size, 64 // comm.size,
                               [int(x) for x in p.i_edges[0]], [int(x) for x in p.o_edges[1]])
            assert (got is None) == (chunks == 1), (chunks, got)
            if got:
                assert sum(w for _, w in got) == 65 and all(b % 8 == 0 for b, _ in got)
            real = pm.create('real', value=data[pm.create('real').slices])
            ck = real.r2c()
            back = ck.c2r(transfer=Transfer.dx1(2))
            ip = real.copy().r2c(out=Ellipsis).c2r(out=Ellipsis, transfer=Transfer.dx1(2))
            res[chunks] = [numpy.asarray(x.value.cpu()) for x in (ck, back, ip)]
        finally:
            comm.Barrier()
            _fft.OVERLAP_CHUNKS = 2
    for chunks in (3, 2):
        for a, b in zip(res[chunks], res[1]):
            assert_allclose(a, b, rtol=0, atol=1e-13 * max(1.0, abs(b).max()))


def case_uneven_blocks_decide_alike(be, comm):
    """Whether a transform pipelines its transposes is decided by every rank for itself — and the decision leads to
    collectives (the probe of asynchronous exchanges, the chunked all-to-alls): all ranks must decide alike.  On
    uneven blocks the ranks' own block sizes differ; with a minimum chunk size between them, ranks that looked at
    their own size disagreed (found by scripts/halo_fuzz_slabs.py: three slab ranks, a 192 x 128 x 512 mesh)."""
    from pmesh_amd import fft as _fft
    from pmesh_amd.pm import ParticleMesh
    if comm.size != 3:
        return
    old = (_fft.OVERLAP_MIN_CHUNK_BYTES, _fft.OVERLAP_CHUNKS)
    comm.Barrier()
    try:
        for Nmesh, np_ in (([64, 64, 128], [3]), ([128, 64, 256], [3]), ([64, 64, 128], [3, 1])):      # (lengths the LDS kernels take)
            pm = ParticleMesh(BoxSize=1.0, Nmesh=Nmesh, comm=comm, dtype='f8', np=np_)
            sizes = comm.allgather(16 * int(numpy.prod(pm.create('complex').shape)))
            comm.Barrier()
            # a threshold that half of the largest block passes and half of the smallest does not (or all do)
            _fft.OVERLAP_MIN_CHUNK_BYTES, _fft.OVERLAP_CHUNKS = (max(sizes) // 2 + min(sizes) // 2) // 2 + 1, 2
            comm.Barrier()
            pm.procmesh.comm._pmx_async_ok = None
            data = numpy.random.RandomState(3).normal(size=Nmesh)
            real = pm.create('real', value=data[pm.create('real').slices])
            back = real.r2c().c2r()
            assert_allclose(numpy.asarray(back), numpy.asarray(real), rtol=0, atol=1e-12)
            comm.Barrier()
    finally:
        comm.Barrier()
        _fft.OVERLAP_MIN_CHUNK_BYTES, _fft.OVERLAP_CHUNKS = old
        comm.Barrier()


def case_fused_transfer_slab(be, comm):
    """c2r(transfer=T) on a slab decomposition (T folded into the first column pass of the
    inverse transform) == apply(T).c2r(), and leaves the complex field untouched out of place"""
    from pmesh_amd.pm import ParticleMesh
    from pmesh_amd.transfer import Transfer
    Nmesh = [64, 64, 128]
    pm = ParticleMesh(BoxSize=[3.0, 2.0, 5.0], Nmesh=Nmesh, comm=comm, dtype='f8', np=[comm.size])
    data = numpy.random.RandomState(23).normal(size=Nmesh)
    ck = pm.create('real', value=data[pm.create('real').slices]).r2c()
    for T in (Transfer.dx1(0), Transfer.dx1(1), Transfer.potential(), Transfer.force(2)):
        assert pm.plans['backwardT'].can_fuse()
        before = numpy.asarray(ck.value.cpu()).copy()
        want = numpy.asarray(ck.apply(T).c2r().value.cpu())
        got = numpy.asarray(ck.c2r(transfer=T).value.cpu())
        assert_allclose(numpy.asarray(ck.value.cpu()), before, rtol=0, atol=0)
        scale = comm.allreduce(float(abs(want).max()) if want.size else 0.0, op='max')
        assert_allclose(got, want, rtol=0, atol=1e-12 * scale)
        got2 = numpy.asarray(ck.copy().c2r(out=Ellipsis, transfer=T).value.cpu())
        assert_allclose(got2, want, rtol=0, atol=1e-12 * scale)


def case_whitenoise(be, comm):
    """generate_whitenoise does not depend on the decomposition (slab and pencil blocks gathered
    == the one-block oracle field), type='real' is its c2r, the mean lands on the k = 0 mode"""
    from pmesh_amd.pm import ParticleMesh
    from oracle import oracle as O
    Nmesh = [12, 16, 10]
    want = O.whitenoise((12, 16, 6), (0, 0, 0), Nmesh, 4242)
    nps = [[comm.size]]
    if comm.size == 4:
        nps.append([2, 2])
    for np_ in nps:
        pm = ParticleMesh(BoxSize=3.0, Nmesh=Nmesh, comm=comm, dtype='f8', np=np_)
        c = pm.generate_whitenoise(4242, mean=2.5)
        full = gather_field(comm, c, (12, 16, 6))
        ref = want.copy()
        ref[0, 0, 0] = 2.5
        assert_allclose(full, ref, rtol=0, atol=1e-14)
        r = pm.generate_whitenoise(4242, type='real', mean=2.5)
        rfull = gather_field(comm, r, tuple(Nmesh))
        assert_allclose(rfull, numpy.fft.irfftn(ref, s=Nmesh, axes=(0, 1, 2)) * numpy.prod(Nmesh), rtol=0, atol=1e-11)
        assert abs(r.cmean() - 2.5) < 1e-12


def case_ravel_resample_preview(be, comm):
    """the C-order redistribution (ravel/unravel), Fourier resampling, collective item access and
    preview on several ranks (test_pm.py:394-454, 553-630, 780-814)"""
    from pmesh_amd.pm import ParticleMesh, RealField, ComplexField
    # ravel / unravel: slab blocks of a 3-d mesh (the transposed complex layout really moves data)
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 6, 4], comm=comm, dtype='f8', np=[comm.size])
    real = RealField(pm)
    truth = numpy.arange(8 * 6 * 4, dtype='f8')
    real[...] = truth.reshape(8, 6, 4)[real.slices]
    unsorted = numpy.asarray(real).copy()
    flat = real.ravel()
    assert len(flat) == real.size
    assert_array_equal(numpy.concatenate(comm.allgather(flat.cpu().numpy())), truth)
    real[...] = 0
    real.unravel(flat)
    assert_array_equal(numpy.asarray(real), unsorted)
    cplx = ComplexField(pm)
    ctruth = numpy.arange(8 * 6 * 3) * (1 + 2j)
    cplx[...] = ctruth.reshape(8, 6, 3)[cplx.slices]
    cflat = cplx.ravel()
    assert_array_equal(numpy.concatenate(comm.allgather(cflat.cpu().numpy())), ctruth)
    # unravel from an arbitrary partition of the flat array: everything on the last rank
    piece = ctruth if comm.rank == comm.size - 1 else ctruth[:0]
    again = pm.unravel(ComplexField, piece)
    assert_array_equal(numpy.asarray(again), numpy.asarray(cplx))
    # Fourier down-sampling == the one-rank result
    pm1 = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8, 8], comm=comm, dtype='f8', np=[comm.size])
    pm2 = ParticleMesh(BoxSize=8.0, Nmesh=[4, 4, 4], comm=comm, dtype='f8', np=[comm.size])
    data = numpy.random.RandomState(3333).normal(size=(8, 8, 8))
    c1 = pm1.create('real', value=data[pm1.create('real').slices]).r2c()
    down = ComplexField(pm2)
    c1.resample(down)
    full = numpy.fft.rfftn(data) / 8 ** 3
    pick = numpy.r_[0:3, 7]                       # modes 0, 1, 2(-> Nyquist, removed), -1
    want = full[numpy.ix_(pick, pick, numpy.arange(3))].copy()
    want[2, :, :] = 0
    want[:, 2, :] = 0
    want[:, :, 2] = 0
    got = gather_field(comm, down, (4, 4, 3))
    assert_allclose(got, want, rtol=0, atol=1e-14)
    rdown = RealField(pm2)
    c1.resample(rdown)
    assert_allclose(gather_field(comm, rdown.r2c(), (4, 4, 3)), want, rtol=0, atol=1e-14)
    # collective item access
    v = down.cgetitem((1, 3, 1))
    assert abs(v - want[1, 3, 1]) < 1e-14
    z = ComplexField(pm2)
    z[...] = 0
    assert z.csetitem((1, 0, 0), 100 + 10j) == 100 + 10j
    assert abs(z.cgetitem((3, 0, 0)) - (100 - 10j)) == 0          # the Hermitian partner was set too
    # preview: the gathered field, and projections of it
    r = pm2.create('real', value=numpy.arange(64.).reshape(4, 4, 4)[pm2.create('real').slices])
    prev = r.preview(axes=(0, 1, 2))
    assert_array_equal(prev, numpy.arange(64.).reshape(4, 4, 4))
    assert_allclose(r.preview(axes=(2, 0)), prev.sum(axis=1).T)
    assert r.preview(Nmesh=2).shape == (2, 2, 2)


def case_untransposed(be, comm):
    """the untransposed complex layout (n0_local, N1, N2c) on several ranks (test_pm.py:128-165):
    r2c / c2r through it, casts between the two layouts, white noise in both"""
    from pmesh_amd.pm import ParticleMesh, UntransposedComplexField, TransposedComplexField
    for Nmesh, dtype, tol in (([8, 12, 10], 'f8', 1e-13), ([9, 7], 'f8', 1e-13), ([64, 64, 128], 'f8', 1e-13),
                              ([16, 8, 8], 'f4', 5e-6)):
        pm = ParticleMesh(BoxSize=1.0, Nmesh=Nmesh, comm=comm, dtype=dtype, np=[comm.size])
        data = numpy.random.RandomState(29).normal(size=Nmesh).astype(dtype)
        real = pm.create('real', value=data[pm.create('real').slices])
        Nc = list(Nmesh[:-1]) + [Nmesh[-1] // 2 + 1]
        ref = numpy.fft.rfftn(data.astype('f8')) / numpy.prod(Nmesh)
        cu = real.r2c(out=UntransposedComplexField(pm))
        assert isinstance(cu, UntransposedComplexField)
        assert tuple(cu.start[1:]) == (0,) * (len(Nmesh) - 1)            # axis 0 is the distributed one
        full = gather_field(comm, cu, Nc)
        assert numpy.sqrt((abs(full - ref) ** 2).sum() / (abs(ref) ** 2).sum()) < tol, Nmesh
        assert_array_equal(numpy.asarray(real), data[real.slices])       # input preserved
        back = cu.c2r()
        assert numpy.sqrt(((numpy.asarray(back) - data[back.slices]) ** 2).sum() /
                          max((data[back.slices] ** 2).sum(), 1e-300)) < 4 * tol
        # the two layouts hold the same modes: casts both ways
        ct = real.r2c()
        as_u = ct.cast(type='untransposedcomplex')
        assert_array_equal(gather_field(comm, as_u, Nc), gather_field(comm, ct, Nc))
        as_t = cu.cast(type='transposedcomplex')
        assert_array_equal(numpy.asarray(as_t), numpy.asarray(as_u.cast(type=TransposedComplexField)))
        assert numpy.sqrt((abs(gather_field(comm, as_t, Nc) - ref) ** 2).sum() / (abs(ref) ** 2).sum()) < tol
        # in place
        cu2 = real.r2c(out=UntransposedComplexField(pm, base=real._base))
        assert numpy.sqrt((abs(gather_field(comm, cu2, Nc) - ref) ** 2).sum() / (abs(ref) ** 2).sum()) < tol
        back2 = cu2.c2r(out=Ellipsis)
        assert numpy.sqrt(((numpy.asarray(back2) - data[back2.slices]) ** 2).sum() /
                          max((data[back2.slices] ** 2).sum(), 1e-300)) < 4 * tol
    # white noise does not depend on the layout (test_pm.py:145-165)
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8, 8], comm=comm, dtype='f4', np=[comm.size])
    f1 = pm.generate_whitenoise(seed=3333, type='untransposedcomplex')
    f2 = pm.generate_whitenoise(seed=3333, type='transposedcomplex')
    assert_array_equal(gather_field(comm, f1, (8, 8, 5)), gather_field(comm, f2, (8, 8, 5)))


def case_c2c(be, comm):
    """complex-to-complex meshes on a slab decomposition (test_pm.py:196-226): fftn / ifftn of the
    gathered field, transposed and untransposed spectra, paint into the real part"""
    from pmesh_amd.pm import ParticleMesh, UntransposedComplexField
    for Nmesh in ([8, 12, 10], [9, 7]):
        pm = ParticleMesh(BoxSize=4.0, Nmesh=Nmesh, comm=comm, dtype='c16', np=[comm.size])
        rs = numpy.random.RandomState(41)
        data = rs.normal(size=Nmesh) + 1j * rs.normal(size=Nmesh)
        real = pm.create('real', value=data[pm.create('real').slices])
        ref = numpy.fft.fftn(data) / numpy.prod(Nmesh)
        ck = real.r2c()
        assert tuple(ck.cshape) == tuple(Nmesh)
        full = gather_field(comm, ck, Nmesh)
        assert numpy.sqrt((abs(full - ref) ** 2).sum() / (abs(ref) ** 2).sum()) < 1e-13
        back = ck.c2r()
        assert_allclose(numpy.asarray(back), data[back.slices], rtol=0, atol=1e-12)     # (ranks may hold nothing)
        cu = real.r2c(out=UntransposedComplexField(pm))
        assert numpy.sqrt((abs(gather_field(comm, cu, Nmesh) - ref) ** 2).sum() / (abs(ref) ** 2).sum()) < 1e-13
        assert_allclose(numpy.asarray(cu.c2r()), data[back.slices], rtol=0, atol=1e-12)
    pm = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8, 8], comm=comm, dtype='c16', np=[comm.size])
    pos = numpy.random.RandomState(3 + comm.rank).uniform(0, 8, size=(50, 3))
    rho = pm.paint(pos, layout=pm.decompose(pos))
    assert abs(rho.csum() - 50 * comm.size) < 1e-10


def case_cycle(be, comm):
    """the whole PM cycle on P ranks == the serial oracle cycle"""
    from pmesh_amd.pm import ParticleMesh
    from pmesh_amd.transfer import Transfer
    from oracle import oracle as O
    N, L = 16, 1000.0
    allpos = O.synth_uniform(N, L)
    # each rank starts with an arbitrary share of the particles
    share = numpy.array_split(numpy.arange(len(allpos)), comm.size)[comm.rank]
    pos = allpos[share]
    pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], comm=comm, dtype='f8', resampler='cic')
    # np=None: the reference's process mesh (pm.py:1317-1325, pfft.split_size_2d) — on 3 ranks [1, 3], which
    # distributes axis 1 of the real field and leaves axis 0 whole, as PFFT lays it out
    from pmesh_amd.fft import split_size_2d
    want_np = list(split_size_2d(comm.size))
    assert pm.np == (want_np if want_np[1] > 1 else [comm.size]), (pm.np, want_np)
    layout = pm.decompose(pos)
    rho = pm.paint(pos, layout=layout)
    if len(pm.np) == 2 and pm.np[0] == 1:
        assert rho.shape[0] == N and rho.shape[1] < N and rho.shape[2] == N, rho.shape
    Ntot = comm.allreduce(len(pos))
    assert Ntot == N ** 3
    rho[...] *= 1.0 * pm.Nmesh.prod() / Ntot                  # nbody.py:205-207
    assert abs(rho.cmean() - 1.0) < 1e-12
    rhok = rho.r2c(out=Ellipsis)
    f = rhok.apply(Transfer.force(0), out=Ellipsis).c2r(out=Ellipsis).readout(pos, layout=layout)
    t = O.make_transfer(laplace_pow=-1, grad_dir=0, grad_kind=1)
    real, ck, back, out = O.pm_cycle(N, L, allpos, kind='tunedcic', transfer=t)
    want = out[share]
    assert abs(f - want).max() <= 1e-11 * abs(out).max()


def case_pencil(be, comm):
    """pencil decomposition np=[P0, P1] (pm.py:1319-1325; config 5): r2c / c2r == numpy.fft on
    the gathered mesh, even and uneven blocks, and the whole cycle == the serial oracle"""
    from pmesh_amd.pm import ParticleMesh
    from pmesh_amd.transfer import Transfer
    from oracle import oracle as O
    shapes = {4: [2, 2], 6: [2, 3], 8: [2, 4]}
    if comm.size not in shapes:
        return
    np_ = shapes[comm.size]
    cases = [([8, 12, 10], 'f8', 1e-13), ([10, 9, 14], 'f8', 1e-13), ([64, 64, 128], 'f8', 1e-13)]
    if be.name == 'hip':
        cases += [([128, 64, 128], 'f4', 5e-6)]
    for Nmesh, dtype, tol in cases:
        pm = ParticleMesh(BoxSize=1.0, Nmesh=Nmesh, comm=comm, dtype=dtype, np=np_)
        data = numpy.random.RandomState(23).normal(size=Nmesh).astype(dtype)
        real = pm.create('real', value=data[pm.create('real').slices])
        ck = real.r2c()
        Nc = list(Nmesh[:-1]) + [Nmesh[-1] // 2 + 1]
        assert tuple(ck.cshape) == tuple(Nc)
        full = gather_field(comm, ck, Nc)
        ref = numpy.fft.rfftn(data.astype('f8')) / numpy.prod(Nmesh)
        err = numpy.sqrt((abs(full - ref) ** 2).sum() / (abs(ref) ** 2).sum())
        assert err < tol, (Nmesh, err)
        assert_array_equal(numpy.asarray(real), data[real.slices])
        back = ck.c2r()
        loc = data[back.slices]
        assert numpy.sqrt(((numpy.asarray(back) - loc) ** 2).sum() / max((loc ** 2).sum(), 1e-300)) < 4 * tol
        ck2 = real.r2c(out=Ellipsis)
        full2 = gather_field(comm, ck2, Nc)
        assert numpy.sqrt((abs(full2 - ref) ** 2).sum() / (abs(ref) ** 2).sum()) < tol
        back2 = ck2.c2r(out=Ellipsis)
        loc = data[back2.slices]
        assert numpy.sqrt(((numpy.asarray(back2) - loc) ** 2).sum() / max((loc ** 2).sum(), 1e-300)) < 4 * tol
    # c2r(transfer=T) on pencils: the transfer rides on the axis-0 pass of the first stage
    Nmesh = [64, 64, 128]
    pm = ParticleMesh(BoxSize=[3.0, 2.0, 5.0], Nmesh=Nmesh, comm=comm, dtype='f8', np=np_)
    data = numpy.random.RandomState(5).normal(size=Nmesh)
    ck = pm.create('real', value=data[pm.create('real').slices]).r2c()
    assert pm.plans['backwardT'].can_fuse()
    for T in (Transfer.dx1(0), Transfer.dx1(2), Transfer.potential(), Transfer.force(1)):
        before = numpy.asarray(ck.value.cpu()).copy()
        want = numpy.asarray(ck.apply(T).c2r().value.cpu())
        got = numpy.asarray(ck.c2r(transfer=T).value.cpu())
        assert_allclose(numpy.asarray(ck.value.cpu()), before, rtol=0, atol=0)
        scale = comm.allreduce(float(abs(want).max()) if want.size else 0.0, op='max')
        assert_allclose(got, want, rtol=0, atol=1e-12 * scale)
        got2 = numpy.asarray(ck.copy().c2r(out=Ellipsis, transfer=T).value.cpu())
        assert_allclose(got2, want, rtol=0, atol=1e-12 * scale)
    # the cycle on pencils
    N, L = 16, 1000.0
    allpos = O.synth_uniform(N, L)
    share = numpy.array_split(numpy.arange(len(allpos)), comm.size)[comm.rank]
    pos = allpos[share]
    pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], comm=comm, dtype='f8', resampler='tsc', np=np_)
    layout = pm.decompose(pos)
    rho = pm.paint(pos, layout=layout)
    assert abs(rho.csum() - N ** 3) < 1e-8
    f = rho.r2c(out=Ellipsis).apply(Transfer.dx1(1), out=Ellipsis).c2r(out=Ellipsis).readout(pos, layout=layout)
    t = O.make_transfer(laplace_pow=-1, grad_dir=1, grad_kind=0)
    real, ck, back, out = O.pm_cycle(N, L, allpos, kind='tunedtsc', transfer=t)
    assert abs(f - out[share]).max() <= 1e-11 * abs(out).max()


def case_pencil_pipelined_equals_single_exchange(be, comm):
    """fft.OVERLAP_CHUNKS: both transposes of the pencil transform cut into chunks of the local planes,
    exchanged asynchronously (the second one into row ranges of the output: comm.alltoall_views), give
    the numbers of the single exchanges bit for bit; r2c / c2r in place and out of place, with the fused
    transfer, equal and uneven last-axis blocks"""
    from pmesh_amd import fft as F, comm as C
    from pmesh_amd.pm import ParticleMesh
    from pmesh_amd.transfer import Transfer
    shapes = {4: [2, 2], 8: [2, 4]}
    if comm.size not in shapes:
        return
    np_ = shapes[comm.size]
    saved = F.OVERLAP_CHUNKS
    try:
        for Nmesh in ([64, 64, 128], [64, 128, 128]):
            res = {}
            for chunks in (1, 2, 3):
                F.OVERLAP_CHUNKS = chunks
                pm = ParticleMesh(BoxSize=[3.0, 2.0, 5.0], Nmesh=Nmesh, comm=comm, dtype='f8', np=np_)
                data = numpy.random.RandomState(31).normal(size=Nmesh)
                real = pm.create('real', value=data[pm.create('real').slices])
                real.r2c()                       # (first use: the probes of the asynchronous exchange)
                rec = C.trace(True) if hasattr(comm, '_dist') else None
                ck = real.r2c()
                if rec is not None:
                    C.trace(False)
                    over = [r for r in rec if r[5]]
                    # pipelined: every data exchange is asynchronous, 2 transposes x chunks of them
                    assert len(over) == (2 * chunks if chunks > 1 else 0), (chunks, len(rec), len(over))
                back = ck.c2r()
                T = Transfer.dx1(1)
                f = ck.c2r(transfer=T)
                ck2 = real.copy().r2c(out=Ellipsis)
                back2 = ck2.copy().c2r(out=Ellipsis)
                res[chunks] = [numpy.asarray(x.value.cpu()).copy() for x in (ck, back, f, ck2, back2)]
                assert_allclose(res[chunks][1], data[back.slices], rtol=0, atol=1e-12)
            for chunks in (2, 3):
                for x, y in zip(res[1], res[chunks]):
                    assert_array_equal(x, y)
    finally:
        F.OVERLAP_CHUNKS = saved


def case_pencil_row_split_equals_two_sweeps(be, comm):
    """fft.ROW_SPLIT: the last-axis split of the pencil transform's first transpose on the row pass itself
    (pmx_rowfft_split) gives the bits of the row pass followed by pmx_slab_pack — single and pipelined exchanges, in
    place and out of place (an out-of-place r2c keeps its input without a copy), with the fused transfer; last axes
    whose mode count does not divide by the row group, and one the split form is not built for"""
    from pmesh_amd import fft as F
    from pmesh_amd.pm import ParticleMesh
    from pmesh_amd.transfer import Transfer
    shapes = {2: [1, 2], 4: [2, 2], 8: [2, 4]}
    if comm.size not in shapes:
        return
    np_ = shapes[comm.size]
    saved = F.ROW_SPLIT, F.OVERLAP_CHUNKS
    T = Transfer.dx1(2)
    try:
        for Nmesh, dtype in (([64, 64, 128], 'f8'), ([64, 128, 256], 'f4'), ([64, 64, 384], 'f8')):
            data = numpy.random.RandomState(17).normal(size=Nmesh)
            for chunks in (1, 2):
                F.OVERLAP_CHUNKS = chunks
                res = {}
                for split in (False, True):
                    F.ROW_SPLIT = split
                    pm = ParticleMesh(BoxSize=[3.0, 2.0, 5.0], Nmesh=Nmesh, comm=comm, dtype=dtype, np=np_)
                    real = pm.create('real', value=data[pm.create('real').slices])
                    before = numpy.asarray(real.value.cpu()).copy()
                    ck = real.r2c()
                    assert_array_equal(numpy.asarray(real.value.cpu()), before)      # the input of r2c is kept
                    back = ck.c2r()
                    f = ck.c2r(transfer=T)
                    ck2 = real.copy().r2c(out=Ellipsis)
                    back2 = ck2.copy().c2r(out=Ellipsis)
                    res[split] = [numpy.asarray(x.value.cpu()).copy() for x in (ck, back, f, ck2, back2)]
                    assert_allclose(res[split][1], data[back.slices], rtol=0, atol=1e-12 if dtype == 'f8' else 2e-5)
                for x, y in zip(res[False], res[True]):
                    assert_array_equal(x, y)
    finally:
        F.ROW_SPLIT, F.OVERLAP_CHUNKS = saved


def case_deferred_last_pass_on_slabs(be, comm):
    """fft.DEFER_LAST_PASS on a slab decomposition: r2c leaves the axis-0 pass on the received block (one exchange) or
    on the chunk buffers of the pipelined exchange; an in-place c2r runs both axis-0 passes and the transfer as one
    kernel per block / chunk and sends the chunks straight back; anything else that looks at the spectrum settles it.
    Same bits as the eager transforms."""
    from pmesh_amd import fft as F
    from pmesh_amd.pm import ParticleMesh
    from pmesh_amd.transfer import Transfer
    saved = F.DEFER_LAST_PASS, F.OVERLAP_CHUNKS
    T = Transfer.dx1(1)
    try:
        for Nmesh in ([64, 64, 128], [128, 64, 128]):
            if Nmesh[0] % comm.size or Nmesh[1] % comm.size:
                continue
            data = numpy.random.RandomState(3).normal(size=Nmesh)
            for chunks in (1, 2):
                F.OVERLAP_CHUNKS = chunks
                res = {}
                for defer in (False, True):
                    F.DEFER_LAST_PASS = defer
                    pm = ParticleMesh(BoxSize=[3.0, 2.0, 5.0], Nmesh=Nmesh, comm=comm, dtype='f8', np=[comm.size])

                    def fresh():
                        return pm.create('real', value=data[pm.create('real').slices])
                    a = numpy.asarray(fresh().r2c(out=Ellipsis).c2r(out=Ellipsis, transfer=T))      # fused
                    b = numpy.asarray(fresh().r2c(out=Ellipsis).c2r(out=Ellipsis))                  # no transfer
                    ck = fresh().r2c(out=Ellipsis)
                    spec = numpy.asarray(ck).copy()                                                 # looked at: settles
                    c = numpy.asarray(ck.c2r(out=Ellipsis, transfer=T))
                    ck = fresh().r2c()                                                              # out of place
                    d = numpy.asarray(ck.c2r(transfer=T))
                    ck = fresh().r2c(out=Ellipsis)                                                  # abandoned, then reused
                    note = getattr(ck._base.storage, '_pmx_pending', None)
                    if defer and ck.size:
                        assert note is not None and note.kind == ('slabpipe' if chunks > 1 else 'slab'), (chunks, note)
                    else:
                        assert note is None
                    e = numpy.asarray(fresh().r2c(out=Ellipsis).c2r(out=Ellipsis))
                    res[defer] = [a, b, spec, c, d, e]
                for x, y in zip(res[False], res[True]):
                    assert_array_equal(x, y)
                assert_allclose(res[True][1], data[pm.create('real').slices], rtol=0, atol=1e-12)
    finally:
        F.DEFER_LAST_PASS, F.OVERLAP_CHUNKS = saved


def case_deferred_last_pass_on_pencils(be, comm):
    """fft.DEFER_LAST_PASS on a 2-d process mesh: r2c leaves the axis-0 pass on the block the second transpose
    delivered (single or pipelined exchanges alike); an in-place c2r of the same partition runs both axis-0 passes
    and the transfer as one kernel; whatever else looks at the spectrum settles it.  Same bits as the eager transforms."""
    from pmesh_amd import fft as F
    from pmesh_amd.pm import ParticleMesh
    from pmesh_amd.transfer import Transfer
    shapes = {4: [2, 2], 6: [2, 3], 8: [2, 4]}
    if comm.size not in shapes:
        return
    np_ = shapes[comm.size]
    saved = F.DEFER_LAST_PASS, F.OVERLAP_CHUNKS
    T = Transfer.dx1(2)
    try:
        for Nmesh in ([64, 64, 128], [64, 72, 128]):          # even blocks (fused axis-1 pass, pipelined) / uneven ones
            data = numpy.random.RandomState(5).normal(size=Nmesh)
            for chunks in (1, 2):
                F.OVERLAP_CHUNKS = chunks
                res = {}
                for defer in (False, True):
                    F.DEFER_LAST_PASS = defer
                    pm = ParticleMesh(BoxSize=[3.0, 2.0, 5.0], Nmesh=Nmesh, comm=comm, dtype='f8', np=np_)

                    def fresh():
                        return pm.create('real', value=data[pm.create('real').slices])
                    a = numpy.asarray(fresh().r2c(out=Ellipsis).c2r(out=Ellipsis, transfer=T))      # fused
                    b = numpy.asarray(fresh().r2c(out=Ellipsis).c2r(out=Ellipsis))                  # no transfer
                    ck = fresh().r2c(out=Ellipsis)
                    spec = numpy.asarray(ck).copy()                                                 # looked at: settles
                    c = numpy.asarray(ck.c2r(out=Ellipsis, transfer=T))
                    ck = fresh().r2c()                                                              # out of place
                    d = numpy.asarray(ck.c2r(transfer=T))
                    ck = fresh().r2c(out=Ellipsis)                                                  # abandoned, then reused
                    note = getattr(ck._base.storage, '_pmx_pending', None)
                    if defer and ck.size:
                        assert note is not None and note.kind == 'slab', (chunks, note)
                    else:
                        assert note is None
                    e = numpy.asarray(fresh().r2c(out=Ellipsis).c2r(out=Ellipsis))
                    cu = fresh().r2c(out=Ellipsis).cast(type='untransposedcomplex')                 # cast settles
                    f = numpy.asarray(cu).copy()
                    res[defer] = [a, b, spec, c, d, e, f]
                for x, y in zip(res[False], res[True]):
                    assert_array_equal(x, y)
                assert_allclose(res[True][1], data[pm.create('real').slices], rtol=0, atol=1e-12)
    finally:
        F.DEFER_LAST_PASS, F.OVERLAP_CHUNKS = saved


def case_pencil_untransposed_and_c2c(be, comm):
    """every plan of the reference on a 2-d process mesh (pm.py:1332-1349 builds all eight): the untransposed
    complex layout — distributed like the real field, (N0 / P0, N1 / P1, N2c) — r2c / c2r through it and casts both
    ways; complex-to-complex meshes (pm.py:1270) transposed and untransposed; even and uneven blocks"""
    from pmesh_amd.pm import ParticleMesh, UntransposedComplexField, TransposedComplexField
    shapes = {4: [2, 2], 6: [2, 3], 8: [2, 4]}
    if comm.size not in shapes:
        return
    np_ = shapes[comm.size]
    for Nmesh, dtype, tol in (([8, 12, 10], 'f8', 1e-13), ([10, 9, 14], 'f8', 1e-13), ([64, 64, 128], 'f8', 1e-13)):
        pm = ParticleMesh(BoxSize=1.0, Nmesh=Nmesh, comm=comm, dtype=dtype, np=np_)
        data = numpy.random.RandomState(37).normal(size=Nmesh).astype(dtype)
        real = pm.create('real', value=data[pm.create('real').slices])
        Nc = list(Nmesh[:-1]) + [Nmesh[-1] // 2 + 1]
        ref = numpy.fft.rfftn(data.astype('f8')) / numpy.prod(Nmesh)
        cu = real.r2c(out=UntransposedComplexField(pm))
        assert isinstance(cu, UntransposedComplexField)
        assert tuple(cu.start[:2]) == tuple(real.start[:2]) and int(cu.start[2]) == 0    # distributed like the real field
        assert tuple(cu.shape) == (real.shape[0], real.shape[1], Nc[2])
        full = gather_field(comm, cu, Nc)
        assert numpy.sqrt((abs(full - ref) ** 2).sum() / (abs(ref) ** 2).sum()) < tol, Nmesh
        assert_array_equal(numpy.asarray(real), data[real.slices])                       # input preserved
        back = cu.c2r()
        assert numpy.sqrt(((numpy.asarray(back) - data[back.slices]) ** 2).sum() /
                          max((data[back.slices] ** 2).sum(), 1e-300)) < 4 * tol
        ct = real.r2c()
        as_u = ct.cast(type='untransposedcomplex')
        assert_array_equal(gather_field(comm, as_u, Nc), gather_field(comm, ct, Nc))
        as_t = cu.cast(type='transposedcomplex')
        assert_array_equal(gather_field(comm, as_t, Nc), gather_field(comm, cu, Nc))
        assert isinstance(as_t, TransposedComplexField)
    for Nmesh in ([8, 12, 10], [16, 8, 12]):
        pm = ParticleMesh(BoxSize=4.0, Nmesh=Nmesh, comm=comm, dtype='c16', np=np_)
        rs = numpy.random.RandomState(43)
        data = rs.normal(size=Nmesh) + 1j * rs.normal(size=Nmesh)
        real = pm.create('real', value=data[pm.create('real').slices])
        ref = numpy.fft.fftn(data) / numpy.prod(Nmesh)
        ck = real.r2c()
        assert tuple(ck.cshape) == tuple(Nmesh)
        assert numpy.sqrt((abs(gather_field(comm, ck, Nmesh) - ref) ** 2).sum() / (abs(ref) ** 2).sum()) < 1e-13
        back = ck.c2r()
        assert_allclose(numpy.asarray(back), data[back.slices], rtol=0, atol=1e-12)
        cu = real.r2c(out=UntransposedComplexField(pm))
        assert numpy.sqrt((abs(gather_field(comm, cu, Nmesh) - ref) ** 2).sum() / (abs(ref) ** 2).sum()) < 1e-13
        assert_allclose(numpy.asarray(cu.c2r()), data[back.slices], rtol=0, atol=1e-12)


def case_length_check_is_collective(be, comm):
    """domain.py:177-179, 240-242: a wrong array on ONE rank raises ValueError on EVERY rank the first
    time a layout is used (the verdict is all-reduced); afterwards the offending rank raises alone,
    after taking part in the exchange, so nobody is left waiting in the all-to-all"""
    import pytest
    from pmesh_amd import domain
    P = comm.size
    dcop = domain.GridND([numpy.linspace(0, 1, P + 1)], comm=comm, periodic=True)
    rng = numpy.random.RandomState(100 + comm.rank)
    pos = rng.uniform(0, 1, size=(40, 1))
    layout = dcop.decompose(pos, smoothing=0.01)
    bad = numpy.ones(41 if comm.rank == P - 1 else 40)
    with pytest.raises(ValueError):
        layout.exchange(bad)                       # first use: every rank raises
    good = layout.exchange(numpy.ones(40))         # the check of this direction has been held
    assert len(good) == layout.recvlength
    with pytest.raises(ValueError):
        layout.gather(numpy.ones(layout.recvlength + (1 if comm.rank == 0 else 0)))
    assert_array_equal(layout.gather(good, mode='any'), numpy.ones(40))
    # later mistakes: the offending rank raises, the others complete the exchange
    if comm.rank == P - 1:
        with pytest.raises(ValueError):
            layout.exchange(bad)
    else:
        layout.exchange(numpy.ones(40))
    comm.Barrier()


def case_promote_and_pack(be, comm):
    """domain.py:50-57 (an empty rank adopts the root's dtype; a trailing shape that differs raises)
    and domain.py:161-166 (pack=True: one all-to-all-v for all arrays, same rows as one by one)"""
    import pytest
    from pmesh_amd import domain, comm as C
    P = comm.size
    a = numpy.zeros((3, 2), dtype='f4') if comm.rank == 0 else numpy.zeros((0, 2), dtype='f8')
    assert domain.promote(a, comm).dtype == numpy.dtype('f4')
    b = numpy.zeros((3, 2)) if comm.rank == 0 else numpy.zeros((3, 3))
    if comm.rank == 0:
        domain.promote(b, comm)
    else:
        with pytest.raises(ValueError):
            domain.promote(b, comm)
    dcop = domain.GridND([numpy.linspace(0, 1, P + 1)], comm=comm, periodic=True)
    rng = numpy.random.RandomState(7 + comm.rank)
    pos = rng.uniform(0, 1, size=(50, 3))
    mass = rng.uniform(size=50)
    ident = numpy.arange(50, dtype='i4') + 1000 * comm.rank
    layout = dcop.decompose(pos[:, :1], smoothing=0.05)
    one = [layout.exchange(pos), layout.exchange(mass), layout.exchange(ident)]
    rec = C.trace(True) if hasattr(comm, '_dist') else None
    packed = layout.exchange(pos, mass, ident)
    if rec is not None:
        C.trace(False)
        assert len(rec) == 1, rec                   # ONE collective for the three arrays
    for x, y in zip(one, packed):
        assert x.dtype == y.dtype and x.shape == y.shape
        assert_array_equal(x, y)
    unpacked = layout.exchange(pos, mass, ident, pack=False)
    for x, y in zip(one, unpacked):
        assert_array_equal(x, y)
    # large exchanges travel array by array (domain.PACK_BYTES_MAX: no packed copy on either side): forced here
    saved = domain.PACK_BYTES_MAX
    try:
        domain.PACK_BYTES_MAX = 0
        rec = C.trace(True) if hasattr(comm, '_dist') else None
        big = layout.exchange(pos, mass, ident)
        if rec is not None:
            C.trace(False)
            assert len(rec) == 3, rec
        for x, y in zip(one, big):
            assert x.dtype == y.dtype and x.shape == y.shape
            assert_array_equal(x, y)
    finally:
        domain.PACK_BYTES_MAX = saved
    # the staging of the exchanges lives on the communicator and is reused by the layouts that follow
    staging = domain._scratch_of(comm)
    held = staging.nbytes()
    layout2 = dcop.decompose(pos[:, :1], smoothing=0.05)
    again = layout2.exchange(pos, mass, ident)
    assert staging.nbytes() == held                 # nothing new was allocated for the same exchange
    for x, y in zip(one, again):
        assert_array_equal(x, y)


def case_async_ghost_exchange(be, comm):
    """Layout.exchange_remote / gather_remote_add with async_op=True (what paint / readout use to run the
    particle exchange under their local work) give what the blocking calls give, packed or not"""
    from pmesh_amd import domain
    P = comm.size
    dcop = domain.GridND([numpy.linspace(0, 1, P + 1)], comm=comm, periodic=True)
    rng = numpy.random.RandomState(50 + comm.rank)
    pos = torch.from_numpy(rng.uniform(0, 1, size=(200, 3))).to(be.device)
    mass = torch.from_numpy(rng.uniform(size=200)).to(be.device)
    a = dcop.decompose(pos[:, :1], smoothing=0.08)
    b = dcop.decompose(pos[:, :1], smoothing=0.08)
    rp, rm = a.exchange_remote(pos, mass)
    h = b.exchange_remote(pos, mass, async_op=True)
    rp2, rm2 = h.wait()
    assert torch.equal(rp, rp2) and torch.equal(rm, rm2)
    assert torch.equal(b.exchange_remote(pos), rp)                      # remembered per source tensor
    vals = rp[:, 0] * 2 + rm
    out1 = torch.zeros(200, dtype=torch.float64, device=be.device)
    out2 = torch.zeros(200, dtype=torch.float64, device=be.device)
    a.gather_remote_add(vals, out1)
    b.gather_remote_add(vals.clone(), None, async_op=True).wait(out2)
    assert torch.allclose(out1, out2, rtol=0, atol=1e-15)
    # array by array (what exchanges of hundreds of MB do, domain.PACK_BYTES_MAX) == packed; the received rows are
    # remembered per source tensor — weakly: a tensor that is gone, or another one at its address, never hits
    saved = domain.PACK_BYTES_MAX
    try:
        domain.PACK_BYTES_MAX = 0
        c = dcop.decompose(pos[:, :1], smoothing=0.08)
        h = c.exchange_remote(pos, mass, async_op=True)
        rp3, rm3 = h.wait()
        assert torch.equal(rp, rp3) and torch.equal(rm, rm3)
    finally:
        domain.PACK_BYTES_MAX = saved
    assert len(c._memo_remote) == 2
    pos2 = pos.clone()
    rq = c.exchange_remote(pos2)
    assert torch.equal(rq, rp) and rq is not rp3
    del pos2, rq
    c.exchange_remote(mass)                                                    # (an exchange sweeps the memo)
    assert all(v[0]() is not None for v in c._memo_remote.values())


def case_readout_into_strided_and_float_out(be, comm):
    """RealField.readout(layout=..., out=F[:, d]) — how a force loop writes its columns — and a float32 out, on several
    ranks: the partial sums of the ghosts land in the caller's rows whatever the stride (the contract of
    pmx_scatter_add is a DENSE out; a strided target takes the strided add).  And two asynchronous exchanges in flight
    on one communicator do not share their staging: a second begin before the first wait leaves the first's rows alone."""
    from pmesh_amd import pm as PM, domain
    N, L = 16, 8.0
    rs = numpy.random.RandomState(70 + comm.rank)
    npart = 301 + 17 * comm.rank
    pos_h = rs.uniform(0, L, size=(npart, 3))
    field = numpy.random.RandomState(6).normal(size=(N, N, N))
    pm = PM.ParticleMesh(BoxSize=L, Nmesh=[N, N, N], comm=comm, dtype='f8', resampler='cic')
    fld = pm.create('real', value=field[pm.create('real').slices])
    pos = torch.from_numpy(pos_h).to(be.device)
    layout = pm.decompose(pos)
    want = [fld.readout(pos, layout=layout, gradient=d) for d in range(3)]
    want = [w if torch.is_tensor(w) else torch.from_numpy(numpy.asarray(w)).to(be.device) for w in want]
    F = torch.full((npart, 3), 7.0, dtype=torch.float64, device=be.device)
    for d in range(3):
        r = fld.readout(pos, layout=layout, gradient=d, out=F[:, d])
        assert r.data_ptr() == F[:, d].data_ptr()
    for d in range(3):
        assert torch.allclose(F[:, d], want[d], rtol=0, atol=1e-12), d
    F4 = torch.zeros((npart, 3), dtype=torch.float32, device=be.device)
    o4 = torch.zeros(npart, dtype=torch.float32, device=be.device)
    fld.readout(pos, layout=layout, gradient=1, out=F4[:, 1])
    fld.readout(pos, layout=layout, gradient=1, out=o4)
    assert torch.allclose(F4[:, 1].double(), want[1], rtol=0, atol=2e-6 * float(want[1].abs().max() + 1))
    assert torch.allclose(o4.double(), want[1], rtol=0, atol=2e-6 * float(want[1].abs().max() + 1))
    assert float(F4[:, 0].abs().max()) == 0.0 and float(F4[:, 2].abs().max()) == 0.0
    # overlapping asynchronous handles on one communicator
    P = comm.size
    dcop = domain.GridND([numpy.linspace(0, 1, P + 1)], comm=comm, periodic=True)
    rng = numpy.random.RandomState(90 + comm.rank)
    x1 = torch.from_numpy(rng.uniform(0, 1, size=(150, 3))).to(be.device)
    m1 = torch.from_numpy(rng.uniform(size=150)).to(be.device)
    x2 = torch.from_numpy(rng.uniform(0, 1, size=(170, 3))).to(be.device)
    m2 = torch.from_numpy(rng.uniform(size=170)).to(be.device)
    la, lb = dcop.decompose(x1[:, :1], smoothing=0.08), dcop.decompose(x2[:, :1], smoothing=0.08)
    ra = la.exchange_remote(x1, m1)
    rb = lb.exchange_remote(x2, m2)
    la2, lb2 = dcop.decompose(x1[:, :1], smoothing=0.08), dcop.decompose(x2[:, :1], smoothing=0.08)
    ha = la2.exchange_remote(x1, m1, async_op=True)
    hb = lb2.exchange_remote(x2, m2, async_op=True)          # begins before the first has been waited for
    ga, gb = ha.wait(), hb.wait()
    assert torch.equal(ga[0], ra[0]) and torch.equal(ga[1], ra[1])
    assert torch.equal(gb[0], rb[0]) and torch.equal(gb[1], rb[1])
    va, vb = ra[0][:, 0] + ra[1], rb[0][:, 1] - rb[1]
    o1, o2 = torch.zeros(150, dtype=torch.float64, device=be.device), torch.zeros(170, dtype=torch.float64, device=be.device)
    la.gather_remote_add(va, o1)
    lb.gather_remote_add(vb, o2)
    p1 = la2.gather_remote_add(va.clone(), None, async_op=True)
    p2 = lb2.gather_remote_add(vb.clone(), None, async_op=True)      # a second begin before wait(out) of the first
    q1, q2 = torch.zeros_like(o1), torch.zeros_like(o2)
    p1.wait(q1)
    p2.wait(q2)
    assert torch.allclose(q1, o1, rtol=0, atol=1e-15) and torch.allclose(q2, o2, rtol=0, atol=1e-15)
    # THREE in flight, the first waited for last: the second must not take the first's hold on the pooled buffer
    # away (the third would then receive into the rows the first has not read yet)
    x3 = torch.from_numpy(rng.uniform(0, 1, size=(190, 3))).to(be.device)
    m3 = torch.from_numpy(rng.uniform(size=190)).to(be.device)
    lc = dcop.decompose(x3[:, :1], smoothing=0.08)
    rc = lc.exchange_remote(x3, m3)
    la3, lb3, lc3 = [dcop.decompose(x[:, :1], smoothing=0.08) for x in (x1, x2, x3)]
    ha = la3.exchange_remote(x1, m1, async_op=True)
    hb = lb3.exchange_remote(x2, m2, async_op=True)
    gb = hb.wait()
    hc = lc3.exchange_remote(x3, m3, async_op=True)
    gc = hc.wait()
    ga = ha.wait()
    for g, r in ((ga, ra), (gb, rb), (gc, rc)):
        assert torch.equal(g[0], r[0]) and torch.equal(g[1], r[1])
    vc = rc[0][:, 2] * rc[1]
    o3 = torch.zeros(190, dtype=torch.float64, device=be.device)
    lc.gather_remote_add(vc, o3)
    p1 = la3.gather_remote_add(va.clone(), None, async_op=True)
    p2 = lb3.gather_remote_add(vb.clone(), None, async_op=True)
    q2 = torch.zeros_like(o2)
    p2.wait(q2)
    p3 = lc3.gather_remote_add(vc.clone(), None, async_op=True)
    q3 = torch.zeros_like(o3)
    p3.wait(q3)
    q1 = torch.zeros_like(o1)
    p1.wait(q1)
    assert torch.allclose(q1, o1, rtol=0, atol=1e-15) and torch.allclose(q2, o2, rtol=0, atol=1e-15)
    assert torch.allclose(q3, o3, rtol=0, atol=1e-15)
    # a handle dropped without wait() gives the staging back: the pool serves the next exchange again
    ld = dcop.decompose(x1[:, :1], smoothing=0.08)
    hd = ld.exchange_remote(x1, m1, async_op=True)
    del hd
    scratch = domain._scratch_of(comm)
    le = dcop.decompose(x2[:, :1], smoothing=0.08)
    ge = le.exchange_remote(x2, m2)
    assert torch.equal(ge[0], rb[0]) and torch.equal(ge[1], rb[1])
    assert not any(scratch.private.values()), scratch.private


def case_comm_trace(be, comm):
    """the record of the data-path collectives that bench.py --gpus N reports (pmesh_amd.comm.trace):
    a slab FFT round trip moves the whole half spectrum twice, (P-1)/P of it off rank"""
    from pmesh_amd import comm as C
    from pmesh_amd.pm import ParticleMesh
    if not hasattr(comm, '_dist'):
        return                                       # thread ranks: no torch.distributed collectives
    N = 16
    pm = ParticleMesh(BoxSize=1.0, Nmesh=[N, N, N], comm=comm, dtype='f8', np=[comm.size])
    rho = pm.create('real', value=1.0)
    rec = C.trace(True)
    back = rho.r2c().c2r()
    C.trace(False)
    assert abs(float(numpy.asarray(back.value).mean()) - 1.0) < 1e-12
    s = C.trace_summary(rec)
    assert s['collectives'] >= 2 and s['bytes_sent'] > 0 and s['sync_ms'] + s['overlapped_window_ms'] > 0
    # both transposes together: 2 x (local share of the N x N x (N/2+1) complex128 spectrum) x (P-1)/P,
    # up to the uneven split of the last rank
    P = comm.size
    total = comm.allreduce(float(s['bytes_sent']))
    full = 2 * N * N * (N // 2 + 1) * 16.0 * (P - 1) / P
    assert 0.7 * full <= total <= 1.3 * full, (total, full)


def case_mesh_of_four_dimensions(be, comm):
    """[r6] a 4-d mesh on several ranks (the reference builds one under mpirun in its CI: pm.reshape(Nmesh=[8, 8, 8, 8],
    BoxSize=8.0), pmesh/tests/test_pm.py:372-389): slabs along axis 0, the local stage one batched 3-d rocFFT plan
    (Plan._execute_slab), the particles classified along the one split axis (GridND._split_axes), the generic n-d
    window kernels on the slab block.  r2c == rfftn / prod(N) and c2r back (numpy.fft: the FFT oracle),
    paint / readout through a layout == the oracle's n-d painter on the whole mesh; the untransposed spectrum; f4."""
    from pmesh_amd.pm import ParticleMesh, UntransposedComplexField
    from oracle import oracle as O
    pm3 = ParticleMesh(BoxSize=8.0, Nmesh=[8, 8, 8], comm=comm, dtype='f8', np=[comm.size])
    pm4 = pm3.reshape(Nmesh=[8, 8, 8, 8], BoxSize=8.0)
    assert pm4.ndim == 4 and list(pm4.np) == [comm.size]
    with pytest_raises(ValueError):
        pm3.reshape(Nmesh=[8, 8, 8, 8])
    N, L = [8, 6, 4, 10], [8.0, 3.0, 2.0, 5.0]
    data = numpy.random.RandomState(3).normal(size=N)
    ref = numpy.fft.rfftn(data) / numpy.prod(N)
    rs = numpy.random.RandomState(40 + comm.rank)
    npart = 150 + 20 * comm.rank
    pos = rs.uniform(-1.0, 9.0, size=(npart, 4)) * numpy.array(L) / 8.0
    mass = rs.uniform(0.5, 1.5, size=npart)
    all_pos = numpy.concatenate(comm.allgather(pos), axis=0)
    all_mass = numpy.concatenate(comm.allgather(mass), axis=0)
    for dtype, tol in (('f8', 1e-13), ('f4', 5e-6)):
        pm = ParticleMesh(BoxSize=L, Nmesh=N, comm=comm, dtype=dtype)
        assert list(pm.np) == [comm.size]
        real = pm.create('real', value=data[pm.create('real').slices].astype(dtype))
        ck = real.r2c()
        got = numpy.asarray(ck)
        scale = comm.allreduce(float(abs(ref).max()), op='max')
        assert float(abs(got - ref[ck.slices]).max()) <= tol * scale if got.size else True
        back = numpy.asarray(ck.c2r())
        assert float(abs(back - data[real.slices]).max()) <= 8 * tol * float(abs(data).max()) if back.size else True
        cu = numpy.asarray(real.r2c(out=UntransposedComplexField(pm)))
        assert float(abs(cu - ref[UntransposedComplexField(pm).slices]).max()) <= tol * scale if cu.size else True
    pm = ParticleMesh(BoxSize=L, Nmesh=N, comm=comm, dtype='f8', resampler='cic')
    aff = O.Affine(4, scale=numpy.array(N) / numpy.array(L), period=N)
    truth = numpy.zeros(N)
    O.Window('tunedcic').paint(truth, all_pos, mass=all_mass, transform=aff)
    layout = pm.decompose(pos)
    rho = pm.paint(pos, mass=mass, layout=layout)
    assert_allclose(gather_field(comm, rho, tuple(N)), truth, rtol=0, atol=1e-12 * abs(truth).max())
    fld = pm.create('real', value=data[rho.slices])
    for grad in (None, 2):
        want = O.Window('tunedcic').readout(data, pos, transform=aff, diffdir=grad)
        assert_allclose(fld.readout(pos, layout=layout, gradient=grad), want, rtol=0, atol=1e-12 * max(1.0, abs(want).max()))
    with pytest_raises(NotImplementedError):
        ParticleMesh(BoxSize=4.0, Nmesh=[4, 4, 4, 4, 4], comm=comm, dtype='f8')


def case_tile_order_on_rank_blocks(be, comm):
    """[r6] ParticleMesh.tile_order on a rank's block (slabs and pencils): a permutation of the rank's rows — on the GPU
    read off the bin plan of the block, rows that touch no local cell last — and painting / reading the re-sorted rows
    through a layout gives what the rows gave as they were"""
    from pmesh_amd import window as W
    from pmesh_amd.pm import ParticleMesh
    N, L = (64, 32, 64), 50.0
    rs = numpy.random.RandomState(60 + comm.rank)
    pos_h = rs.uniform(-5.0, 55.0, size=(40000 + 500 * comm.rank, 3))
    old = (W.BINNED, W.BINNED_MIN_PARTICLES)
    try:
        comm.Barrier()
        W.BINNED, W.BINNED_MIN_PARTICLES = 'auto', 1000
        comm.Barrier()
        for np_ in ([comm.size], None):
            pm = ParticleMesh(BoxSize=L, Nmesh=N, comm=comm, dtype='f8', resampler='tsc', np=np_)
            pos = torch.from_numpy(pos_h).to(be.device)
            o = pm.tile_order(pos)
            oh = o.cpu().numpy()
            assert oh.dtype == numpy.int64 and numpy.array_equal(numpy.sort(oh), numpy.arange(len(pos_h)))
            a = pm.paint(pos, layout=pm.decompose(pos))
            ps = pos[o].contiguous()
            lay = pm.decompose(ps)
            b = pm.paint(ps, layout=lay)
            av, bv = numpy.asarray(a), numpy.asarray(b)
            assert float(abs(av - bv).max()) <= 1e-12 * max(1.0, float(abs(av).max())) if av.size else True
            ra = numpy.asarray(a.readout(pos, layout=pm.decompose(pos)).cpu())
            rb = numpy.asarray(a.readout(ps, layout=lay).cpu())
            assert float(abs(ra[oh] - rb).max()) <= 1e-12 * max(1.0, float(abs(ra).max()))
    finally:
        comm.Barrier()
        W.BINNED, W.BINNED_MIN_PARTICLES = old
        W.clear_bin_cache()


class pytest_raises(object):
    """(the cases also run outside pytest: python tests/mp_cases.py under torch.distributed.run)"""
    def __init__(self, exc):
        self.exc = exc

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        assert et is not None and issubclass(et, self.exc), 'expected %s, got %r' % (self.exc.__name__, ev)
        return True


CASES = [case_mesh_of_four_dimensions, case_tile_order_on_rank_blocks, case_comm_trace, case_async_ghost_exchange, case_readout_into_strided_and_float_out, case_length_check_is_collective, case_promote_and_pack, case_pencil,
         case_pencil_pipelined_equals_single_exchange, case_pencil_row_split_equals_two_sweeps, case_pencil_untransposed_and_c2c, case_deferred_last_pass_on_slabs, case_deferred_last_pass_on_pencils, case_exchange, case_period_empty_ranks, case_paint_distributed_equals_serial, case_halo_merge_left_to_the_slab_row_pass,
         case_ghosts_only_equals_literal, case_slab_fft, case_pipelined_equals_single_exchange, case_uneven_blocks_decide_alike, case_fused_transfer_slab, case_whitenoise, case_ravel_resample_preview, case_untransposed, case_c2c, case_cycle]


def main():
    be, comm = setup()
    names = sys.argv[1:]
    for case in CASES:
        if names and case.__name__ not in names:
            continue
        case(be, comm)
        comm.Barrier()
        if comm.rank == 0:
            print('ok', case.__name__, 'on', comm.size, 'ranks', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
