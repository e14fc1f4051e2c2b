"""The LDS-resident row / column FFT kernels (csrc/pmx_colfft.hip) on their own,
against numpy.fft, for every length they are built for, both precisions, ragged
batch sizes (tiles with fewer than W columns / rows), scaling and the fused transfer.
Runs against the HIP library under -m gpu and against the numpy double otherwise
(which then only checks the test itself and the host plumbing)."""
import numpy
import pytest
import torch
from numpy.testing import assert_allclose

from pmesh_amd import _abi


def rel(a, b):
    return numpy.sqrt((abs(a - b) ** 2).sum() / max((abs(b) ** 2).sum(), 1e-300))


@pytest.mark.parametrize('elsize,tol', [(8, 2e-15), (4, 1e-6)])
@pytest.mark.parametrize('N', [64, 128, 256, 512, 1024, 2048, 192, 384, 768, 1536, 320, 640, 1280])
def test_colfft_lengths(be, elsize, tol, N):
    if not be.colfft_supported(N, elsize):
        pytest.skip('length not built for this precision')
    cdt = 'c16' if elsize == 8 else 'c8'
    rs = numpy.random.RandomState(N)
    for A, B in ((1, 37), (3, 8), (2, 129)):
        x = (rs.normal(size=(A, N, B)) + 1j * rs.normal(size=(A, N, B))).astype(cdt)
        for inverse in (False, True):
            t = torch.view_as_real(torch.from_numpy(x.copy())).reshape(-1).to(be.device)
            be.colfft(elsize, inverse, t, A, N, B, scale=0.5)
            got = t.cpu().numpy().view(cdt).reshape(A, N, B)
            want = (numpy.fft.ifft(x.astype('c16'), axis=1) * N if inverse else numpy.fft.fft(x.astype('c16'), axis=1)) * 0.5
            assert rel(got, want) < tol * numpy.log2(N), (N, A, B, inverse)


@pytest.mark.parametrize('elsize,N', [(8, 1024), (4, 1024), (8, 768), (8, 640), (8, 512)])
def test_colfft_more_tiles_than_compute_units(be, oracle, elsize, N):
    """Batches of several hundred tiles per plane with a ragged last one: where a tile fills a CU the column kernel is
    a persistent workgroup that prefetches its next tile (first tile peeled, whole and ragged tiles stored by different
    paths) — the small batches of test_colfft_lengths give every workgroup one tile.  Then the round-trip kernel on the
    same batch (its own tile width at these lengths) against the two passes it replaces, bit for bit."""
    if be.name != 'hip':
        pytest.skip('the persistent forms exist in the HIP kernels only')
    be.colfft_configure(1)      # (a multi-rank test earlier in this process may have switched them off)
    if not be.colfft_supported(N, elsize):
        pytest.skip('length not built for this precision')
    cdt = 'c16' if elsize == 8 else 'c8'
    tol = 2e-15 if elsize == 8 else 1e-6
    W = 128 // (2 * elsize)
    A, B = 2, W * 300 + 3
    rs = numpy.random.RandomState(N + elsize)
    x = (rs.normal(size=(A, N, B)) + 1j * rs.normal(size=(A, N, B))).astype(cdt)
    for inverse in (False, True):
        t = torch.view_as_real(torch.from_numpy(x.copy())).reshape(-1).to(be.device)
        be.colfft(elsize, inverse, t, A, N, B, scale=0.5)
        got = t.cpu().numpy().view(cdt).reshape(A, N, B)
        want = (numpy.fft.ifft(x.astype('c16'), axis=1) * N if inverse else numpy.fft.fft(x.astype('c16'), axis=1)) * 0.5
        assert rel(got, want) < tol * numpy.log2(N), (N, inverse)
        for a in range(A):      # the last column of a plane (the ragged tile) and the first of the next
            assert rel(got[a, :, -3:], want[a, :, -3:]) < tol * numpy.log2(N) * 4
        # the same kernels with one workgroup per tile (what plans on several ranks select): same bits
        be.colfft_configure(0)
        t2 = torch.view_as_real(torch.from_numpy(x.copy())).reshape(-1).to(be.device)
        be.colfft(elsize, inverse, t2, A, N, B, scale=0.5)
        be.colfft_configure(1)
        assert torch.equal(t, t2)
    if not be.colfft_roundtrip_supported(N, elsize):
        return
    n1, n2 = 3, B // 3
    assert n1 * n2 == B
    nmesh, box, start = (N, 6, 2 * n2), (100.0, 50.0, 70.0), (0, 3, 0)
    tr = oracle.make_transfer(laplace_pow=-1, grad_dir=0)
    one = torch.view_as_real(torch.from_numpy(x[0].copy())).reshape(-1).to(be.device)
    two = one.clone()
    be.colfft_roundtrip(elsize, one, N, B, scale=1.0 / N, transfer=tr, n1=n1, n2=n2, start=start, nmesh=nmesh, boxsize=box)
    be.colfft(elsize, False, two, 1, N, B, scale=1.0 / N)
    be.colfft(elsize, True, two, 1, N, B, transfer=tr, n1=n1, n2=n2, start=start, nmesh=nmesh, boxsize=box)
    assert torch.equal(one, two)


@pytest.mark.parametrize('elsize,tol', [(8, 2e-15), (4, 1e-6)])
@pytest.mark.parametrize('n', [128, 256, 512, 1024, 2048, 384, 768, 1536, 640, 1280])
def test_rowfft_lengths(be, elsize, tol, n):
    if not be.rowfft_supported(n, elsize):
        pytest.skip('length not built for this precision')
    rdt, cdt = ('f8', 'c16') if elsize == 8 else ('f4', 'c8')
    rs = numpy.random.RandomState(n)
    for nrows, pitch in ((5, n // 2 + 1), (19, n // 2 + 8)):
        buf = numpy.zeros((nrows, 2 * pitch), dtype=rdt)
        x = rs.normal(size=(nrows, n)).astype(rdt)
        buf[:, :n] = x
        t = torch.from_numpy(buf.copy()).reshape(-1).to(be.device)
        be.rowfft(elsize, False, t, nrows, n, pitch, scale=2.0)
        got = t.cpu().numpy().view(cdt).reshape(nrows, pitch)[:, :n // 2 + 1]
        want = numpy.fft.rfft(x.astype('f8'), axis=1) * 2.0
        assert rel(got, want) < tol * numpy.log2(n)
        # and back (unnormalised: n * x)
        be.rowfft(elsize, True, t, nrows, n, pitch, scale=1.0 / (2.0 * n))
        back = t.cpu().numpy().reshape(nrows, 2 * pitch)[:, :n]
        assert rel(back, x) < 2 * tol * numpy.log2(n)


@pytest.mark.parametrize('elsize,tol', [(8, 2e-15), (4, 1e-6)])
@pytest.mark.parametrize('n', [256, 2048, 384, 640])
def test_rowfft_c2r_ignores_imag_of_dc_and_nyquist(be, elsize, tol, n):
    """c2r of rows whose DC / Nyquist modes carry an imaginary part (a spectrum that is not
    exactly Hermitian, e.g. after a gradient transfer on the Nyquist planes): FFTW's c2r behind
    PFFT and numpy.fft.irfft ignore those imaginary parts; so must this kernel."""
    rdt, cdt = ('f8', 'c16') if elsize == 8 else ('f4', 'c8')
    # (256: threads that keep one position along the row; 2048 in double, 384, 640: threads that walk along it and carry
    # the Nyquist mode in the DC slot)
    if not be.rowfft_supported(n, elsize):
        pytest.skip('length not built for this precision')
    nrows, pitch = 11, n // 2 + 8
    rs = numpy.random.RandomState(4)
    X = (rs.normal(size=(nrows, n // 2 + 1)) + 1j * rs.normal(size=(nrows, n // 2 + 1))).astype(cdt)
    buf = numpy.zeros((nrows, pitch), dtype=cdt)
    buf[:, :n // 2 + 1] = X
    t = torch.view_as_real(torch.from_numpy(buf.copy())).reshape(-1).to(be.device)
    be.rowfft(elsize, True, t, nrows, n, pitch)
    got = t.cpu().numpy().view(rdt).reshape(nrows, 2 * pitch)[:, :n]
    want = numpy.fft.irfft(X.astype('c16'), n=n, axis=1) * n
    assert rel(got, want) < 2 * tol * numpy.log2(n)


def _mode_ranges(n, nparts, kind):
    M1 = n // 2 + 1
    if kind == 'even':                    # what fft.block_edges gives a row group of nparts ranks
        return [(M1 * q) // nparts for q in range(nparts + 1)]
    if kind == 'empty':                   # more ranks than the split leaves modes for some: empty blocks, also at the end
        e = [0, 0, 5, 5, M1 - 1, M1, M1, M1]
        return e[:nparts] + [M1] if nparts < len(e) else e + [M1] * (nparts + 1 - len(e))
    rs = numpy.random.RandomState(n + nparts)
    return [0] + sorted(int(v) for v in rs.randint(0, M1 + 1, size=nparts - 1)) + [M1]


@pytest.mark.parametrize('elsize,tol', [(8, 2e-15), (4, 1e-6)])
@pytest.mark.parametrize('n,nparts,kind', [(128, 1, 'even'), (128, 4, 'even'), (256, 3, 'ragged'), (512, 4, 'even'),
                                           (512, 16, 'ragged'), (1024, 2, 'even'), (2048, 8, 'ragged'), (256, 7, 'empty'),
                                           (512, 5, 'empty'), (2048, 4, 'even'), (384, 4, 'even'), (768, 3, 'ragged'),
                                           (1536, 2, 'even'), (640, 4, 'ragged'), (1280, 5, 'empty')])
def test_rowfft_split(be, elsize, tol, n, nparts, kind):
    """the row pass with the last-axis split of a pencil transform's first transpose on it: bit for bit what the plain
    row pass followed by slab_pack (forward) / slab_unpack followed by the plain pass (inverse) deliver, the input kept"""
    assert be.rowfft_split_supported(n, elsize, nparts) == be.rowfft_supported(n, elsize)
    if not be.rowfft_supported(n, elsize):
        pytest.skip('length not built for this precision')
    rdt, cdt = ('f8', 'c16') if elsize == 8 else ('f4', 'c8')
    M1 = n // 2 + 1
    e = _mode_ranges(n, nparts, kind)
    assert len(e) == nparts + 1 and e[0] == 0 and e[-1] == M1
    rs = numpy.random.RandomState(n * 31 + nparts)
    for nrows, pitch in ((5, M1), (19, M1 + 7), (64, M1)):
        buf = numpy.zeros((nrows, 2 * pitch), dtype=rdt)
        x = rs.normal(size=(nrows, n)).astype(rdt)
        buf[:, :n] = x
        src = torch.from_numpy(buf.copy()).reshape(-1).to(be.device)
        keep = src.clone()
        dst = torch.full((2 * nrows * M1 + 8,), 7.0, dtype=src.dtype, device=be.device)
        be.rowfft_split(elsize, False, src, dst, nrows, n, pitch, e, scale=2.0)
        assert torch.equal(src, keep)                               # input preserved
        assert bool((dst[2 * nrows * M1:] == 7.0).all())            # nothing written past the blocks
        got = dst[:2 * nrows * M1].cpu().numpy().view(cdt)
        want = numpy.fft.rfft(x.astype('f8'), axis=1) * 2.0
        for q in range(nparts):
            blk = got[nrows * e[q]:nrows * e[q + 1]].reshape(nrows, e[q + 1] - e[q])
            if blk.size:
                assert rel(blk, want[:, e[q]:e[q + 1]]) < tol * numpy.log2(n), q
        # the two-sweep form, bit for bit
        plain = keep.clone()
        be.rowfft(elsize, False, plain, nrows, n, pitch, scale=2.0)
        dense = torch.view_as_real(torch.view_as_complex(plain.view(nrows, pitch, 2))[:, :M1].contiguous()).reshape(-1)
        packed = torch.zeros_like(dense)
        be.slab_pack(dense, packed, nrows, M1, 1, e, 2 * elsize)
        assert torch.equal(packed, dst[:2 * nrows * M1])
        # inverse: blocks -> rows (the pad of every row is not written: compare the n reals)
        back = torch.full_like(keep, 3.0)
        be.rowfft_split(elsize, True, dst, back, nrows, n, pitch, e, scale=1.0 / (2.0 * n))
        assert rel(back.cpu().numpy().reshape(nrows, 2 * pitch)[:, :n], x) < 2 * tol * numpy.log2(n)
        two = torch.zeros((nrows, pitch, 2), dtype=src.dtype, device=be.device)
        unp = torch.zeros_like(dense)
        be.slab_pack(dst[:2 * nrows * M1].contiguous(), unp, nrows, M1, 1, e, 2 * elsize, inverse=True)
        two[:, :M1] = unp.view(nrows, M1, 2)
        two = two.reshape(-1)
        be.rowfft(elsize, True, two, nrows, n, pitch, scale=1.0 / (2.0 * n))
        assert torch.equal(two.view(nrows, 2 * pitch)[:, :n], back.view(nrows, 2 * pitch)[:, :n])


def test_rowfft_split_rejects(be):
    src = torch.zeros(2 * 4 * 65, dtype=torch.float64, device=be.device)
    dst = torch.zeros_like(src)
    from pmesh_amd.backend import PmxError
    with pytest.raises(PmxError):                   # offsets that do not end at n/2 + 1
        be.rowfft_split(8, False, src, dst, 4, 128, 65, [0, 30, 64])
    with pytest.raises(PmxError):                   # decreasing
        be.rowfft_split(8, False, src, dst, 4, 128, 65, [0, 40, 30, 65])
    with pytest.raises(PmxError):                   # in place
        be.rowfft_split(8, False, src, src, 4, 128, 65, [0, 65])
    assert not be.rowfft_split_supported(96, 8, 2)              # (not a length of pmx_rowfft)
    assert not be.rowfft_split_supported(512, 8, _abi.PMX_MAXSEG + 1)
    assert be.rowfft_split_supported(512, 8, _abi.PMX_MAXSEG)


@pytest.mark.parametrize('elsize,tol', [(8, 2e-15), (4, 1e-6)])
@pytest.mark.parametrize('N,nsplit', [(64, 8), (128, 64), (256, 1), (512, 64), (64, 64)])
def test_colfft_split(be, elsize, tol, N, nsplit):
    """the axis-1 pass fused with the slab pack (forward) / unpack (inverse): the split layout
    is [range r][a][line in range][b], exactly what slab_pack builds from the plain array"""
    if not be.colfft_supported(N, elsize):
        pytest.skip('length not built for this precision')
    cdt = 'c16' if elsize == 8 else 'c8'
    rs = numpy.random.RandomState(N + nsplit)
    R = N // nsplit
    for A, B in ((3, 9), (1, 33)):
        x = (rs.normal(size=(A, N, B)) + 1j * rs.normal(size=(A, N, B))).astype(cdt)
        src = torch.view_as_real(torch.from_numpy(x.copy())).reshape(-1).to(be.device)
        dst = torch.zeros_like(src)
        be.colfft_split(elsize, False, src, dst, A, N, B, nsplit, scale=0.25)
        got = dst.cpu().numpy().view(cdt).reshape(R, A, nsplit, B)
        want = (numpy.fft.fft(x.astype('c16'), axis=1) * 0.25).reshape(A, R, nsplit, B).transpose(1, 0, 2, 3)
        assert rel(got, want) < tol * numpy.log2(N)
        assert numpy.array_equal(src.cpu().numpy().view(cdt).reshape(A, N, B), x)     # input preserved
        # the same as the plain pass followed by slab_pack
        if R <= 64:         # slab_pack takes at most 64 ranges (PMX_MAXRANKS)
            plain = src.clone()
            be.colfft(elsize, False, plain, A, N, B, scale=0.25)
            packed = torch.zeros_like(plain)
            be.slab_pack(plain, packed, A, N, B, [r * nsplit for r in range(R + 1)], 2 * elsize)
            assert numpy.array_equal(packed.cpu().numpy(), dst.cpu().numpy())
        # inverse: split -> plain, undoing the forward up to N * scale
        back = torch.zeros_like(src)
        be.colfft_split(elsize, True, dst, back, A, N, B, nsplit, scale=4.0 / N)
        assert rel(back.cpu().numpy().view(cdt).reshape(A, N, B), x) < 2 * tol * numpy.log2(N)


@pytest.mark.parametrize('elsize,tol', [(8, 2e-15), (4, 1e-6)])
@pytest.mark.parametrize('N,nin,nout', [(64, 16, 32), (128, 64, 16), (256, 128, 0), (64, 0, 8), (192, 64, 0)])
def test_colfft_resplit(be, elsize, tol, N, nin, nout):
    """the axis-1 pass of the pencil transform: input in the split layout of one transpose,
    output in the split layout of the other == unpack (slab_pack inverse), colfft, slab_pack"""
    if not be.colfft_supported(N, elsize):
        pytest.skip('length not built for this precision')
    cdt = 'c16' if elsize == 8 else 'c8'
    rs = numpy.random.RandomState(N + nin + nout)

    def to_split(a, ns):       # (A, N, B) -> flat split layout
        A, _, B = a.shape
        return a if ns == 0 else numpy.ascontiguousarray(a.reshape(A, N // ns, ns, B).transpose(1, 0, 2, 3))

    for A, B in ((3, 9), (1, 33)):
        x = (rs.normal(size=(A, N, B)) + 1j * rs.normal(size=(A, N, B))).astype(cdt)
        for inverse in (False, True):
            src = torch.view_as_real(torch.from_numpy(to_split(x, nin).reshape(-1).copy())).reshape(-1).to(be.device)
            dst = torch.zeros_like(src)
            be.colfft_resplit(elsize, inverse, src, dst, A, N, B, nin, nout, scale=0.5)
            y = (numpy.fft.ifft(x.astype('c16'), axis=1) * N if inverse else numpy.fft.fft(x.astype('c16'), axis=1)) * 0.5
            got = dst.cpu().numpy().view(cdt).reshape(-1)
            assert rel(got, to_split(y, nout).reshape(-1)) < tol * numpy.log2(N), (A, B, inverse)
            assert numpy.array_equal(src.cpu().numpy().view(cdt).reshape(-1), to_split(x, nin).reshape(-1))


def test_colfft_fused_transfer(be, oracle):
    """element (i0, i1, i2) * T(k) before the inverse axis-0 pass == apply_transfer + ifft"""
    N0, n1, n2 = 64, 6, 9
    nmesh = (N0, 12, 16)
    box = (100.0, 50.0, 70.0)
    start = (0, 3, 0)
    rs = numpy.random.RandomState(5)
    x = rs.normal(size=(N0, n1, n2)) + 1j * rs.normal(size=(N0, n1, n2))
    for t in (oracle.make_transfer(laplace_pow=-1, grad_dir=1), oracle.make_transfer(amplitude=-2.0, laplace_pow=-1),
              oracle.make_transfer(laplace_pow=1, grad_dir=0), oracle.make_transfer(grad_dir=2)):
        want = numpy.fft.ifft(oracle.apply_transfer(t, x.copy(), start, nmesh, box), axis=0) * N0
        d = torch.view_as_real(torch.from_numpy(x.copy())).reshape(-1).to(be.device)
        be.colfft(8, True, d, 1, N0, n1 * n2, transfer=t, n1=n1, n2=n2, start=start, nmesh=nmesh, boxsize=box)
        got = d.cpu().numpy().view('c16').reshape(N0, n1, n2)
        assert rel(got, want) < 1e-14


@pytest.mark.parametrize('elsize,tol', [(8, 2e-15), (4, 1e-6)])
def test_colfft_chunk(be, oracle, elsize, tol):
    """the axis-0 pass on a chunk [coff, coff+cw) of the last axis of an (N, n1, pitch) block,
    through the dense chunk buffer of a pipelined transpose: scatter (r2c side), gather (c2r
    side), gather with the fused transfer; everything outside the chunk stays untouched"""
    cdt = 'c16' if elsize == 8 else 'c8'
    tdt = torch.complex128 if elsize == 8 else torch.complex64
    N, n1, pitch = 64, 5, 21
    nmesh, box, start = (N, 10, 40), (30.0, 20.0, 50.0), (0, 5, 0)
    rs = numpy.random.RandomState(9)
    full_h = (rs.normal(size=(N, n1, pitch)) + 1j * rs.normal(size=(N, n1, pitch))).astype(cdt)
    for coff, cw in ((0, 8), (8, 13), (3, 7), (0, 21)):
        chunk_h = (rs.normal(size=(N, n1, cw)) + 1j * rs.normal(size=(N, n1, cw))).astype(cdt)
        # to_full: FFT of the chunk lands in the columns of the block
        full = torch.from_numpy(full_h.copy()).to(be.device)
        chunk = torch.from_numpy(chunk_h.copy()).to(be.device)
        be.colfft_chunk(elsize, False, torch.view_as_real(chunk).reshape(-1), torch.view_as_real(full).reshape(-1),
                        N, n1, cw, pitch, coff, True, scale=0.5)
        got = full.cpu().numpy()
        want = full_h.copy()
        want[:, :, coff:coff + cw] = numpy.fft.fft(chunk_h.astype('c16'), axis=0) * 0.5
        assert rel(got[:, :, coff:coff + cw], want[:, :, coff:coff + cw]) < tol * 6
        mask = numpy.ones(pitch, bool)
        mask[coff:coff + cw] = False
        assert numpy.array_equal(got[:, :, mask], full_h[:, :, mask])
        assert numpy.array_equal(chunk.cpu().numpy(), chunk_h)
        # gather: inverse FFT of the block's columns into the chunk buffer, block untouched
        full = torch.from_numpy(full_h.copy()).to(be.device)
        chunk = torch.zeros((N, n1, cw), dtype=tdt, device=be.device)
        be.colfft_chunk(elsize, True, torch.view_as_real(chunk).reshape(-1), torch.view_as_real(full).reshape(-1),
                        N, n1, cw, pitch, coff, False)
        want = numpy.fft.ifft(full_h[:, :, coff:coff + cw].astype('c16'), axis=0) * N
        assert rel(chunk.cpu().numpy(), want) < tol * 6
        assert numpy.array_equal(full.cpu().numpy(), full_h)
        # gather with the transfer function of the block's global coordinates
        t = oracle.make_transfer(laplace_pow=-1, grad_dir=2)
        be.colfft_chunk(elsize, True, torch.view_as_real(chunk).reshape(-1), torch.view_as_real(full).reshape(-1),
                        N, n1, cw, pitch, coff, False, transfer=t, start=start, nmesh=nmesh, boxsize=box)
        tk = oracle.apply_transfer(t, full_h.astype('c16'), start, nmesh, box)
        want = numpy.fft.ifft(tk[:, :, coff:coff + cw], axis=0) * N
        assert rel(chunk.cpu().numpy(), want) < tol * 6
