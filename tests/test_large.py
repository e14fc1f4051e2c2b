"""Maximum sizes (SURVEY.md 8c edge cases): a mesh with more than 2^32 cells on one GPU.

2048^3 in fp32 is 34 GB — ordinary for the 288 GB of an MI355X, and beyond 32-bit cell and element
indices everywhere: the tile-binned and the direct paint/readout must agree, mass must be conserved
with the last planes holding their share, and r2c -> c2r must return the field (rocFFT's own 3-d
transform returns wrong numbers at this size, which is why pmx_fft_create refuses it and the own
row/column kernels cover 2048).  Size-independent properties only: no oracle run at this size.
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_mesh_beyond_32bit_cells():
    from pmesh_amd import backend, window
    from pmesh_amd._arrays import vec
    from pmesh_amd.pm import ParticleMesh
    backend.reset()
    be = backend.get()
    free, _ = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip('needs ~100 GB of free HBM')
    N, side, L = 2048, 640, 1000.0
    pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f4')
    n = side ** 3
    pos = torch.empty((n, 3), dtype=torch.float64, device=be.device)
    pv = vec(pos)
    be.call('synth_uniform', C.byref(pv), side, L, 7, 0, n, be.stream())
    saved = window.BINNED
    try:
        window.BINNED = 'always'
        a = pm.paint(pos)
        assert abs(a.csum() / n - 1) < 1e-6
        window.BINNED = 'never'
        b = pm.paint(pos)
        assert abs(b.csum() / n - 1) < 1e-6
        d = 0.0
        for i in range(0, N, 256):                  # plane chunks: no 34 GB temporaries
            d = max(d, float((a.value[i:i + 256] - b.value[i:i + 256]).abs().max()))
        assert d < 2e-6, d
        share = float(a.value[N - 64:].double().sum()) / (n * 64.0 / N)
        assert abs(share - 1) < 1e-2
        window.BINNED = 'always'
        ra = a.readout(pos)
        window.BINNED = 'never'
        rb = a.readout(pos)
        assert float((ra - rb).abs().max()) < 2e-6
        del b, rb, ra
        one = pm.create('real', value=1.0)
        window.BINNED = 'always'
        assert float((one.readout(pos) - 1).abs().max()) < 1e-6
        del one
    finally:
        window.BINNED = saved
    hi, lo = a.value[N - 8:].clone(), a.value[:8].clone()
    back = a.r2c(out=Ellipsis).c2r(out=Ellipsis)
    assert float((back.value[N - 8:] - hi).abs().max()) < 1e-5
    assert float((back.value[:8] - lo).abs().max()) < 1e-5
    backend.reset()


def test_rocfft_refuses_spans_it_gets_wrong():
    from pmesh_amd import backend, _abi
    backend.reset()
    be = backend.get()
    N = 2048
    with pytest.raises(backend.PmxError, match='2\\^32'):
        be.fft_create(_abi.PMX_FFT_R2C, 4, [N, N, N], [N * (N + 2), N + 2, 1], N * N * (N + 2),
                      [N * (N // 2 + 1), N // 2 + 1, 1], N * N * (N // 2 + 1), 1, 1.0, True)
    backend.reset()
