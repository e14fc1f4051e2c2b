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


@pytest.mark.parametrize('config', ['C4', 'C5'])
def test_multi_gpu_configs_decomposed_on_thread_ranks(config):
    """BASELINE.json configs 4 and 5 in their DECOMPOSED form, at the largest size one GPU holds, with the
    8 ranks as threads of one process driving the real kernels (tests/distributed_cycle.py):
      C4: 1024^3 mesh, 1024^3 uniform particles, CIC fp64, slab np=[8], ghosts-only routing, pipelined
          transposes, fused transfer — config 4 exactly, minus the wire;
      C5: 2 x 4 pencils, PCS, 2 x 1024^3 Zel'dovich-displaced particles with a per-particle fp64 mass on a
          1024^3 mesh (config 5 is 2048^3 / 2 x 2048^3: 8x this; positions in fp32 so that the distributed
          set, its one-rank copy and both sets of bin lists fit the 288 GB together).
    Every rank's readout equals the one-rank cycle (pinned to the oracle at this size by the test above) to
    1e-11 of the result's scale, and the first planes of rank 0's painted block equal the oracle's paint of
    every particle of every rank that touches them to 1e-12."""
    import os
    import subprocess
    import sys
    # the child process needs the memory this (pytest) process still holds in its allocator cache and plans
    from pmesh_amd import backend, window
    window.bin_cache().destroy(backend.get()) if torch.cuda.is_available() else None
    backend.reset()
    import gc
    gc.collect()                                   # (fields and meshes of earlier tests that wait in reference cycles)
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    need = 130e9 if config == 'C4' else 240e9      # measured peaks 114 / 225 GB (the "peak device memory" line the child prints; profiles/r04_b_multirank8_*.log)
    if free < need:
        pytest.skip('needs %.0f GB of free HBM, %.0f are free' % (need / 1e9, free / 1e9))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'tests', 'distributed_cycle.py'), '--ranks', '8', '--mesh', '1024',
           '--steps', '1', '--warmup', '1', '--check', '1', '--oracle-planes', '2']
    if config == 'C4':
        cmd += ['--window', 'cic']
    else:
        # (--migrate 1, as bench.py --gpus N: without it three quarters of the rows of every rank travel as "ghosts" on the
        # pencil mesh — that case runs at 512^3 in test_multirank.py — and this one needs 300 of the device's 309 GB)
        cmd += ['--np', '2x4', '--window', 'pcs', '--data', 'clustered', '--double', '1', '--mass', 'array',
                '--pos-dtype', 'f4', '--migrate', '1']
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=1500)
    print(out.stdout[-2000:])
    assert out.returncode == 0, out.stdout[-2000:] + '\n' + out.stderr[-4000:]
    assert 'vs one rank' in out.stdout and 'vs oracle' in out.stdout and 'peak device memory' in out.stdout


def _slab_subset(pos, N, L, k0, nplanes, margin):
    """rows of `pos` whose x grid coordinate lies within `margin` cells of planes [k0, k0 + nplanes)"""
    out = []
    step = 1 << 29                                   # torch.nonzero cannot take more than 2^31 elements
    for i in range(0, pos.shape[0], step):
        gx = pos[i:i + step, 0] * (N / L)
        sel = (gx >= k0 - margin) & (gx < k0 + nplanes + margin)
        out.append(torch.nonzero(sel)[:, 0] + i)
    return torch.cat(out)


@pytest.mark.parametrize('config', ['C4', 'C5'])
def test_multi_gpu_configs_per_gpu_load_on_one_gpu(config):
    """BASELINE.json configs 4 and 5 need 8 GPUs; what ONE of them holds fits one MI355X and is run here
    at full size through the production path, checked by size-independent properties and, on slabs of
    planes, against the CPU oracle (every particle whose window touches the slab):
      C4: 1024^3 mesh, 1024^3 uniform particles, CIC, fp64 (the whole problem of config 4);
      C5: 1024^3 cells, 2 x 1024^3 Zel'dovich-displaced particles (more than 2^31 rows), PCS,
          per-particle fp64 mass (the per-GPU share of config 5's 2048^3 / 2 x 2048^3)."""
    from oracle import oracle as O
    from pmesh_amd import backend, window
    from pmesh_amd._arrays import vec
    from pmesh_amd.pm import ParticleMesh
    from pmesh_amd.transfer import Transfer
    import numpy
    backend.reset()
    be = backend.get()
    free, _ = torch.cuda.mem_get_info()
    need = 60e9 if config == 'C4' else 150e9
    if free < need:
        pytest.skip('needs %.0f GB of free HBM' % (need / 1e9))
    N, L = 1024, 1000.0
    name = 'cic' if config == 'C4' else 'pcs'
    S = 2 if config == 'C4' else 4
    nlat = N ** 3
    copies = 1 if config == 'C4' else 2
    pos = torch.empty((copies * nlat, 3), dtype=torch.float64, device=be.device)
    if config == 'C4':
        pv = vec(pos)
        be.call('synth_uniform', C.byref(pv), N, L, 42, 0, nlat, be.stream())
        mass, mass_h = 1.0, None
        mtot = float(nlat)
    else:
        modes = O.zeldovich_modes(N, L)
        for c in range(copies):
            pv = vec(pos[c * nlat:(c + 1) * nlat])
            be.call('synth_clustered', C.byref(pv), N, L, modes.ctypes.data_as(C.POINTER(C.c_double)), len(modes),
                    0.5 * c, 0, nlat, be.stream())
        mass = 0.5 + (torch.arange(copies * nlat, device=be.device, dtype=torch.int64) % 1024).to(torch.float64) / 1024.0
        mtot = float(mass.sum())
    pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler=name)
    saved, saved_exact = window.BINNED, window.EXACT
    try:
        window.BINNED = 'auto'
        window.EXACT = True            # (the readout below is compared with the oracle's bit for bit)
        rho = pm.paint(pos, mass=mass)
        assert any(e[3] for e in window.bin_cache().entries), 'the tile-binned path was not taken'
        assert abs(rho.csum() / mtot - 1) < 1e-10                      # mass conservation
        peak = float(rho.value.max())
        # paint against the oracle on two slabs of 4 planes (one at the periodic wrap)
        oaff = lambda k0: O.Affine(3, scale=N / L, translate=[-k0, 0, 0], period=N)
        for k0 in (0, 517):
            sub = _slab_subset(pos, N, L, k0, 4, S + 1)
            if k0 == 0:                                                # + the particles that wrap onto plane 0
                sub = torch.cat([sub, _slab_subset(pos, N, L, N, 4, S + 1)])
            ph = pos[sub].cpu().numpy()
            mh = mass[sub].cpu().numpy() if config == 'C5' else 1.0
            want = numpy.zeros((4, N, N))
            O.Window('tuned' + name).paint(want, ph, mass=mh, transform=oaff(k0))
            got = rho.value[k0:k0 + 4].cpu().numpy()
            err = abs(got - want).max() / max(1.0, abs(want).max())
            print('%s paint planes %d..%d vs oracle: %.2e (%d particles)' % (config, k0, k0 + 3, err, len(ph)))
            assert err < 1e-12
        # binned == direct
        window.BINNED = 'never'
        direct = pm.paint(pos, mass=mass)
        d = 0.0
        for i in range(0, N, 128):
            d = max(d, float((rho.value[i:i + 128] - direct.value[i:i + 128]).abs().max()))
        assert d <= 1e-12 * peak, (d, peak)
        del direct
        window.BINNED = 'auto'
        one = pm.create('real', value=1.0)                              # partition of unity
        v = one.readout(pos)
        assert float((v - 1.0).abs().max()) < 1e-13
        del one, v
        # the FFT returns the field
        keep = rho.value[300:304].clone()
        ck = rho.r2c(out=Ellipsis)
        back = ck.c2r(out=Ellipsis)
        assert float((back.value[300:304] - keep).abs().max()) < 1e-11 * peak
        # the cycle: transfer fused into c2r, readout bit-identical to the oracle's on the device's field
        f = back.r2c(out=Ellipsis).c2r(out=Ellipsis, transfer=Transfer.dx1(0)).readout(pos)
        assert bool(torch.isfinite(f).all())
        k0 = 771
        sub = _slab_subset(pos, N, L, k0 + S, 2, 0.0)                   # windows inside planes [k0, k0 + 2 S + 2)
        ph = pos[sub].cpu().numpy()
        planes = back.value[k0:k0 + 2 * S + 3].cpu().numpy()
        want = O.Window('tuned' + name).readout(planes, ph, transform=oaff(k0))
        got = f[sub].cpu().numpy()
        assert numpy.array_equal(got, want), abs(got - want).max()
        print('%s readout of %d particles bit-identical to the oracle' % (config, len(ph)))
    finally:
        window.BINNED, window.EXACT = saved, saved_exact
        window.clear_bin_cache()
        backend.reset()
