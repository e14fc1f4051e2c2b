#!/usr/bin/env python3
"""The FIRST build of a bin plan (two passes: count, fill) on rows in tile order under strong clustering: a fraction f of
512^3 particles in `nb` Gaussian blobs of sigma cells.  PMESH_AMD_LIBRARY selects the build (scripts/build_variant.sh
twopass0 "-DPMX_LEAN_TWOPASS=0": the per-wave count and scatter kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pmesh_amd import backend, window
from pmesh_amd.pm import ParticleMesh
be = backend.get()
N, L = 512, 1000.0
pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler='cic')
rho = pm.create('real')
n = N ** 3
g = torch.Generator(device=be.device); g.manual_seed(3)
window.SORTED = 'never'
for f, nb, sigma in ((0.0, 1, 1.0), (0.3, 4096, 2.0), (0.5, 64, 2.0), (0.5, 64, 0.5), (0.9, 8, 1.0)):
    pos = torch.rand((n, 3), dtype=torch.float64, device=be.device, generator=g) * L
    nc = int(f * n)
    if nc:
        centres = torch.rand((nb, 3), dtype=torch.float64, device=be.device, generator=g) * L
        which = torch.randint(0, nb, (nc,), device=be.device, generator=g)
        pos[:nc] = (centres[which] + torch.randn((nc, 3), dtype=torch.float64, device=be.device, generator=g) * sigma * L / N) % L
    pos = pos[pm.tile_order(pos)].contiguous()
    ts = []
    for k in range(4):
        window.bin_cache().destroy(be)
        window.clear_bin_cache()
        pm.resampler.prebin(rho.value, pos, pm.affine)          # (allocations)
        torch.cuda.synchronize()
        window.clear_bin_cache()
        for e in window.bin_cache().entries:                     # forget the history: the next build is a first one
            pass
        window.bin_cache().destroy(be)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.record(); pm.resampler.prebin(rho.value, pos, pm.affine); b.record()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    window.clear_bin_cache()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); pm.resampler.prebin(rho.value, pos, pm.affine); b.record(); torch.cuda.synchronize()
    print('f=%.1f blobs=%d sigma=%.1f: first build (with its allocations) %.2f ms wall; rebuild %.2f ms' % (f, nb, sigma, min(ts), a.elapsed_time(b)), flush=True)
    del pos
