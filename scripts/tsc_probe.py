#!/usr/bin/env python3
"""TSC / CIC / PCS tile-kernel times vs the jitter pattern of a lattice-ordered set: does the
cost of TSC come from neighbouring lanes that round to the same base cell (same-address LDS atomics)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pmesh_amd import backend, window
from pmesh_amd.pm import ParticleMesh
be = backend.get()
N, L = 512, 1000.0
g = torch.Generator(device=be.device); g.manual_seed(1)
idx = torch.arange(N, device=be.device, dtype=torch.float64)
lat = torch.stack(torch.meshgrid(idx, idx, idx, indexing='ij'), dim=-1).reshape(-1, 3)
def timeit(fn, k=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
for name in ('cic', 'tsc', 'pcs'):
    pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler=name)
    rho = pm.create('real')
    for off, jit in ((0.5, 0.8), (0.25, 0.4), (0.75, 0.4), (0.5, 0.0), (0.0, 0.8)):
        pos = ((lat + off + jit * (torch.rand(lat.shape, dtype=torch.float64, device=be.device, generator=g) - 0.5)) * (L / N)) % L
        window.clear_bin_cache()
        pm.resampler.prebin(rho.value, pos, pm.affine)
        tp = timeit(lambda: pm.paint(pos, out=rho))
        tr = timeit(lambda: rho.readout(pos))
        print('%s offset %.2f jitter %.1f: paint %.3f ms readout %.3f ms' % (name, off, jit, tp, tr), flush=True)
        del pos
