#!/usr/bin/env python3
"""direct (atomic) vs tile-binned paint+readout as a function of the batch size, for a
uniform batch and for a ghost-like batch (a band one cell thick at a slab face)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pmesh_amd import backend, window
from pmesh_amd.pm import ParticleMesh

be = backend.get()
N, L = 512, 1000.0
pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler=sys.argv[1] if len(sys.argv) > 1 else 'cic')
rho = pm.create('real')
# a 64-plane local block like one of 8 slab ranks
from pmesh_amd.window import Affine
blk = torch.zeros((64, N, N), dtype=torch.float64, device=be.device)
aff = Affine(3, scale=N / L, translate=[-128, 0, 0], period=N)
g = torch.Generator(device=be.device); g.manual_seed(1)

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for shape in ('uniform', 'band'):
    for lg in range(16, 24):
        n = 1 << lg
        pos = torch.rand((n, 3), dtype=torch.float64, device=be.device, generator=g) * L
        if shape == 'uniform':
            pos[:, 0] = (128 + pos[:, 0] / L * 64) * L / N
        else:
            pos[:, 0] = (127.0 + pos[:, 0] / L * 1.0) * L / N
        res = {}
        for mode in ('never', 'always'):
            window.BINNED = mode
            def run():
                window.clear_bin_cache()
                pm.resampler.paint(blk, pos, transform=aff)
                return pm.resampler.readout(blk, pos, transform=aff)
            try:
                res[mode] = timeit(run)
            except Exception as e:
                res[mode] = float('nan')
        print('%-8s n=2^%d direct %8.1f us  binned %8.1f us' % (shape, lg, res['never'], res['always']), flush=True)
