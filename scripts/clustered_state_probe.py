#!/usr/bin/env python3
"""Where does a strongly clustered particle set — the state scripts/nbody_long.py reaches after S steps — spend its PM
cycle?  Evolves the long run's flow untimed, then times the stages on the final positions with the device drained
between them: the rows as the run left them and re-sorted into tile order, one field and three per readout, with the
population statistics of the tiles (mean 4096 particles per tile of 8 x 16 x 32 cells at one particle per cell).
    python scripts/clustered_state_probe.py [N=512] [steps=100]"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy
import torch
import bench
from pmesh_amd import backend, window
from pmesh_amd._arrays import vec
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
be = backend.get()
L = float(N)
pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler='cic')


def lattice(rms):
    x = torch.empty((N ** 3, 3), dtype=torch.float64, device=be.device)
    pv = vec(x)
    modes = bench.zeldovich_modes(numpy, N, L, rms_cells=rms)
    be.call('synth_clustered', C.byref(pv), N, L, modes.ctypes.data_as(C.POINTER(C.c_double)), len(modes), 0.0, 0, N ** 3, be.stream())
    return x


def force(x):
    rhok = pm.paint(x).r2c(out=Ellipsis)
    return pm.readout([rhok.c2r(transfer=Transfer.force(d)) for d in range(3)], x)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t) / reps * 1e3


def report(tag, x):
    window.clear_bin_cache()
    rho0 = pm.create('real')
    _, tb = timed(lambda: (window.clear_bin_cache(), pm.resampler.prebin(rho0.value, x, pm.affine)))
    pm.resampler.prebin(rho0.value, x, pm.affine)
    rho, tp = timed(lambda: pm.paint(x))
    rhok = rho.r2c()
    comps = [rhok.c2r(transfer=Transfer.force(d)) for d in range(3)]
    out1 = torch.empty(len(x), dtype=torch.float64, device=x.device)
    _, t1 = timed(lambda: comps[0].readout(x, out=out1))
    out3 = torch.empty((len(x), 3), dtype=torch.float64, device=x.device)
    _, t3 = timed(lambda: pm.readout(comps, x, out=out3))
    print('%-34s bin (rebuild) %6.2f  paint %6.2f  readout of one field %6.2f  of three %6.2f ms' % (tag, tb, tp, t1, t3), flush=True)


q = lattice(1e-9)
x = lattice(0.4)
d = torch.remainder(x - q + 0.5 * L, L) - 0.5 * L
v = d * (0.05 / 0.4)
F = force(x)
g = 0.03 * float(v.pow(2).mean().sqrt()) / float(F.pow(2).mean().sqrt())
report('initial state (0.4 cells rms)', x)
for s in range(1, steps + 1):
    v += 0.5 * g * F
    x = torch.remainder(x + v, L)
    F = force(x)
    v += 0.5 * g * F
del F, v
disp = float((torch.remainder(x - q + 0.5 * L, L) - 0.5 * L).pow(2).sum(dim=1).mean().sqrt())
# tile populations
cell = torch.floor(x).to(torch.int64) % N
tile = (cell[:, 0] // 8) * ((N // 16) * (N // 32)) + (cell[:, 1] // 16) * (N // 32) + cell[:, 2] // 32
counts = torch.bincount(tile, minlength=(N // 8) * (N // 16) * (N // 32))
cellid = (cell[:, 0] * N + cell[:, 1]) * N + cell[:, 2]
ccounts = torch.bincount(cellid, minlength=N ** 3)
del cell, tile, cellid
print('after %d steps: rms displacement %.1f cells; tiles: max %d, %d above 16384, %d above 8192, %d below 1024 (of %d); '
      'cells: max %d, empty %.1f %%, particles in cells of more than 8: %.1f %%'
      % (steps, disp, int(counts.max()), int((counts > 16384).sum()), int((counts > 8192).sum()), int((counts < 1024).sum()),
         counts.numel(), int(ccounts.max()), 100.0 * float((ccounts == 0).sum()) / ccounts.numel(),
         100.0 * float(ccounts[ccounts > 8].sum()) / len(x)), flush=True)
del counts, ccounts
report('evolved, rows as the run left them', x)
window.SORTED = 'always'
report('... with the plan\'s tile-ordered copy', x)
window.SORTED = 'auto'
o = pm.tile_order(x)
xs = x[o].contiguous()
del o
report('evolved, rows in tile order', xs)
# rows sorted by tile AND cell (what a caller that keeps its particles in cell / Peano-Hilbert order hands over)
cell = torch.floor(xs).to(torch.int64) % N
key = ((cell[:, 0] // 8) * (N // 16) + cell[:, 1] // 16) * (N // 32) + cell[:, 2] // 32
key = key * 4096 + ((cell[:, 0] % 8) * 16 + cell[:, 1] % 16) * 32 + cell[:, 2] % 32
xc = xs[torch.argsort(key, stable=True)].contiguous()
del cell, key
report('evolved, rows in cell order', xc)
del xc
g2 = torch.Generator(device=x.device).manual_seed(1)
xr = x[torch.randperm(len(x), device=x.device, generator=g2)].contiguous()
report('evolved, rows in random order', xr)
