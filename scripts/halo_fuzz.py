#!/usr/bin/env python3
"""Random meshes / windows / canvas types / particle sets: r2c of a field whose halo merge was left to the transform
(pm.HALO_DEFER = 'fresh': gathered by the row pass, in place and out of place, blocked and not) against r2c of the
eagerly merged field.  python scripts/halo_fuzz.py [cases=200] [seed=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
import pmesh_amd.pm as pmod
from pmesh_amd import fft as _fft, window
from pmesh_amd.pm import ParticleMesh

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = numpy.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
window.BINNED = 'always'
N0s = [64, 72, 96, 128, 192, 320]          # multiples of the tile extent 8 that the column kernels take
N1s = [64, 128, 192, 320]                   # ... of 16
N2s = [128, 256, 512, 1024]                 # rows the gather is built for (2048 in float too, left out for time)
worst = 0.0
deferred = 0
for c in range(cases):
    nmesh = (int(rs.choice(N0s)), int(rs.choice(N1s)), int(rs.choice(N2s)))
    if numpy.prod(nmesh) > 2 ** 25:
        continue
    dtype = rs.choice(['f8', 'f4'])
    name = rs.choice(['cic', 'tsc', 'pcs'])
    box = rs.uniform(0.5, 300.0, size=3)
    pm = ParticleMesh(Nmesh=nmesh, BoxSize=box, dtype=dtype, resampler=name)
    n = int(rs.uniform(0.2, 1.5) * numpy.prod(nmesh))
    g = torch.Generator(device='cpu').manual_seed(int(rs.randint(1 << 30)))
    pos = torch.rand(n, 3, generator=g, dtype=torch.float64) * torch.as_tensor(box)
    if rs.rand() < 0.3:         # a blob: crowded tiles
        pos[: n // 3] = (torch.as_tensor(box) * 0.37 + torch.randn(n // 3, 3, generator=g, dtype=torch.float64) *
                         torch.as_tensor(box / numpy.array(nmesh) * 1.5)) % torch.as_tensor(box)
    pos = pos.cuda()
    mass = (torch.rand(n, generator=g, dtype=torch.float64) + 0.5).cuda() if rs.rand() < 0.5 else 1.0
    _fft.L3_BLOCK_BYTES = int(rs.choice([0, 3, 7, 33])) * nmesh[1] * (nmesh[2] + 16) * (8 if dtype == 'f8' else 4)
    oop = rs.rand() < 0.4
    pmod.HALO_DEFER = 'never'
    window.clear_bin_cache()
    e = pm.paint(pos, mass=mass)
    ek = (e.r2c() if oop else e.r2c(out=Ellipsis)).value.clone()
    pmod.HALO_DEFER = 'fresh'
    window.clear_bin_cache()
    l = pm.paint(pos, mass=mass)
    owed = getattr(l._base.storage, '_pmx_halo', None) is not None
    deferred += owed
    lk = (l.r2c() if oop else l.r2c(out=Ellipsis)).value
    err = float((lk - ek).abs().max()) / float(ek.abs().max())
    tol = 2e-13 if dtype == 'f8' else 4e-6
    worst = max(worst, err / tol)
    if err > tol or not owed:
        print('FAILED case %d: %s %s %s n=%d oop=%s block=%d: err %.2e owed %s' % (c, nmesh, dtype, name, n, oop,
              _fft.L3_BLOCK_BYTES, err, owed), flush=True)
        sys.exit(1)
print('%d cases, %d deferred, worst error / tolerance %.3f' % (cases, deferred, worst))
