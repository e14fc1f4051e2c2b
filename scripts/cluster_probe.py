#!/usr/bin/env python3
"""paint / readout under strong clustering: a fraction f of the particles sits in Gaussian blobs
(sigma cells) around `nb` centres; the rest is uniform.  512^3 mesh, 512^3 particles, CIC f8."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pmesh_amd import backend, window
from pmesh_amd.pm import ParticleMesh
be = backend.get()
N, L = 512, 1000.0
name = sys.argv[1] if len(sys.argv) > 1 else 'cic'
pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler=name)
rho = pm.create('real')
n = N ** 3
g = torch.Generator(device=be.device); g.manual_seed(3)
def timeit(fn, k=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
for f, nb, sigma in ((0.0, 1, 1.0), (0.3, 4096, 2.0), (0.5, 512, 2.0), (0.5, 64, 2.0), (0.5, 64, 0.5), (0.9, 8, 1.0)):
    pos = torch.rand((n, 3), dtype=torch.float64, device=be.device, generator=g) * L
    nc = int(f * n)
    if nc:
        centres = torch.rand((nb, 3), dtype=torch.float64, device=be.device, generator=g) * L
        which = torch.randint(0, nb, (nc,), device=be.device, generator=g)
        pos[:nc] = (centres[which] + torch.randn((nc, 3), dtype=torch.float64, device=be.device, generator=g) * sigma * L / N) % L
    out = {}
    for mode in ('always', 'never'):
        window.BINNED = mode
        def run():
            window.clear_bin_cache()
            pm.paint(pos, out=rho)
            return rho.readout(pos)
        run()
        out[mode] = timeit(run)
    window.BINNED = 'always'
    window.clear_bin_cache()
    pm.paint(pos, out=rho)
    mx = float(rho.value.max())
    print('%s f=%.1f blobs=%d sigma=%.1f: max cell %.0f  binned (bin+paint+readout) %.2f ms   direct %.2f ms' % (name, f, nb, sigma, mx, out['always'], out['never']), flush=True)
    del pos
