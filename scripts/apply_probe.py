#!/usr/bin/env python3
"""Field.apply with a caller's numpy-style transfer function (examples/nbody.py:162-171) at 512^3: on device arrays
(pmesh_amd/_devarr.py) against the host evaluation it used to fall back to, and against the fused Transfer kernel"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pm = ParticleMesh(BoxSize=1000.0, Nmesh=[N, N, N], dtype='f8')
rho = pm.create('real')
rho.value.normal_()
ck = rho.r2c()


def force_transfer(direction):
    def filter(k, v):
        k2 = sum(ki ** 2 for ki in k)
        k2[k2 == 0] = 1.0
        C = (v.BoxSize / v.Nmesh)[direction]
        w = k[direction] * C
        kfinite = 1.0 / C * 1 / 6.0 * (8 * numpy.sin(w) - numpy.sin(2 * w))
        return 1j * kfinite / k2 * v
    return filter


def timeit(fn, k=3):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3, r

out = pm.create('complex')
t_dev, a = timeit(lambda: ck.apply(force_transfer(0), out=out))
a = a.value.clone()
t_fused, b = timeit(lambda: ck.apply(Transfer.force(0), out=out))
err = float((a - b.value).abs().max() / b.value.abs().max())
host = pm.create('complex')
t0 = time.perf_counter(); type(ck)._apply_host(ck, force_transfer(0), 'wavenumber', host.value); torch.cuda.synchronize()
t_host = (time.perf_counter() - t0) * 1e3
errh = float((a - host.value).abs().max() / b.value.abs().max())
print('N=%d: callable on device arrays %.1f ms, host slab loop %.0f ms, fused Transfer kernel %.2f ms; device vs fused %.1e, device vs host %.1e'
      % (N, t_dev, t_host, t_fused, err, errh))
