"""rocFFT 3-d in-place r2c/c2r against a staged transform (1-d batched rocFFT per axis) at sizes
beyond the own kernels (2048) and beyond 2^32 elements."""
import sys, time
sys.path.insert(0, '.')
import torch
from pmesh_amd import backend, fft as F
from pmesh_amd.pm import ParticleMesh
be = backend.get()

def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t) * 1e3

import itertools
cases = [([1024, 1024, 1024], 'f4'), ([2048, 1024, 1024], 'f4'), ([1024, 1024, 2048], 'f4'), ([2048, 2048, 1024], 'f4'),
         ([2048, 2048, 2048], 'f4'), ([2048, 2048, 2048], 'f8'), ([1024, 2048, 512], 'f8')]
for shape, dt in cases:
    for mode in ('auto', 'never'):
        F.COLFFT = mode
        pm = ParticleMesh(BoxSize=1000.0, Nmesh=shape, dtype=dt)
        a = pm.create('real')
        g = torch.Generator(device=be.device); g.manual_seed(1)
        for i in range(0, shape[0], 256):
            a.value[i:i + 256] = torch.randn(a.value[i:i + 256].shape, device=be.device, generator=g, dtype=a.value.dtype)
        ref = a.value[-4:].clone(); ref0 = a.value[:4].clone()
        c = b = None
        try:
            c, t0 = timed(lambda: a.r2c(out=Ellipsis))
            b, t1 = timed(lambda: c.c2r(out=Ellipsis))
            e = max(float((b.value[-4:] - ref).abs().max()), float((b.value[:4] - ref0).abs().max()))
            c, t2 = timed(lambda: b.r2c(out=Ellipsis))
            b, t3 = timed(lambda: c.c2r(out=Ellipsis))
            print(shape, dt, 'COLFFT=%s' % mode, 'first r2c %.0f c2r %.0f ms; second r2c %.1f c2r %.1f ms; round trip err %.2e' % (t0, t1, t2, t3, e), flush=True)
        except Exception as ex:
            print(shape, mode, 'FAILED', repr(ex)[:300], flush=True)
        del a, c, b, pm
        torch.cuda.empty_cache()
