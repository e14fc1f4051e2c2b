"""time r2c / c2r of one mesh (in place, second call): python scripts/fft_only.py N0 N1 N2 f4|f8"""
import sys, time
sys.path.insert(0, '.')
import torch
from pmesh_amd import backend
from pmesh_amd.pm import ParticleMesh
be = backend.get()
shape = [int(x) for x in sys.argv[1:4]]; dt = sys.argv[4]
pm = ParticleMesh(BoxSize=1.0, Nmesh=shape, dtype=dt)
a = pm.create('real')
g = torch.Generator(device=be.device); g.manual_seed(1)
for i in range(0, shape[0], 128):
    a.value[i:i + 128] = torch.randn(a.value[i:i + 128].shape, device=be.device, generator=g, dtype=a.value.dtype)
ref = a.value[-2:].clone()
c = a.r2c(out=Ellipsis); b = c.c2r(out=Ellipsis)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); c = b.r2c(out=Ellipsis); torch.cuda.synchronize(); t1 = time.perf_counter()
    b = c.c2r(out=Ellipsis); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t) * 1e3, (t2 - t1) * 1e3))
print(shape, dt, 'r2c %.1f ms c2r %.1f ms' % min(ts), 'err %.2e' % float((b.value[-2:] - ref).abs().max()))
