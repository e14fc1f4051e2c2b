"""Maximum-size check on one GPU: a 2048^3 fp32 mesh (8.6e9 cells, beyond 32-bit cell indices; 34 GB)
with ~8e8 particles.  Tile-binned kernels against the direct kernels, mass conservation, readout of a
constant field, r2c -> c2r round trip (rocFFT: 2048 is outside the own FFT kernels)."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import torch
from pmesh_amd import backend, window
from pmesh_amd._arrays import vec
from pmesh_amd.pm import ParticleMesh
be = backend.get()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
side = int(sys.argv[2]) if len(sys.argv) > 2 else 928
L = 1000.0
pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f4')
n = side ** 3
pos = torch.empty((n, 3), dtype=torch.float64, device=be.device)
pv = vec(pos)
be.call('synth_uniform', C.byref(pv), side, L, 7, 0, n, be.stream())
torch.cuda.synchronize()

def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t) * 1e3

window.BINNED = 'always'
a, t = timed(lambda: pm.paint(pos)); a, t = timed(lambda: pm.paint(pos, out=a))
print('binned paint %.1f ms, csum/n - 1 = %.2e' % (t, a.csum() / n - 1))
window.BINNED = 'never'
b, t = timed(lambda: pm.paint(pos)); b, t = timed(lambda: pm.paint(pos, out=b))
print('direct paint %.1f ms, csum/n - 1 = %.2e' % (t, b.csum() / n - 1))
d = 0.0
for i in range(0, N, 128):                      # plane chunks: no 34 GB temporaries
    d = max(d, float((a.value[i:i + 128] - b.value[i:i + 128]).abs().max()))
print('max |binned - direct| = %.3e (cell values up to %.2f)' % (d, float(a.value[-64:].max())))
assert d < 2e-5
hi = float(a.value[N - 64:].double().sum()) / (n * 64.0 / N) - 1
print('last 64 planes hold their share of the mass to %.2e' % hi)
assert abs(hi) < 1e-2
window.BINNED = 'always'
ra, t = timed(lambda: a.readout(pos)); ra, t = timed(lambda: a.readout(pos))
print('binned readout %.1f ms' % t)
window.BINNED = 'never'
rb, t = timed(lambda: a.readout(pos)); rb, t = timed(lambda: a.readout(pos))
print('direct readout %.1f ms; max diff %.3e' % (t, float((ra - rb).abs().max())))
assert float((ra - rb).abs().max()) < 2e-5
del b, rb
one = pm.create('real', value=1.0)
window.BINNED = 'always'
r1 = one.readout(pos)
print('readout of a constant field: max |r - 1| = %.2e' % float((r1 - 1).abs().max()))
assert float((r1 - 1).abs().max()) < 1e-5
del one, r1
ref = a.value[N - 8:].clone()
c, t = timed(lambda: a.r2c(out=Ellipsis)); 
print('r2c %.1f ms' % t)
back, t = timed(lambda: c.c2r(out=Ellipsis))
print('c2r %.1f ms' % t)
e = float((back.value[N - 8:] - ref).abs().max())
print('round trip error in the last planes %.2e' % e)
assert e < 1e-4
print('OK')
