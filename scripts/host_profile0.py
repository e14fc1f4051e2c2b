"""cProfile of the host side of the one-rank headline cycle exactly as bench.py issues it (no layout: prebin, paint,
r2c, apply, c2r, readout), on a mesh so small that the device is idle: where the 0.4 ms of host time per cycle go."""
import cProfile, io, pstats, sys, time
sys.path.insert(0, '.')
import torch
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer
from pmesh_amd import window
dev = torch.device('cuda')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pm = ParticleMesh(BoxSize=1000.0, Nmesh=[N, N, N], dtype='f8')
pos = torch.rand((N ** 3, 3), dtype=torch.float64, device=dev) * 1000.0
rho = pm.create('real')
T = Transfer.dx1(0)
res = torch.empty(len(pos), dtype=torch.float64, device=dev)
stamp = [0.0] * 7
def cycle(split=False):
    t = time.perf_counter
    a = t(); window.clear_bin_cache()
    pm.resampler.prebin(rho.value, pos, pm.affine); b = t(); stamp[0] += b - a
    painted = pm.paint(pos, mass=1.0, hold=False, out=rho); a = t(); stamp[1] += a - b
    rhok = painted.r2c(out=Ellipsis); b = t(); stamp[2] += b - a
    rhok.apply(T, out=Ellipsis); a = t(); stamp[3] += a - b
    back = rhok.c2r(out=Ellipsis); b = t(); stamp[4] += b - a
    f = back.readout(pos, out=res); a = t(); stamp[5] += a - b
    return f
for _ in range(5): cycle()
torch.cuda.synchronize()
K = 200
for i in range(7): stamp[i] = 0.0
t = time.perf_counter()
for _ in range(K): cycle()
ti = (time.perf_counter() - t) / K
torch.cuda.synchronize()
print('host issue %.3f ms / cycle: prebin %.1f paint %.1f r2c %.1f apply %.1f c2r %.1f readout %.1f us' % (
    (ti * 1e3,) + tuple(1e6 * s / K for s in stamp[:6])))
pr = cProfile.Profile(); pr.enable()
for _ in range(K): cycle()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(40); print(s.getvalue()[:8000])
