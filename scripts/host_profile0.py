"""cProfile of the host side of the one-rank headline cycle exactly as bench.py issues it (no layout: prebin, paint,
r2c, apply, c2r, readout), on a mesh so small that the device is idle: where the 0.4 ms of host time per cycle go."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer
from pmesh_amd import window
dev = torch.device('cuda')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pm = ParticleMesh(BoxSize=1000.0, Nmesh=[N, N, N], dtype='f8')
# rows in lattice order, as the headline's (rows in random order take the plan's other form: more launches)
import ctypes as C
from pmesh_amd import backend as _b
from pmesh_amd._arrays import vec as _vec
pos = torch.empty((N ** 3, 3), dtype=torch.float64, device=dev)
_pv = _vec(pos)
_b.get().call('synth_uniform', C.byref(_pv), N, 1000.0, 42, 0, N ** 3, _b.get().stream())
rho = pm.create('real')
T = Transfer.dx1(0)
res = torch.empty(len(pos), dtype=torch.float64, device=dev)
stamp = [0.0] * 7
BENCH = len(sys.argv) > 2 and sys.argv[2] == 'bench'      # as bench.py issues it: a fresh field per paint, an event per stage
marks = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
def cycle(split=False):
    t = time.perf_counter
    a = t(); window.clear_bin_cache()
    if BENCH: marks[0].record()
    pm.resampler.prebin(rho.value, pos, pm.affine); b = t(); stamp[0] += b - a
    if BENCH: marks[1].record()
    painted = pm.paint(pos, mass=1.0, hold=False, out=None if BENCH else rho); a = t(); stamp[1] += a - b
    if BENCH:
        marks[2].record(); rhok = painted.r2c(out=Ellipsis); b = t(); stamp[2] += b - a
        marks[3].record(); rhok.apply(T, out=Ellipsis); a = t(); stamp[3] += a - b
        marks[4].record(); back = rhok.c2r(out=Ellipsis); b = t(); stamp[4] += b - a
        marks[5].record(); f = back.readout(pos, out=res); a = t(); stamp[5] += a - b
        marks[6].record()
        return f
    rhok = painted.r2c(out=Ellipsis); b = t(); stamp[2] += b - a
    rhok.apply(T, out=Ellipsis); a = t(); stamp[3] += a - b
    back = rhok.c2r(out=Ellipsis); b = t(); stamp[4] += b - a
    f = back.readout(pos, out=res); a = t(); stamp[5] += a - b
    return f
for _ in range(5): cycle()
torch.cuda.synchronize()
K = 200
for i in range(7): stamp[i] = 0.0
ti = 0.0
for k in range(K):
    if k % 5 == 0: torch.cuda.synchronize()       # (the device must stay idle: a full queue makes every launch wait)
    t = time.perf_counter(); cycle(); ti += time.perf_counter() - t
ti /= K
torch.cuda.synchronize()
print('host issue %.3f ms / cycle: prebin %.1f paint %.1f r2c %.1f apply %.1f c2r %.1f readout %.1f us' % (
    (ti * 1e3,) + tuple(1e6 * s / K for s in stamp[:6])))
pr = cProfile.Profile()
for k in range(K):
    if k % 5 == 0: torch.cuda.synchronize()
    pr.enable(); cycle(); pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(40); print(s.getvalue()[:8000])
# host time inside the library, per entry point (ctypes call included)
from pmesh_amd import backend
be = backend.get()
acc = {}
orig = be.call
def timed(name, *a):
    t0 = time.perf_counter()
    r = orig(name, *a)
    d = time.perf_counter() - t0
    e = acc.setdefault(name, [0, 0.0]); e[0] += 1; e[1] += d
    return r
be.call = timed
for k in range(K):
    if k % 5 == 0: torch.cuda.synchronize()
    cycle()
torch.cuda.synchronize()
be.call = orig
tot = 0.0
for name, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('%-28s %5.1f calls/cycle %7.1f us/call %7.1f us/cycle' % (name, n / K, 1e6 * t / n, 1e6 * t / K)); tot += t
print('library calls: %.1f us per cycle' % (1e6 * tot / K))
