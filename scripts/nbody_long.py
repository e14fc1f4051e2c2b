#!/usr/bin/env python3
"""What a time-stepping caller sees over a LONG run: >= 100 kick-drift-kick steps from a lattice with a growing
Zel'dovich flow (the rows start in lattice order and lose it by degrees: rms displacement 0.4 -> ~8 cells), the PM
force through the drop-in surface (paint -> r2c -> three fused c2r -> one readout of the three components), ms per step
reported every 10 steps — with and without re-sorting the particle arrays into tile order every K steps
(ParticleMesh.tile_order; the gather of x, v and the sort itself are inside the timed steps).

    python scripts/nbody_long.py [Nmesh=512] [steps=100] [K=0,25]

DESIGN.md section 4 quotes the table this prints (the steady-state cycle time a real caller
sees, not the step-1 time)."""
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
Ks = [int(k) for k in (sys.argv[3] if len(sys.argv) > 3 else '0,25').split(',')]
GROWTH = float(os.environ.get('NBODY_GROWTH', '0.03'))      # velocities grow by this fraction per step at the start

be = backend.get()
L = float(N)
pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler='cic')


def initial():
    """lattice q and x0 = q + psi (Zel'dovich plane waves, rms 0.4 cells), rows in lattice order"""
    n = N ** 3
    q = torch.empty((n, 3), dtype=torch.float64, device=be.device)
    pv = vec(q)
    modes = bench.zeldovich_modes(numpy, N, L, rms_cells=1e-9)
    be.call('synth_clustered', C.byref(pv), N, L, modes.ctypes.data_as(C.POINTER(C.c_double)), len(modes), 0.0, 0, n, be.stream())
    x = torch.empty_like(q)
    pv = vec(x)
    modes = bench.zeldovich_modes(numpy, N, L, rms_cells=0.4)
    be.call('synth_clustered', C.byref(pv), N, L, modes.ctypes.data_as(C.POINTER(C.c_double)), len(modes), 0.0, 0, n, be.stream())
    return q, x


def force(x):
    rhok = pm.paint(x).r2c(out=Ellipsis)
    comps = [rhok.c2r(transfer=Transfer.force(d)) for d in range(3)]
    return pm.readout(comps, x)


def wrap(d):
    return torch.remainder(d + 0.5 * L, L) - 0.5 * L


def stages(x):
    """one force evaluation with the device drained between its stages (diagnostic, outside the timed steps): ms of
    bin + paint, r2c, the three c2r, the readout"""
    def timed(fn):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        return r, (time.perf_counter() - t) * 1e3
    window.clear_bin_cache()
    _, tb = timed(lambda: pm.resampler.prebin(pm.create('real').value, x, pm.affine))
    rho, tp = timed(lambda: pm.paint(x))
    rhok, tf = timed(lambda: rho.r2c(out=Ellipsis))
    comps, tc = timed(lambda: [rhok.c2r(transfer=Transfer.force(d)) for d in range(3)])
    _, tr = timed(lambda: pm.readout(comps, x))
    return 'bin %.2f paint %.2f r2c %.2f 3 x c2r %.2f readout(3) %.2f' % (tb, tp, tf, tc, tr)


for K in Ks:
    window.clear_bin_cache()
    q, x = initial()
    psi = wrap(x - q)
    v = psi * (0.05 / 0.4)                          # the first drifts move the particles by 0.05 cells rms
    F = force(x)
    # the kick that makes the flow grow by GROWTH per step at the start (F is parallel to psi in the linear regime)
    g = GROWTH * float(v.pow(2).mean().sqrt()) / float(F.pow(2).mean().sqrt())
    del psi
    torch.cuda.synchronize()
    print('N=%d, %d particles, re-sorted into tile order every %s steps' % (N, len(x), K if K else 'inf (never)'), flush=True)
    t0 = time.perf_counter()
    tsort = 0.0
    for s in range(1, steps + 1):
        v += 0.5 * g * F
        x = torch.remainder(x + v, L)
        if K and s % K == 0:
            torch.cuda.synchronize(); ts = time.perf_counter()
            o = pm.tile_order(x)
            x, v, q = x[o], v[o], q[o]
            del o
            torch.cuda.synchronize(); tsort += time.perf_counter() - ts
        F = force(x)
        v += 0.5 * g * F
        if s % 10 == 0:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            disp = float(wrap(x - q).pow(2).sum(dim=1).mean().sqrt())
            sorted_plans = window.bin_cache().sorted_plans(be)
            overflows = window.bin_cache().overflows(be)
            single = two = 0
            for e in window.bin_cache().entries:
                a, b = C.c_uint32(0), C.c_uint32(0)
                be.call('binplan_builds', e[1], C.byref(a), C.byref(b))
                single, two = single + a.value, two + b.value
            print('  steps %3d-%3d: %7.2f ms per step%s   rms displacement %5.2f cells, |v| rms %.3f cells/step, plans with the tile-ordered copy: %d, plan builds so far: %d in one pass (%d of them repaired), %d in two, %.1f GB reserved'
                  % (s - 9, s, (t1 - t0) / 10 * 1e3, (' (of which re-sorting %.2f)' % (tsort / 10 * 1e3)) if K else '', disp,
                     float(v.pow(2).sum(dim=1).mean().sqrt()), sorted_plans, single, overflows, two, torch.cuda.memory_reserved() / 1e9), flush=True)
            if os.environ.get('NBODY_STAGES'):
                print('      stages: ' + stages(x), flush=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tsort = 0.0
    del q, x, v, F
    torch.cuda.empty_cache()
