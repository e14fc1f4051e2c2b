#!/usr/bin/env python3
"""A caller's time step as examples/nbody.py writes it (force(): paint -> r2c -> three x (apply -> c2r -> readout),
nbody.py:199-218), on device-resident particles: per-step time with the caller's own numpy-style transfer functions
(evaluated on device arrays, pmesh_amd/_devarr.py) and with the fused Transfer objects riding on c2r's first pass.
    python scripts/nbody_steps.py [Nmesh=512] [steps=5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pm = ParticleMesh(BoxSize=float(N), Nmesh=[N, N, N], dtype='f8', resampler='cic')
Q = pm.generate_uniform_particle_grid(shift=0.5)
g = torch.Generator(device=Q.device).manual_seed(3)
X = (Q + 0.3 * torch.randn(Q.shape, generator=g, dtype=Q.dtype, device=Q.device)) % float(N)
V = torch.zeros_like(X)


def force_transfer(direction):            # the four-point finite-difference force kernel of examples/nbody.py:162-171
    def kernel(k, v):
        ksq = k[0] ** 2 + k[1] ** 2 + k[2] ** 2
        ksq[ksq == 0] = 1.0
        cell = (v.BoxSize / v.Nmesh)[direction]
        phase = k[direction] * cell
        return 1j * ((8 * numpy.sin(phase) - numpy.sin(2 * phase)) / (6.0 * cell)) / ksq * v
    return kernel


def force(X, fused):
    layout = pm.decompose(X)
    rho = pm.paint(X, layout=layout)
    rhok = rho.r2c(out=Ellipsis)
    # fused == 2: the components of the force one after the other in memory (a (3, n) array, used as its (n, 3) view):
    # a column of an (n, 3) array is written in 8-byte pieces 24 bytes apart — every 64-byte piece of the array is read,
    # patched and written back by the memory system, 1.4 ms more per readout at 512^3 than a dense vector, whichever
    # kernel writes it (measured: scripts/r05/nbody_kstats.sh, 2.54 against 1.12 ms)
    if fused == 3:
        # [r6] the three components kept as fields and read by ONE launch into the rows of F (ParticleMesh.readout /
        # pmx_readout_binned_multi): every row of F written once, the positions of a tile fetched once
        return pm.readout([rhok.c2r(transfer=Transfer.force(d)) for d in range(3)], X, layout=None)
    F = torch.empty((3, len(X)), dtype=X.dtype, device=X.device).t() if fused == 2 else torch.empty_like(X)
    for d in range(3):
        # (the result goes straight into its column of the force array: readout's `out`, window.py:165-221)
        if fused:
            rhok.c2r(transfer=Transfer.force(d)).readout(X, layout=layout, out=F[:, d])
        else:
            rhok.apply(force_transfer(d)).c2r(out=Ellipsis).readout(X, layout=layout, out=F[:, d])
    return F


for fused in (False, True, 2, 3):
    x, v = X.clone(), V.clone()
    F = force(x, fused)

    def step(x, v, F):                    # kick - drift - kick
        v += 0.5e-3 * F
        x = (x + v) % float(N)
        F = force(x, fused)
        v += 0.5e-3 * F
        return x, v, F
    x, v, F = step(x, v, F)               # (untimed: the first use of every kernel loads its code object)
    torch.cuda.synchronize(); t = time.perf_counter()
    st0 = torch.cuda.memory_stats()
    for _ in range(steps):
        x, v, F = step(x, v, F)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / steps * 1e3
    st1 = torch.cuda.memory_stats()
    print('   allocator in the timed loop: %d device allocations, %d frees, reserved %.1f GB'
          % (st1['num_device_alloc'] - st0['num_device_alloc'], st1['num_device_free'] - st0['num_device_free'],
             st1['reserved_bytes.all.current'] / 1e9), flush=True)
    print('N=%d: %s: %.2f ms per step (one paint, one r2c, three c2r + readout; %d particles), |F| max %.3e'
          % (N, ('fused Transfer.force on c2r' + (', force components contiguous' if fused == 2 else (', one readout of the three fields into (n, 3) rows' if fused == 3 else ''))) if fused else "the caller's numpy-style force_transfer on device arrays",
             ms, len(x), float(F.abs().max())), flush=True)
