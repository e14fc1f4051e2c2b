#!/usr/bin/env python3
"""time of the tile binning (rebuild from history) per particle vs the mesh size"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ctypes as C
from pmesh_amd import backend, window
from pmesh_amd._arrays import vec
from pmesh_amd.pm import ParticleMesh
be = backend.get()
for N in (128, 192, 256, 320, 384, 512, 640):
    pm = ParticleMesh(BoxSize=1000.0, Nmesh=[N, N, N], dtype='f8')
    pos = torch.empty((N ** 3, 3), dtype=torch.float64, device=be.device)
    pv = vec(pos)
    be.call('synth_uniform', C.byref(pv), N, 1000.0, 42, 0, N ** 3, be.stream())
    rho = pm.create('real')
    def f():
        window.clear_bin_cache()
        pm.resampler.prebin(rho.value, pos, pm.affine)
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    K = 20
    for _ in range(K): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
    print('N=%d: bin %.3f ms = %.2f ps/particle (%.2f TB/s of positions)' % (N, dt * 1e3, dt / N ** 3 * 1e12, 24 * N ** 3 / dt / 1e12), flush=True)
    del pos, rho, pm
