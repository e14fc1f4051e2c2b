#!/usr/bin/env python3
"""One rank's share of the 512^3 headline at P ranks, ALONE on the GPU: the slab-local block (512/P planes of the periodic
mesh), the particles of that slab — bin / paint / readout times against 1/P of the one-rank cycle.  What a GPU of an
8-GPU run spends in these kernels (thread ranks on one GPU run concurrently and hide launch-size effects)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
from pmesh_amd import backend, window
from pmesh_amd._arrays import vec
from pmesh_amd.window import windows, Affine

be = backend.get()
N, L = 512, 1000.0
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
name = sys.argv[2] if len(sys.argv) > 2 else 'cic'
W = windows[name]
n0 = N // P
allpos = torch.empty((N ** 3, 3), dtype=torch.float64, device=be.device)
pv = vec(allpos)
be.call('synth_uniform', C.byref(pv), N, L, 42, 0, N ** 3, be.stream())
x = allpos[:, 0] * (N / L)
for rank in (0, 3):
    lo, hi = rank * n0, (rank + 1) * n0
    pos = allpos[(x >= lo) & (x < hi)].contiguous()
    canvas = torch.zeros((n0, N, N), dtype=torch.float64, device=be.device)
    aff = Affine(3, scale=N / L, translate=[-float(lo), 0.0, 0.0], period=[N, N, N])
    out = torch.empty(len(pos), dtype=torch.float64, device=be.device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ts = []
    for it in range(12):
        window.clear_bin_cache()
        ev[0].record(); W.prebin(canvas, pos, aff)
        ev[1].record(); W.paint(canvas, pos, transform=aff, _overwrite=True)
        ev[2].record(); W.readout(canvas, pos, transform=aff, out=out)
        ev[3].record(); torch.cuda.synchronize()
        if it >= 3:
            ts.append([ev[i].elapsed_time(ev[i + 1]) for i in range(3)])
    m = [sum(t[i] for t in ts) / len(ts) for i in range(3)]
    print('%s rank %d of %d: %d particles: bin %.3f paint %.3f readout %.3f ms  (x %d = %.2f / %.2f / %.2f)' % (
        name, rank, P, len(pos), m[0], m[1], m[2], P, P * m[0], P * m[1], P * m[2]))
