#!/usr/bin/env python3
"""What an overflowing single-pass rebuild costs, and the steps after it: 512^3 particles in lattice order; from step 3
on a fraction of them sits in a blob (tiles there outgrow the slack of their ranges).  Bin time per step (events),
builds by kind (pmx_binplan_builds), overflows seen."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
from pmesh_amd import backend, window
from pmesh_amd._arrays import vec
from pmesh_amd.pm import ParticleMesh
be = backend.get()
N, L = 512, 1000.0
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler='cic')
rho = pm.create('real')
A = torch.empty((N ** 3, 3), dtype=torch.float64, device=be.device)
pv = vec(A)
be.call('synth_uniform', C.byref(pv), N, L, 42, 0, N ** 3, be.stream())
g = torch.Generator(device=be.device); g.manual_seed(5)
B = A.clone()
k = int(frac * N ** 3)
# the rows that fall into the blob: a contiguous range (neighbours in the array, as collapsing matter is) or, with a
# third argument, scattered through the array (the adversarial case: every wave holds one of them)
if len(sys.argv) > 2:
    idx = torch.randperm(N ** 3, device=be.device, generator=g)[:k]
else:
    idx = torch.arange(40000000, 40000000 + k, device=be.device)
B[idx] = (torch.tensor([300.0, 500.0, 700.0], device=be.device, dtype=torch.float64) +
          torch.randn((k, 3), dtype=torch.float64, device=be.device, generator=g) * 6.0 * L / N) % L
def builds():
    t = [0, 0]
    for e in window.bin_cache().entries:
        a, b = C.c_uint32(0), C.c_uint32(0)
        be.call('binplan_builds', e[1], C.byref(a), C.byref(b)); t[0] += a.value; t[1] += b.value
    return t
seq = [A, A, A] + [B + 0.001 * i for i in range(12)]
for step, pos in enumerate(seq):
    window.clear_bin_cache()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); pm.resampler.prebin(rho.value, pos, pm.affine); b.record()
    torch.cuda.synchronize()
    print('step %2d (%s): bin %.2f ms  builds single/two-pass %s  overflows %d' % (step, 'lattice' if step < 3 else 'blob', a.elapsed_time(b), builds(), window.bin_cache().overflows(be)), flush=True)
