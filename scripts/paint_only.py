"""paint the standard 512^3 set a few times with one window (for kernel-level timing experiments)"""
import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
from pmesh_amd import backend, window
from pmesh_amd._arrays import vec
from pmesh_amd.pm import ParticleMesh
be = backend.get()
N, L = 512, 1000.0
pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler=sys.argv[1])
pos = torch.empty((N ** 3, 3), dtype=torch.float64, device=be.device)
pv = vec(pos)
be.call('synth_uniform', C.byref(pv), N, L, 42, 0, N ** 3, be.stream())
rho = pm.create('real')
for _ in range(6):
    pm.paint(pos, out=rho, hold=False)
torch.cuda.synchronize()
