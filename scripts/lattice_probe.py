"""TSC / PCS / CIC paint and readout on a PERFECT lattice (cell centres + a constant offset) against the jittered lattice of
the benchmark: how much of the TSC / PCS paint time is the same-address / bank pattern of the LDS atomics?"""
import sys, time
sys.path.insert(0, '.')
import ctypes as C
import torch
from pmesh_amd import backend
from pmesh_amd._arrays import vec
from pmesh_amd.pm import ParticleMesh
be = backend.get()
N, L = 512, 1000.0
def t(fn, k=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
i = torch.arange(N, device=be.device, dtype=torch.float64)
for name in ('cic', 'tsc', 'pcs'):
    pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler=name)
    rho = pm.create('real')
    for label, off in (('centres', 0.5), ('centres + 0.25', 0.75), ('cell corners + 0.01', 0.01)):
        g = (i + off) * (L / N)
        pos = torch.stack(torch.meshgrid(g, g, g, indexing='ij'), dim=-1).reshape(-1, 3).contiguous()
        pm.paint(pos, out=rho)
        tp = t(lambda: pm.paint(pos, out=rho))
        tr = t(lambda: rho.readout(pos))
        print('%s perfect lattice (%s): paint %.2f ms readout %.2f ms (plan reused: kernels only)' % (name, label, tp, tr))
        del pos
    pos = torch.empty((N ** 3, 3), dtype=torch.float64, device=be.device)
    pv = vec(pos)
    be.call('synth_uniform', C.byref(pv), N, L, 42, 0, N ** 3, be.stream())
    pm.paint(pos, out=rho)
    print('%s jittered lattice: paint %.2f ms readout %.2f ms' % (name, t(lambda: pm.paint(pos, out=rho)), t(lambda: rho.readout(pos))))
    del pos
