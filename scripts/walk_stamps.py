"""per-phase wave time of paint_walk_kernel (diagnostic build with s_memtime stamps:
PMESH_AMD_LIBRARY=pmesh_amd/libpmesh_amd_stamp.so)"""
import ctypes as C, sys
import torch
from pmesh_amd import backend, window
from pmesh_amd._arrays import vec
from pmesh_amd.pm import ParticleMesh
be = backend.get()
name = sys.argv[1] if len(sys.argv) > 1 else 'tsc'
N = 512
pos = torch.empty((N ** 3, 3), dtype=torch.float64, device=be.device)
pv = vec(pos)
be.call('synth_uniform', C.byref(pv), N, 1000.0, 42, 0, N ** 3, be.stream())
pm = ParticleMesh(BoxSize=1000.0, Nmesh=[N, N, N], dtype='f8', resampler=name)
window.WALK = 'always'
rho = pm.create('real')
for i in range(3):
    pm.paint(pos, hold=False, out=rho)
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
be.lib.pmx_walk_debug_stamps(out, 1)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
pm.paint(pos, hold=False, out=rho)
e1.record()
torch.cuda.synchronize()
be.lib.pmx_walk_debug_stamps(out, 1)
units = out[8]
names = ['top', 'issue', 'sort', 'accumulate', 'flush', 'write_out', 'vm_drain', 'barrier']
tot = sum(out[i] for i in range(8))
print('paint %.3f ms; %d units; per wave and unit (100 MHz ticks):' % (e0.elapsed_time(e1), units))
for i in range(8):
    print('  %-11s %9.0f  %5.1f %%' % (names[i], out[i] / (8.0 * units), 100.0 * out[i] / tot))
print('  total per unit %.0f ticks = %.1f us' % (tot / (8.0 * units), tot / (8.0 * units) / 100.0))
