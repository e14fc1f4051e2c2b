"""one rank, with and without a layout: which kernels run (rocprofv3 --kernel-trace --stats around this)"""
import sys, time
sys.path.insert(0, '.')
import torch
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer
from pmesh_amd import window
dev = torch.device('cuda')
N = 128
use_layout = int(sys.argv[1])
pm = ParticleMesh(BoxSize=1000.0, Nmesh=[N, N, N], dtype='f8')
pos = torch.rand((N ** 3, 3), dtype=torch.float64, device=dev) * 1000.0
rho = pm.create('real')
layout = pm.decompose(pos) if use_layout else None
T = Transfer.dx1(0)
def cycle():
    window.clear_bin_cache()
    if layout is not None:
        layout._memo = None
    pm.paint(pos, hold=False, layout=layout, out=rho)
    rhok = rho.r2c(out=Ellipsis)
    back = rhok.c2r(out=Ellipsis, transfer=T)
    return back.readout(pos, layout=layout)
for _ in range(5): cycle()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): cycle()
torch.cuda.synchronize()
print('layout %d: %.3f ms per cycle' % (use_layout, (time.perf_counter() - t) / 20 * 1e3))
