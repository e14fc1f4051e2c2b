import cProfile, io, pstats, sys, time
sys.path.insert(0, '.')
import torch
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer
from pmesh_amd import window
dev = torch.device('cuda')
N = 128
pm = ParticleMesh(BoxSize=1000.0, Nmesh=[N, N, N], dtype='f8')
pos = torch.rand((N ** 3, 3), dtype=torch.float64, device=dev) * 1000.0
rho = pm.create('real')
layout = pm.decompose(pos)
T = Transfer.dx1(0)
def cycle():
    window.clear_bin_cache()
    layout._memo = None
    pm.paint(pos, hold=False, layout=layout, out=rho)
    rhok = rho.r2c(out=Ellipsis)
    back = rhok.c2r(out=Ellipsis, transfer=T)
    return back.readout(pos, layout=layout)
for _ in range(5): cycle()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(30): cycle()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:3000])
