"""Host-side cost of one PM cycle: run a mesh so small that the GPU is idle most of the time."""
import sys, time
sys.path.insert(0, '.')
import torch
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer
from pmesh_amd import window
dev = torch.device('cuda')
for N, use_layout in ((64, False), (64, True), (128, False), (128, True)):
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
    K = 50
    for _ in range(K): cycle()
    t_issue = (time.perf_counter() - t) / K
    torch.cuda.synchronize(); t_total = (time.perf_counter() - t) / K
    print('N=%d layout=%s: host issue %.3f ms / cycle, wall %.3f ms / cycle' % (N, use_layout, t_issue * 1e3, t_total * 1e3))
