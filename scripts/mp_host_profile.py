"""Host-side cost of the multi-process PM cycle: N ranks share cuda:0 over gloo on a mesh so small
that the GPU is idle; rank 0 prints a cProfile of the timed cycles.
  PMESH_AMD_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29701 scripts/mp_host_profile.py [mesh]
"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, '.')
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('gloo')
from pmesh_amd.comm import default_comm
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer
from pmesh_amd import window
comm = default_comm()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda')
pm = ParticleMesh(BoxSize=1000.0, Nmesh=[N, N, N], dtype='f8', comm=comm, np=[comm.size])
g = torch.Generator(device=dev); g.manual_seed(comm.rank)
pos = torch.rand((N ** 3 // comm.size, 3), dtype=torch.float64, device=dev, generator=g) * 1000.0
rho = pm.create('real')
layout = pm.decompose(pos)
T = Transfer.dx1(0)
res = torch.empty(len(pos), dtype=torch.float64, device=dev)

def cycle():
    window.clear_bin_cache()
    layout._memo = None; layout._memo_remote = None
    pm.resampler.prebin(rho.value, pos, pm.affine)
    pm.paint(pos, hold=False, layout=layout, out=rho)
    rhok = rho.r2c(out=Ellipsis)
    back = rhok.c2r(out=Ellipsis, transfer=T)
    return back.readout(pos, layout=layout, out=res)

for _ in range(5): cycle()
torch.cuda.synchronize(); comm.Barrier()
K = 30
t = time.perf_counter()
for _ in range(K): cycle()
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / K
pr = cProfile.Profile(); pr.enable()
for _ in range(K): cycle()
torch.cuda.synchronize()
pr.disable()
if comm.rank == 0:
    print('ranks=%d mesh=%d wall %.3f ms / cycle' % (comm.size, N, wall * 1e3))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(35); print(s.getvalue()[:6000])
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:7000])
