import sys, time, os
sys.path.insert(0, '.')
import torch, ctypes as C
from pmesh_amd import backend, window
from pmesh_amd._arrays import vec
from pmesh_amd.pm import ParticleMesh
be = backend.get()
N = 512
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
for _ in range(10): f()
torch.cuda.synchronize(); print('PMX_BIN_DEBUG=%s: bin %.3f ms' % (os.environ.get('PMX_BIN_DEBUG', '0'), (time.perf_counter() - t) / 10 * 1e3))
