"""forward vs inverse column passes alone, on the 512^3 fp64 one-rank layout (axis-1 and axis-0 shapes)"""
import sys, time
sys.path.insert(0, '.')
import torch
from pmesh_amd import backend
be = backend.get()
N, pitch = 512, 264
plane = N * pitch + 8
buf = torch.randn(2 * N * plane, dtype=torch.float64, device=be.device)
def t(fn, k=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
for inv in (False, True, False, True):
    a1 = t(lambda: be.colfft(8, inv, buf, N, N, pitch, a_stride=plane, scale=1.0 / N if inv else 1.0))
    a0 = t(lambda: be.colfft(8, inv, buf, 1, N, N * pitch, n_stride=plane, scale=1.0 / N if inv else 1.0))
    print('inverse' if inv else 'forward', 'axis-1 %.0f us, axis-0 %.0f us' % (a1, a0))
