"""column / row pass rate against the working set: 512-point f64 passes over A planes of the 512^3 one-rank layout
(A * 2.16 MB; 32 MB of L2 in total, 256 MB of Infinity Cache)"""
import sys, time
sys.path.insert(0, '.')
import torch
from pmesh_amd import backend
be = backend.get()
N, pitch = 512, 264
plane = N * pitch + 8
buf = torch.randn(2 * N * plane, dtype=torch.float64, device=be.device)
def t(fn, k=20):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
for A in (4, 8, 16, 32, 64, 96, 128, 256, 512):
    nbytes = 2.0 * A * N * 257 * 16
    for inv in (False, True):
        a1 = t(lambda: be.colfft(8, inv, buf, A, N, pitch, a_stride=plane, scale=1.0))
        rw = t(lambda: be.rowfft(8, inv, buf, A * N, N, pitch, 1.0, rows_per_plane=N, plane_pitch=plane))
        a1 = t(lambda: be.colfft(8, inv, buf, A, N, pitch, a_stride=plane, scale=1.0))
        print('A %3d (%6.1f MB) %s: axis-1 pass %.0f us = %.2f TB/s   row pass %.0f us = %.2f TB/s' % (
            A, nbytes / 2e6, 'inv' if inv else 'fwd', a1, nbytes / a1 / 1e6, rw, nbytes / max(rw, 1e-9) / 1e6))
