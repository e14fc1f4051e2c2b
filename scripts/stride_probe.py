#!/usr/bin/env python3
"""axis-0 column pass (1, N, B): time vs the line stride B (channel / TLB aliasing of the 2 MB plane stride)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pmesh_amd import backend
be = backend.get()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
base = N * (N // 2 + 8)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for pad in (0, 8, 16, 24, 32, 40, 64, 72, 128, 136, 264, 520, 1032):
    B = base + pad
    t = torch.randn(2 * N * B, dtype=torch.float64, device=be.device)
    us = timeit(lambda: be.colfft(8, False, t, 1, N, B))
    us2 = timeit(lambda: be.colfft(8, True, t, 1, N, B))
    print('N %d B %d (+%d): fwd %.1f us inv %.1f us  -> %.2f TB/s' % (N, B, pad, us, us2, 2 * 16 * N * B / us / 1e6), flush=True)
# the axis-1 pass for reference
B1 = N // 2 + 8
t = torch.randn(2 * N * N * B1, dtype=torch.float64, device=be.device)
us = timeit(lambda: be.colfft(8, False, t, N, N, B1))
print('axis-1 pass (A=%d, N=%d, B=%d): %.1f us' % (N, N, B1, us))
