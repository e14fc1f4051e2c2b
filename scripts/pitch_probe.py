#!/usr/bin/env python3
"""per-rank slab FFT passes at P = 8 of 512^3: 257-column (dense) vs 264-column (128-byte) rows"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pmesh_amd import backend
be = backend.get()
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
A, N = 64, 512
for B in (257, 264):
    t = torch.randn(2 * A * N * B, dtype=torch.float64, device=be.device)
    w = torch.empty_like(t)
    print('B=%d: rowfft fwd %.1f us inv %.1f us; colfft axis-1 %.1f us; colfft_split fwd %.1f inv %.1f us' % (
        B, timeit(lambda: be.rowfft(8, False, t, A * N, 512, B)), timeit(lambda: be.rowfft(8, True, t, A * N, 512, B)),
        timeit(lambda: be.colfft(8, False, t, A, N, B)),
        timeit(lambda: be.colfft_split(8, False, t, w, A, N, B, 64)), timeit(lambda: be.colfft_split(8, True, w, t, A, N, B, 64))), flush=True)
# the axis-0 pass of the transposed block (N0, n1loc = 64, 257)
for B in (64 * 257, 64 * 264):
    t = torch.randn(2 * N * B, dtype=torch.float64, device=be.device)
    print('axis-0 pass B=%d: %.1f us' % (B, timeit(lambda: be.colfft(8, False, t, 1, N, B))), flush=True)
