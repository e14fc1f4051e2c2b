#!/usr/bin/env python3
"""One column pass under a profiler: pmx_colfft forward along axis 1 of (A, N, B), `reps` launches.
    python scripts/r06/col_one.py N B [elsize=8] [GB=1] [reps=4]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pmesh_amd import backend

be = backend.get()
N, B = int(sys.argv[1]), int(sys.argv[2])
elsize = int(sys.argv[3]) if len(sys.argv) > 3 else 8
GB = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
A = max(1, int(GB * 1e9 / (2 * elsize * N * B)))
x = torch.randn(A * N * B * 2, dtype=torch.float64 if elsize == 8 else torch.float32, device=be.device) * 1e-3
for _ in range(reps):
    be.colfft(elsize, False, x, A, N, B, scale=1.0 / N)
torch.cuda.synchronize()
print('A=%d N=%d B=%d: %.3f GB read and %.3f GB written per launch (algorithmic)' % (A, N, B, x.numel() * elsize / 1e9, x.numel() * elsize / 1e9))
