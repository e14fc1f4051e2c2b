#!/usr/bin/env python3
"""The column passes on their own, one GPU, 4 GB read + written per sweep: pmx_colfft (axis 1 of (A, N, B): B = the
contiguous run, as the y pass; A = 1, large B: as the x pass) and pmx_colfft_roundtrip (forward, scale, inverse in one
kernel) for N = 512, 1024, 2048 in both precisions, with pmx_colfft_configure(1 / 0) (persistent / one workgroup per tile).
    python scripts/r06/col_micro.py [GB=2]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pmesh_amd import backend

be = backend.get()
GB = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0


def timed(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


for elsize in (8, 4):
    dt = torch.float64 if elsize == 8 else torch.float32
    for N, Bc in ((512, 257), (512, 264), (512, 65), (512, 72), (1024, 513), (1024, 520), (2048, 1025), (2048, 1032)):
        # Bc: the contiguous run of a y pass — the modes of a row, dense (the wire format of the transposes) or padded to
        # whole 128-byte lines (the one-rank layout); 65 / 72: a pencil rank's quarter of them
        A = max(1, int(GB * 1e9 / (2 * elsize * N * Bc)))
        x = torch.randn(A * N * Bc * 2, dtype=dt, device=be.device) * 1e-3
        nbytes = 2 * x.numel() * elsize
        out = []
        for persistent in (1,):
            be.colfft_configure(persistent)
            ty = timed(lambda: be.colfft(elsize, False, x, A, N, Bc, scale=1.0 / N))
            tyi = timed(lambda: be.colfft(elsize, True, x, A, N, Bc, scale=1.0))
            tx = timed(lambda: be.colfft(elsize, False, x, 1, N, A * Bc, scale=1.0 / N))
            tr = timed(lambda: be.colfft_roundtrip(elsize, x, N, A * Bc, scale=1.0 / N)) if be.colfft_roundtrip_supported(N, elsize) else float('nan')
            out.append('%s: y fwd %.3f (%.2f TB/s) y inv %.3f x fwd %.3f (%.2f TB/s) round trip %.3f (%.2f TB/s)'
                       % ('persistent' if persistent else 'per tile', ty, nbytes / ty / 1e9, tyi, tx, nbytes / tx / 1e9, tr, nbytes / tr / 1e9))
        be.colfft_configure(1)
        print('f%d N=%4d (A=%d, B=%d; %.2f GB per sweep)  ' % (elsize, N, A, Bc, nbytes / 1e9) + ' | '.join(out), flush=True)
        del x
        torch.cuda.empty_cache()
