#!/usr/bin/env python3
"""One rank's row pass of a pencil transform, alone on the GPU: pmx_rowfft + pmx_slab_pack (two sweeps) against
pmx_rowfft_split (one), forward and inverse, for the row lengths of the configurations (512, 1024, 2048; both precisions).
    python scripts/r06/row_split_micro.py [rows_bytes_GB=2]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pmesh_amd import backend

be = backend.get()
GB = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0


def timed(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


for elsize in (8, 4):
    dt = torch.float64 if elsize == 8 else torch.float32
    for n in (512, 1024, 2048):
        M1 = n // 2 + 1
        nrows = int(GB * 1e9 / (2 * M1 * elsize)) // 64 * 64
        P1 = 4
        e = [(M1 * q) // P1 for q in range(P1 + 1)]
        src = torch.randn(nrows * 2 * M1, dtype=dt, device=be.device)
        a = torch.empty_like(src)
        b = torch.empty_like(src)
        bytes_sweep = 2 * src.numel() * elsize
        t_row = timed(lambda: be.rowfft_to(elsize, False, src, a, nrows, n, M1))
        t_pack = timed(lambda: be.slab_pack(a, b, nrows, M1, 1, e, 2 * elsize))
        t_split = timed(lambda: be.rowfft_split(elsize, False, src, b, nrows, n, M1, e))
        t_unpack = timed(lambda: be.slab_pack(b, a, nrows, M1, 1, e, 2 * elsize, inverse=True))
        t_rowi = timed(lambda: be.rowfft(elsize, True, a, nrows, n, M1))
        t_spliti = timed(lambda: be.rowfft_split(elsize, True, b, a, nrows, n, M1, e))
        print('f%d n=%4d rows=%8d (%.2f GB per sweep, read + write): forward row %.3f + pack %.3f = %.3f ms, split %.3f ms (%.2f TB/s); '
              'inverse unpack %.3f + row %.3f = %.3f ms, split %.3f ms (%.2f TB/s)'
              % (elsize, n, nrows, bytes_sweep / 1e9, t_row, t_pack, t_row + t_pack, t_split, bytes_sweep / t_split / 1e9,
                 t_unpack, t_rowi, t_unpack + t_rowi, t_spliti, bytes_sweep / t_spliti / 1e9), flush=True)
        del src, a, b
        torch.cuda.empty_cache()
