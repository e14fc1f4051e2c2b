#!/usr/bin/env python3
"""time generate_whitenoise on the device (512^3 half spectrum)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pmesh_amd.pm import ParticleMesh
for N in (128, 256, 512):
    pm = ParticleMesh(BoxSize=1000.0, Nmesh=[N, N, N], dtype='f8')
    pm.generate_whitenoise(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c = pm.generate_whitenoise(7)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    v = c.value
    print('N=%d: %.1f ms (%.2e modes/s); std re %.4f im %.4f' % (N, 1e3 * t, v.numel() / t, float(v.real.std()), float(v.imag.std())), flush=True)
