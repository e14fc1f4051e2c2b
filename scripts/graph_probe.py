#!/usr/bin/env python3
"""capture one PM cycle into a HIP graph (torch.cuda.graph) and compare replay with eager issue"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer
from pmesh_amd import window
dev = torch.device('cuda')
for N in (64, 128, 256, 512):
    pm = ParticleMesh(BoxSize=1000.0, Nmesh=[N, N, N], dtype='f8')
    pos = torch.rand((N ** 3, 3), dtype=torch.float64, device=dev) * 1000.0
    rho = pm.create('real')
    out = torch.empty(N ** 3, dtype=torch.float64, device=dev)
    T = Transfer.dx1(0)
    def cycle():
        window.clear_bin_cache()
        pm.paint(pos, hold=False, out=rho)
        back = rho.r2c(out=Ellipsis).c2r(out=Ellipsis, transfer=T)
        back.readout(pos, out=out)
    def timeit(fn, k=30):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(k): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
    for _ in range(3): cycle()
    t_eager = timeit(cycle)
    ref = out.clone()
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): cycle()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cycle()
        t_graph = timeit(g.replay)
        ok = bool(torch.allclose(out, ref, rtol=1e-12, atol=1e-12 * float(ref.abs().max())))
        print('N=%d: eager %.3f ms, graph replay %.3f ms, same result %s' % (N, t_eager, t_graph, ok), flush=True)
    except Exception as e:
        print('N=%d: eager %.3f ms, capture failed: %r' % (N, t_eager, e), flush=True)
