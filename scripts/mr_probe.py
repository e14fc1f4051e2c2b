#!/usr/bin/env python3
"""Multi-rank PM cycle on ONE GPU: P ranks as P threads (tests/thread_comm.py), the real HIP
kernels, collectives as device copies.  Under `rocprofv3 --kernel-trace --stats` the summed
kernel time per cycle is the *compute* cost of the distributed algorithm (ghost particles,
packing, transposed FFT stages) — what is left once the wire is free.  Wall time here is
inflated by the thread communicator's synchronisations and is only indicative.

    python scripts/mr_probe.py --ranks 8 --mesh 512 --steps 5
"""
import argparse
import os
import sys
import time
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ranks', type=int, default=2)
    ap.add_argument('--mesh', type=int, default=512)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--window', default='cic')
    ap.add_argument('--ghosts-only', type=int, default=1)
    ap.add_argument('--fuse', type=int, default=1)
    ap.add_argument('--np', default='', help="process mesh, e.g. 2x4 (pencils); default: [ranks] slabs")
    ap.add_argument('--check', type=int, default=0,
                    help='1: compare the result of every rank with the one-rank cycle on the same particles')
    args = ap.parse_args()

    import torch
    from thread_comm import run_ranks
    from pmesh_amd import backend, pm as PM, window as _window
    from pmesh_amd._arrays import vec
    from pmesh_amd.transfer import Transfer

    be = backend.get()
    N, L, P = args.mesh, 1000.0, args.ranks
    ntot = N ** 3
    if not args.ghosts_only:
        PM.GHOSTS_ONLY = 'never'
    results = {}

    def rank_main(comm):
        r = comm.rank
        g0, g1 = r * ntot // P, (r + 1) * ntot // P
        pos = torch.empty((g1 - g0, 3), dtype=torch.float64, device=be.device)
        pv = vec(pos)
        be.call('synth_uniform', C.byref(pv), N, L, 42, g0, g1 - g0, be.stream())
        np_ = [int(x) for x in args.np.split('x')] if args.np else [P]
        pm = PM.ParticleMesh(BoxSize=L, Nmesh=[N, N, N], comm=comm, dtype='f8', resampler=args.window, np=np_)
        T = Transfer.dx1(0)
        rho = pm.create('real')
        layout = pm.decompose(pos)

        def cycle():
            _window.clear_bin_cache()
            layout._memo = None
            layout._memo_remote = None
            pm.paint(pos, layout=layout, out=rho)
            ck = rho.r2c(out=Ellipsis)
            if args.fuse:
                back = ck.c2r(out=Ellipsis, transfer=T)
            else:
                back = ck.apply(T, out=Ellipsis).c2r(out=Ellipsis)
            return back.readout(pos, layout=layout)
        for _ in range(args.warmup):
            cycle()
        torch.cuda.synchronize()
        comm.Barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            f = cycle()
        torch.cuda.synchronize()
        comm.Barrier()
        results[r] = (time.perf_counter() - t0, float(f.sum()), int(layout.remote_recvlength))
        if args.check:
            parts[r] = f.clone()

    parts = {}
    run_ranks(P, rank_main)
    if args.check:
        # the same cycle on one rank (the single-GPU path, itself pinned to the oracle at this size
        # by tests/test_binned.py::test_baseline_cycle_equals_oracle)
        _window.clear_bin_cache()
        pos = torch.empty((ntot, 3), dtype=torch.float64, device=be.device)
        pv = vec(pos)
        be.call('synth_uniform', C.byref(pv), N, L, 42, 0, ntot, be.stream())
        pm1 = PM.ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler=args.window)
        one = pm1.paint(pos).r2c(out=Ellipsis).c2r(out=Ellipsis, transfer=Transfer.dx1(0)).readout(pos)
        scale = float(one.abs().max())
        worst = 0.0
        for r in range(P):
            g0, g1 = r * ntot // P, (r + 1) * ntot // P
            worst = max(worst, float((parts[r] - one[g0:g1]).abs().max()))
        print('distributed (%d ranks) vs one rank: max |diff| = %.3e (result scale %.3e) -> %.2e relative'
              % (P, worst, scale, worst / scale))
        assert worst <= 1e-11 * scale
    t = max(v[0] for v in results.values()) / args.steps
    print('ranks %d mesh %d: %.3f ms wall per cycle (all ranks on one GPU), ghosts received per rank %s, '
          'checksum %.6e' % (P, N, 1e3 * t, [v[2] for v in results.values()][:4],
                             sum(v[1] for v in results.values())), flush=True)


if __name__ == '__main__':
    main()
