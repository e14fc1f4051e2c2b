#!/usr/bin/env python3
"""The multi-rank PM cycle on ONE GPU (TEST INFRASTRUCTURE; `scripts/mr_probe.py` is its command line).

P ranks run as P threads (tests/thread_comm.py) that drive the real HIP kernels of the distributed
path — decompose, ghosts-only particle routing, slab or pencil FFT with its pack / unpack and
pipelined transposes, fused transfer, readout — with the collectives as device copies.  This is how
BASELINE.json's multi-GPU configurations are exercised in their decomposed form on the 1-GPU box:

    config 4:  --ranks 8 --mesh 1024 --window cic                      (slab np=[8], 1024^3 particles)
    config 5:  --ranks 8 --np 2x4 --mesh 1024 --window pcs --data clustered --double 1 --mass array
               --pos-dtype f4     (2 x 4 pencils, 2 x 1024^3 Zel'dovich particles, per-particle mass:
               the largest pencil case one GPU holds; config 5 itself is 2048^3 / 2 x 2048^3)

--check 1: every rank's readout must equal the one-rank cycle on the same particles (the single-GPU
path, itself pinned to the oracle at full size by tests/test_binned.py and tests/test_large.py) to
1e-11 of the result's scale.  --oracle-planes K: the planes [0, K) of rank 0's painted block are
compared with the CPU oracle's paint of every particle (of all ranks) whose window touches them.

Under `rocprofv3 --kernel-trace --stats` the summed kernel time per cycle is the compute cost of the
distributed algorithm (ghost particles, packing, transposed FFT stages) — what is left once the wire
is free.  Wall time here is inflated by the thread communicator's synchronisations.
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, 'tests') not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, 'tests'))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--ranks', type=int, default=2)
    ap.add_argument('--mesh', type=int, default=512)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--window', default='cic')
    ap.add_argument('--ghosts-only', type=int, default=1)
    ap.add_argument('--fuse', type=int, default=1)
    ap.add_argument('--np', default='', help="process mesh, e.g. 2x4 (pencils); default: [ranks] slabs")
    ap.add_argument('--data', default='uniform', choices=['uniform', 'clustered'])
    ap.add_argument('--double', type=int, default=0, help='1: 2 x mesh^3 particles (the lattice and a copy shifted by half a cell)')
    ap.add_argument('--mass', default='scalar', choices=['scalar', 'array'])
    ap.add_argument('--pos-dtype', default='f8', choices=['f4', 'f8'])
    ap.add_argument('--check', type=int, default=0,
                    help='1: compare the result of every rank with the one-rank cycle on the same particles')
    ap.add_argument('--backend', default='hip', choices=['hip', 'double'],
                    help="double: the oracle double of tests/oracle_backend.py (a CPU rehearsal of this script's flow)")
    ap.add_argument('--migrate', type=int, default=0,
                    help='1: every particle moves to the rank that owns its cell first, once (bench.py --gpus N does: what a '
                         'time-stepping code does after its first decompose); 0: the ranks keep their slabs of lattice ids and '
                         'on a pencil mesh three quarters of the rows travel as "ghosts" in every cycle')
    ap.add_argument('--decompose', type=int, default=0, help='1: pm.decompose inside every cycle, as a time-stepping caller writes it (examples/nbody.py:199-204)')
    ap.add_argument('--out-field', type=int, default=0, help='1: paint into a field the caller keeps across cycles')
    ap.add_argument('--oracle-planes', type=int, default=0,
                    help='K > 0: the first K planes of rank 0 painted block against the CPU oracle')
    return ap.parse_args(argv)


def _generate(be, args, torch, rank, P, modes):
    """this rank's share of the particles: lattice ids [rank, rank + 1) * N^3 / P (and their shifted copies)"""
    from pmesh_amd._arrays import vec
    N, L = args.mesh, 1000.0
    nlat = N ** 3
    g0, g1 = rank * nlat // P, (rank + 1) * nlat // P
    copies = 2 if args.double else 1
    tdt = torch.float64 if args.pos_dtype == 'f8' else torch.float32
    pos = torch.empty((copies * (g1 - g0), 3), dtype=tdt, device=be.device)
    if args.data == 'uniform':
        if args.double:
            raise SystemExit('--double is defined for --data clustered')
        pv = vec(pos)
        be.call('synth_uniform', C.byref(pv), N, L, 42, g0, g1 - g0, be.stream())
    else:
        for c in range(copies):
            pv = vec(pos[c * (g1 - g0):(c + 1) * (g1 - g0)])
            be.call('synth_clustered', C.byref(pv), N, L, modes.ctypes.data_as(C.POINTER(C.c_double)), len(modes),
                    0.5 * c, g0, g1 - g0, be.stream())
    mass = 1.0
    if args.mass == 'array':
        # deterministic, order 1, exactly summable; a function of the lattice id so that it does not depend on P
        ids = torch.arange(g0, g1, device=be.device, dtype=torch.int64).repeat(copies)
        mass = 0.5 + (ids % 1024).to(torch.float64) / 1024.0
    return pos, mass


def run(args):
    import numpy
    import torch
    from thread_comm import run_ranks
    from pmesh_amd import backend, pm as PM, window as _window
    from pmesh_amd.transfer import Transfer

    if args.backend == 'double':
        import oracle_backend
        be = oracle_backend.install()
    else:
        be = backend.get()
    sync = torch.cuda.synchronize if be.device.type == 'cuda' else (lambda: None)
    N, L, P = args.mesh, 1000.0, args.ranks
    modes = None
    if args.data == 'clustered':
        import bench
        modes = bench.zeldovich_modes(numpy, N, L)
    if not args.ghosts_only:
        PM.GHOSTS_ONLY = 'never'
    np_ = [int(x) for x in args.np.split('x')] if args.np else [P]
    results, parts, kept, block0, staging = {}, {}, {}, {}, {}

    def rank_main(comm):
        r = comm.rank
        pos, mass = _generate(be, args, torch, r, P, modes)
        pm = PM.ParticleMesh(BoxSize=L, Nmesh=[N, N, N], comm=comm, dtype='f8', resampler=args.window, np=np_)
        if args.migrate and P > 1:
            home = pm.domain.decompose(pos, smoothing=0, _scale=pm.affine.scale)
            if args.mass == 'array':
                pos, mass = home.exchange(pos, mass)
            else:
                pos = home.exchange(pos)
            del home
            from pmesh_amd.domain import release_staging
            release_staging(comm)
        T = Transfer.dx1(0)
        rho0 = pm.create('real')
        layout0 = pm.decompose(pos)
        result = torch.empty(len(pos), dtype=torch.float64, device=be.device)      # lives across cycles, as a caller keeps it

        def cycle(keep_block=False):
            _window.clear_bin_cache()
            if args.decompose:
                layout = pm.decompose(pos)
            else:
                layout = layout0
                layout._memo = None
                layout._memo_remote = None
            # as the one-rank cycle of bench.py: a field of the paint's own making (no fill, and the merge of the tile
            # halos left to the row pass of r2c), unless --out-field 1 hands it the caller's
            rho = pm.paint(pos, mass=mass, layout=layout, out=rho0 if args.out_field else None)
            if keep_block and r == 0 and args.oracle_planes:
                block0['value'] = rho.value[:args.oracle_planes].clone().cpu().numpy()
                block0['start'] = [int(x) for x in rho.start]
            ck = rho.r2c(out=Ellipsis)
            if args.fuse:
                back = ck.c2r(out=Ellipsis, transfer=T)
            else:
                back = ck.apply(T, out=Ellipsis).c2r(out=Ellipsis)
            return back.readout(pos, layout=layout, out=result)
        for k in range(args.warmup):
            cycle(keep_block=(k == 0))
        if args.warmup == 0 and args.oracle_planes:
            cycle(keep_block=True)
        sync()
        comm.Barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            f = cycle()
        sync()
        comm.Barrier()
        results[r] = (time.perf_counter() - t0, float(f.sum()), int(layout0.remote_recvlength))
        from pmesh_amd.domain import _scratch_of
        staging[r] = _scratch_of(comm).nbytes()
        if args.check or args.oracle_planes:
            parts[r] = f
            kept[r] = (pos, mass)
        _window.bin_cache().destroy(be)           # the rank's plans die with its thread

    # the device's own account of its memory (torch's allocator AND the hipMalloc'd bin plans / FFT work buffers of
    # libpmesh_amd.so), sampled while the ranks run: the peak is what a box must have free for this case
    low = {'free': None, 'stop': False}

    def sampler():
        while not low['stop']:
            f_b, _ = torch.cuda.mem_get_info()
            low['free'] = f_b if low['free'] is None else min(low['free'], f_b)
            time.sleep(0.02)
    watch = None
    if be.device.type == 'cuda':
        import threading
        watch = threading.Thread(target=sampler, daemon=True)
        watch.start()
    run_ranks(P, rank_main)
    t = max(v[0] for v in results.values()) / max(1, args.steps)
    if be.device.type == 'cuda':
        # what the decomposed cycle needed at its peak, all ranks together on the one device (torch's allocator;
        # the bin plans and FFT work buffers of libpmesh_amd.so come on top: hipMalloc, see mem_get_info below)
        free_b, total_b = torch.cuda.mem_get_info()
        print('peak device memory of the %d ranks: %.1f GB in use on the device at the worst sample (%.1f GB allocated by '
              'torch at its peak, %.1f GB reserved; exchange staging %.1f GB of it)'
              % (P, (total_b - (low['free'] if low['free'] is not None else free_b)) / 1e9,
                 torch.cuda.max_memory_allocated() / 1e9, torch.cuda.max_memory_reserved() / 1e9,
                 sum(staging.values()) / 1e9), flush=True)
    print('ranks %d (np %s) mesh %d %s %s%s: %.3f ms wall per cycle (all ranks on one GPU), ghosts received per rank %s, '
          'checksum %.6e' % (P, np_, N, args.window, args.data, ' x2' if args.double else '', 1e3 * t,
                             [v[2] for v in results.values()][:4], sum(v[1] for v in results.values())), flush=True)

    if args.oracle_planes:
        # rank 0's block starts at (0, 0, 0): planes [0, K) x its axis-1 range x all of axis 2, against
        # the oracle's paint of every particle of every rank whose window can touch it (also across the wrap)
        from oracle import oracle as O
        K = args.oracle_planes
        S = {'nnb': 1, 'cic': 2, 'tsc': 3, 'pcs': 4}[args.window]
        got = block0['value']
        n1 = got.shape[1]
        sel_p, sel_m = [], []
        for r in range(P):
            pos, mass = kept[r]
            g = pos.double() * (N / L)
            x, y = g[:, 0], g[:, 1]
            m = ((x >= -(S + 1)) & (x < K + S + 1)) | (x >= N - (S + 1))
            if n1 < N:
                m &= ((y >= -(S + 1)) & (y < n1 + S + 1)) | (y >= N - (S + 1))
            idx = torch.nonzero(m)[:, 0]
            sel_p.append(pos[idx].double().cpu().numpy())
            if args.mass == 'array':
                sel_m.append(mass[idx].cpu().numpy())
            del g, x, y, m, idx
        ph = numpy.concatenate(sel_p)
        mh = numpy.concatenate(sel_m) if args.mass == 'array' else 1.0
        if args.pos_dtype == 'f4':
            ph = ph.astype('f4')                  # the oracle reads the same float rows the device read
        want = numpy.zeros((K, n1, N))
        # the block is the window [0, K) x [0, n1) of the periodic mesh: paint onto it as a local block
        aff = O.Affine(3, scale=N / L, translate=[0, 0, 0], period=N)
        O.Window('tuned' + args.window).paint(want, ph, mass=mh, transform=aff)
        err = abs(got - want).max() / max(1.0, abs(want).max())
        print('rank 0 paint, planes 0..%d of its block vs oracle: %.2e (%d particles)' % (K - 1, err, len(ph)), flush=True)
        assert err < 1e-12, err

    if args.check:
        # the same cycle on one rank (the single-GPU path, itself pinned to the oracle at full size)
        _window.clear_bin_cache()
        # the ranks' particles in one array each, moved rank by rank (never two copies of the whole set)
        sizes = [len(kept[r][0]) for r in range(P)]
        first = kept[0]
        pos = torch.empty((sum(sizes),) + tuple(first[0].shape[1:]), dtype=first[0].dtype, device=be.device)
        mass = torch.empty(sum(sizes), dtype=first[1].dtype, device=be.device) if args.mass == 'array' else 1.0
        del first
        off = 0
        for r in range(P):
            p_r, m_r = kept.pop(r)
            pos[off:off + sizes[r]] = p_r
            if args.mass == 'array':
                mass[off:off + sizes[r]] = m_r
            off += sizes[r]
            del p_r, m_r
            if be.device.type == 'cuda':
                torch.cuda.empty_cache()
        pm1 = PM.ParticleMesh(BoxSize=L, Nmesh=[N, N, N], dtype='f8', resampler=args.window)
        one = pm1.paint(pos, mass=mass).r2c(out=Ellipsis).c2r(out=Ellipsis, transfer=Transfer.dx1(0)).readout(pos)
        scale = float(one.abs().max())
        worst, off = 0.0, 0
        for r in range(P):
            worst = max(worst, float((parts[r] - one[off:off + sizes[r]]).abs().max()))
            off += sizes[r]
        print('distributed (%d ranks) vs one rank: max |diff| = %.3e (result scale %.3e) -> %.2e relative'
              % (P, worst, scale, worst / scale), flush=True)
        assert worst <= 1e-11 * scale
    if watch is not None:
        low['stop'] = True
        watch.join()
        _, total_b = torch.cuda.mem_get_info()
        print('peak device memory of the whole run (with the one-rank check): %.1f GB of %.1f'
              % ((total_b - low['free']) / 1e9, total_b / 1e9), flush=True)
    return t


def main(argv=None):
    run(parse(argv))


if __name__ == '__main__':
    main()
