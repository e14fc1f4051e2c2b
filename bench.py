#!/usr/bin/env python3
"""PM-cycle benchmark: particles/s of paint -> r2c -> apply(transfer) -> c2r -> readout.

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--mesh 512] [--window cic] [--dtype f8] [--data uniform|clustered]

Contract (driver): with --gpus N > 1 it is launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank
per GPU over RCCL.  W untimed warm-up cycles, then exactly K cycles timed between
barrier + synchronize on both sides; the MAX over ranks is used; rank 0 prints ONE
JSON line.  The workload is BASELINE.json's headline: 512^3 mesh, 512^3 uniform
particles (lattice + hashed jitter), CIC, f64, generated in HBM —
a "step" is one PM cycle over that particle set, inputs resident in HBM.  Consecutive
cycles see the particles moved by N(0, --drift) cells (default 0.1) like a time-stepping caller:
the bin plan is rebuilt from moved positions every cycle (`bin_overflows` counts the rebuilds
that needed the two-pass repair).

For N > 1 the mesh is slab-decomposed, particles start on the rank that generated
them (rank r: lattice ids [r, r+1) * N^3 / P) and every cycle includes the particle
exchange (layout.exchange before paint and readout, layout.gather after readout) and
the FFT's global transposes; `decompose` is done once, outside the timed region, and
its time is reported separately (exchange reported included/excluded).

Extra objects on the JSON line:
  roofline     : dominant kernel's achieved algorithmic GB/s (HIP events on the launch
                 stream, averaged over the timed cycles) against the 8 TB/s HBM peak.
  cpu_baseline : the reference's own window kernels (oracle/_ref, compiled from the
                 reference's _window_imp.c) + pocketfft on ONE host core, on a bounded
                 sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

# Runtime switches of the HSA / HIP layer are read when the runtime initialises, i.e. at the first
# torch.cuda call: set them before anything imports torch.  (dmabuf IPC is the only form the host
# driver of this pool supports; without it RCCL fails with `hipIpcGetMemHandle: invalid argument`.)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def algorithmic_bytes(stage, e, pe, nu=1.0, me=0):
    """compulsory HBM bytes per particle: e mesh element, pe position element,
    me per-particle mass element (0: scalar mass)"""
    return {
        'paint': 3 * pe + me + 2 * e / nu,     # positions [+ mass] + one read and one write per cell
        'readout': 3 * pe + e + e / nu,        # positions + result + each cell once
        'r2c': 2 * e / nu, 'c2r': 2 * e / nu,  # real in, half-complex out (single pass)
        'apply': 2 * e / nu,                   # complex read + write
        'zero': e / nu, 'bin': 3 * pe,
    }[stage]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--mesh', type=int, default=512)
    ap.add_argument('--particles', type=int, default=None, help='lattice size per side (default: mesh)')
    ap.add_argument('--window', default='cic', choices=['nnb', 'cic', 'tsc', 'pcs'])
    ap.add_argument('--dtype', default='f8', choices=['f4', 'f8'])
    ap.add_argument('--data', default='uniform', choices=['uniform', 'clustered', 'shuffled'],
                    help="uniform / clustered: SURVEY.md 8d, in lattice order; shuffled: the uniform set in "
                         "random order (diagnostic: no spatial coherence between neighbouring rows)")
    ap.add_argument('--gradient', type=int, default=None, help='gradient readout direction')
    ap.add_argument('--double', type=int, default=0,
                    help="1: 2 x particles^3 particles — the lattice plus a copy shifted by half a cell, same "
                         "displacement field (SURVEY.md 8d, config 5)")
    ap.add_argument('--mass', default='scalar', choices=['scalar', 'array'],
                    help='array: a per-particle fp64 mass (config 5) instead of the scalar 1.0')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-mesh', type=int, default=256)
    ap.add_argument('--binned', type=int, default=-1, help='1/0 force the tile-binned kernels on/off')
    ap.add_argument('--colfft', type=int, default=1, help='0: all-rocFFT 3-d transforms; 1: LDS column FFT')
    ap.add_argument('--tile-order', type=int, default=0,
                    help='1: reorder the particle rows once with ParticleMesh.tile_order before the cycles '
                         '(the remedy for --data shuffled); the reordering time is reported, not timed')
    ap.add_argument('--exchange', type=int, default=0,
                    help='1: route the particles through a decompose() layout even on one rank')
    ap.add_argument('--ghosts-only', type=int, default=1,
                    help='0: the literal exchange-everything scheme of the reference for paint/readout '
                         'with a layout; 1: own particles in place, only ghosts travel')
    ap.add_argument('--drift', type=float, default=0.1,
                    help='rms displacement, in mesh cells per axis, of the particles between two cycles (a '
                         'time-stepping caller): the cycles walk through 3 position sets, each the previous one '
                         'plus N(0, drift) — the single-pass rebuild of the bin plan then works on moved '
                         'particles and its overflow / repair path can trigger; 0: identical positions every cycle')
    ap.add_argument('--host-arrays', type=int, default=0,
                    help='1: positions and results are numpy arrays in host memory, as an unmodified nbodykit / '
                         'fastpm caller passes them: every paint / readout stages them over PCIe (diagnostic: '
                         'the reported rate then is PCIe inclusive and is NOT the headline metric); 2: the same with '
                         'the position arrays registered once (ParticleMesh.stage) and refreshed once per cycle')
    ap.add_argument('--cold-plan', type=int, default=0,
                    help='1: the bin plans are destroyed before every cycle (diagnostic: every build is the first build of a '
                         'plan — allocation, the exact two-pass build, no slot ranges of a previous step to reuse)')
    ap.add_argument('--count-jitter', type=float, default=0.0,
                    help='a diagnostic (one rank, scalar mass): the k-th of the position sets has this fraction times k '
                         'FEWER rows than the first — a rank whose particle count changes from step to step (migration)')
    ap.add_argument('--pos-columns', type=int, default=3,
                    help='a diagnostic (one rank): > 3: the positions are the first three columns of an array with this many '
                         '(a phase-space array: rows with a pitch, not dense)')
    ap.add_argument('--out-field', type=int, default=0,
                    help='0: pm.paint(pos) returns a new field every cycle, as the callers of the reference write it '
                         '(fastpm: pm.paint(x, layout=layout)) — on one rank the halo merge of the tile kernels then rides '
                         'on the row pass of r2c (pm.HALO_DEFER); 1: the cycle paints into one field object of the '
                         "caller's (out=rho), whose value must be complete when paint returns: merge kernel inside paint")
    ap.add_argument('--fuse-apply', type=int, default=1,
                    help='1: the transfer multiplication rides on the first pass of c2r (c2r(transfer=))')
    ap.add_argument('--deterministic', type=int, default=0,
                    help='1: window.DETERMINISTIC — bit-reproducible paint (integer sums, one more sweep of the block)')
    ap.add_argument('--migrate', type=int, default=1,
                    help='1 (--gpus N > 1): move every particle to the rank that owns its cell once, before the '
                         'timed cycles; 0: the particles stay on the rank that generated them')
    ap.add_argument('--decomp', default='slab', choices=['slab', 'pencil'],
                    help="process mesh for --gpus N > 1: slab = [N] (one transpose per transform), pencil = "
                         "pfft's split_size_2d(N), e.g. [2, 4] on 8 ranks (the reference's default for 3-d)")
    ap.add_argument('--config', default=None, choices=['c2', 'c3', 'c4', 'c5'],
                    help="BASELINE.json's other configurations as presets (the default run is the headline metric's "
                         "512^3 workload): c2 = 256^3 CIC fp64; c3 = 512^3 TSC fp32, gradient readout; c4 = 1024^3 "
                         "mesh / 1024^3 particles CIC fp64 on slabs; c5 = 2048^3 mesh / 2 x 2048^3 clustered particles, "
                         "PCS, per-particle mass, pencils (needs 8 GPUs: 52 GB of particles per GPU)")
    args = ap.parse_args()
    presets = {
        'c2': dict(mesh=256),
        'c3': dict(mesh=512, window='tsc', dtype='f4', gradient=0),
        'c4': dict(mesh=1024, decomp='slab'),
        'c5': dict(mesh=2048, window='pcs', data='clustered', double=1, mass='array', decomp='pencil'),
    }
    for k, v in presets.get(args.config, {}).items():
        setattr(args, k, v)
    return args


def cpu_baseline(args):
    """Time the reference path on ONE core and on ALL cores of the host and report the faster
    (some hosts — sandboxes, SMT siblings, cgroup quotas — do not scale with threads)."""
    import copy
    warm = copy.copy(args)
    warm.cpu_sample_mesh = 32
    cpu_baseline_run(warm, 1)                  # load libraries, touch the allocator
    one = cpu_baseline_run(args, 1)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    if cores <= 1:
        return one
    try:
        big = copy.copy(args)
        # with many cores the bounded sample is the workload itself (mesh^3 particles):
        # ~1 s of wall time, tens of core-seconds; it needs ~10 GB of host memory
        try:
            import psutil
            mem_ok = psutil.virtual_memory().available > 24e9
        except Exception:
            mem_ok = False
        if cores >= 16 and mem_ok:
            big.cpu_sample_mesh = min(args.mesh, 512)
        many = cpu_baseline_run(big, cores)
    except Exception as ex:
        one['sample'] += '; the %d-core run failed: %r' % (cores, ex)
        return one
    best, other = (many, one) if many['value'] > one['value'] else (one, many)
    best['sample'] += '; for comparison on %d core(s): %.3e particles/s' % (other['cores'], other['value'])
    return best


def cpu_baseline_run(args, cores):
    """The reference path on `cores` host cores: one slab worker per core, as the reference runs
    one MPI rank per core (its kernels are single threaded, pmesh/_window.pyx:157-165).
    Worker r owns mesh planes [a_r, b_r) and paints the particles whose window can touch them
    (ghost duplicates across slab boundaries; cells outside the slab are dropped by the
    kernel, pmesh/_window_generics.h:144-167) with the reference's own compiled window
    kernels (oracle/_ref; the oracle port if that is absent); r2c / c2r are scipy's pocketfft
    with workers = cores (PFFT/FFTW cannot be installed here); the transfer and the readout
    are chunked over the same workers."""
    import numpy
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    try:
        import scipy.fft as sfft
    except Exception:
        sfft = None
    which = 'ref' if O.have_ref() else 'oracle'
    n = args.cpu_sample_mesh
    P = max(1, min(cores, n // 8))
    L = 1000.0
    kind = {'nnb': 'tunednnb', 'cic': 'tunedcic', 'tsc': 'tunedtsc', 'pcs': 'tunedpcs'}[args.window]
    S = {'nnb': 1, 'cic': 2, 'tsc': 3, 'pcs': 4}[args.window]
    pos = O.synth_uniform(n, L, dtype=args.dtype)          # lattice order: plane i = rows [i n^2, (i+1) n^2)
    edges = [r * n // P for r in range(P + 1)]
    pool = ThreadPoolExecutor(P)                            # ctypes releases the GIL inside the kernels

    def paint_slab(r):
        a, b = edges[r], edges[r + 1]
        W = O.Window(kind, which=which)
        aff = O.Affine(3, scale=1.0 * n / L, translate=[-a, 0, 0], period=n)
        # jitter is +-0.4 cells: particles of lattice planes [a-S, b+S) can touch the slab
        lo, hi = a - S, b + S
        parts = []
        if P == 1:
            parts, lo, hi = [pos], 0, 0                      # the slab is the whole period: no ghosts
        elif lo < 0:
            parts.append(pos[(lo % n) * n * n:])
            lo = 0
        top = min(hi, n)
        if P > 1:
            parts.append(pos[lo * n * n:top * n * n])
        if P > 1 and hi > n:
            parts.append(pos[:(hi - n) * n * n])
        for q in parts:
            if len(q):
                W.paint(real[a:b], q, transform=aff)

    def readout_chunk(r):
        lo, hi = r * len(pos) // P, (r + 1) * len(pos) // P
        W = O.Window(kind, which=which)
        aff = O.Affine(3, scale=1.0 * n / L, period=n)
        W.readout(back, pos[lo:hi], out=out[lo:hi], transform=aff, diffdir=args.gradient)

    cdt = 'c16' if args.dtype == 'f8' else 'c8'
    t0 = time.perf_counter()
    real = numpy.zeros((n, n, n), dtype=args.dtype)
    list(pool.map(paint_slab, range(P)))
    t1 = time.perf_counter()
    if sfft is not None:
        ck = sfft.rfftn(real, workers=P) / float(n) ** 3
    else:
        ck = numpy.fft.rfftn(real) / float(n) ** 3
    ck = numpy.ascontiguousarray(ck, dtype=cdt)
    t2 = time.perf_counter()
    tr = O.make_transfer(laplace_pow=-1, grad_dir=0)

    def transfer_slab(r):
        a, b = edges[r], edges[r + 1]
        O.apply_transfer(tr, ck[a:b], (a, 0, 0), (n, n, n), (L, L, L), out=ck[a:b])
    list(pool.map(transfer_slab, range(P)))
    t3 = time.perf_counter()
    if sfft is not None:
        back = sfft.irfftn(ck, s=(n, n, n), workers=P) * float(n) ** 3
    else:
        back = numpy.fft.irfftn(ck, s=(n, n, n), axes=(0, 1, 2)) * float(n) ** 3
    back = numpy.ascontiguousarray(back, dtype=args.dtype)
    t4 = time.perf_counter()
    out = numpy.zeros(len(pos), dtype='f8')
    list(pool.map(readout_chunk, range(P)))
    t5 = time.perf_counter()
    pool.shutdown()
    total = t5 - t0
    msum = float(real.sum(dtype='f8'))
    assert abs(msum - n ** 3) < 1e-6 * n ** 3, msum          # the slab/ghost scheme conserves mass
    return {
        'value': n ** 3 / total, 'unit': 'particles/s', 'cores': P,
        'kind': 'reference' if which == 'ref' else 'port',
        'sample': '%d^3 mesh, %d^3 uniform particles, %s %s, one PM cycle on %d host cores (one slab '
                  'worker per core, ghost particles across slabs): paint+readout by %s, r2c/c2r by scipy '
                  'pocketfft (workers=%d) standing in for PFFT; %.2f s (paint %.2f r2c %.2f apply %.2f '
                  'c2r %.2f readout %.2f)'
                  % (n, n, args.window, args.dtype, P,
                     "the reference's _window_imp.c (oracle/_ref)" if which == 'ref' else 'the oracle port',
                     P, total, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4),
    }


def main():
    args = parse()
    import numpy
    import torch
    import ctypes as C

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`'
                             % (args.gpus, args.gpus))
    # PMESH_AMD_SHARE_GPU=1 + PMESH_AMD_DIST_BACKEND=gloo: a rehearsal of the multi-process run on a
    # box with one GPU (all ranks on cuda:0, exchanges staged by gloo) -- it checks the flow and the
    # host-side cost, its throughput means nothing.  The real run is one rank per GPU over RCCL.
    share = os.environ.get('PMESH_AMD_SHARE_GPU') == '1'
    torch.cuda.set_device(0 if share else local_rank)
    launched = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ      # under torch.distributed.run
    if world > 1 or launched:
        import datetime
        import torch.distributed as dist
        # a rank that dies must not leave the others in an all-to-all for ever: collectives time out (default 5 min)
        # and, with RCCL's asynchronous error handling, a timed-out rank's process exits non-zero — the launcher
        # (torch.distributed.run) then ends the job.  (Never re-exec a process that has touched the GPU.)
        os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '1')
        dist.init_process_group(os.environ.get('PMESH_AMD_DIST_BACKEND', 'nccl'),
                                timeout=datetime.timedelta(seconds=int(os.environ.get('PMESH_AMD_COLLECTIVE_TIMEOUT_S', '300'))))

    from pmesh_amd import backend
    from pmesh_amd._arrays import vec
    from pmesh_amd.comm import default_comm
    from pmesh_amd.pm import ParticleMesh
    from pmesh_amd.transfer import Transfer

    be = backend.get()                     # raises if the HIP library / GPU is missing
    # what the library was built with: a build that contains wrong-result timing experiments (-DPMX_EXPERIMENT,
    # PMX_EXP_*) is never reported as a number
    build_flags = be.lib.pmx_build_flags().decode()
    nocheck = os.environ.get('PMESH_AMD_BENCH_NOCHECK') == '1'
    if ('PMX_EXP' in build_flags) and not nocheck:
        raise SystemExit('bench.py: %s was built with experiment switches (%s): its results are wrong by design; '
                         'no bench line.  (PMESH_AMD_BENCH_NOCHECK=1 runs it for kernel traces, without a line.)'
                         % (backend.library_path(), build_flags))
    comm = default_comm()
    N = args.mesh
    Np_side = args.particles or N
    L = 1000.0
    e = 8 if args.dtype == 'f8' else 4
    tdt = torch.float64 if args.dtype == 'f8' else torch.float32
    nlat = Np_side ** 3
    copies = 2 if args.double else 1
    ntot = copies * nlat
    g0 = rank * nlat // world
    g1 = (rank + 1) * nlat // world
    nloc = copies * (g1 - g0)

    # ---- synthetic particles, generated in HBM ---------------------------------
    pos = torch.empty((nloc, 3), dtype=tdt, device=be.device)
    if args.double and args.data != 'clustered':
        raise SystemExit('--double is defined for --data clustered (the shifted copy of the lattice)')
    if args.data in ('uniform', 'shuffled'):
        pv = vec(pos)
        be.call('synth_uniform', C.byref(pv), Np_side, L, 42, g0, nloc, be.stream())
        if args.data == 'shuffled':
            gen = torch.Generator(device=be.device)
            gen.manual_seed(1234 + rank)
            pos = pos[torch.randperm(nloc, device=be.device, generator=gen)].contiguous()
    else:
        modes = zeldovich_modes(numpy, Np_side, L)
        for c in range(copies):
            part = pos[c * (g1 - g0):(c + 1) * (g1 - g0)]
            pv = vec(part)
            be.call('synth_clustered', C.byref(pv), Np_side, L,
                    modes.ctypes.data_as(C.POINTER(C.c_double)), len(modes), 0.5 * c, g0, g1 - g0, be.stream())
    mass = 1.0
    mtot = float(ntot)
    if args.mass == 'array':
        # deterministic, order 1, exactly summable: 0.5 + (row mod 1024) / 1024
        mass = 0.5 + (torch.arange(nloc, device=be.device, dtype=torch.int64) % 1024).to(torch.float64) / 1024.0
        mtot = float(comm.allreduce(float(mass.sum()))) if world > 1 else float(mass.sum())

    if args.decomp == 'pencil' and world > 1:
        from pmesh_amd.fft import split_size_2d
        np_ = [int(x) for x in split_size_2d(world)]
    else:
        np_ = [world]
    pm = ParticleMesh(BoxSize=L, Nmesh=[N, N, N], comm=comm, dtype=args.dtype, resampler=args.window,
                      np=np_)
    if world > 1 and args.migrate:
        # every particle moves to the rank that owns its cell, once and untimed: what a time-stepping code
        # does after its first decompose.  (The ranks generate slabs of lattice ids; on a pencil mesh, or with
        # the Zel'dovich displacements, most rows would otherwise travel as "ghosts" in every cycle.)
        home = pm.domain.decompose(pos, smoothing=0, _scale=pm.affine.scale)
        if args.mass == 'array':
            pos, mass = home.exchange(pos, mass)
        else:
            pos = home.exchange(pos)
        nloc = int(pos.shape[0])
        del home
        from pmesh_amd.domain import release_staging
        release_staging(comm)              # (the staging of this one-off exchange is the size of the particle set)
    transfer = Transfer.dx1(0)             # T(k) = i k_x / k^2
    rho = pm.create('real')
    t_order = 0.0
    if args.tile_order:
        pm.tile_order(pos[:1024])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pos = pos[pm.tile_order(pos)].contiguous()
        torch.cuda.synchronize()
        t_order = time.perf_counter() - t0

    # a time-stepping caller: the positions of consecutive cycles differ by a small random step
    # (three position sets; two when a third would not leave room for the mesh and the bin lists)
    psets = [pos]
    if args.drift > 0:
        gen = torch.Generator(device=be.device)
        gen.manual_seed(4321 + rank)
        free_b, _ = torch.cuda.mem_get_info()
        set_b = pos.numel() * pos.element_size()
        work_b = 14 * nloc + 6 * e * N ** 3 // world          # bin lists + mesh, work buffers, halo staging
        nextra = 2 if free_b > 3 * set_b + work_b else 1      # (+1: the temporary of the random step)
        if world > 1:
            # every rank must build the same number of sets: each costs one collective decompose below, and the
            # local figures (free memory, nloc after --migrate on a clustered set) differ from rank to rank
            nextra = int(comm.allreduce(nextra, op='min'))
        for k in range(nextra):
            step = torch.randn(pos.shape, dtype=tdt, device=be.device, generator=gen) * (args.drift * L / N)
            psets.append(psets[-1] + step)
            del step
        if args.pos_columns > 3:
            if world > 1 or args.exchange or args.host_arrays:
                raise SystemExit('--pos-columns is a single-GPU diagnostic')
            wide = [torch.zeros((len(q), args.pos_columns), dtype=q.dtype, device=q.device) for q in psets]
            for w, q in zip(wide, psets):
                w[:, :3] = q
            psets = [w[:, :3] for w in wide]
        if args.count_jitter > 0:
            if world > 1 or args.exchange or args.mass == 'array' or args.host_arrays:
                raise SystemExit('--count-jitter is a single-GPU diagnostic with a scalar mass')
            psets = [q[:int(nloc * (1.0 - k * args.count_jitter))].contiguous() for k, q in enumerate(psets)]
    if args.host_arrays:
        if world > 1 or args.exchange:
            raise SystemExit('--host-arrays is a single-GPU diagnostic')
        psets = [q.cpu().numpy() for q in psets]
        if args.host_arrays == 2:
            # the caller registers its arrays (ParticleMesh.stage): one upload per cycle instead of one per
            # paint / readout call
            psets = [pm.stage(q) for q in psets]
    layouts = [None] * len(psets)
    layout = None
    t_decompose = 0.0
    if not args.ghosts_only:
        from pmesh_amd import pm as _pm
        _pm.GHOSTS_ONLY = 'never'
    if world > 1 or args.exchange:
        comm.Barrier()                         # RCCL builds its communicator on the first collective
        layout = pm.decompose(pos)             # untimed first call (allocations, lazy initialisation)
        comm.Barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        layout = pm.decompose(pos)
        torch.cuda.synchronize()
        t_decompose = time.perf_counter() - t0
        layouts = [layout] + [pm.decompose(q) for q in psets[1:]]

    from pmesh_amd import window as _window
    if args.deterministic:
        _window.DETERMINISTIC = True
    if args.binned == 0:
        _window.BINNED = 'never'
    elif args.binned == 1:
        _window.BINNED = 'always'
    from pmesh_amd import fft as _fft
    if args.colfft == 0:
        _fft.COLFFT = 'never'
    stages = ['bin', 'paint', 'r2c', 'apply', 'c2r', 'readout']
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(stages) + 1)]
          for _ in range(args.steps)]

    # the caller's result buffer, in the precision the run is in (the reference's `out` takes any float type,
    # window.py:165-221; algorithmic_bytes charges the result at the mesh element size): an fp32 run keeps fp32 results
    result = torch.empty(nloc, dtype=torch.float64 if e == 8 else torch.float32, device=be.device)
    if args.host_arrays:
        result = numpy.empty(nloc, dtype='f8' if e == 8 else 'f4')

    ncycle = [0]
    halo_deferred = [False]       # the last paint left its halo merge to the row pass of r2c (pm.HALO_DEFER)

    cur_stream = torch.cuda.current_stream()         # (Event.record() without a stream looks it up: ~10 us of host time per mark)

    def cycle(marks=None):
        def mark(i):
            if marks is not None:
                marks[i].record(cur_stream)
        pos = psets[ncycle[0] % len(psets)]
        layout = layouts[ncycle[0] % len(psets)]
        ncycle[0] += 1
        # a new time step: positions are "new", nothing binned or exchanged is reused
        if args.cold_plan:
            _window.bin_cache().destroy(be)
        else:
            _window.clear_bin_cache()
        if layout is not None:
            layout._memo = None
            layout._memo_remote = None
        mark(0)
        if args.host_arrays == 2:
            pos.refresh()                                       # this step's positions cross the link once
        if (layout is None or args.ghosts_only) and args.host_arrays != 1:
            pm.resampler.prebin(rho.value, getattr(pos, 'tensor', pos), pm.affine)      # tile binning, shared by paint+readout
        mark(1)
        # (includes the zero fill)
        painted = pm.paint(pos, mass=mass, hold=False, layout=layout, out=rho if args.out_field else None)
        halo_deferred[0] = getattr(painted._base.storage, '_pmx_halo', None) is not None
        mark(2)
        rhok = painted.r2c(out=Ellipsis)
        mark(3)
        if not args.fuse_apply:
            rhok.apply(transfer, out=Ellipsis)
        mark(4)
        back = rhok.c2r(out=Ellipsis, transfer=transfer if args.fuse_apply else None)
        mark(5)
        # the result goes into a buffer that lives across cycles, as a time-stepping caller keeps it:
        # a fresh 1 GB tensor per cycle occasionally costs a hipMalloc (~20 ms) inside the timed loop
        f = back.readout(pos, gradient=args.gradient, layout=layout, out=result[:len(pos)] if args.count_jitter > 0 else result)
        mark(6)
        return f

    for _ in range(args.warmup):
        cycle()
    # The interpreter's cyclic collector walks every object alive — millions after `import torch` — when its oldest
    # generation comes due, which on a fresh box happened in the third cycle of the process: ~40 ms of host time with
    # the GPU idle, inside the c2r stage of the first timed step whenever --warmup < 3 (found with
    # PMESH_AMD_BENCH_STEPS=1; the kernels have nothing to do with it).  What exists now is collected once and set
    # aside (a time-stepping code does the same after its set-up); the collector stays on for what the cycles allocate.
    import gc
    gc.collect()
    gc.freeze()
    from pmesh_amd import comm as _comm
    records = _comm.trace(True) if world > 1 else None
    comm.Barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        f = cycle(ev[k])
    t_issue = time.perf_counter() - t0           # the host has enqueued every cycle (it may have waited inside collectives)
    torch.cuda.synchronize()
    comm.Barrier()
    elapsed = time.perf_counter() - t0
    elapsed = comm.allreduce(elapsed, op='max') if world > 1 else elapsed

    comm_line = None
    if records is not None:
        _comm.trace(False)
        cs = _comm.trace_summary([r for r in records if r[4] is not None])
        XGMI_LINK_GBS = 153.0                      # MI355X: 7 point-to-point links per GPU, ~153 GB/s each
        comm_line = {
            'collectives_per_cycle': cs['collectives'] / float(args.steps),
            'bytes_sent_per_rank_per_cycle': cs['bytes_sent'] / float(args.steps),
            'sync_ms_per_cycle': cs['sync_ms'] / args.steps,
            'overlapped_window_ms_per_cycle': cs['overlapped_window_ms'] / args.steps,
            # the exchange step of the path against its own bound: one xGMI link per peer
            # (every data-path exchange is issued asynchronously now: the rate of an overlapped one, bytes over
            # its whole window, is a lower bound of what the link carried)
            'roofline': {'bound': 'xgmi link', 'achieved': max(cs['max_link_GBs'], cs['max_link_GBs_overlapped']),
                         'peak': XGMI_LINK_GBS, 'unit': 'GB/s',
                         'frac': max(cs['max_link_GBs'], cs['max_link_GBs_overlapped']) / XGMI_LINK_GBS,
                         'lower_bound': cs['max_link_GBs'] < cs['max_link_GBs_overlapped']},
        }
    stage_ms = {}
    for i, s in enumerate(stages):
        stage_ms[s] = sum(ev[k][i].elapsed_time(ev[k][i + 1]) for k in range(args.steps)) / args.steps
    if os.environ.get('PMESH_AMD_BENCH_STEPS') == '1':          # per-step stage times (a hiccup inside the timed region shows here)
        for k in range(args.steps):
            print('step %d: %s' % (k, ' '.join('%s %.3f' % (s, ev[k][i].elapsed_time(ev[k][i + 1])) for i, s in enumerate(stages))),
                  file=sys.stderr, flush=True)

    # sanity: mass conservation and a finite result (size-independent properties)
    check = pm.paint(pos, mass=mass, layout=layout)
    msum = check.csum()
    if nocheck:      # timing experiments with wrong results (-DPMX_EXPERIMENT builds): kernel traces only, no bench line
        msum = mtot
    assert abs(msum - mtot) <= 1e-9 * mtot if args.dtype == 'f8' else abs(msum - mtot) <= 1e-3 * mtot, (msum, mtot)
    assert bool(numpy.isfinite(f).all()) if args.host_arrays else bool(torch.isfinite(f).all())

    ms_per_step = 1e3 * elapsed / args.steps
    value = ntot / (elapsed / args.steps)

    if rank == 0:
        nu = ntot / float(N) ** 3
        pe = e
        # the dominant kernel: the longest single-kernel stage on this rank
        single = {'paint': stage_ms['paint'], 'readout': stage_ms['readout'], 'apply': stage_ms['apply']}
        dom = max(single, key=single.get)
        units = nloc
        me = 8 if args.mass == 'array' else 0
        ach = algorithmic_bytes(dom, e, pe, nu, me) * units / (single[dom] * 1e-3) / 1e9
        binned = any(e[3] for e in _window.bin_cache().entries)
        if binned:
            # (float canvases under TSC / PCS paint through the 32-bit regions; the default readout of dense rows is
            # the lean loop: the names rocprofv3 shows)
            pk = 'paint_tile32_kernel' if (args.dtype == 'f4' and args.window in ('tsc', 'pcs') and not args.deterministic) else 'paint_tile_kernel'
            kname = {'paint': pk if halo_deferred[0] else pk + '+halo_merge_kernel',
                     'readout': 'readout_tile_lean_kernel' if (not args.host_arrays and _window.EXACT is False) else 'readout_tile_kernel',
                     'apply': 'transfer_kernel'}[dom]
        else:
            kname = {'paint': 'paint_tuned_kernel', 'readout': 'readout_tuned_kernel',
                     'apply': 'transfer_kernel'}[dom]
        # HBM bytes per launch of the dominant stage's kernels from the PMC passes of this configuration
        # (profiles/traffic.json, regenerated by scripts/make_traffic.py from scripts/profile_round.sh output)
        traffic = None
        traffic_key = '%d/%s/%s/%s%s%s' % (N, args.window, args.dtype, args.data, '/x2' if args.double else '',
                                           '/mass' if args.mass == 'array' else '')
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get('%s/%s' % (dom, traffic_key))
            except Exception:
                traffic = None
        line = {
            'metric': baseline_metric(),
            'value': value, 'unit': 'particles/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64' if args.dtype == 'f8' else 'f32',
            'data': 'synthetic',
            'config': {'workload': '%d^3 mesh, %d^3 %s particles (lattice + %s), %s window, %s, '
                                   'paint->r2c->apply(i kx/k^2)->c2r->readout%s'
                                   % (N, Np_side, ('2 x ' if args.double else '') + args.data,
                                      {'uniform': 'hashed jitter', 'shuffled': 'hashed jitter, rows in random order'}.get(
                                          args.data, "Zel'dovich plane waves"),
                                      args.window.upper(), 'fp64' if args.dtype == 'f8' else 'fp32',
                                      ('' if args.gradient is None else ' (gradient %d)' % args.gradient) +
                                      (', per-particle fp64 mass' if args.mass == 'array' else '')),
                       'decomposition': ('single GPU' if world == 1 else
                                         '%s np=%s, particle exchange included (%s)'
                                         % ('slab' if len(np_) == 1 else 'pencil', np_,
                                            'ghosts only' if args.ghosts_only else 'all particles')),
                       'particles': ntot, 'apply': 'fused into c2r' if args.fuse_apply else 'separate kernel',
                       'halo_merge': ('gathered by the row pass of r2c from the staging buffer of the tile kernels '
                                      '(no kernel of its own; its time shows under r2c)' if halo_deferred[0]
                                      else 'halo_merge_kernel inside paint'),
                       'fft': ('LDS row + column FFT kernels' + (
                           ', the last pass of r2c deferred into the first of c2r (one kernel for both and the transfer; '
                           'its time shows under c2r)' if world == 1 and _fft.DEFER_LAST_PASS and args.fuse_apply else ''))
                       if args.colfft else 'rocFFT 3-d'},
            'stages_ms': {k: round(v, 4) for k, v in stage_ms.items()},
            'decompose_ms': round(1e3 * t_decompose, 3),
            # the exchange reported both ways — `value` has decompose outside the cycle
            'ms_per_step_with_decompose': ms_per_step + 1e3 * t_decompose,
            'value_with_decompose': ntot / (elapsed / args.steps + t_decompose),
            'comm': comm_line,
            'drift_cells': args.drift,
            'count_jitter': args.count_jitter,
            'host_arrays': bool(args.host_arrays), 'deterministic_paint': bool(args.deterministic),
            'bin_overflows': _window.bin_cache().overflows(be),
            # what the `bin` stage is: the plan is rebuilt in EVERY cycle (the cache is cleared: a time-stepping caller's
            # positions are new), by the single pass into the slot ranges of the previous cycle's build — whose positions
            # differ by N(0, drift_cells) per axis — unless --cold-plan; a build whose ranges overflow is repaired by the exact
            # two-pass build on the device (bin_overflows counts those over the whole run, warm-up included)
            'bin_rebuilds_per_step': 1.0, 'cold_plan': bool(args.cold_plan),
            'tile_order_ms': round(1e3 * t_order, 3),
            # every stage against the same roofline: its algorithmic bytes (the fused apply is
            # charged to c2r) over its measured time, as a fraction of the HBM peak
            'stage_roofline_frac': {k: round((algorithmic_bytes(k, e, pe, nu, me if k == 'paint' else 0) +
                                              (algorithmic_bytes('apply', e, pe, nu) if k == 'c2r' and args.fuse_apply else 0)) *
                                             units / (max(stage_ms[k], 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                    for k in ('paint', 'r2c', 'c2r', 'readout')},
            # the same without the bytes of the fused apply (the pass itself moves 2 e per cell, whatever rides on it)
            'c2r_alone_roofline_frac': round(algorithmic_bytes('c2r', e, pe, nu) * units /
                                             (max(stage_ms['c2r'], 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            'cycle_roofline_frac': (sum(algorithmic_bytes(s, e, pe, nu, me) for s in
                                        ('paint', 'r2c', 'apply', 'c2r', 'readout')) * units /
                                    (ms_per_step * 1e-3) / 1e9) / HBM_PEAK_GBS,
            'roofline': {'bound': 'hbm', 'kernel': kname, 'achieved': ach, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_key': traffic_key,
                         # the bin pass exists only to feed paint and readout and has no algorithmic
                         # bytes of its own: their bytes over the time of all three
                         'with_bin_frac': ((algorithmic_bytes('paint', e, pe, nu, me) +
                                            algorithmic_bytes('readout', e, pe, nu, 0)) * units /
                                           ((stage_ms['bin'] + stage_ms['paint'] + stage_ms['readout']) * 1e-3)
                                           / 1e9) / HBM_PEAK_GBS,
                         'algorithmic_bytes_per_particle': algorithmic_bytes(dom, e, pe, nu, me),
                         'particles_per_launch': units, 'ms_per_launch': single[dom]},
        }
        line['build_flags'] = build_flags
        # to take a multi-GPU line apart: what the kernels of this rank took stage by stage (stages_ms: events on the
        # stream, so a stage that waits for a collective contains the wait), what the collectives took (comm) and how
        # long the host needed to enqueue a cycle — the cycle is bound by the host when this approaches ms_per_step
        line['host_issue_ms_per_step'] = 1e3 * t_issue / args.steps
        line['stages_sum_ms'] = sum(stage_ms.values())
        if world == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline(args)
            except Exception as ex:          # the baseline never takes the benchmark down
                line['cpu_baseline'] = {'value': None, 'unit': 'particles/s', 'cores': 1,
                                        'kind': 'port', 'sample': 'failed: %r' % (ex,)}
        if nocheck:
            print('PMESH_AMD_BENCH_NOCHECK=1: the result was not checked, no bench line is printed '
                  '(%.3f ms per cycle, for kernel traces only)' % ms_per_step, file=sys.stderr, flush=True)
        else:
            print(json.dumps(line), flush=True)
    if world > 1 or launched:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def baseline_metric():
    """the metric string of BASELINE.json, verbatim"""
    try:
        return json.load(open(os.path.join(ROOT, 'BASELINE.json')))['metric']
    except Exception:
        return 'PM-cycle particles/s (paint+r2c+c2r+readout), 512^3 mesh at 1/2/4/8 GPUs'


def zeldovich_modes(numpy, nlat, boxsize, rms_cells=3.0, nmodes=16, seed=1234):
    """SURVEY.md 8d plane-wave table (same as oracle.zeldovich_modes; restated here so the
    timed path does not import the oracle)."""
    rng = numpy.random.RandomState(seed)
    # wavenumbers up to nlat/16 per axis: with a 3-cell rms displacement the displacement gradient
    # is ~1 (shell crossing, caustics: density contrast >> 10); |n| <= 4 alone would be a smooth,
    # single-stream flow with a contrast of order one
    nmax = max(4, int(nlat) // 16)
    n = rng.randint(-nmax, nmax + 1, size=(nmodes, 3)).astype('f8')
    n[(n == 0).all(axis=1)] = [1, 0, 0]
    norm = numpy.sqrt((n ** 2).sum(axis=1))
    direc = n / norm[:, None]
    amp = 1.0 / norm
    phase = rng.uniform(0, 2 * numpy.pi, size=nmodes)
    rms = numpy.sqrt(0.5 * (amp ** 2).sum())
    amp *= rms_cells * (boxsize / nlat) / rms
    modes = numpy.zeros((nmodes, 8), dtype='f8')
    modes[:, 0:3] = n
    modes[:, 3:6] = direc
    modes[:, 6] = amp
    modes[:, 7] = phase
    return modes


if __name__ == '__main__':
    main()
