#!/usr/bin/env python3
"""halo_fuzz.py for slab ranks: random meshes / rank counts (uneven slabs too) / windows / canvas types / particle sets,
the ranks as threads of this process on the one GPU.  pm.paint(pos, layout=...) leaves the halo merge of a rank's tile
kernels to the row pass of its slab r2c (fft.Plan._slab_row_forward) — against the eagerly merged field.
python scripts/halo_fuzz_slabs.py [cases=60] [seed=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
import pmesh_amd.pm as pmod
from pmesh_amd import backend, window
from pmesh_amd.pm import ParticleMesh
from tests import thread_comm

be = backend.get()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = numpy.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N0s, N1s, N2s = [64, 72, 96, 128, 192], [64, 128, 192], [128, 256, 512]
window.BINNED = 'auto'
window.BINNED_MIN_PARTICLES = 1 << 30        # set per case: own particles through the tile kernels, ghosts direct
stat = {'worst': 0.0, 'deferred': 0, 'ranks': 0}
for c in range(cases):
    nmesh = (int(rs.choice(N0s)), int(rs.choice(N1s)), int(rs.choice(N2s)))
    P = int(rs.choice([2, 3, 4, 8]))
    dtype = str(rs.choice(['f8', 'f4']))
    name = str(rs.choice(['cic', 'tsc', 'pcs']))
    box = rs.uniform(0.5, 300.0, size=3)
    ntot = int(rs.uniform(0.4, 1.2) * numpy.prod(nmesh))
    seed = int(rs.randint(1 << 30))
    with_mass = rs.rand() < 0.5
    out = {}
    print('case %d: %s P=%d %s %s n=%d mass=%s' % (c, nmesh, P, dtype, name, ntot, with_mass), flush=True)

    def body(comm):
        g = torch.Generator(device='cpu').manual_seed(seed + comm.rank)
        n = ntot // P
        pos0 = (torch.rand(n, 3, generator=g, dtype=torch.float64) * torch.as_tensor(box) * 1.2 - 0.1 * torch.as_tensor(box)).cuda()
        mass0 = (torch.rand(n, generator=g, dtype=torch.float64) + 0.5).cuda()
        pm = ParticleMesh(Nmesh=nmesh, BoxSize=box, dtype=dtype, resampler=name, comm=comm, np=[P])
        home = pm.decompose(pos0, smoothing=0)
        pos, mass = home.exchange(pos0), home.exchange(mass0)
        if not with_mass:
            mass = 1.0
        layout = pm.decompose(pos)
        nown = comm.allreduce(len(pos), op='min')
        nghost = comm.allreduce(int(layout.remote_recvlength), op='max')
        if comm.rank == 0:
            window.BINNED_MIN_PARTICLES = max(nghost + 1, 1000)
        comm.Barrier()
        binned = nown >= window.BINNED_MIN_PARTICLES
        res = {}
        for mode in ('never', 'fresh'):
            comm.Barrier()
            if comm.rank == 0:
                pmod.HALO_DEFER = mode
            comm.Barrier()
            window.clear_bin_cache()
            f = pm.paint(pos, mass=mass, layout=layout)
            owed = getattr(f._base.storage, '_pmx_halo', None) is not None
            res[mode] = (f.r2c(out=Ellipsis).value.clone(), owed)
        ek, lk = res['never'][0], res['fresh'][0]
        scale = comm.allreduce(float(ek.abs().max()) if ek.numel() else 0.0, op='max')
        err = (float((lk - ek).abs().max()) if ek.numel() else 0.0) / scale
        out[comm.rank] = (err, res['fresh'][1], res['never'][1], binned, tuple(ek.shape))
        comm.Barrier()

    thread_comm.run_ranks(P, body)
    tol = 2e-13 if dtype == 'f8' else 4e-6
    for r, (err, owed, eager_owed, binned, shape) in sorted(out.items()):
        stat['worst'] = max(stat['worst'], err / tol)
        stat['deferred'] += owed
        stat['ranks'] += 1
        if err > tol or eager_owed:
            print('FAILED case %d rank %d of %d: %s %s %s: err %.2e owed %s / %s block %s' % (c, r, P, nmesh, dtype, name, err, owed, eager_owed, shape), flush=True)
            sys.exit(1)
print('%d cases, %d rank-paints of %d deferred, worst error / tolerance %.3f' % (cases, stat['deferred'], stat['ranks'], stat['worst']))
