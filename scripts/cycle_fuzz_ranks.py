#!/usr/bin/env python3
"""The whole PM cycle (decompose, paint, r2c, transfer, c2r, readout) on P thread ranks of the one GPU against the same
cycle on one rank: random meshes (cubic or not, lengths the LDS kernels take and lengths they leave to rocFFT), rank
counts and process meshes (slabs, pencils, uneven blocks), windows, canvas types, particle sets (uniform / a blob,
every rank starting with an arbitrary share), masses, fused or separate transfer, gradient readouts.
python scripts/cycle_fuzz_ranks.py [cases=40] [seed=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from pmesh_amd import backend, window
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer
from tests import thread_comm

be = backend.get()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else None      # run just this case (the draws before it are made)
rs = numpy.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
SIDES = [32, 48, 64, 72, 96, 100, 128, 160, 192, 256]
worst = 0.0
for c in range(cases):
    nmesh = tuple(int(rs.choice(SIDES)) for _ in range(3))
    if rs.rand() < 0.3:
        nmesh = (nmesh[0],) * 3
    P = int(rs.choice([2, 3, 4, 6, 8]))
    nps = [[P]] + [[a, P // a] for a in (2, 3, 4) if P % a == 0 and P // a > 1] + [None]
    np_ = nps[rs.randint(len(nps))]
    dtype = str(rs.choice(['f8', 'f4']))
    name = str(rs.choice(['cic', 'tsc', 'pcs', 'nnb']))
    box = rs.uniform(1.0, 500.0, size=3)
    n = int(rs.uniform(0.3, 1.5) * numpy.prod(nmesh))
    seed = int(rs.randint(1 << 30))
    with_mass, fuse, blob = rs.rand() < 0.5, rs.rand() < 0.5, rs.rand() < 0.3
    grad = [None, 0, 1, 2][rs.randint(4)]
    tdir = int(rs.randint(3))
    if ONLY is not None and c != ONLY:
        rs.permutation(n)            # (the draw the case would have made: the stream stays the one of a full run)
        continue
    print('case %d: %s P=%d np=%s %s %s n=%d mass=%s fuse=%s blob=%s grad=%s' % (c, nmesh, P, np_, dtype, name, n, with_mass, fuse, blob, grad), flush=True)
    g = torch.Generator(device='cpu').manual_seed(seed)
    tb = torch.as_tensor(box)
    pos = torch.rand(n, 3, generator=g, dtype=torch.float64) * tb * 1.3 - 0.15 * tb
    if blob:
        pos[: n // 3] = tb * 0.61 + torch.randn(n // 3, 3, generator=g, dtype=torch.float64) * (tb / torch.as_tensor(numpy.array(nmesh, dtype='f8')) * 2.0)
    mass = torch.rand(n, generator=g, dtype=torch.float64) + 0.5
    pos, mass = pos.cuda(), mass.cuda()
    T = Transfer.dx1(tdir)

    def cycle(pm, p, m, layout):
        rho = pm.paint(p, mass=m if with_mass else 1.0, layout=layout)
        ck = rho.r2c(out=Ellipsis)
        back = ck.c2r(out=Ellipsis, transfer=T) if fuse else ck.apply(T, out=Ellipsis).c2r(out=Ellipsis)
        return back.readout(p, gradient=grad, layout=layout)
    window.clear_bin_cache()
    one = cycle(ParticleMesh(Nmesh=nmesh, BoxSize=box, dtype=dtype, resampler=name), pos, mass, None)
    one = torch.as_tensor(numpy.asarray(one.cpu() if hasattr(one, 'cpu') else one))
    shares = numpy.array_split(rs.permutation(n), P)
    out = {}

    def body(comm):
        idx = torch.as_tensor(shares[comm.rank]).cuda()
        p, m = pos[idx].contiguous(), mass[idx].contiguous()
        pm = ParticleMesh(Nmesh=nmesh, BoxSize=box, dtype=dtype, resampler=name, comm=comm, np=np_)
        layout = pm.decompose(p)
        f = cycle(pm, p, m, layout)
        f2 = cycle(pm, p, m, layout)                 # (a second cycle: plans and buffers reused)
        out[comm.rank] = [torch.as_tensor(numpy.asarray(x.cpu() if hasattr(x, 'cpu') else x)) for x in (f, f2)]
        comm.Barrier()
    thread_comm.run_ranks(P, body)
    scale = float(one.abs().max()) or 1.0        # (the derivative of the NNB window is zero: compare absolutely)
    tol = 1e-11 if dtype == 'f8' else 2e-4
    for r in range(P):
        for k, f in enumerate(out[r]):
            err = float((f - one[shares[r]]).abs().max()) / scale
            worst = max(worst, err / tol)
            if not err <= tol:
                if dtype == 'f4':
                    # float meshes: some configurations (a gradient readout of a differentiated field: cancellation) sit
                    # this far from the double-precision cycle on ONE rank already; the P-rank result must not be further
                    window.clear_bin_cache()
                    truth = cycle(ParticleMesh(Nmesh=nmesh, BoxSize=box, dtype='f8', resampler=name), pos, mass, None)
                    truth = torch.as_tensor(numpy.asarray(truth.cpu() if hasattr(truth, 'cpu') else truth)).double()
                    ts = float(truth.abs().max()) or 1.0
                    e1 = float((one.double() - truth).abs().max()) / ts
                    eP = float((f.double() - truth[shares[r]]).abs().max()) / ts
                    print('  case %d rank %d: %.2e from one rank; against the f8 cycle: one rank %.2e, %d ranks %.2e' % (c, r, err, e1, P, eP), flush=True)
                    if eP <= 2 * e1 + 1e-5:
                        continue
                print('FAILED case %d rank %d cycle %d: err %.2e (tolerance %.0e)' % (c, r, k, err, tol), flush=True)
                sys.exit(1)
print('%d cases ok, worst error / tolerance %.3f' % (cases, worst))
