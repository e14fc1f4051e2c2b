#!/usr/bin/env python3
"""Random mesh shapes through r2c / c2r on the GPU against numpy.fft, and the deferred last pass against the eager
transforms bit for bit (fft.DEFER_LAST_PASS on / off): a wider net than the fixed shapes of tests/test_pm.py.
    python scripts/fft_fuzz.py [cases] [seed]"""
import sys
import numpy
sys.path.insert(0, '.')
import torch
from pmesh_amd import fft as F
from pmesh_amd.pm import ParticleMesh
from pmesh_amd.transfer import Transfer

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = numpy.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lengths = [64, 128, 256, 512, 192, 384, 320, 96, 100, 1024, 768, 640]
worst = 0.0
for case in range(ncase):
    while True:
        Nmesh = [int(rs.choice(lengths)) for _ in range(3)]
        if numpy.prod(Nmesh, dtype='f8') <= 2.0 ** 27:
            break
    dtype = rs.choice(['f8', 'f4'])
    tol = 1e-13 if dtype == 'f8' else 2e-5
    T = Transfer.dx1(int(rs.randint(3))) if rs.rand() < 0.7 else None
    data = rs.normal(size=Nmesh).astype(dtype)
    res = {}
    for defer in (False, True):
        F.DEFER_LAST_PASS = defer
        pm = ParticleMesh(BoxSize=[3.0, 2.0, 5.0], Nmesh=Nmesh, dtype=dtype)
        real = pm.create('real', value=data)
        ck = real.r2c(out=Ellipsis)
        if defer is False:
            spec = numpy.asarray(ck).astype('c16')
        back = ck.c2r(out=Ellipsis, transfer=T) if T is not None else ck.c2r(out=Ellipsis)
        res[defer] = numpy.asarray(back).copy()
        del pm, real, ck, back
    ref = numpy.fft.rfftn(data.astype('f8')) / numpy.prod(Nmesh)
    e1 = numpy.sqrt((abs(spec - ref) ** 2).sum() / (abs(ref) ** 2).sum())
    same = numpy.array_equal(res[False], res[True])
    if T is None:
        e2 = numpy.sqrt(((res[True] - data) ** 2).sum() / (data.astype('f8') ** 2).sum())
    else:
        e2 = 0.0
    worst = max(worst, e1 / tol, e2 / tol)
    flag = 'ok' if (e1 < tol and e2 < 4 * tol and same) else 'FAIL'
    print('%s %-18s %s transfer=%s spectrum %.1e roundtrip %.1e deferred==eager %s' % (flag, Nmesh, dtype, T is not None, e1, e2, same), flush=True)
    if flag != 'ok':
        sys.exit(1)
print('all %d cases ok (worst error / tolerance %.2f)' % (ncase, worst))
