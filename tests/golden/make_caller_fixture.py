"""Fixture for tests/test_caller_nbody.py: the force of examples/nbody.py:199-218 on small seeded inputs,
computed on the CPU from the reference's own window kernels (oracle/_ref: pmesh/_window_imp.c compiled
where it lies, through oracle.Window(which='ref')) and numpy.fft under the reference's normalisation
(pm.py:692: r2c = rfftn / prod(Nmesh); pm.py:1017: c2r = irfftn * prod(Nmesh)), with the transfer function
of examples/nbody.py:162-171 evaluated by numpy on the full wavenumber grid.

    python tests/golden/make_caller_fixture.py      (in the build container; needs oracle/_ref)
"""
import os
import sys

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O      # noqa: E402

Om0 = 0.31


def force_numpy(Nmesh, BoxSize, Q, S, kind, which):
    N = Nmesh
    X = S + Q
    W = O.Window(kind, which=which)
    aff = O.Affine(3, scale=1.0 * N / BoxSize, translate=0, period=N)
    rho = numpy.zeros((N, N, N))
    W.paint(rho, X, transform=aff)
    rho *= 1.0 * N ** 3 / len(X)
    rhok = numpy.fft.rfftn(rho) / float(N) ** 3
    k1 = 2 * numpy.pi / BoxSize * numpy.fft.fftfreq(N, 1.0 / N)
    k1[N // 2] = -abs(k1[N // 2])                      # pm.py:1200-1226: the Nyquist mode is negative
    kz = 2 * numpy.pi / BoxSize * numpy.arange(N // 2 + 1)
    kz[-1] = -kz[-1]
    k = [k1[:, None, None], k1[None, :, None], kz[None, None, :]]
    k2 = k[0] ** 2 + k[1] ** 2 + k[2] ** 2
    k2[k2 == 0] = 1.0
    F = numpy.empty_like(Q)
    C = BoxSize / N
    for d in range(3):
        w = k[d] * C
        kfinite = 1.0 / C * 1 / 6.0 * (8 * numpy.sin(w) - numpy.sin(2 * w))
        fk = 1j * kfinite / k2 * rhok
        f = numpy.fft.irfftn(fk, s=(N, N, N), axes=(0, 1, 2)) * float(N) ** 3
        F[..., d] = W.readout(numpy.ascontiguousarray(f), X, transform=aff)
    return 1.5 * Om0 * F


def main():
    which = 'ref' if O.have_ref() else 'oracle'
    out = {}
    for tag, N, nside, kind in (('n16_cic', 16, 16, 'tunedcic'), ('n64_cic', 64, 20, 'tunedcic'),
                                ('n16_tsc', 16, 12, 'tunedtsc')):
        rs = numpy.random.RandomState(1000 + N + nside)
        BoxSize = 100.0
        g = (numpy.arange(nside) + 0.5) * BoxSize / nside
        Q = numpy.stack(numpy.meshgrid(g, g, g, indexing='ij'), axis=-1).reshape(-1, 3)
        S = rs.normal(0, 0.6 * BoxSize / N, size=Q.shape)
        out[tag + '_Q'] = Q
        out[tag + '_S'] = S
        out[tag + '_F'] = force_numpy(N, BoxSize, Q, S, kind, which)
        out[tag + '_meta'] = numpy.array([N, BoxSize, Om0])
    out['kernels'] = numpy.array(which)
    numpy.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'caller_nbody.npz'), **out)
    print('wrote caller_nbody.npz with the %s kernels' % which)


if __name__ == '__main__':
    main()
