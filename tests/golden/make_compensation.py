#!/usr/bin/env python3
"""Golden vectors of ResampleWindow.get_compensation (pmesh/window.py:65-80) from the REFERENCE itself.

Built like tests/golden/make_golden.py (the reference's Cython `_window` extension compiled in a scratch
directory outside the repository, its own window.py imported from there); run in the build container:

    python tests/golden/make_compensation.py

Stores, for every tuned and generic window, the function returned by get_compensation() applied to a seeded
complex block `v` on a 3-d grid of circular frequencies `w` (what ComplexField.apply(kind='circular') hands
it), at native support and resized to 6.
"""
import os
import shutil
import sys
import tempfile

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G      # noqa: E402


def main():
    scratch = tempfile.mkdtemp(prefix='pmesh_ref_build_')
    out = {}
    try:
        G.build_reference(scratch)
        G.install_mpi_placeholder()
        sys.path.insert(0, scratch)
        from pmesh import window
        rs = numpy.random.RandomState(4242)
        n = (8, 6, 5)
        w = [2 * numpy.pi * numpy.fft.fftfreq(n[0]).reshape(-1, 1, 1),
             2 * numpy.pi * numpy.fft.fftfreq(n[1]).reshape(1, -1, 1),
             2 * numpy.pi * numpy.arange(n[2]).reshape(1, 1, -1) / (2.0 * (n[2] - 1))]
        v = rs.normal(size=n) + 1j * rs.normal(size=n)
        out['w0'], out['w1'], out['w2'], out['v'] = w[0], w[1], w[2], v
        for name in G.TUNED + G.GENERIC:
            W = window.windows[name]
            out['%s/native' % name] = W.get_compensation()(w, v)
            out['%s/resize6' % name] = W.resize(6).get_compensation()(w, v)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    numpy.savez_compressed(os.path.join(HERE, 'compensation.npz'), **out)
    print('compensation.npz', os.path.getsize(os.path.join(HERE, 'compensation.npz')), 'bytes')


if __name__ == '__main__':
    main()
