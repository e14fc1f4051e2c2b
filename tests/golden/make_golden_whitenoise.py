#!/usr/bin/env python3
"""Golden vectors for the white-noise generator, from the REFERENCE itself.

Run in the build container only (needs /root/reference, gcc and Cython):

    python tests/golden/make_golden_whitenoise.py

Builds the reference's Cython extension `pmesh._whitenoise` (sources as listed in the
reference's setup.py:36-43: _whitenoise.pyx, _whitenoise_imp.c and the vendored GSL RANLUX
under pmesh/gsl) in a scratch directory OUTSIDE the repository, imports the reference's own
`pmesh/whitenoise.py` from that scratch copy and stores `generate(...)` outputs for seeded
cases: inputs (Nmesh, start, shape, seed, unitary, dtype) + the filled block.  The fixture is
data; tests/test_whitenoise.py checks oracle/pmesh_oracle.c against it bit for bit and the HIP
kernel within the libm tolerance.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # Nmesh, start, shape, seed, unitary, dtype
    ((4, 4, 4), (0, 0, 0), (4, 4, 3), 5463, False, 'c16'),       # the N-GenIC check of the reference's tests
    ((8, 8, 8), (0, 0, 0), (8, 8, 5), 1, False, 'c16'),
    ((8, 8, 8), (0, 0, 0), (8, 8, 5), 1, True, 'c16'),
    ((16, 16, 16), (0, 0, 0), (16, 16, 9), 8, True, 'c16'),
    ((16, 16, 16), (0, 0, 0), (16, 16, 9), 8, False, 'c8'),
    ((16, 16, 16), (3, 5, 2), (9, 7, 5), 8, False, 'c16'),       # a block: decomposition invariance
    ((32, 32, 32), (16, 0, 0), (16, 32, 17), 8, True, 'c16'),    # a slab of a finer mesh: scale invariance
    ((12, 8, 10), (0, 0, 0), (12, 8, 6), 77, False, 'c16'),      # non-cubic (the reference mixes N0 / N1)
    ((6, 10, 8), (1, 2, 0), (5, 8, 5), 4000000000, False, 'c16'),  # seed with bit 31 set
    ((9, 9, 9), (0, 0, 0), (9, 9, 5), 3, False, 'c16'),          # odd mesh: unvisited columns keep seed 0
    ((16, 16, 16), (0, 0, 0), (16, 16, 9), 0, False, 'c16'),     # seed 0 -> 1
]


def main():
    scratch = tempfile.mkdtemp(prefix='pmesh_ref_wn_')
    try:
        pk = os.path.join(scratch, 'pmesh')
        os.makedirs(pk)
        for fn in os.listdir(os.path.join(REF, 'pmesh')):
            if fn == 'whitenoise.py' or fn.startswith('_whitenoise'):
                shutil.copy(os.path.join(REF, 'pmesh', fn), pk)
        shutil.copytree(os.path.join(REF, 'pmesh', 'gsl'), os.path.join(pk, 'gsl'))
        open(os.path.join(pk, '__init__.py'), 'w').close()
        setup = '''
import numpy
from setuptools import setup, Extension
from Cython.Build import cythonize
ext = [Extension("pmesh._whitenoise", ["pmesh/gsl/ranlxd.c", "pmesh/gsl/missing.c", "pmesh/gsl/rng.c",
                 "pmesh/_whitenoise_imp.c", "pmesh/_whitenoise.pyx"], libraries=["m"],
                 include_dirs=["pmesh/gsl", "pmesh", numpy.get_include()])]
setup(name="pmesh", ext_modules=cythonize(ext), packages=["pmesh"])
'''
        open(os.path.join(scratch, 'setup_probe.py'), 'w').write(setup)
        subprocess.check_call([sys.executable, 'setup_probe.py', '-q', 'build_ext', '--inplace'],
                              cwd=scratch, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        sys.path.insert(0, scratch)
        from pmesh.whitenoise import generate
        out = {'ncases': numpy.array(len(CASES))}
        for n, (nmesh, start, shape, seed, unitary, dtype) in enumerate(CASES):
            value = numpy.zeros(shape, dtype=dtype)
            generate(value, start, nmesh, seed, unitary)
            out['%d/nmesh' % n] = numpy.array(nmesh)
            out['%d/start' % n] = numpy.array(start)
            out['%d/seed' % n] = numpy.array(seed, dtype='u8')
            out['%d/unitary' % n] = numpy.array(unitary)
            out['%d/value' % n] = value
        numpy.savez_compressed(os.path.join(HERE, 'whitenoise.npz'), **out)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    print('whitenoise.npz', os.path.getsize(os.path.join(HERE, 'whitenoise.npz')), 'bytes')


if __name__ == '__main__':
    main()
