#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference, gcc and Cython):

    python tests/golden/make_golden.py

What it does
  1. builds the reference's Cython extensions `pmesh._window` and
     `pmesh._domain` in a scratch directory OUTSIDE the repository (the
     reference tree is read-only and nothing of it is copied into the repo),
     following SURVEY.md Appendix D;
  2. imports the reference's own `pmesh/window.py` and `pmesh/domain.py` from
     that scratch copy.  `domain.py` does `from mpi4py import MPI`, which is
     not installed here; GridND.decompose only needs `comm.size`, `comm.rank`
     and the count `Alltoall` inside Layout.__init__, so a minimal in-process
     communicator object is handed to it (class VirtualComm below: it is an
     argument value, no reference code is altered);
  3. calls ResampleWindow.paint/readout, get_fwindow and GridND.decompose on
     seeded inputs and stores inputs + outputs as .npz files.

The fixtures are DATA (inputs and the reference's outputs).  tests/test_oracle.py
checks oracle/pmesh_oracle.c against them bit for bit; the GPU tests check the
HIP kernels against the oracle and against these files.
"""
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def build_reference(scratch):
    pk = os.path.join(scratch, 'pmesh')
    os.makedirs(pk)
    for fn in os.listdir(os.path.join(REF, 'pmesh')):
        if fn in ('window.py', 'domain.py') or fn.startswith('_window') or fn == '_domain.pyx':
            shutil.copy(os.path.join(REF, 'pmesh', fn), pk)
    open(os.path.join(pk, '__init__.py'), 'w').close()  # avoid `from .pm import ParticleMesh`
    setup = '''
import numpy
from setuptools import setup, Extension
from Cython.Build import cythonize
ext = [Extension("pmesh._domain", ["pmesh/_domain.pyx"], include_dirs=["./", numpy.get_include()]),
       Extension("pmesh._window", ["pmesh/_window.pyx", "pmesh/_window_imp.c"], libraries=["m"],
                 include_dirs=["./", numpy.get_include()])]
setup(name="pmesh", ext_modules=cythonize(ext), packages=["pmesh"])
'''
    open(os.path.join(scratch, 'setup_probe.py'), 'w').write(setup)
    subprocess.check_call([sys.executable, 'setup_probe.py', '-q', 'build_ext', '--inplace'],
                          cwd=scratch, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


class VirtualComm(object):
    """What GridND.decompose touches of a communicator (domain.py:92-123, 380-383)."""
    def __init__(self, size, rank=0):
        self.size = size
        self.rank = rank

    def Barrier(self):
        pass

    def Alltoall(self, send, recv):
        recv[...] = send  # not used by the fixtures

    def allgather(self, x):
        return [x] * self.size


def install_mpi_placeholder():
    mpi4py = types.ModuleType('mpi4py')
    MPI = types.ModuleType('mpi4py.MPI')
    MPI.COMM_WORLD = VirtualComm(1)
    MPI.SUM = None
    mpi4py.MPI = MPI
    sys.modules['mpi4py'] = mpi4py
    sys.modules['mpi4py.MPI'] = MPI


TUNED = ['nnb', 'cic', 'tsc', 'pcs']
GENERIC = ['nearest', 'linear', 'quadratic', 'cubic']


def window_fixtures(window):
    Affine = window.Affine
    out = {}
    rs = numpy.random.RandomState(20240801)

    # ---- case A: 3-d, anisotropic affine, period, slab-local block, masses, gradients
    shape = (6, 8, 10)
    period = (12, 8, 10)
    npart = 400
    posA = rs.uniform(-6, 30, size=(npart, 3))
    massA = rs.uniform(0.5, 1.5, size=npart)
    scaleA = numpy.array([0.5, 1.0, 0.75])
    translA = numpy.array([-3.0, 0.0, 0.25])
    fieldA = rs.normal(size=shape)
    out['A/pos'] = posA
    out['A/mass'] = massA
    out['A/scale'] = scaleA
    out['A/translate'] = translA
    out['A/period'] = numpy.array(period)
    out['A/field'] = fieldA
    for name in TUNED + GENERIC:
        W = window.windows[name]
        for dt in ('f8', 'f4'):
            for diffdir in (None, 0, 1, 2):
                aff = Affine(3, scale=scaleA, translate=translA, period=period)
                real = numpy.zeros(shape, dtype=dt)
                W.paint(real, posA, mass=massA, diffdir=diffdir, transform=aff)
                v = W.readout(fieldA.astype(dt), posA, diffdir=diffdir, transform=aff)
                key = 'A/%s/%s/%s' % (name, dt, 'n' if diffdir is None else diffdir)
                out[key + '/paint'] = real
                out[key + '/readout'] = v

    # ---- case B: f4 positions, f4 mass, f4 out, scalar mass, non periodic, out of range
    shapeB = (5, 7, 4)
    posB = rs.uniform(-2, 9, size=(300, 3)).astype('f4')
    fieldB = rs.normal(size=shapeB).astype('f4')
    out['B/pos'] = posB
    out['B/field'] = fieldB
    for name in TUNED + GENERIC:
        W = window.windows[name]
        real = numpy.zeros(shapeB, dtype='f4')
        W.paint(real, posB, mass=2.5)
        o = numpy.zeros(len(posB), dtype='f4')
        W.readout(fieldB, posB, out=o)
        out['B/%s/paint' % name] = real
        out['B/%s/readout' % name] = o

    # ---- case C: 2-d and 1-d
    posC2 = rs.uniform(-3, 12, size=(200, 2))
    posC1 = rs.uniform(-3, 12, size=(100, 1))
    massC = rs.uniform(0, 2, size=200)
    f2 = rs.normal(size=(9, 7))
    f1 = rs.normal(size=(11,))
    out['C/pos2'] = posC2
    out['C/pos1'] = posC1
    out['C/mass'] = massC
    out['C/field2'] = f2
    out['C/field1'] = f1
    for name in TUNED + GENERIC:
        W = window.windows[name]
        for diffdir in (None, 0, 1):
            aff = Affine(2, scale=[1.0, 0.5], translate=[0.5, -1], period=[9, 7])
            real = numpy.zeros((9, 7))
            W.paint(real, posC2, mass=massC, diffdir=diffdir, transform=aff)
            key = 'C/%s/2/%s' % (name, 'n' if diffdir is None else diffdir)
            out[key + '/paint'] = real
            out[key + '/readout'] = W.readout(f2, posC2, diffdir=diffdir, transform=aff)
        for diffdir in (None, 0):
            aff = Affine(1, scale=[0.9], translate=[0.1], period=[11])
            real = numpy.zeros((11,))
            W.paint(real, posC1, mass=massC[:100], diffdir=diffdir, transform=aff)
            key = 'C/%s/1/%s' % (name, 'n' if diffdir is None else diffdir)
            out[key + '/paint'] = real
            out[key + '/readout'] = W.readout(f1, posC1, diffdir=diffdir, transform=aff)

    # ---- case D: hsml (per particle) and resized windows: generic path, quirk Q6
    posD = rs.uniform(0, 12, size=(150, 3))
    hsmlD = rs.uniform(0.4, 2.2, size=150)
    fieldD = rs.normal(size=(12, 12, 12))
    out['D/pos'] = posD
    out['D/hsml'] = hsmlD
    out['D/field'] = fieldD
    for name in TUNED + GENERIC:
        W = window.windows[name]
        aff = Affine(3, period=12)
        real = numpy.zeros((12, 12, 12))
        W.paint(real, posD, hsml=hsmlD, transform=aff)
        out['D/%s/paint' % name] = real
        out['D/%s/readout' % name] = W.readout(fieldD, posD, hsml=hsmlD, transform=aff)
        out['D/%s/readout_g1' % name] = W.readout(fieldD, posD, hsml=hsmlD, transform=aff, diffdir=1)
        W6 = W.resize(6)
        real = numpy.zeros((12, 12, 12))
        W6.paint(real, posD, transform=aff)
        out['D/%s/resize6/paint' % name] = real
        out['D/%s/resize6/readout' % name] = W6.readout(fieldD, posD, transform=aff)
        out['D/%s/resize6/support' % name] = numpy.array([W6.support, W6.nativesupport])

    # ---- case E: dyadic positions / integer masses: every partial sum is exact, so the
    # result does not depend on summation order (bit-exact target for GPU atomics)
    shapeE = (8, 8, 8)
    posE = rs.randint(-64, 3 * 128, size=(2000, 3)) / 16.0
    massE = rs.randint(1, 5, size=2000).astype('f8')
    fieldE = rs.randint(-8, 9, size=shapeE).astype('f8')
    out['E/pos'] = posE
    out['E/mass'] = massE
    out['E/field'] = fieldE
    for name in TUNED:
        W = window.windows[name]
        for dt in ('f8', 'f4'):
            aff = Affine(3, period=8)
            real = numpy.zeros(shapeE, dtype=dt)
            W.paint(real, posE, mass=massE, transform=aff)
            out['E/%s/%s/paint' % (name, dt)] = real
            out['E/%s/%s/readout' % (name, dt)] = W.readout(fieldE.astype(dt), posE, transform=aff)

    # ---- case F: cell-boundary positions (x = k L/N, +-1 ulp, negative, >= L, -0.0)
    N, L = 16, 1000.0
    k = numpy.arange(-3, N + 4, dtype='f8')
    base = k * L / N
    edge = numpy.concatenate([base, numpy.nextafter(base, numpy.inf), numpy.nextafter(base, -numpy.inf),
                              [-0.0, 0.0, L, -L, 2 * L, 0.5 * L / N, (N - 0.5) * L / N]])
    posF = numpy.stack([edge, numpy.roll(edge, 7), numpy.roll(edge, 13)], axis=-1)
    out['F/pos'] = posF
    out['F/N'] = numpy.array([N])
    out['F/L'] = numpy.array([L])
    fieldF = rs.normal(size=(N, N, N))
    out['F/field'] = fieldF
    for name in TUNED:
        W = window.windows[name]
        aff = Affine(3, scale=1.0 * N / L, period=N)
        real = numpy.zeros((N, N, N))
        W.paint(real, posF, transform=aff)
        out['F/%s/paint' % name] = real
        out['F/%s/readout' % name] = W.readout(fieldF, posF, transform=aff)

    # ---- case G: strided canvas (test_window.py:145-153 generalised) and complex canvas
    posG = rs.uniform(0, 6, size=(80, 2))
    out['G/pos'] = posG
    for name in TUNED:
        W = window.windows[name]
        big = numpy.zeros((18, 12))
        view = big[::3, ::2]
        W.paint(view, posG, transform=Affine(2, period=[6, 6]))
        out['G/%s/big' % name] = big
        cplx = numpy.zeros((6, 6), dtype='c16')
        W.paint(cplx, posG, transform=Affine(2, period=[6, 6]))
        out['G/%s/complex' % name] = cplx

    # ---- fwindow
    w = numpy.linspace(-numpy.pi, numpy.pi, 33)
    out['W/w'] = w
    for name in TUNED + GENERIC:
        W = window.windows[name]
        out['W/%s/fwindow' % name] = W.get_fwindow(w)
        out['W/%s/resize6/fwindow' % name] = W.resize(6).get_fwindow(w)
        out['W/%s/support' % name] = numpy.array([W.support, W.nativesupport])
    return out


def decompose_fixtures(domain):
    out = {}
    rs = numpy.random.RandomState(777)
    N = 16
    cases = {
        'slab1': ([numpy.linspace(0, N, 2), [0, N], [0, N]], 1),
        'slab2': ([numpy.linspace(0, N, 3), [0, N], [0, N]], 2),
        'slab4': ([numpy.linspace(0, N, 5), [0, N], [0, N]], 4),
        'slab8': ([numpy.linspace(0, N, 9), [0, N], [0, N]], 8),
        'pencil2x4': ([numpy.linspace(0, N, 3), numpy.linspace(0, N, 5), [0, N]], 8),
        'uneven3': ([numpy.array([0., 6., 11., 16.]), [0, N], [0, N]], 3),
        'degenerate': ([numpy.array([0., 8., 8., 16.]), [0, N], [0, N]], 3),
        'grid2x2x2': ([numpy.linspace(0, N, 3)] * 3, 8),
    }
    pos = rs.uniform(-N, 2 * N, size=(3000, 3))
    # add exact-edge and near-edge positions
    e = numpy.array([0., 2., 4., 6., 8., 11., 12., 16., -0.0, 15.999999999999998, 7.999999999999999])
    pe = numpy.stack([numpy.resize(e, 66), numpy.resize(numpy.roll(e, 3), 66),
                      numpy.resize(numpy.roll(e, 5), 66)], axis=-1)
    pos = numpy.concatenate([pos, pe], axis=0)
    out['pos'] = pos
    out['pos_f4'] = pos.astype('f4')
    for cname, (edges, P) in cases.items():
        edges = [numpy.asarray(g, dtype='f8') for g in edges]
        for d, g in enumerate(edges):
            out['%s/edges%d' % (cname, d)] = g
        out['%s/nranks' % cname] = numpy.array([P])
        for periodic in (True, False):
            comm = VirtualComm(P)
            grid = domain.GridND(edges, comm=comm, periodic=periodic)
            out['%s/assign' % cname] = grid.DomainAssign
            out['%s/degenerate' % cname] = grid.DomainDegenerate
            for sm in (0.0, 1.0, 1.5, 2.0, [0.5, 1.0, 3.0]):
                for scale in (1.0, 0.5):
                    for ptag, pp in (('f8', pos), ('f4', out['pos_f4'])):
                        if ptag == 'f4' and (scale != 1.0 or sm != 1.0):
                            continue
                        layout = grid.decompose(pp, smoothing=sm,
                                                transform=lambda x, s=scale: s * x)
                        tag = '%s/%s/sm%s/sc%s/%s' % (cname, 'per' if periodic else 'nonper',
                                                      str(sm).replace(' ', ''), scale, ptag)
                        out[tag + '/counts'] = numpy.asarray(layout.sendcounts)
                        out[tag + '/indices'] = numpy.asarray(layout.indices)
    # a custom DomainAssign (pm.py:1447-1461 passes one) with fewer ranks than domains
    edges = [numpy.linspace(0, N, 5), numpy.linspace(0, N, 3), [0, N]]
    assign = numpy.array([3, 2, 1, 1, 0, 3, 2, 3], dtype='int32')
    grid = domain.GridND(edges, comm=VirtualComm(4), periodic=True, DomainAssign=assign)
    for d, g in enumerate(edges):
        out['assigned/edges%d' % d] = numpy.asarray(g, dtype='f8')
    out['assigned/nranks'] = numpy.array([4])
    out['assigned/assign'] = grid.DomainAssign
    out['assigned/degenerate'] = grid.DomainDegenerate
    layout = grid.decompose(pos, smoothing=1.5)
    out['assigned/per/sm1.5/sc1.0/f8/counts'] = numpy.asarray(layout.sendcounts)
    out['assigned/per/sm1.5/sc1.0/f8/indices'] = numpy.asarray(layout.indices)
    return out


def cycle_fixture(window):
    """paint -> r2c -> transfer -> c2r -> readout at 16^3 with the reference
    kernels and numpy.fft under the reference's normalisation (pm.py:692:
    r2c = rfftn / N^3; c2r = irfftn * N^3).  The transfer functions restate
    examples/nbody.py:154-175 (dx1_transfer, force_transfer, pot_transfer)."""
    out = {}
    N, L = 16, 1000.0
    rs = numpy.random.RandomState(99)
    q = (numpy.indices((N, N, N)).reshape(3, -1).T + 0.5) * (L / N)
    pos = q + rs.uniform(-0.4, 0.4, size=q.shape) * (L / N)
    out['pos'] = pos
    out['N'] = numpy.array([N])
    out['L'] = numpy.array([L])
    ki = [2 * numpy.pi / L * numpy.where(numpy.arange(N) >= N // 2, numpy.arange(N) - N,
                                         numpy.arange(N)).astype('f8')]
    k0 = ki[0].reshape(-1, 1, 1)
    k1 = ki[0].reshape(1, -1, 1)
    k2_ = ki[0][:N // 2 + 1].copy().reshape(1, 1, -1)
    k = [k0, k1, k2_]

    def dx1(direction, v):
        k2 = sum(kk ** 2 for kk in k)
        k2[k2 == 0] = 1.0
        return 1j * k[direction] / k2 * v

    def force(direction, v):
        k2 = sum(kk ** 2 for kk in k)
        k2[k2 == 0] = 1.0
        C = L / N
        w = k[direction] * C
        kf = 1.0 / C * 1 / 6.0 * (8 * numpy.sin(w) - numpy.sin(2 * w))
        return 1j * kf / k2 * v

    def pot(v):
        k2 = sum(kk ** 2 for kk in k)
        k2[k2 == 0] = 1.0
        return -1. / k2 * v

    Affine = window.Affine
    for name in TUNED:
        W = window.windows[name]
        aff = Affine(3, scale=1.0 * N / L, period=N)
        real = numpy.zeros((N, N, N))
        W.paint(real, pos, transform=aff)
        ck = numpy.fft.rfftn(real) / N ** 3
        out['%s/paint' % name] = real
        out['%s/r2c' % name] = ck
        for tname, tf in (('dx1_0', lambda v: dx1(0, v)), ('force_2', lambda v: force(2, v)),
                          ('pot', pot)):
            tk = tf(ck)
            back = numpy.fft.irfftn(tk, s=(N, N, N), axes=(0, 1, 2)) * N ** 3
            out['%s/%s/c2r' % (name, tname)] = back
            out['%s/%s/readout' % (name, tname)] = W.readout(back, pos, transform=aff)
            out['%s/%s/readout_g0' % (name, tname)] = W.readout(back, pos, transform=aff, diffdir=0)
    return out


def main():
    scratch = tempfile.mkdtemp(prefix='pmesh_ref_build_')
    try:
        build_reference(scratch)
        install_mpi_placeholder()
        sys.path.insert(0, scratch)
        from pmesh import window, domain
        numpy.savez_compressed(os.path.join(HERE, 'window.npz'), **window_fixtures(window))
        numpy.savez_compressed(os.path.join(HERE, 'decompose.npz'), **decompose_fixtures(domain))
        numpy.savez_compressed(os.path.join(HERE, 'cycle16.npz'), **cycle_fixture(window))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    for fn in ('window.npz', 'decompose.npz', 'cycle16.npz'):
        print(fn, os.path.getsize(os.path.join(HERE, fn)), 'bytes')


if __name__ == '__main__':
    main()
