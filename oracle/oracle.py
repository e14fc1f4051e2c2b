"""CPU oracle bindings — TEST INFRASTRUCTURE ONLY.

numpy-level access to oracle/liboracle.so (the C restatement, prefix ``pmo_``)
and, when present, oracle/_ref/libpmesh_ref.so (the reference's own
``_window_imp.c`` compiled by oracle/Makefile, prefix ``ref_``).  Only tests/,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py import
this module; nothing under pmesh_amd/ does.

The Python-level argument handling restates pmesh/window.py:106-221 (order[]
from diffdir, scalar mass broadcast, default f8 output) and
pmesh/domain.py:561-652 (Layout construction), so that a test can write the
reference's call and get the reference's answer.
"""
import ctypes as C
import os
import subprocess

import numpy

from pmesh_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False):
    """(Re)build liboracle.so and, if /root/reference exists, _ref."""
    out = subprocess.run(['make', '-C', _HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError('oracle build failed:\n' + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


_libs = {}


def lib(which='oracle'):
    if which not in _libs:
        if which == 'oracle':
            path = os.path.join(_HERE, 'liboracle.so')
            prefix = 'pmo_'
            table = _abi.PROTOTYPES
        elif which == 'ref':
            path = os.path.join(_HERE, '_ref', 'libpmesh_ref.so')
            prefix = 'ref_'
            table = {k: _abi.PROTOTYPES[k] for k in ('paint', 'readout', 'paint_nd', 'readout_nd', 'window_info', 'fwindow')}
        else:
            raise ValueError(which)
        if not os.path.exists(path):
            if which == 'oracle':
                build()
            else:
                raise FileNotFoundError(path)
        L = C.CDLL(path)
        missing = _abi.declare(L, prefix, table)
        if missing:
            raise RuntimeError('oracle library lacks symbols: %s' % missing)
        _libs[which] = (L, prefix)
    return _libs[which]


def have_ref():
    return os.path.exists(os.path.join(_HERE, '_ref', 'libpmesh_ref.so'))


def _fn(which, name):
    L, prefix = lib(which)
    return getattr(L, prefix + name)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed: %s' % (what, _abi.STATUS_NAMES.get(rc, rc)))


def vec_of(a, ncol=None):
    """pmx_vec over a numpy array (1-d or 2-d, f4/f8, any strides)."""
    if a is None:
        return None
    assert a.dtype in (numpy.float32, numpy.float64), a.dtype
    v = _abi.Vec()
    v.data = a.ctypes.data
    v.elsize = a.dtype.itemsize
    if a.ndim == 1:
        v.ncol = 1
        v.stride0 = a.strides[0]
        v.stride1 = 0
    else:
        v.ncol = a.shape[1]
        v.stride0 = a.strides[0]
        v.stride1 = a.strides[1]
    return v


def make_painter(kind, support, real, order, scale, translate, period):
    # (meshes of more than three dimensions: pmx_painter_nd and the *_nd entry points)
    p = _abi.Painter() if real.ndim <= _abi.PMX_MAXDIM else _abi.PainterND()
    p.kind = _abi.KINDS[kind] if isinstance(kind, str) else int(kind)
    p.support = int(support)
    p.ndim = real.ndim
    p.canvas_elsize = real.dtype.itemsize
    for d in range(real.ndim):
        p.order[d] = int(order[d])
        p.scale[d] = float(scale[d])
        p.translate[d] = float(translate[d])
        p.period[d] = int(period[d])
        p.size[d] = real.shape[d]
        p.strides[d] = real.strides[d]
    return p


class Affine(object):
    """window.py:18-55"""
    def __init__(self, ndim, scale=None, translate=None, period=None):
        def mk(v, default, dtype):
            if v is None:
                v = default
            r = numpy.empty(ndim, dtype)
            r[...] = v
            return r
        self.scale = mk(scale, 1.0, 'f8')
        self.translate = mk(translate, 0, 'f8')
        self.period = mk(period, 0, 'intp')
        self.ndim = ndim


class Window(object):
    """The reference's ResampleWindow.paint/readout contract on numpy arrays
    (window.py:57-221), computed by the oracle (`which='oracle'`) or by the
    compiled reference kernels (`which='ref'`)."""

    def __init__(self, kind, support=-1, which='oracle'):
        self.kind = kind
        self.which = which
        ns, es = C.c_int32(), C.c_int32()
        k = _abi.KINDS[kind] if isinstance(kind, str) else int(kind)
        _check(_fn(which, 'window_info')(k, support, C.byref(ns), C.byref(es)), 'window_info')
        self.nativesupport = ns.value
        self.support = es.value
        self._k = k

    def resize(self, support):
        return Window(self.kind, support, self.which)

    def get_fwindow(self, w):
        w1 = numpy.ascontiguousarray(numpy.reshape(w, -1), dtype='f8')
        out = numpy.zeros_like(w1)
        _check(_fn(self.which, 'fwindow')(
            self._k, self.support, w1.ctypes.data_as(C.POINTER(C.c_double)), len(w1),
            out.ctypes.data_as(C.POINTER(C.c_double))), 'fwindow')
        return out.reshape(numpy.shape(w))

    def paint(self, real, pos, hsml=None, mass=None, diffdir=None, transform=None):
        if transform is None:
            transform = Affine(real.ndim)
        order = numpy.zeros(real.ndim, dtype=int)
        if diffdir is not None:
            order[diffdir] = 1
        pos = numpy.asarray(pos)
        if pos.dtype.kind != 'f':
            pos = pos.astype('f8')
        if pos.ndim == 1:
            pos = pos.reshape(-1, 1)
        mass_scalar = 1.0
        massv = None
        if mass is not None:
            mass = numpy.asarray(mass)
            if mass.ndim == 0:
                mass_scalar = float(mass)
            else:
                if mass.dtype.kind != 'f':
                    raise TypeError('mass must be floating point')
                massv = vec_of(mass)
        hs = None
        if hsml is not None:
            hsml = numpy.asarray(hsml, dtype=None)
            if hsml.ndim == 0:
                hsml = numpy.full(len(pos), float(hsml))
            hs = vec_of(hsml)
        if numpy.iscomplexobj(real):
            real = real.real
        assert real.dtype.kind == 'f'
        p = make_painter(self._k, self.support, real, order, transform.scale, transform.translate,
                         transform.period)
        pv = vec_of(pos)
        _check(_fn(self.which, 'paint' if real.ndim <= _abi.PMX_MAXDIM else 'paint_nd')(
            C.byref(p), real.ctypes.data, C.byref(pv),
            C.byref(massv) if massv is not None else None, mass_scalar,
            C.byref(hs) if hs is not None else None, len(pos), None), 'paint')

    def readout(self, real, pos, hsml=None, out=None, diffdir=None, transform=None):
        if transform is None:
            transform = Affine(real.ndim)
        order = numpy.zeros(real.ndim, dtype=int)
        if diffdir is not None:
            order[diffdir] = 1
        pos = numpy.asarray(pos)
        if pos.dtype.kind != 'f':
            pos = pos.astype('f8')
        if out is None:
            out = numpy.zeros(pos.shape[:-1], dtype='f8')
        hs = None
        if hsml is not None:
            hsml = numpy.asarray(hsml)
            if hsml.ndim == 0:
                hsml = numpy.full(len(pos), float(hsml))
            hs = vec_of(hsml)
        if numpy.iscomplexobj(real):
            real = real.real
        p = make_painter(self._k, self.support, real, order, transform.scale, transform.translate,
                         transform.period)
        pv = vec_of(pos)
        ov = vec_of(out)
        _check(_fn(self.which, 'readout' if real.ndim <= _abi.PMX_MAXDIM else 'readout_nd')(
            C.byref(p), real.ctypes.data, C.byref(pv),
            C.byref(hs) if hs is not None else None, C.byref(ov), len(pos), None), 'readout')
        return out


# ------------------------------------------------------------------ decompose

class GridSpec(object):
    """The arrays of a GridND (domain.py:370-407) laid out for pmx_grid."""

    def __init__(self, edges, nranks, periodic=True, DomainAssign=None):
        self.edges = [numpy.ascontiguousarray(e, dtype='f8') for e in edges]
        self.shape = numpy.array([len(e) - 1 for e in self.edges], dtype='int32')
        self.ndim = len(self.shape)
        self.size = int(numpy.prod(self.shape))
        self.nranks = nranks
        self.periodic = periodic
        if DomainAssign is None:
            if nranks >= self.size:
                DomainAssign = numpy.arange(self.size, dtype='int32')
            else:
                DomainAssign = numpy.empty(self.size, dtype='int32')
                for i in range(nranks):
                    DomainAssign[i * self.size // nranks:(i + 1) * self.size // nranks] = i
        self.DomainAssign = numpy.ascontiguousarray(DomainAssign, dtype='int32')
        dd = numpy.zeros(self.shape, dtype='int16')
        for i, edge in enumerate(self.edges):
            dd1 = edge[1:] == edge[:-1]
            dd1 = dd1.reshape([-1 if ii == i else 1 for ii in range(self.ndim)])
            dd[...] |= dd1
        self.DomainDegenerate = numpy.ascontiguousarray(dd.ravel())

    def cgrid(self):
        g = _abi.Grid()
        g.ndim = self.ndim
        g.periodic = int(bool(self.periodic))
        g.nranks = self.nranks
        for d in range(self.ndim):
            g.shape[d] = int(self.shape[d])
            g.edges[d] = self.edges[d].ctypes.data
        g.assign = self.DomainAssign.ctypes.data
        g.degenerate = self.DomainDegenerate.ctypes.data
        return g


def decompose(grid, pos, smoothing, scale=None, index_dtype='int32'):
    """GridND.decompose (domain.py:561-652) -> (counts int32[P], indices int32[sum])."""
    pos = numpy.asarray(pos)
    if pos.dtype.kind != 'f':
        pos = pos.astype('f8')
    n = len(pos)
    sm = numpy.empty(grid.ndim, 'f8')
    sm[:] = smoothing
    sc = numpy.ones(grid.ndim, 'f8')
    if scale is not None:
        sc[:] = scale
    masks = numpy.zeros(max(n, 1), dtype='u8')
    counts = numpy.zeros(grid.nranks, dtype='i8')
    g = grid.cgrid()
    pv = vec_of(pos) if n else _abi.Vec()
    _check(_fn('oracle', 'decompose_count')(
        C.byref(g), C.byref(pv), sc.ctypes.data_as(C.POINTER(C.c_double)),
        sm.ctypes.data_as(C.POINTER(C.c_double)), n, masks.ctypes.data, counts.ctypes.data, None),
        'decompose_count')
    offsets = numpy.zeros(grid.nranks, dtype='i8')
    offsets[1:] = numpy.cumsum(counts)[:-1]
    indices = numpy.zeros(int(counts.sum()), dtype=index_dtype)
    _check(_fn('oracle', 'decompose_fill')(
        grid.nranks, masks.ctypes.data, n, offsets.ctypes.data, indices.ctypes.data,
        indices.dtype.itemsize, None), 'decompose_fill')
    return counts.astype('int32'), indices


def take_rows(data, indices):
    data = numpy.ascontiguousarray(data)
    indices = numpy.ascontiguousarray(indices)
    out = numpy.empty((len(indices),) + data.shape[1:], dtype=data.dtype)
    row = data.dtype.itemsize * int(numpy.prod(data.shape[1:], dtype='i8'))
    _check(_fn('oracle', 'take_rows')(data.ctypes.data, row, row, indices.ctypes.data,
                                      indices.dtype.itemsize, len(indices), out.ctypes.data, None),
           'take_rows')
    return out


def scatter_add(values, indices, nout):
    values = numpy.ascontiguousarray(values)
    indices = numpy.ascontiguousarray(indices)
    ncol = int(numpy.prod(values.shape[1:], dtype='i8'))
    out = numpy.zeros((nout,) + values.shape[1:], dtype=values.dtype)
    _check(_fn('oracle', 'scatter_add')(values.ctypes.data, values.dtype.itemsize, ncol,
                                        indices.ctypes.data, indices.dtype.itemsize, len(indices),
                                        out.ctypes.data, nout, None), 'scatter_add')
    return out


# ------------------------------------------------------------------ transfer

def make_transfer(amplitude=1.0, laplace_pow=0, grad_dir=-1, grad_kind=0, deconv_pow=0, gauss_r=0.0):
    t = _abi.Transfer()
    t.amplitude = amplitude
    t.laplace_pow = laplace_pow
    t.grad_dir = grad_dir
    t.grad_kind = grad_kind
    t.deconv_pow = deconv_pow
    t.gauss_r = gauss_r
    return t


def apply_transfer(t, cin, start, nmesh, boxsize, out=None):
    cin = numpy.asarray(cin)
    assert cin.dtype.kind == 'c'
    if out is None:
        out = numpy.empty_like(cin)
    nd = cin.ndim
    _check(_fn('oracle', 'apply_transfer')(
        C.byref(t), nd, cin.dtype.itemsize // 2, cin.ctypes.data, _abi.i64arr(cin.strides, 3),
        out.ctypes.data, _abi.i64arr(out.strides, 3), _abi.i64arr(cin.shape, 3),
        _abi.i64arr(start, 3), _abi.i64arr(nmesh, 3), _abi.f64arr(boxsize, 3), None),
        'apply_transfer')
    return out


# ------------------------------------------------------------------ synthetic

def synth_uniform(nlat, boxsize, seed=42, g0=0, npart=None, dtype='f8'):
    if npart is None:
        npart = nlat ** 3
    pos = numpy.empty((npart, 3), dtype=dtype)
    pv = vec_of(pos)
    _check(_fn('oracle', 'synth_uniform')(C.byref(pv), nlat, boxsize, seed, g0, npart, None),
           'synth_uniform')
    return pos


def zeldovich_modes(nlat, boxsize, rms_cells=3.0, nmodes=16, seed=1234):
    """The plane-wave table of SURVEY.md 8(d): integer wavevectors, unit
    directions, amplitudes ~ 1/|n| scaled to an rms displacement of
    `rms_cells` cells, phases — from numpy.random.RandomState(seed)."""
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
    # rms of sum_m A_m sin(.) n_m  = sqrt(sum A_m^2 / 2) for independent phases
    rms = numpy.sqrt(0.5 * (amp ** 2).sum())
    amp *= rms_cells * (boxsize / nlat) / rms
    modes = numpy.zeros((nmodes, 8), dtype='f8')
    modes[:, 0:3] = n
    modes[:, 3:6] = direc
    modes[:, 6] = amp
    modes[:, 7] = phase
    return modes


def synth_clustered(nlat, boxsize, modes, shift=0.0, g0=0, npart=None, dtype='f8'):
    if npart is None:
        npart = nlat ** 3
    pos = numpy.empty((npart, 3), dtype=dtype)
    pv = vec_of(pos)
    modes = numpy.ascontiguousarray(modes, dtype='f8')
    _check(_fn('oracle', 'synth_clustered')(
        C.byref(pv), nlat, boxsize, modes.ctypes.data_as(C.POINTER(C.c_double)), len(modes),
        shift, g0, npart, None), 'synth_clustered')
    return pos


# ------------------------------------------------------------------ the cycle

def r2c(real):
    """RealField.r2c contract (pm.py:655-694): rfftn / prod(Nmesh)."""
    return numpy.fft.rfftn(real) / numpy.prod(real.shape, dtype='f8')


def c2r(cplx, shape):
    """ComplexField.c2r contract (pm.py:987-1019): irfftn * prod(Nmesh)."""
    return numpy.fft.irfftn(cplx, s=shape, axes=tuple(range(len(shape)))) * numpy.prod(shape, dtype='f8')


def pm_cycle(nmesh, boxsize, pos, kind='tunedcic', transfer=None, gradient=None, mass=1.0,
             dtype='f8', which='oracle'):
    """paint -> r2c -> apply(transfer) -> c2r -> readout on one block
    (examples/nbody.py:199-218 with one readout), all on the CPU."""
    W = Window(kind, which=which)
    shape = (nmesh,) * 3
    aff = Affine(3, scale=1.0 * nmesh / boxsize, translate=0, period=nmesh)
    real = numpy.zeros(shape, dtype=dtype)
    W.paint(real, pos, mass=mass, transform=aff)
    ck = r2c(real.astype('f8')).astype('c16' if dtype == 'f8' else 'c8')
    if transfer is not None:
        ck = apply_transfer(transfer, ck, (0, 0, 0), shape, (boxsize,) * 3)
    back = c2r(ck.astype('c16'), shape).astype(dtype)
    out = W.readout(back, pos, transform=aff, diffdir=gradient)
    return real, ck, back, out


# ---------------------------------------------------------------- white noise

def whitenoise(shape, start, nmesh, seed, unitary=False, dtype='c16', out=None):
    """pmesh.whitenoise.generate (3-d) by the oracle port: the local block `shape` at `start`
    of the half spectrum of an `nmesh` mesh."""
    value = numpy.zeros(shape, dtype=dtype) if out is None else out
    assert value.ndim == 3 and value.dtype.kind == 'c'
    L, prefix = lib('oracle')
    fn = getattr(L, prefix + 'whitenoise')
    _check(fn(int(seed) & 0xFFFFFFFF, int(bool(unitary)), _abi.i64arr(nmesh, 3), _abi.i64arr(start, 3),
              _abi.i64arr(value.shape, 3), _abi.i64arr(value.strides, 3), value.dtype.itemsize,
              value.ctypes.data_as(C.c_void_p), None), 'whitenoise')
    return value


class _RefGenerator(C.Structure):
    """struct PMeshWhiteNoiseGenerator of pmesh/_whitenoise_imp.h:1-15"""
    _fields_ = [('ndim', C.c_int), ('seed', C.c_uint), ('unitary', C.c_uint),
                ('Nmesh', C.c_ssize_t * 32), ('start', C.c_ssize_t * 32),
                ('canvas', C.c_void_p), ('canvas_dtype_elsize', C.c_int),
                ('size', C.c_ssize_t * 32), ('strides', C.c_ssize_t * 32),
                ('seedtable', (C.c_void_p * 2) * 2)]


def have_whitenoise_ref():
    return os.path.exists(os.path.join(_HERE, '_ref', 'libwhitenoise_ref.so'))


_wn_ref = None


def whitenoise_ref(shape, start, nmesh, seed, unitary=False, dtype='c16'):
    """the same through the reference's own compiled C (oracle/_ref/libwhitenoise_ref.so =
    pmesh/_whitenoise_imp.c + pmesh/gsl), driven as pmesh/_whitenoise.pyx:25-45 does"""
    global _wn_ref
    if _wn_ref is None:
        _wn_ref = C.CDLL(os.path.join(_HERE, '_ref', 'libwhitenoise_ref.so'))
    value = numpy.zeros(shape, dtype=dtype)
    g = _RefGenerator()
    g.canvas = value.ctypes.data
    g.canvas_dtype_elsize = value.dtype.itemsize
    g.ndim = 3
    for d in range(3):
        g.size[d] = value.shape[d]
        g.start[d] = int(start[d])
        g.strides[d] = value.strides[d]
        g.Nmesh[d] = int(nmesh[d])
    g.unitary = int(bool(unitary))
    g.seed = int(seed) & 0xFFFFFFFF
    _wn_ref.pmesh_whitenoise_generator_init(C.byref(g))
    _wn_ref.pmesh_whitenoise_generator_fill(C.byref(g))
    return value
