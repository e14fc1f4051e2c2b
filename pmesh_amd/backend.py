"""Kernel backend: the one place where the host layer meets native code.

The product backend is :class:`HipBackend`: it binds ``libpmesh_amd.so`` (the
C ABI of include/pmesh_amd.h, hand-written HIP kernels for gfx950 + rocFFT)
through the Cython shim ``pmesh_amd._pmx`` and runs on a real GPU.  There is NO CPU implementation in this package: if the
library or the GPU is missing, :func:`get` raises — it never falls back.

``use(backend)`` exists so that tests can drive the host logic (argument
handling, layouts, the distributed FFT schedule, torch.distributed collectives
over gloo) against the CPU oracle on machines without a GPU; the object they
install lives in tests/, not here.
"""
import ctypes as C
import os

import torch

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBNAME = 'libpmesh_amd.so'


class PmxError(RuntimeError):
    def __init__(self, what, code, msg=''):
        self.code = code
        RuntimeError.__init__(self, '%s failed: %s%s' % (
            what, _abi.STATUS_NAMES.get(code, code), (' — ' + msg) if msg else ''))


def library_path():
    # PMESH_AMD_LIBRARY: another build of the same ABI (A/B timing of compile-time switches)
    return os.environ.get('PMESH_AMD_LIBRARY') or os.path.join(_HERE, LIBNAME)


def load_library(path=None, binding=None):
    """Load the C ABI and bind every entry point of include/pmesh_amd.h; raises if the library is missing or lacks a
    symbol the header declares.  The binding is the Cython shim `pmesh_amd._pmx` (generated from the header by
    csrc/gen_pyx.py, built by the same `make`): the returned object has one callable `pmx_<name>` per entry point.
    PMESH_AMD_BINDING=ctypes binds the same library through the ctypes table of _abi.py instead (the binding the
    tests' CPU double uses; ~10 us slower per call) — a debugging aid, never a fallback: a missing shim raises."""
    path = path or library_path()
    if not os.path.exists(path):
        raise ImportError(
            '%s not found: build it with `make -C pmesh_amd/csrc` (hipcc, gfx950). '
            'There is no CPU fallback.' % path)
    binding = binding or os.environ.get('PMESH_AMD_BINDING', 'cython')
    if binding == 'ctypes':
        lib = C.CDLL(path)
        missing = _abi.declare(lib, 'pmx_', _abi.PROTOTYPES)
        missing += _abi.declare(lib, 'pmx_', _abi.DEVICE_ONLY)
    else:
        try:
            from . import _pmx as lib
        except ImportError as e:
            raise ImportError('the Cython shim pmesh_amd/_pmx is not built (`make -C pmesh_amd/csrc`): %s' % e)
        missing = lib.bind(path)
    if missing:
        raise ImportError('%s lacks symbols declared in include/pmesh_amd.h: %s' % (path, missing))
    return lib


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
if _raw_stream is None:
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream


class HipBackend(object):
    """libpmesh_amd.so on cuda:<index> (ROCm)."""
    name = 'hip'
    prefix = 'pmx_'

    def __init__(self, device_index=None):
        self.lib = load_library()
        if self.lib.pmx_device_count() < 1 or not torch.cuda.is_available():
            raise RuntimeError('pmesh_amd needs an AMD GPU (gfx950): no HIP device is visible. '
                               'There is no CPU fallback.')
        if device_index is None:
            device_index = torch.cuda.current_device()
        self.device = torch.device('cuda', device_index)

    # -- plumbing ---------------------------------------------------------
    def stream(self):
        # (the raw handle of torch's current stream: torch.cuda.current_stream() builds a Stream object per call,
        # 6 us of the host's ~15 us per kernel launch)
        return C.c_void_p(_raw_stream(self.device.index))

    def call(self, name, *args):
        rc = getattr(self.lib, 'pmx_' + name)(*args)
        if rc != 0:
            raise PmxError('pmx_' + name, rc, self.lib.pmx_last_error().decode())

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    # -- FFT plans --------------------------------------------------------
    def fft_create(self, kind, elsize, n, istride, idist, ostride, odist, batch, scale, inplace):
        plan = C.c_void_p()
        nd = len(n)
        self.call('fft_create', C.byref(plan), kind, elsize, nd, _abi.i64arr(n), _abi.i64arr(istride),
                  idist, _abi.i64arr(ostride), odist, batch, float(scale), int(bool(inplace)))
        return plan

    def fft_execute(self, plan, tin, tout):
        self.call('fft_execute', plan, tin.data_ptr(), tout.data_ptr(), self.stream())

    def fft_destroy(self, plan):
        self.call('fft_destroy', plan)

    # -- LDS-resident column FFT -----------------------------------------
    def colfft_supported(self, n, elsize):
        return self.lib.pmx_colfft_supported(int(n), int(elsize)) == 0

    def colfft(self, elsize, inverse, data, A, N, B, scale=1.0, transfer=None, n1=1, n2=1,
               start=(0, 0, 0), nmesh=(1, 1, 1), boxsize=(1.0, 1.0, 1.0), a_stride=0, n_stride=0):
        """in-place FFT along the middle axis of the (A, N, B) complex array in `data`
        (a_stride / n_stride: padded strides between successive a / lines n, 0 = dense)"""
        self.call('colfft', elsize, int(bool(inverse)), data.data_ptr(), A, N, B, float(scale),
                  C.byref(transfer) if transfer is not None else None, n1, n2,
                  _abi.i64arr(start, 3), _abi.i64arr(nmesh, 3), _abi.f64arr(boxsize, 3),
                  int(a_stride), int(n_stride), self.stream())

    def colfft_configure(self, persistent):
        """process-wide: persistent (prefetching) column passes where a tile fills a CU, or one workgroup per tile
        (for transforms that overlap with collectives) — pmx_colfft_configure"""
        self.call('colfft_configure', int(bool(persistent)))

    def colfft_roundtrip_supported(self, n, elsize):
        return self.lib.pmx_colfft_roundtrip_supported(int(n), int(elsize)) == 0

    def colfft_roundtrip(self, elsize, data, N, B, scale=1.0, transfer=None, n1=1, n2=1,
                         start=(0, 0, 0), nmesh=(1, 1, 1), boxsize=(1.0, 1.0, 1.0), n_stride=0):
        """forward column FFT x scale [x transfer] x inverse column FFT in one kernel, in place on (N, B)"""
        self.call('colfft_roundtrip', elsize, data.data_ptr(), N, B, float(scale),
                  C.byref(transfer) if transfer is not None else None, n1, n2,
                  _abi.i64arr(start, 3), _abi.i64arr(nmesh, 3), _abi.f64arr(boxsize, 3), int(n_stride), self.stream())

    def colfft_split(self, elsize, inverse, src, dst, A, N, B, nsplit, scale=1.0, plain_pitch=0):
        """column FFT fused with the slab pack (forward: plain -> split) / unpack (inverse);
        plain_pitch: elements per line of the plain side (0 = B)"""
        self.call('colfft_split', elsize, int(bool(inverse)), src.data_ptr(), dst.data_ptr(), A, N, B,
                  int(nsplit), float(scale), int(plain_pitch), self.stream())

    def colfft_resplit(self, elsize, inverse, src, dst, A, N, B, nsplit_in, nsplit_out, scale=1.0):
        """column FFT between two split layouts (the axis-1 pass of a pencil transform with the
        unpack before and the pack after it fused in); nsplit 0 = plain"""
        self.call('colfft_resplit', elsize, int(bool(inverse)), src.data_ptr(), dst.data_ptr(), A, N, B,
                  int(nsplit_in), int(nsplit_out), float(scale), self.stream())

    def colfft_chunk(self, elsize, inverse, chunk, full, N, n1, cw, pitch, coff, to_full, scale=1.0,
                     transfer=None, start=(0, 0, 0), nmesh=(1, 1, 1), boxsize=(1.0, 1.0, 1.0)):
        """axis-0 pass on the columns [coff, coff+cw) of the (N, n1, pitch) block `full`, through the
        dense (N, n1, cw) buffer `chunk` (pipelined slab transposes)"""
        self.call('colfft_chunk', elsize, int(bool(inverse)), chunk.data_ptr(), full.data_ptr(), N, n1, cw,
                  pitch, coff, int(bool(to_full)), float(scale),
                  C.byref(transfer) if transfer is not None else None,
                  _abi.i64arr(start, 3), _abi.i64arr(nmesh, 3), _abi.f64arr(boxsize, 3), self.stream())

    def rowfft_supported(self, n, elsize):
        return self.lib.pmx_rowfft_supported(int(n), int(elsize)) == 0

    def rowfft(self, elsize, inverse, data, nrows, n, pitch, scale=1.0, rows_per_plane=0, plane_pitch=0):
        """in-place r2c / c2r of `nrows` rows of n reals at a pitch of `pitch` complex elements
        (rows_per_plane > 0: planes of that many rows, `plane_pitch` complex elements apart)"""
        self.call('rowfft', elsize, int(bool(inverse)), data.data_ptr(), nrows, n, pitch, float(scale),
                  int(rows_per_plane), int(plane_pitch), self.stream())

    def rowfft_halo(self, elsize, data, nrows, n, pitch, rows_per_plane, plane_pitch, plan, canvas_ptr, x0, last,
                    scale=1.0, dst=None):
        """the forward rowfft on planes x0 ... of a canvas whose paint left its halo merge to this pass
        (pmx_paint_binned_defer): the staged halos are added to the rows as they are loaded; last: the plan is
        released; dst: where the rows are written (None: in place)"""
        self.call('rowfft_halo', elsize, data.data_ptr(), dst.data_ptr() if dst is not None else None, nrows, n, pitch,
                  float(scale), int(rows_per_plane), int(plane_pitch), plan, C.c_void_p(canvas_ptr), int(x0),
                  int(bool(last)), self.stream())

    def rowfft_to(self, elsize, inverse, src, dst, nrows, n, pitch, scale=1.0, rows_per_plane=0, plane_pitch=0):
        """rowfft from `src` into `dst` (same layout)"""
        self.call('rowfft_to', elsize, int(bool(inverse)), src.data_ptr(), dst.data_ptr(), nrows, n, pitch, float(scale),
                  int(rows_per_plane), int(plane_pitch), self.stream())

    def colfft_to(self, elsize, inverse, src, dst, A, N, B, scale=1.0, transfer=None, n1=1, n2=1,
                  start=(0, 0, 0), nmesh=(1, 1, 1), boxsize=(1.0, 1.0, 1.0), a_stride=0, n_stride=0):
        """colfft from `src` into `dst` (same layout)"""
        self.call('colfft_to', elsize, int(bool(inverse)), src.data_ptr(), dst.data_ptr(), A, N, B, float(scale),
                  C.byref(transfer) if transfer is not None else None, n1, n2,
                  _abi.i64arr(start, 3), _abi.i64arr(nmesh, 3), _abi.f64arr(boxsize, 3),
                  int(a_stride), int(n_stride), self.stream())

    # -- slab transposes --------------------------------------------------
    def rowfft_split_supported(self, n, elsize, nparts):
        return self.lib.pmx_rowfft_split_supported(int(n), int(elsize), int(nparts)) == 0

    def rowfft_split(self, elsize, inverse, src, dst, nrows, n, pitch, offsets, scale=1.0):
        """the row pass with the last-axis split of a pencil transform's first transpose on it (forward: rows ->
        blocks by mode range; inverse: blocks -> rows); out of place"""
        self.call('rowfft_split', elsize, int(bool(inverse)), src.data_ptr(), dst.data_ptr(), nrows, n, pitch, float(scale),
                  _abi.i64arr(offsets), len(offsets) - 1, self.stream())

    def slab_pack(self, src, dst, n0, n1, n2, n1_offsets, elbytes, inverse=False):
        """(n0, n1, n2) -> blocks by n1 range (inverse: blocks -> (n0, n1, n2))"""
        self.call('slab_unpack' if inverse else 'slab_pack', src.data_ptr(), dst.data_ptr(), n0, n1, n2,
                  _abi.i64arr(n1_offsets), len(n1_offsets) - 1, elbytes, self.stream())


_current = None


def use(backend):
    """Install a backend object (tests only; see module docstring)."""
    global _current
    _current = backend
    return backend


def get():
    """The active backend; creates the HIP backend on first use and raises if
    that is impossible."""
    global _current
    if _current is None:
        _current = HipBackend()
    return _current


def reset():
    global _current
    _current = None
