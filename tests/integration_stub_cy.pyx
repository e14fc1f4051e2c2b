# cython: language_level=3
"""The reference-side binding of INTEGRATION.md section 2 as CYTHON — the language pmesh's own native layer is bound in
(pmesh/_window.pyx): what a maintainer adds next to `_window.pyx` to keep `pmesh/*.py` and swap only the native layer.

A drop-in for `pmesh._window.ResampleWindow` (pmesh/_window.pyx:67-205: a cdef class around the C painter, `paint` /
`readout` / `get_fwindow` with the same argument lists) over `libpmesh_amd.so`: where the reference's class loops over the
particles and calls `pmesh_painter_paint` once per particle (_window.pyx:157-165), this one makes ONE batched call.
Nothing of pmesh_amd's Python side is imported; arrays are any device arrays with the CUDA array interface.
tests/test_integration_stub.py builds this file (cython + the host compiler, linked against the library) and drives it
under `-m gpu` beside the ctypes form of the same binding (tests/integration_stub.py).
"""
from libc.stdint cimport int32_t, int64_t
from libc.string cimport memset

import numpy

cdef extern from "pmesh_amd.h" nogil:
    ctypedef struct pmx_painter:
        int32_t kind
        int32_t support
        int32_t ndim
        int32_t canvas_elsize
        int32_t order[3]
        double scale[3]
        double translate[3]
        int64_t period[3]
        int64_t size[3]
        int64_t strides[3]
    ctypedef struct pmx_vec:
        void *data
        int32_t elsize
        int32_t ncol
        int64_t stride0
        int64_t stride1
    int pmx_window_info(int32_t kind, int32_t support, int32_t *nativesupport, int32_t *eff_support)
    int pmx_fwindow(int32_t kind, int32_t support, const double *w, int64_t n, double *out)
    int pmx_paint(const pmx_painter *p, void *canvas, const pmx_vec *pos, const pmx_vec *mass, double mass_scalar,
                  const pmx_vec *hsml, int64_t n, void *stream)
    int pmx_readout(const pmx_painter *p, const void *canvas, const pmx_vec *pos, const pmx_vec *hsml,
                    const pmx_vec *out, int64_t n, void *stream)
    const char *pmx_last_error()

_KINDS = {'nearest': 0, 'linear': 1, 'quadratic': 2, 'cubic': 3,
          'tunednnb': 4, 'tunedcic': 5, 'tunedtsc': 6, 'tunedpcs': 7}


cdef _interface(a):
    ai = a.__cuda_array_interface__               # (cupy / torch / numba device arrays)
    itemsize = numpy.dtype(ai['typestr']).itemsize
    shape = tuple(ai['shape'])
    strides = ai.get('strides')
    if strides is None:                           # C contiguous
        strides, acc = [], itemsize
        for n in reversed(shape):
            strides.insert(0, acc)
            acc *= n
    return ai['data'][0], itemsize, shape, tuple(strides)


cdef int _vec(pmx_vec *v, a) except -1:
    memset(v, 0, sizeof(pmx_vec))
    if a is None:
        return 0
    ptr, itemsize, shape, strides = _interface(a)
    v.data = <void *><size_t>ptr
    v.elsize = itemsize
    v.ncol = shape[1] if len(shape) > 1 else 1
    v.stride0 = strides[0] if len(shape) > 0 else 0
    v.stride1 = strides[1] if len(shape) > 1 else 0
    return 0


cdef int _check(int rc) except -1:
    if rc:
        raise RuntimeError((<bytes>pmx_last_error()).decode())
    return 0


cdef class ResampleWindow:
    cdef pmx_painter painter[1]                   # (the reference keeps its PMeshPainter the same way, _window.pyx:70)
    cdef readonly int support
    cdef readonly int nativesupport
    cdef readonly object kind

    def __init__(self, kind, int support=-1):
        cdef int32_t ns = 0, es = 0
        self.kind = kind
        memset(self.painter, 0, sizeof(pmx_painter))
        self.painter.kind = _KINDS[kind]
        _check(pmx_window_info(self.painter.kind, support, &ns, &es))
        self.nativesupport, self.support = ns, es
        self.painter.support = es

    cdef size_t _bind(self, real, order, scale, translate, period) except? 0:
        """the geometric part of the painter for this call (what pmesh_painter_init is given, _window.pyx:140-156)"""
        ptr, itemsize, shape, strides = _interface(real)
        cdef int d
        self.painter.ndim = len(shape)
        self.painter.canvas_elsize = itemsize
        for d in range(len(shape)):
            self.painter.order[d] = order[d]
            self.painter.scale[d] = scale[d]
            self.painter.translate[d] = translate[d]
            self.painter.period[d] = period[d]
            self.painter.size[d] = shape[d]
            self.painter.strides[d] = strides[d]
        return ptr

    def paint(self, real, pos, hsml, mass, order, scale, translate, period):      # _window.pyx:128-165
        cdef pmx_vec pv, mv, hv
        cdef size_t canvas = self._bind(real, order, scale, translate, period)
        cdef int64_t n = pos.shape[0]
        cdef int rc
        _vec(&pv, pos); _vec(&mv, mass); _vec(&hv, hsml)
        if mass.shape[0] == 1:
            mv.stride0 = 0                        # the reference broadcasts a length-1 mass (window.py:146)
        with nogil:
            rc = pmx_paint(self.painter, <void *>canvas, &pv, &mv, 1.0, &hv if hv.data != NULL else NULL, n, NULL)
        _check(rc)

    def readout(self, real, pos, hsml, out, order, scale, translate, period):     # _window.pyx:167-205
        cdef pmx_vec pv, ov, hv
        cdef size_t canvas = self._bind(real, order, scale, translate, period)
        cdef int64_t n = pos.shape[0]
        cdef int rc
        _vec(&pv, pos); _vec(&ov, out); _vec(&hv, hsml)
        with nogil:
            rc = pmx_readout(self.painter, <const void *>canvas, &pv, &hv if hv.data != NULL else NULL, &ov, n, NULL)
        _check(rc)

    def get_fwindow(self, w):                                                      # _window.pyx:116-126
        cdef double[::1] wv = numpy.ascontiguousarray(w, dtype='f8').ravel()
        T = numpy.empty(wv.shape[0], dtype='f8')
        cdef double[::1] tv = T
        _check(pmx_fwindow(self.painter.kind, self.support, &wv[0], wv.shape[0], &tv[0]))
        return T.reshape(numpy.shape(w))
