"""The reference-side binding of INTEGRATION.md section 2, in executable form.

What a pmesh maintainer would add next to `pmesh/_window.pyx` to keep `pmesh/*.py` and swap only the native
layer: a drop-in for `pmesh._window.ResampleWindow` (pmesh/_window.pyx:67-205) that speaks to
`libpmesh_amd.so` through ctypes and nothing else — no pmesh_amd Python code is imported.  Arrays are any
device arrays with the CUDA array interface (torch / cupy / numba).  tests/test_integration_stub.py drives it
under `-m gpu` with the argument lists of _window.pyx:128-205.
"""
import ctypes as C
import os

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = C.CDLL(os.environ.get('PMESH_AMD_LIBRARY') or
              os.path.join(os.path.dirname(_HERE), 'pmesh_amd', 'libpmesh_amd.so'))


class pmx_painter(C.Structure):                   # include/pmesh_amd.h: struct pmx_painter
    _fields_ = [("kind", C.c_int32), ("support", C.c_int32), ("ndim", C.c_int32),
                ("canvas_elsize", C.c_int32), ("order", C.c_int32 * 3), ("_pad", C.c_int32),
                ("scale", C.c_double * 3), ("translate", C.c_double * 3),
                ("period", C.c_int64 * 3), ("size", C.c_int64 * 3), ("strides", C.c_int64 * 3)]


class pmx_vec(C.Structure):                       # include/pmesh_amd.h: struct pmx_vec
    _fields_ = [("data", C.c_void_p), ("elsize", C.c_int32), ("ncol", C.c_int32),
                ("stride0", C.c_int64), ("stride1", C.c_int64)]


_lib.pmx_paint.argtypes = [C.POINTER(pmx_painter), C.c_void_p, C.POINTER(pmx_vec),
                           C.POINTER(pmx_vec), C.c_double, C.POINTER(pmx_vec), C.c_int64, C.c_void_p]
_lib.pmx_readout.argtypes = [C.POINTER(pmx_painter), C.c_void_p, C.POINTER(pmx_vec),
                             C.POINTER(pmx_vec), C.POINTER(pmx_vec), C.c_int64, C.c_void_p]
_lib.pmx_window_info.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
_lib.pmx_fwindow.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double)]
_lib.pmx_last_error.restype = C.c_char_p
_KINDS = {'nearest': 0, 'linear': 1, 'quadratic': 2, 'cubic': 3,
          'tunednnb': 4, 'tunedcic': 5, 'tunedtsc': 6, 'tunedpcs': 7}


def _interface(a):
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


def _vec(a):
    ptr, itemsize, shape, strides = _interface(a)
    v = pmx_vec()
    v.data, v.elsize = ptr, itemsize
    v.ncol = shape[1] if len(shape) > 1 else 1
    v.stride0 = strides[0] if len(shape) > 0 else 0
    v.stride1 = strides[1] if len(shape) > 1 else 0
    return v


def _check(rc):
    if rc:
        raise RuntimeError(_lib.pmx_last_error().decode())


class ResampleWindow(object):
    def __init__(self, kind, support=-1):
        self.kind, self._k = kind, _KINDS[kind]
        ns, es = C.c_int32(), C.c_int32()
        _check(_lib.pmx_window_info(self._k, support, C.byref(ns), C.byref(es)))
        self.nativesupport, self.support = ns.value, es.value

    def _painter(self, real, order, scale, translate, period):
        ptr, itemsize, shape, strides = _interface(real)
        p = pmx_painter()
        p.kind, p.support, p.ndim, p.canvas_elsize = self._k, self.support, len(shape), itemsize
        for d in range(len(shape)):
            p.order[d], p.scale[d], p.translate[d] = int(order[d]), float(scale[d]), float(translate[d])
            p.period[d], p.size[d], p.strides[d] = int(period[d]), shape[d], strides[d]
        return p, ptr

    def paint(self, real, pos, hsml, mass, order, scale, translate, period):      # _window.pyx:128-165
        p, canvas = self._painter(real, order, scale, translate, period)
        pv, mv = _vec(pos), _vec(mass)
        if mass.shape[0] == 1:
            mv.stride0 = 0                        # the reference broadcasts a length-1 mass (window.py:146)
        hv = _vec(hsml) if hsml is not None else None
        _check(_lib.pmx_paint(C.byref(p), canvas, C.byref(pv), C.byref(mv), 1.0,
                              C.byref(hv) if hv is not None else None, pos.shape[0], None))

    def readout(self, real, pos, hsml, out, order, scale, translate, period):     # _window.pyx:167-205
        p, canvas = self._painter(real, order, scale, translate, period)
        pv, ov = _vec(pos), _vec(out)
        hv = _vec(hsml) if hsml is not None else None
        _check(_lib.pmx_readout(C.byref(p), canvas, C.byref(pv), C.byref(hv) if hv is not None else None,
                                C.byref(ov), pos.shape[0], None))

    def get_fwindow(self, w):                                                      # _window.pyx:116-126
        w = numpy.ascontiguousarray(w, dtype='f8')
        T = numpy.empty_like(w)
        _check(_lib.pmx_fwindow(self._k, self.support, w.ctypes.data_as(C.POINTER(C.c_double)), w.size,
                                T.ctypes.data_as(C.POINTER(C.c_double))))
        return T
