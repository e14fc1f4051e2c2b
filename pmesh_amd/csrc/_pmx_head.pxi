# The hand-written head of _pmx.pyx (the generated per-entry-point wrappers follow it; see gen_pyx.py).
#
# What a pmesh maintainer binds today is Cython (pmesh/_window.pyx:67-205: a cdef class around the C painter, a typed
# loop over the particles with the GIL released inside); this is the same layer over the batched C ABI.
from libc.stdint cimport int32_t, int64_t, uint32_t, uint64_t
from libc.string cimport memset, memcpy
from cpython.bytes cimport PyBytes_AsString

import ctypes as _ct
import os as _os

cdef extern from "dlfcn.h" nogil:
    void *dlopen(const char *path, int flags)
    void *dlsym(void *handle, const char *name)
    char *dlerror()
    int RTLD_NOW
    int RTLD_GLOBAL

cdef extern from "pmesh_amd.h" nogil:
    int PMX_MAXDIM
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
    ctypedef struct pmx_painter_nd:
        pass
    ctypedef struct pmx_vec:
        void *data
        int32_t elsize
        int32_t ncol
        int64_t stride0
        int64_t stride1
    ctypedef struct pmx_grid:
        pass
    ctypedef struct pmx_transfer:
        double amplitude
        int32_t laplace_pow
        int32_t grad_dir
        int32_t grad_kind
        int32_t deconv_pow
        double gauss_r
    ctypedef struct pmx_binplan:
        pass
    ctypedef struct pmx_fft:
        pass

_bound = None

cdef object _c_void_p = _ct.c_void_p
cdef object _CArg = type(_ct.byref(_ct.c_int()))
cdef object _addressof = _ct.addressof
cdef object _Pointer = _ct._Pointer
cdef object _cast = _ct.cast


def _fsencode(path):
    return _os.fsencode(path)


cdef class Struct:
    """Memory for one C struct of the ABI (pmx_painter, pmx_vec, ...) owned by the shim: `addr` is its address, the
    ctypes mirror class of pmesh_amd/_abi.py built over it with from_address() gives the host code named fields —
    filled once, passed many times without a byref() / addressof() per call."""
    cdef public size_t addr
    cdef public object view        # the ctypes structure living in this memory
    cdef bytearray store

    def __cinit__(self, ctype):
        n = _ct.sizeof(ctype)
        self.store = bytearray(n + 16)
        cdef char *p = self.store
        self.addr = (<size_t>p + 15) & ~<size_t>15
        self.view = ctype.from_address(self.addr)


cdef inline size_t _ptr(object o) except? <size_t>-1:
    """the address an argument stands for: None (NULL), an int, a ctypes c_void_p / byref(x) / structure / array /
    POINTER(T) instance, or a Struct of this module"""
    if o is None:
        return 0
    t = type(o)
    if t is int:
        return <size_t>o
    if t is Struct:
        return (<Struct>o).addr
    if t is _c_void_p:
        v = o.value
        return 0 if v is None else <size_t>v
    if t is _CArg:
        return <size_t>_addressof(o._obj)
    if isinstance(o, _Pointer):
        v = _cast(o, _c_void_p).value
        return 0 if v is None else <size_t>v
    if isinstance(o, int):
        return <size_t>int(o)
    return <size_t>_addressof(o)


def address(o):
    """the address `o` stands for as an argument of an entry point (tests)"""
    return _ptr(o)


def bound():
    """path of the library the entry points are bound to, or None"""
    return _bound


def fill_painter(p, int kind, int support, int elsize, shape, strides, order,
                 const double[::1] scale, const double[::1] translate, period):
    """pmx_painter at `p` (a Struct or a ctypes Painter) <- window kind / support, the canvas block (shape, strides in
    ELEMENTS, element size) and the affine transform (the fields of window.Affine): what ResampleWindow._painter does
    with ~30 ctypes attribute stores, as one typed call.  Three dimensions at most (pmx_painter_nd is filled by the
    Python path)."""
    cdef pmx_painter *q = <pmx_painter *>_ptr(p)
    cdef int d, nd = len(shape)
    if nd > 3 or q == NULL:
        raise ValueError('fill_painter: 1 to 3 dimensions')
    memset(q, 0, sizeof(pmx_painter))
    q.kind = kind
    q.support = support
    q.ndim = nd
    q.canvas_elsize = elsize
    for d in range(nd):
        q.order[d] = <int32_t>order[d]
        q.scale[d] = scale[d]
        q.translate[d] = translate[d]
        q.period[d] = <int64_t>period[d]
        q.size[d] = <int64_t>shape[d]
        q.strides[d] = <int64_t>strides[d] * elsize


def fill_vec(v, size_t data, int elsize, int ncol, long long stride0, long long stride1):
    """pmx_vec at `v` <- a strided per-particle column set"""
    cdef pmx_vec *q = <pmx_vec *>_ptr(v)
    q.data = <void *>data
    q.elsize = elsize
    q.ncol = ncol
    q.stride0 = stride0
    q.stride1 = stride1
