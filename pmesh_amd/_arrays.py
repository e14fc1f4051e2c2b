"""Moving caller arrays to the device and describing them to the C ABI.

The reference is numpy-in / numpy-out.  Here particle and mesh data live in HBM
as torch tensors (torch is only the allocator / handle); numpy arrays and lists
are accepted for drop-in compatibility and staged through the device (the
PCIe-inclusive path; results are copied back into the caller's array).
"""
import ctypes as C
import os

import numpy
import torch

from . import _abi
try:
    from . import _pmx
except ImportError:
    _pmx = None

_FLOATS = (torch.float32, torch.float64)


def is_tensor(x):
    return isinstance(x, torch.Tensor)


def touched(*tensors):
    """Tell torch that the library wrote into these tensors through their raw pointers (the
    kernels do not go through torch, so nothing else bumps the version counter).  The bin-plan
    cache (window._BinCache) and the Layout memos key on (data_ptr, _version): without the bump
    a tensor rewritten by readout(out=), gather or an FFT would be served a stale plan."""
    for t in tensors:
        if isinstance(t, torch.Tensor) and not t.is_inference():
            torch.autograd.graph.increment_version(t)


_unversioned = [0]


def version_of(t):
    """the version counter of a tensor for cache keys; tensors made under torch.inference_mode() have
    none: they get a fresh value every time, i.e. they are never found in a cache"""
    if t.is_inference():
        _unversioned[0] -= 1
        return _unversioned[0]
    return t._version


# ---- host <-> device staging for numpy callers -------------------------------------------
# The reference is numpy in / numpy out; a drop-in caller hands pageable host arrays on every
# call.  Uploads go through torch's own pageable path (measured on the MI355X box: 3.2 GB of
# positions in ~60 ms, ~53 GB/s — a hand-rolled pair of pinned bounce buffers on a side stream
# was 5x slower, scripts in the history of this file); downloads land directly in the
# caller's array when there is one, instead of in a fresh host tensor that is copied again.
def upload(host, device):
    """host tensor -> device"""
    return host.to(device)


def to_numpy(t, out=None):
    """device (or host) tensor -> numpy array; `out`: a numpy array to fill in place"""
    if out is not None:
        if (t.device.type == 'cuda' and out.flags.c_contiguous and out.flags.writeable and
                out.dtype == numpy_dtype_of(t) and out.shape == tuple(t.shape)):
            torch.from_numpy(out).copy_(t)
        else:
            out[...] = t.cpu().numpy()
        return out
    return t.cpu().numpy()


def numpy_dtype_of(t):
    try:
        return numpy_dtype(t.dtype)
    except KeyError:
        return None


class Staged(object):
    """A host array registered for repeated use (`ParticleMesh.stage(X)` / `pmesh_amd.stage(X)`): its
    device copy is made once and every later paint / readout / decompose that is handed this object
    works on that copy — examples/nbody.py's force() uploads the same positions for the paint and for
    each of its three readouts, 3.2 GB each at 512^3 particles.  Results still come back as numpy arrays:
    to the library it is a host array whose upload has already happened.  The caller promises not to
    change the host array while the handle is in use (`refresh()` after an update in place)."""

    def __init__(self, host, device):
        self.host = numpy.asarray(host)
        self.tensor = None
        self.device = device
        self.refresh()

    def refresh(self):
        """upload the host array again (after it was changed in place)"""
        t, _ = to_device(self.host, self.device, 'staged array', allow_int=True)
        if self.tensor is not None and self.tensor.shape == t.shape and self.tensor.dtype == t.dtype:
            self.tensor.copy_(t)                 # same storage: plans and layouts keyed on it see a new version
        else:
            self.tensor = t
        return self

    # enough of the array protocol for callers that go on using the handle like the array it wraps
    def __len__(self):
        return len(self.host)

    def __array__(self, dtype=None, copy=None):
        return self.host if dtype is None else self.host.astype(dtype)

    @property
    def shape(self):
        return self.host.shape

    @property
    def dtype(self):
        return self.host.dtype

    @property
    def ndim(self):
        return self.host.ndim


def to_device(x, device, what='array', allow_int=False):
    """-> (tensor on `device`, came_from_host).  Lists become float64."""
    if isinstance(x, Staged):
        t = x.tensor
        if not allow_int and t.dtype not in _FLOATS:
            raise TypeError('%s must be float32 or float64, got %s' % (what, t.dtype))
        return t, True
    if is_tensor(x):
        t = x
        host = False
    else:
        a = numpy.asarray(x)
        if a.dtype == object:
            raise TypeError('%s: unsupported dtype object' % what)
        # (No implicit memo of device copies: `flags.writeable == False` does not mean the contents cannot change — a
        # read-only view of a writable base, setflags, mode='r' memory maps.  A caller who passes the same array to
        # many calls registers it once with ParticleMesh.stage, which is explicit about who refreshes it.)
        if not a.flags.writeable:
            a = a.copy()
        if a.dtype.byteorder not in ('=', '|', '<'):
            a = a.astype(a.dtype.newbyteorder('='))
        # negative strides are not representable by from_numpy
        if any(s < 0 for s in a.strides):
            a = numpy.ascontiguousarray(a)
        t = torch.from_numpy(a) if a.ndim else torch.tensor(a.item(), dtype=_torch_dtype(a.dtype))
        host = True
    if not allow_int and t.dtype not in _FLOATS:
        # the reference's fused types are f4/f8 only (_window.pyx:6-16)
        raise TypeError('%s must be float32 or float64, got %s '
                        '(Function call with ambiguous argument types)' % (what, t.dtype))
    if t.device != device:
        t = upload(t, device) if host else t.to(device)
    return t, host


def _torch_dtype(dt):
    return {'f4': torch.float32, 'f8': torch.float64, 'i4': torch.int32, 'i8': torch.int64,
            'c8': torch.complex64, 'c16': torch.complex128, 'i2': torch.int16,
            'u1': torch.uint8, 'b1': torch.bool}[numpy.dtype(dt).str[1:]]


def torch_dtype(dt):
    if isinstance(dt, torch.dtype):
        return dt
    return _torch_dtype(dt)


def numpy_dtype(dt):
    return {torch.float32: numpy.dtype('f4'), torch.float64: numpy.dtype('f8'),
            torch.complex64: numpy.dtype('c8'), torch.complex128: numpy.dtype('c16'),
            torch.int32: numpy.dtype('i4'), torch.int64: numpy.dtype('i8'),
            torch.int16: numpy.dtype('i2')}[dt]


def vec(t):
    """pmx_vec describing a 1-d or 2-d float tensor (any strides)."""
    v = _abi.Vec()
    if t is None:
        return v
    es = t.element_size()
    if _pmx is not None:
        nd = t.dim()
        _pmx.fill_vec(v, t.data_ptr(), es, t.shape[1] if nd > 1 else 1, t.stride(0) * es if nd else 0,
                      t.stride(1) * es if nd > 1 else 0)
        return v
    v.data = t.data_ptr()
    v.elsize = es
    if t.dim() == 0:
        v.ncol = 1
        v.stride0 = 0
        v.stride1 = 0
    elif t.dim() == 1:
        v.ncol = 1
        v.stride0 = t.stride(0) * es
        v.stride1 = 0
    else:
        v.ncol = t.shape[1]
        v.stride0 = t.stride(0) * es
        v.stride1 = t.stride(1) * es
    return v


def vec_ref(v):
    return C.byref(v) if v is not None else None


def real_view(t):
    """The float view of a canvas: complex canvases paint into their real part
    (window.py:161-162 `real = real.real`)."""
    if t.is_complex():
        return torch.view_as_real(t)[..., 0]
    return t
