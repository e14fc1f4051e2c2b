"""A numpy-flavoured handle on a device tensor for the callables of ``Field.apply`` (pm.py:617-648).

The reference hands its transfer functions numpy arrays: ``func(k, v)`` with ``k`` a list of broadcastable
coordinate arrays and ``v`` the values of a slab (pm.py:87-120), and callers write them in numpy — ``numpy.sin(w)``,
``kk[kk == 0] = 1``, ``mask = (kk == 0).nonzero(); b[mask] = 0`` (examples/nbody.py:151-197, fastpm's kernels).  Here
the values live in HBM.  Handing the callable bare ``torch`` tensors works for arithmetic and broadcasting, but numpy
functions refuse device tensors (the callable then falls back to the host: the whole field over PCIe and back), and a
few spellings mean something else in torch — ``Tensor.nonzero()`` is an (n, ndim) matrix, not numpy's tuple of index
arrays, so ``b[mask] = 0`` with it quietly indexes rows.

``DevArr`` wraps the tensor and answers in numpy's terms: ufuncs (``__array_ufunc__``) and the handful of numpy
functions transfer functions use (``__array_function__``) run as the matching torch operations on the device and
return ``DevArr``; operators come from numpy's mixin and therefore follow the same path; indexing, masks and
``nonzero()`` follow numpy's conventions; anything that is not mapped raises ``TypeError``, which ``Field.apply`` takes
as "this callable needs real numpy arrays" and evaluates it on the host as before.  Nothing here is on the fused
path: ``Transfer`` objects never see it.
"""
import numpy
import torch
from numpy.lib.mixins import NDArrayOperatorsMixin

from ._arrays import numpy_dtype, torch_dtype

_UFUNCS = {
    'add': torch.add, 'subtract': torch.sub, 'multiply': torch.mul, 'true_divide': torch.true_divide,
    'divide': torch.true_divide, 'floor_divide': torch.floor_divide, 'negative': torch.neg, 'positive': torch.positive,
    'power': torch.pow, 'float_power': torch.float_power, 'square': torch.square, 'sqrt': torch.sqrt,
    'cbrt': lambda x: torch.sign(x) * torch.abs(x) ** (1.0 / 3.0), 'reciprocal': torch.reciprocal,
    'exp': torch.exp, 'exp2': torch.exp2, 'expm1': torch.expm1, 'log': torch.log, 'log2': torch.log2, 'log10': torch.log10,
    'log1p': torch.log1p, 'sin': torch.sin, 'cos': torch.cos, 'tan': torch.tan, 'arcsin': torch.asin,
    'arccos': torch.acos, 'arctan': torch.atan, 'arctan2': torch.atan2, 'sinh': torch.sinh, 'cosh': torch.cosh,
    'tanh': torch.tanh, 'arcsinh': torch.asinh, 'arccosh': torch.acosh, 'arctanh': torch.atanh, 'hypot': torch.hypot,
    'absolute': torch.abs, 'fabs': torch.abs, 'sign': torch.sign, 'conjugate': torch.conj_physical,
    'isfinite': torch.isfinite, 'isnan': torch.isnan, 'isinf': torch.isinf, 'signbit': torch.signbit,
    'less': torch.lt, 'less_equal': torch.le, 'greater': torch.gt, 'greater_equal': torch.ge, 'equal': torch.eq,
    'not_equal': torch.ne, 'logical_and': torch.logical_and, 'logical_or': torch.logical_or,
    'logical_not': torch.logical_not, 'logical_xor': torch.logical_xor, 'bitwise_and': torch.bitwise_and,
    'bitwise_or': torch.bitwise_or, 'bitwise_xor': torch.bitwise_xor, 'invert': torch.bitwise_not,
    'maximum': torch.maximum, 'minimum': torch.minimum, 'fmax': torch.fmax, 'fmin': torch.fmin,
    'remainder': torch.remainder, 'mod': torch.remainder, 'fmod': torch.fmod, 'floor': torch.floor, 'ceil': torch.ceil,
    'rint': torch.round, 'trunc': torch.trunc, 'heaviside': torch.heaviside, 'copysign': torch.copysign,
    'deg2rad': torch.deg2rad, 'rad2deg': torch.rad2deg,
}
_REDUCE = {'add': torch.sum, 'multiply': torch.prod, 'maximum': torch.amax, 'minimum': torch.amin,
           'logical_and': torch.all, 'logical_or': torch.any}


def _device_of(args):
    for a in args:
        if isinstance(a, DevArr):
            return a.t.device
        if isinstance(a, (list, tuple)):
            d = _device_of(a)
            if d is not None:
                return d
    return None


def _unwrap(a, device):
    """DevArr -> tensor, numpy array -> tensor on `device`, python / numpy scalars stay scalars"""
    if isinstance(a, DevArr):
        return a.t
    if isinstance(a, torch.Tensor):
        return a
    if isinstance(a, numpy.ndarray):
        if a.ndim == 0:
            return a.item()
        return torch.as_tensor(a, device=device)
    if isinstance(a, numpy.generic):
        return a.item()
    if isinstance(a, (list, tuple)) and any(isinstance(x, (DevArr, torch.Tensor)) for x in a):
        return type(a)(_unwrap(x, device) for x in a)
    return a


def _store(dst, r):
    """dst[...] = r with numpy's 'same_kind' rule: a complex result does not go into a real array (torch's copy_ would
    drop the imaginary parts with a warning; numpy raises — and Field.apply then evaluates the callable on the host)"""
    if isinstance(r, torch.Tensor) and r.is_complex() and not dst.is_complex():
        raise TypeError("cannot cast a complex result into a real device array")
    if isinstance(r, complex) and not dst.is_complex():
        raise TypeError("cannot cast a complex result into a real device array")
    if isinstance(r, torch.Tensor):
        dst.copy_(r)
    else:
        dst[...] = r


def _wrap(r):
    if isinstance(r, torch.Tensor):
        return DevArr(r)
    if isinstance(r, tuple):
        return tuple(_wrap(x) for x in r)
    return r


def _index(idx, device):
    if isinstance(idx, tuple):
        return tuple(_index(i, device) for i in idx)
    if isinstance(idx, DevArr):
        return idx.t
    if isinstance(idx, numpy.ndarray):
        return torch.as_tensor(idx, device=device)
    if isinstance(idx, list) and any(isinstance(i, DevArr) for i in idx):
        return [_index(i, device) for i in idx]
    return idx


class DevArr(NDArrayOperatorsMixin):
    """a device tensor that behaves like the numpy array the reference would hand a transfer function"""
    __array_priority__ = 1000

    def __init__(self, t):
        self.t = t

    # -- what callers read off an array ---------------------------------------------------------------------------
    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def ndim(self):
        return self.t.dim()

    @property
    def size(self):
        return self.t.numel()

    @property
    def dtype(self):
        return numpy_dtype(self.t.dtype)

    @property
    def real(self):
        return DevArr(self.t.real if self.t.is_complex() else self.t)

    @real.setter
    def real(self, v):
        (self.t.real if self.t.is_complex() else self.t)[...] = _unwrap(v, self.t.device)

    @property
    def imag(self):
        if self.t.is_complex():
            return DevArr(self.t.imag)
        return DevArr(torch.zeros_like(self.t))

    @imag.setter
    def imag(self, v):
        if not self.t.is_complex():
            raise TypeError('array does not have imaginary part to set')
        self.t.imag[...] = _unwrap(v, self.t.device)

    @property
    def T(self):
        return DevArr(self.t.permute(*reversed(range(self.t.dim()))))

    @property
    def flat(self):
        raise TypeError('DevArr.flat: needs a host array')

    def __len__(self):
        return self.t.shape[0]

    def __iter__(self):
        for i in range(self.t.shape[0]):
            yield DevArr(self.t[i])

    def __repr__(self):
        return 'DevArr(%r)' % (self.t,)

    def __bool__(self):
        return bool(self.t)

    def __float__(self):
        return float(self.t)

    def __int__(self):
        return int(self.t)

    def __complex__(self):
        return complex(self.t)

    def __index__(self):
        return int(self.t)

    def __array__(self, dtype=None, copy=None):
        # a numpy function that is not mapped below: the callable needs host arrays (Field.apply's fallback)
        raise TypeError('this numpy operation is not available on device arrays')

    # -- numpy protocols ------------------------------------------------------------------------------------------
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        dev = _device_of(inputs) or self.t.device
        out = kwargs.pop('out', None)
        where = kwargs.pop('where', True)
        if where is not True:
            raise TypeError('ufunc where= is not available on device arrays')
        if method == '__call__':
            fn = _UFUNCS.get(ufunc.__name__)
            if fn is None:
                raise TypeError('numpy.%s is not available on device arrays' % ufunc.__name__)
            kwargs.pop('casting', None)
            dt = kwargs.pop('dtype', None)
            if kwargs:
                raise TypeError('ufunc arguments %r are not available on device arrays' % (sorted(kwargs),))
            args = [_unwrap(a, dev) for a in inputs]
            if not any(isinstance(a, torch.Tensor) for a in args):
                raise TypeError('no device array among the operands')
            # numpy's (NEP 50) promotion for numpy SCALARS: `x * numpy.float64(c)` with a float32 array computes in
            # float64 — python floats are weak, numpy scalars are not (torch treats both as weak)
            strong = [a.dtype for a in inputs if isinstance(a, numpy.generic) and a.dtype.kind in 'fc']
            if strong:
                have = [numpy_dtype(a.dtype) for a in args if isinstance(a, torch.Tensor) and (a.is_floating_point() or a.is_complex())]
                if have:
                    target = torch_dtype(numpy.result_type(*(have + strong)))
                    args = [a.to(target) if isinstance(a, torch.Tensor) and (a.is_floating_point() or a.is_complex())
                            and a.dtype != target and torch.promote_types(a.dtype, target) == target else a for a in args]
            if len(args) > 1 and ufunc.__name__ not in ('add', 'subtract', 'multiply', 'true_divide', 'divide', 'power'):
                # (most binary torch functions take tensors on both sides; a 0-d tensor promotes like a python scalar)
                args = [a if isinstance(a, torch.Tensor) else torch.as_tensor(a, device=dev) for a in args]
            r = fn(*args)
            if dt is not None:
                r = r.to(torch_dtype(numpy.dtype(dt)))
        elif method == 'reduce':
            fn = _REDUCE.get(ufunc.__name__)
            if fn is None:
                raise TypeError('numpy.%s.reduce is not available on device arrays' % ufunc.__name__)
            axis = kwargs.pop('axis', 0)
            keepdims = kwargs.pop('keepdims', False)
            rdt = kwargs.pop('dtype', None)
            if kwargs.pop('initial', None) is not None or kwargs:
                raise TypeError('reduce arguments are not available on device arrays')
            t = _unwrap(inputs[0], dev)
            if rdt is not None:
                t = t.to(torch_dtype(numpy.dtype(rdt)))          # (the accumulator's type, as numpy's dtype= asks)
            if axis is None:
                r = fn(t)
                if keepdims:
                    r = r.reshape([1] * t.dim())
            else:
                r = fn(t, dim=axis, keepdim=keepdims)
        else:
            raise TypeError('ufunc method %r is not available on device arrays' % method)
        if out is not None:
            o = out[0] if isinstance(out, tuple) else out
            if not isinstance(o, DevArr):
                raise TypeError('out= must be a device array')
            _store(o.t, r)
            return o
        return _wrap(r)

    def __array_function__(self, func, types, args, kwargs):
        fn = _FUNCTIONS.get(func)
        if fn is None:
            raise TypeError('numpy.%s is not available on device arrays' % getattr(func, '__name__', func))
        return fn(*args, **kwargs)

    # -- indexing with numpy's conventions --------------------------------------------------------------------------
    def __getitem__(self, idx):
        return DevArr(self.t[_index(idx, self.t.device)])

    def __setitem__(self, idx, value):
        v = _unwrap(value, self.t.device)
        if not self.t.is_complex() and (isinstance(v, complex) or (isinstance(v, torch.Tensor) and v.is_complex())):
            raise TypeError("cannot cast a complex value into a real device array")      # (numpy's 'same_kind' rule: see _store)
        self.t[_index(idx, self.t.device)] = v

    def nonzero(self):
        return tuple(DevArr(x) for x in self.t.nonzero(as_tuple=True))

    # -- the ndarray methods transfer functions use -------------------------------------------------------------------
    def _reduce(self, fn, axis, keepdims, dtype=None):
        t = self.t if dtype is None else self.t.to(torch_dtype(numpy.dtype(dtype)))      # (numpy's dtype=: the accumulator's type)
        if axis is None:
            r = fn(t)
            return DevArr(r.reshape([1] * t.dim())) if keepdims else DevArr(r)
        return DevArr(fn(t, dim=axis, keepdim=keepdims))

    def sum(self, axis=None, dtype=None, out=None, keepdims=False):
        return self._reduce(torch.sum, axis, keepdims, dtype)

    def prod(self, axis=None, dtype=None, out=None, keepdims=False):
        return self._reduce(torch.prod, axis, keepdims, dtype)

    def mean(self, axis=None, dtype=None, out=None, keepdims=False):
        return self._reduce(torch.mean, axis, keepdims, dtype)

    def max(self, axis=None, out=None, keepdims=False):
        return self._reduce(torch.amax, axis, keepdims)

    def min(self, axis=None, out=None, keepdims=False):
        return self._reduce(torch.amin, axis, keepdims)

    def any(self, axis=None, out=None, keepdims=False):
        return self._reduce(torch.any, axis, keepdims)

    def all(self, axis=None, out=None, keepdims=False):
        return self._reduce(torch.all, axis, keepdims)

    def conj(self):
        return DevArr(torch.conj_physical(self.t))

    conjugate = conj

    def copy(self, order='C'):
        return DevArr(self.t.clone())

    def astype(self, dtype, copy=True):
        return DevArr(self.t.to(torch_dtype(numpy.dtype(dtype)), copy=copy))

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return DevArr(self.t.reshape(shape))

    def ravel(self):
        return DevArr(self.t.reshape(-1))

    def flatten(self):
        return DevArr(self.t.reshape(-1).clone())

    def squeeze(self, axis=None):
        return DevArr(self.t.squeeze() if axis is None else self.t.squeeze(axis))

    def clip(self, a_min=None, a_max=None, out=None):
        return DevArr(torch.clamp(self.t, min=_unwrap(a_min, self.t.device), max=_unwrap(a_max, self.t.device)))

    def fill(self, value):
        self.t.fill_(value)

    def item(self):
        return self.t.item()

    def cpu(self):
        """the values as a host tensor (what callers of this package do with the tensors it hands out elsewhere)"""
        return self.t.cpu()

    def tolist(self):
        return self.t.tolist()

    def __abs__(self):
        return DevArr(torch.abs(self.t))


def _like(maker):
    def f(a, dtype=None, **kw):
        t = a.t if isinstance(a, DevArr) else a
        return DevArr(maker(t, dtype=torch_dtype(numpy.dtype(dtype)) if dtype is not None else None))
    return f


def _where(cond, *xy):
    dev = _device_of((cond,) + xy)
    c = _unwrap(cond, dev)
    if not xy:
        return tuple(DevArr(x) for x in c.nonzero(as_tuple=True))
    x, y = (_unwrap(a, dev) for a in xy)
    ref = x if isinstance(x, torch.Tensor) else (y if isinstance(y, torch.Tensor) else None)
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x, device=dev, dtype=ref.dtype if ref is not None and not isinstance(x, complex) else None)
    if not isinstance(y, torch.Tensor):
        y = torch.as_tensor(y, device=dev, dtype=ref.dtype if ref is not None and not isinstance(y, complex) else None)
    return DevArr(torch.where(c, x, y))


def _reducer(name):
    def f(a, axis=None, dtype=None, out=None, keepdims=False, **kw):
        if kw or out is not None:
            raise TypeError('numpy.%s arguments are not available on device arrays' % name)
        return getattr(a, name)(axis=axis, keepdims=keepdims)
    return f


def _sinc(x):
    return DevArr(torch.sinc(x.t))


_FUNCTIONS = {
    numpy.where: _where,
    numpy.zeros_like: _like(torch.zeros_like), numpy.ones_like: _like(torch.ones_like),
    numpy.empty_like: _like(torch.empty_like),
    numpy.sum: _reducer('sum'), numpy.prod: _reducer('prod'), numpy.mean: _reducer('mean'),
    numpy.amax: _reducer('max'), numpy.amin: _reducer('min'), numpy.max: _reducer('max'), numpy.min: _reducer('min'),
    numpy.any: _reducer('any'), numpy.all: _reducer('all'),
    numpy.sinc: _sinc,
    numpy.real: lambda a: a.real, numpy.imag: lambda a: a.imag, numpy.conj: lambda a: a.conj(),
    numpy.conjugate: lambda a: a.conj(), numpy.copy: lambda a, **kw: a.copy(), numpy.abs: lambda a: abs(a),
    numpy.absolute: lambda a: abs(a), numpy.shape: lambda a: a.shape, numpy.ndim: lambda a: a.ndim,
    numpy.size: lambda a, axis=None: a.size if axis is None else a.shape[axis],
    numpy.nonzero: lambda a: a.nonzero(), numpy.isscalar: lambda a: False,
    numpy.clip: lambda a, a_min=None, a_max=None, **kw: a.clip(a_min, a_max),
    numpy.broadcast_to: lambda a, shape, **kw: DevArr(torch.broadcast_to(a.t, tuple(shape))),
    numpy.iscomplexobj: lambda a: a.t.is_complex(), numpy.isrealobj: lambda a: not a.t.is_complex(),
    numpy.squeeze: lambda a, axis=None: a.squeeze(axis), numpy.ravel: lambda a, **kw: a.ravel(),
    numpy.reshape: lambda a, shape, **kw: a.reshape(shape),
}


def unwrap(r, device):
    """what a callable returned -> a tensor (or a scalar) for the caller to store"""
    return _unwrap(r, device)
