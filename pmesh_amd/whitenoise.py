"""White noise in Fourier space (pmesh/whitenoise.py:4-43).

3-d: the Gadget / N-GenIC compatible generator of the reference (pmesh/_whitenoise_imp.c,
_whitenoise_generics.h), on the device (csrc/pmx_whitenoise.hip): independent of the domain
decomposition, large scales invariant under a change of the mesh size, Hermitian.
1-d / 2-d: the reference's own numpy definition (a normal field through fftn; "only used for
testing", whitenoise.py:24-41) evaluated on the host and copied into the block.
"""

import os

import numpy
import torch

from . import _abi, backend
from ._arrays import is_tensor


#: where the master seed stream runs (one sequential chain of N0 * N1 draws): False = one host core (default, a few ms at
#: 512^2), True = one device thread (no copy, no wait, ~35 x slower; include/pmesh_amd.h: pmx_whitenoise_master)
MASTER_ON_DEVICE = os.environ.get('PMESH_AMD_WN_MASTER', 'host') == 'device'


def generate(complex, start, Nmesh, seed, unitary):
    """
        The result is always hermitian.

        complex : the local block (a complex device tensor, any strides; or a numpy array,
        which is filled through a device buffer) of the half spectrum starting at `start`.

        unitary : bool
            True for a unitary gaussian field (amplitude is fixed to 1)
            False for a true gaussian field
    """
    ndim = complex.ndim if not is_tensor(complex) else complex.dim()
    _start = numpy.empty(ndim, dtype='intp')
    _Nmesh = numpy.empty(ndim, dtype='intp')
    _start[:] = start
    _Nmesh[:] = Nmesh
    be = backend.get()
    if not is_tensor(complex):
        dev = torch.zeros(tuple(complex.shape), dtype=torch.complex64 if complex.dtype.itemsize == 8
                          else torch.complex128, device=be.device)
        generate(dev, _start, _Nmesh, seed, unitary)
        complex[...] = dev.cpu().numpy()
        return
    if ndim == 3:
        be.call('whitenoise_master', 1 if MASTER_ON_DEVICE else 0)
        if not complex.is_complex():
            raise TypeError('complex must be a complex array')
        es = complex.element_size()
        strides = [s * es for s in complex.stride()]
        be.call('whitenoise', int(seed) & 0xFFFFFFFF, int(bool(unitary)), _abi.i64arr(_Nmesh, 3),
                _abi.i64arr(_start, 3), _abi.i64arr(list(complex.shape), 3), _abi.i64arr(strides, 3), es,
                complex.data_ptr() if complex.numel() else None, be.stream())
    elif ndim <= 2:
        # FIXME of the reference kept: not scale invariant, but invariant against the partition
        rng = numpy.random.RandomState(seed)
        real = rng.normal(size=_Nmesh)
        full = numpy.fft.fftn(real)
        full[...] *= numpy.prod(_Nmesh) ** -0.5
        slices = tuple([slice(a, a + b) for a, b in zip(_start, complex.shape)])
        block = full[slices]
        if unitary:
            block = numpy.exp(1j * numpy.angle(block))
        complex.copy_(torch.from_numpy(numpy.ascontiguousarray(block)).to(complex.dtype))
    else:
        raise ValueError("Only knows how to make a whitenoise up to 3d")
