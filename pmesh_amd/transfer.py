"""Fused transfer functions for ``ComplexField.apply``.

The reference applies arbitrary Python callables slab by slab
(pmesh/pm.py:617-648); the PM cycle only ever uses a few closed forms
(examples/nbody.py:154-181; pmesh/transfer.py:69-112, 232-240; window
compensation pmesh/window.py:65-80).  A :class:`Transfer` describes such a form
and runs as ONE kernel over the complex field (csrc/pmx_transfer.hip, one
complex read + one write per mode, wavenumbers recomputed from the index):

    T(k) = amplitude * (k^2)^laplace_pow * exp(-k^2 r^2 / 2) / prod_d sinc(w_d/2)^deconv_pow
           * [ i * D(k_dir) ]            k^2(0) := 1 as in nbody.py:156-157

with D(k) = k ("dx1_transfer") or the 4-point finite difference
(8 sin w - sin 2w) / (6 C), w = k C, C = L/N ("force_transfer").

A Transfer is also an ordinary ``func(k, v)`` callable (same formula written
with array operators), so it can be passed anywhere the reference takes a
filter, and the two evaluations are tested against each other.
"""
import ctypes as C

import numpy
import torch

from . import _abi, backend


class Transfer(object):
    def __init__(self, amplitude=1.0, laplace_pow=0, grad_dir=None, grad_kind='spectral',
                 deconv_pow=0, gauss_r=0.0):
        self.amplitude = float(amplitude)
        self.laplace_pow = int(laplace_pow)
        self.grad_dir = -1 if grad_dir is None else int(grad_dir)
        if grad_kind not in ('spectral', 'finite4'):
            raise ValueError("grad_kind must be 'spectral' or 'finite4'")
        self.grad_kind = grad_kind
        self.deconv_pow = int(deconv_pow)
        self.gauss_r = float(gauss_r)

    # the closed forms of the reference, by name
    @classmethod
    def dx1(cls, direction):
        """ 1j * k_d / k^2  (examples/nbody.py:154-160) """
        return cls(laplace_pow=-1, grad_dir=direction, grad_kind='spectral')

    @classmethod
    def force(cls, direction):
        """ 1j * D4(k_d) / k^2  (examples/nbody.py:162-171) """
        return cls(laplace_pow=-1, grad_dir=direction, grad_kind='finite4')

    @classmethod
    def potential(cls):
        """ -1 / k^2  (examples/nbody.py:173-176; transfer.py:232-240) """
        return cls(amplitude=-1.0, laplace_pow=-1)

    @classmethod
    def lowpass(cls, r):
        """ exp(-k^2 r^2 / 2)  (examples/nbody.py:177-181; transfer.py:97-112) """
        return cls(gauss_r=r)

    @classmethod
    def compensation(cls, resampler):
        """ 1 / prod_d sinc(w_d/2)^p, p the native support of the window
            (ResampleWindow.get_compensation, window.py:65-80) """
        from .window import FindResampler
        return cls(deconv_pow=FindResampler(resampler).nativesupport)

    def fusable(self):
        """closed forms without per-element transcendentals can ride on the first pass of c2r ([r4] the
        finite-difference gradient along axes 1 and 2 too: its factor belongs to the column and comes from a table;
        along axis 0 it would cost the fused kernels a load per element and stays a kernel of its own)"""
        return (self.gauss_r == 0.0 and self.deconv_pow == 0 and -1 <= self.laplace_pow <= 1 and
                (self.grad_dir < 0 or self.grad_kind == 'spectral' or self.grad_dir > 0))

    def _cstruct(self):
        t = _abi.Transfer()
        t.amplitude = self.amplitude
        t.laplace_pow = self.laplace_pow
        t.grad_dir = self.grad_dir
        t.grad_kind = 0 if self.grad_kind == 'spectral' else 1
        t.deconv_pow = self.deconv_pow
        t.gauss_r = self.gauss_r
        return t

    def _launch(self, field, outv):
        be = backend.get()
        v = field.value
        es = v.element_size()
        nd = v.dim()
        t = self._cstruct()
        be.call('apply_transfer', C.byref(t), nd, es // 2, v.data_ptr(),
                _abi.i64arr([s * es for s in v.stride()], 3), outv.data_ptr(),
                _abi.i64arr([s * es for s in outv.stride()], 3), _abi.i64arr(v.shape, 3),
                _abi.i64arr(field.start, 3), _abi.i64arr(field.Nmesh, 3),
                _abi.f64arr(field.BoxSize, 3), be.stream())

    def __call__(self, k, v):
        """ the same transfer as a reference-style filter func(k, v), kind='wavenumber' """
        xp_sin, xp_exp = (torch.sin, torch.exp) if isinstance(v, torch.Tensor) else (numpy.sin, numpy.exp)
        k2 = sum(ki ** 2 for ki in k)
        r = self.amplitude
        if self.laplace_pow:
            q = k2 + (k2 == 0) * 1.0
            r = r * q ** self.laplace_pow
        if self.gauss_r:
            r = r * xp_exp(-0.5 * k2 * self.gauss_r ** 2)
        if self.deconv_pow:
            BoxSize, Nmesh = v.BoxSize, v.Nmesh
            for ki, L, N in zip(k, BoxSize, Nmesh):
                w = ki * (float(L) / float(N))
                half = 0.5 * w
                s = xp_sin(half) / (half + (half == 0) * 1.0) + (half == 0) * 1.0
                r = r / s ** self.deconv_pow
        if self.grad_dir >= 0:
            d = self.grad_dir
            if self.grad_kind == 'spectral':
                D = k[d]
            else:
                Cc = float(v.BoxSize[d]) / float(v.Nmesh[d])
                w = k[d] * Cc
                D = 1.0 / Cc * 1 / 6.0 * (8 * xp_sin(w) - xp_sin(2 * w))
            r = 1j * (r * D)
        return r * v
