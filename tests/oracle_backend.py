"""A test double for pmesh_amd.backend: the same C ABI served by the CPU oracle.

TEST INFRASTRUCTURE ONLY.  It lets the `-m "not gpu"` suite exercise the host
layer of pmesh_amd (argument handling, Field/Layout logic, the distributed FFT
schedule and the torch.distributed collectives over gloo) on machines without
a GPU, against oracle/liboracle.so (prefix pmo_, host pointers) and numpy.fft.
The product never installs it: pmesh_amd.backend.get() only ever builds the
HIP backend.
"""
import ctypes as C

import numpy
import torch

from pmesh_amd import _abi, backend
from oracle import oracle as O


class _Plan(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


class OracleBackend(object):
    name = 'oracle-double'
    prefix = 'pmo_'

    def __init__(self):
        self.lib, _ = O.lib('oracle')
        self.device = torch.device('cpu')

    def stream(self):
        return None

    def synchronize(self):
        pass

    def call(self, name, *args):
        if name in ('window_set_table', 'whitenoise_master'):
            return                      # the reference library has its tables compiled in; one form of the master stream
        if name in ('paint', 'readout', 'paint_nd', 'readout_nd', 'window_info', 'fwindow'):
            kind = args[0] if name in ('window_info', 'fwindow') else args[0]._obj.kind
            if kind >= 8:               # table-driven kinds: only the compiled reference has them
                if not O.have_ref():
                    import pytest
                    pytest.skip('oracle/_ref (compiled reference) is not available')
                lib, prefix = O.lib('ref')
                rc = getattr(lib, prefix + name)(*args)
                if rc != 0:
                    raise backend.PmxError('ref_' + name, rc)
                return
        rc = getattr(self.lib, 'pmo_' + name)(*args)
        if rc != 0:
            raise backend.PmxError('pmo_' + name, rc)

    # ---- FFT: numpy.fft with the plan's strided/batched geometry ---------
    def fft_create(self, kind, elsize, n, istride, idist, ostride, odist, batch, scale, inplace):
        return _Plan(kind=kind, elsize=elsize, n=tuple(int(x) for x in n),
                     istride=tuple(int(x) for x in istride), idist=int(idist),
                     ostride=tuple(int(x) for x in ostride), odist=int(odist), batch=int(batch),
                     scale=float(scale), inplace=bool(inplace))

    @staticmethod
    def _view(t, is_complex, elsize, n, stride, dist, batch):
        """numpy strided view (batch, *n) over the storage of tensor t, in elements
        of the requested kind (real or complex)."""
        rdt = numpy.dtype('f4' if elsize == 4 else 'f8')
        cdt = numpy.dtype('c8' if elsize == 4 else 'c16')
        dt = cdt if is_complex else rdt
        base = t.detach().numpy()
        if not base.flags.c_contiguous:
            raise ValueError('FFT buffers must be contiguous storage')
        flat = base.reshape(-1).view(dt)
        shape = (batch,) + tuple(n)
        strides = (dist * dt.itemsize,) + tuple(s * dt.itemsize for s in stride)
        return numpy.lib.stride_tricks.as_strided(flat, shape=shape, strides=strides)

    def fft_execute(self, plan, tin, tout):
        n = plan.n
        nd = len(n)
        axes = tuple(range(1, nd + 1))
        nc = n[:-1] + (n[-1] // 2 + 1,)
        if plan.kind == _abi.PMX_FFT_R2C:
            src = self._view(tin, False, plan.elsize, n, plan.istride, plan.idist, plan.batch)
            res = numpy.fft.rfftn(numpy.array(src, dtype='f8'), axes=axes) * plan.scale
            dst = self._view(tout, True, plan.elsize, nc, plan.ostride, plan.odist, plan.batch)
            dst[...] = res
        elif plan.kind == _abi.PMX_FFT_C2R:
            src = self._view(tin, True, plan.elsize, nc, plan.istride, plan.idist, plan.batch)
            res = numpy.fft.irfftn(numpy.array(src, dtype='c16'), s=n, axes=axes)
            res *= numpy.prod(n, dtype='f8') * plan.scale  # unnormalised backward
            dst = self._view(tout, False, plan.elsize, n, plan.ostride, plan.odist, plan.batch)
            dst[...] = res
        else:
            src = self._view(tin, True, plan.elsize, n, plan.istride, plan.idist, plan.batch)
            a = numpy.array(src, dtype='c16')
            if plan.kind == _abi.PMX_FFT_C2C_FWD:
                res = numpy.fft.fftn(a, axes=axes)
            else:
                res = numpy.fft.ifftn(a, axes=axes) * numpy.prod(n, dtype='f8')
            dst = self._view(tout, True, plan.elsize, n, plan.ostride, plan.odist, plan.batch)
            dst[...] = res * plan.scale

    def fft_destroy(self, plan):
        pass

    # ---- column FFT (numpy restatement of csrc/pmx_colfft.hip) ------------
    def colfft_supported(self, n, elsize):
        from pmesh_amd.fft import _col_length_ok
        return _col_length_ok(n, elsize)

    def colfft(self, elsize, inverse, data, A, N, B, scale=1.0, transfer=None, n1=1, n2=1,
               start=(0, 0, 0), nmesh=(1, 1, 1), boxsize=(1.0, 1.0, 1.0), a_stride=0, n_stride=0):
        cdt = 'c8' if elsize == 4 else 'c16'
        flat = data.detach().numpy().reshape(-1).view(cdt)
        sa = a_stride or N * B
        sn = n_stride or B
        arr = numpy.lib.stride_tricks.as_strided(flat, (A, N, B), (sa * flat.itemsize, sn * flat.itemsize,
                                                                   flat.itemsize))
        x = arr.astype('c16')
        if transfer is not None:
            blk = numpy.ascontiguousarray(x.reshape(N, n1, n2))
            blk = O.apply_transfer(transfer, blk, start, nmesh, boxsize)
            x = blk.reshape(A, N, B)
        y = numpy.fft.ifft(x, axis=1) * N if inverse else numpy.fft.fft(x, axis=1)
        arr[...] = y * scale

    def colfft_configure(self, persistent):
        pass

    def colfft_roundtrip_supported(self, n, elsize):
        return self.colfft_supported(n, elsize)

    def colfft_roundtrip(self, elsize, data, N, B, scale=1.0, transfer=None, n1=1, n2=1,
                         start=(0, 0, 0), nmesh=(1, 1, 1), boxsize=(1.0, 1.0, 1.0), n_stride=0):
        self.colfft(elsize, False, data, 1, N, B, scale=scale, n_stride=n_stride)
        self.colfft(elsize, True, data, 1, N, B, transfer=transfer, n1=n1, n2=n2, start=start, nmesh=nmesh,
                    boxsize=boxsize, n_stride=n_stride)

    def colfft_split(self, elsize, inverse, src, dst, A, N, B, nsplit, scale=1.0, plain_pitch=0):
        cdt = 'c8' if elsize == 4 else 'c16'
        pp = plain_pitch or B
        plain_t, split_t = (dst, src) if inverse else (src, dst)
        pflat = plain_t.detach().numpy().reshape(-1).view(cdt)      # may start at a column offset
        isz = pflat.itemsize
        plain = numpy.lib.stride_tricks.as_strided(pflat, (A, N, B), (N * pp * isz, pp * isz, isz))
        split = split_t.detach().numpy().reshape(-1).view(cdt)[:A * N * B]
        R = N // nsplit
        if inverse:
            x = split.reshape(R, A, nsplit, B).transpose(1, 0, 2, 3).reshape(A, N, B).astype('c16')
            plain[...] = numpy.fft.ifft(x, axis=1) * N * scale
        else:
            y = numpy.fft.fft(plain.astype('c16'), axis=1) * scale
            split.reshape(R, A, nsplit, B)[...] = y.reshape(A, R, nsplit, B).transpose(1, 0, 2, 3)

    def colfft_resplit(self, elsize, inverse, src, dst, A, N, B, nsplit_in, nsplit_out, scale=1.0):
        cdt = 'c8' if elsize == 4 else 'c16'

        def view(t, ns):
            flat = t.detach().numpy().reshape(-1).view(cdt)[:A * N * B]
            if ns == 0:
                return flat.reshape(A, N, B), None
            return flat.reshape(N // ns, A, ns, B), ns
        sv, sn = view(src, nsplit_in)
        x = (sv if sn is None else sv.transpose(1, 0, 2, 3).reshape(A, N, B)).astype('c16')
        y = (numpy.fft.ifft(x, axis=1) * N if inverse else numpy.fft.fft(x, axis=1)) * scale
        dv, dn = view(dst, nsplit_out)
        if dn is None:
            dv[...] = y
        else:
            dv[...] = y.reshape(A, N // dn, dn, B).transpose(1, 0, 2, 3)

    def colfft_chunk(self, elsize, inverse, chunk, full, N, n1, cw, pitch, coff, to_full, scale=1.0,
                     transfer=None, start=(0, 0, 0), nmesh=(1, 1, 1), boxsize=(1.0, 1.0, 1.0)):
        cdt = 'c8' if elsize == 4 else 'c16'
        ch = chunk.detach().numpy().reshape(-1).view(cdt)[:N * n1 * cw].reshape(N, n1, cw)
        fu = full.detach().numpy().reshape(-1).view(cdt)[:N * n1 * pitch].reshape(N, n1, pitch)[:, :, coff:coff + cw]
        x = (ch if to_full else fu).astype('c16')
        if transfer is not None:
            st = (start[0], start[1], start[2] + coff)
            x = O.apply_transfer(transfer, numpy.ascontiguousarray(x), st, nmesh, boxsize)
        y = (numpy.fft.ifft(x, axis=0) * N if inverse else numpy.fft.fft(x, axis=0)) * scale
        if to_full:
            fu[...] = y
        else:
            ch[...] = y

    def rowfft_supported(self, n, elsize):
        from pmesh_amd.fft import _row_length_ok
        return _row_length_ok(n)

    def rowfft(self, elsize, inverse, data, nrows, n, pitch, scale=1.0, rows_per_plane=0, plane_pitch=0):
        rdt, cdt = ('f4', 'c8') if elsize == 4 else ('f8', 'c16')
        flat = data.detach().numpy().reshape(-1)
        if rows_per_plane:
            # planes of rows_per_plane rows, plane_pitch complex elements apart
            for a in range(nrows // rows_per_plane):
                sub = data.reshape(-1)[2 * a * plane_pitch:]
                self.rowfft(elsize, inverse, sub, rows_per_plane, n, pitch, scale=scale)
            return
        real = flat.view(rdt)[:nrows * 2 * pitch].reshape(nrows, 2 * pitch)
        cplx = flat.view(cdt)[:nrows * pitch].reshape(nrows, pitch)
        if inverse:
            x = numpy.fft.irfft(cplx[:, :n // 2 + 1].astype('c16'), n=n, axis=1) * n
            real[:, :n] = x * scale
        else:
            y = numpy.fft.rfft(real[:, :n].astype('f8'), axis=1)
            cplx[:, :n // 2 + 1] = y * scale

    def rowfft_split_supported(self, n, elsize, nparts):
        return self.rowfft_supported(n, elsize) and 1 <= nparts <= _abi.PMX_MAXSEG

    def rowfft_split(self, elsize, inverse, src, dst, nrows, n, pitch, offsets, scale=1.0):
        """rows <-> the blocks of a pencil transform's first transpose (csrc/pmx_colfft.hip: pmx_rowfft_split)"""
        rdt, cdt = ('f4', 'c8') if elsize == 4 else ('f8', 'c16')
        M1 = n // 2 + 1
        e = [int(v) for v in offsets]
        if (not self.rowfft_split_supported(n, elsize, len(e) - 1) or e[0] != 0 or e[-1] != M1 or
                any(b < a for a, b in zip(e, e[1:])) or src.data_ptr() == dst.data_ptr()):
            raise backend.PmxError('pmx_rowfft_split', _abi.PMX_EINVAL, 'bad arguments')
        rows = (dst if inverse else src).detach().numpy().reshape(-1)
        blocks = (src if inverse else dst).detach().numpy().reshape(-1).view(cdt)
        real = rows.view(rdt)[:nrows * 2 * pitch].reshape(nrows, 2 * pitch)
        if inverse:
            X = numpy.empty((nrows, M1), dtype='c16')
            for a, b in zip(e, e[1:]):
                X[:, a:b] = blocks[nrows * a:nrows * b].reshape(nrows, b - a)
            real[:, :n] = numpy.fft.irfft(X, n=n, axis=1) * n * scale
        else:
            y = (numpy.fft.rfft(real[:, :n].astype('f8'), axis=1) * scale).astype(cdt)
            for a, b in zip(e, e[1:]):
                blocks[nrows * a:nrows * b] = y[:, a:b].reshape(-1)

    # ---- slab transposes (numpy restatement of csrc/pmx_fft.hip kernels) ----
    def slab_pack(self, src, dst, n0, n1, n2, n1_offsets, elbytes, inverse=False):
        cdt = 'c8' if elbytes == 8 else 'c16'
        full = (dst if inverse else src).detach().numpy().reshape(-1).view(cdt)[:n0 * n1 * n2].reshape(n0, n1, n2)
        blocks = (src if inverse else dst).detach().numpy().reshape(-1).view(cdt)
        for r in range(len(n1_offsets) - 1):
            a, b = int(n1_offsets[r]), int(n1_offsets[r + 1])
            blk = blocks[a * n0 * n2:b * n0 * n2].reshape(n0, b - a, n2)
            if inverse:
                full[:, a:b, :] = blk
            else:
                blk[...] = full[:, a:b, :]


def install():
    return backend.use(OracleBackend())
