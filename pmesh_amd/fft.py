"""Mesh partition, buffers and FFT plans: what ``pfft-python`` provides to
``pmesh/pm.py`` (pfft.ProcMesh / Partition / LocalBuffer / Plan; call sites
pm.py:226, 1321, 1332-1349, 1393-1441, 689, 1017), rebuilt for one process per GPU.

PFFT = serial FFTW per rank + MPI all-to-all transposes.  Here

* one rank: a single in-device 3-D (or 1-/2-D) rocFFT R2C / C2R on the padded
  in-place layout; the forward normalisation 1/prod(Nmesh) (pm.py:692) is the
  plan's scale factor, so no extra pass over the mesh;
* P ranks, slab decomposition (``np=[P]``): batched 2-D R2C over the local
  planes -> pack kernel (split axis 1 by destination) -> RCCL all-to-all over
  xGMI, received straight into the output buffer -> ONE strided batched 1-D C2C
  along axis 0.  The complex field comes out "transposed": axis 1 is
  distributed (PFFT's TRANSPOSED_OUT, the reference's default ``ComplexField``)
  and the local block (N0, n1_local, N2c) is plain C order, because the blocks
  that arrive from the P ranks are exactly its row ranges — no unpack pass.

Partitions are block distributions with block = ceil(N / P) (FFTW-MPI / PFFT
default): rank r owns [r*block, min((r+1)*block, N)).
"""
import os
import weakref
import warnings

import numpy
import torch

from . import _abi, backend
from ._arrays import torch_dtype


# 'auto': 3-d transforms whose axis lengths are 2^k up to 2048, 3 * 2^k up to 1536 or 5 * 2^k up to 1280 run as
# rocFFT (unit-stride R2C/C2R along the contiguous axis) + the LDS-resident column FFT of
# csrc/pmx_colfft.hip along the other two; 'never': everything through rocFFT.
COLFFT = 'auto'
# one-rank 3-d transforms: bytes of mesh planes whose row pass and axis-1 pass run back to back so
# that the second pass finds part of them in the 256 MiB Infinity Cache (0 = whole-array passes).
# Measured on MI355X (scripts/fft_block.sh, profiles/r02_fft_l3_block.txt): 512^3 fp64 r2c
# 1.38 -> 1.19 ms and c2r 1.32 -> 1.23 ms at 224-256 MB, nothing at 96 MB (a block is read AND
# written between its two uses, and launches of a few dozen planes fill the chip badly), a loss
# for meshes of several GB (768^3: +5 %; 1024^3: +15 %): applied to arrays up to L3_BLOCK_MAX_ARRAY.
L3_BLOCK_BYTES = int(os.environ.get('PMESH_AMD_L3_BLOCK_MB', '224')) << 20
L3_BLOCK_MAX_ARRAY = 3 << 29          # 1.5 GiB


def split_size_2d(s):
    """pfft.split_size_2d: the most square factorisation a*d = s with a <= d."""
    a = int(s ** 0.5) + 1
    d = s
    while a > 0:
        if s % a == 0:
            d = s // a
            break
        a = a - 1
    return a, d


def block_edges(n, p):
    blk = -(-n // p)
    e = [min(r * blk, n) for r in range(p + 1)]
    return numpy.array(e, dtype='intp')


class ProcMesh(object):
    """pfft.ProcMesh(np, comm): this rank's coordinates on the process mesh."""
    def __init__(self, np, comm):
        self.np = [int(x) for x in np]
        self.comm = comm
        size = 1
        for x in self.np:
            size *= x
        if size != comm.size:
            raise ValueError('process mesh %s does not match the communicator size %d' % (self.np, comm.size))
        self.this = numpy.unravel_index(comm.rank, self.np) if len(self.np) else ()
        self.rank = comm.rank
        self._sub = None

    def subcomms(self):
        """(row communicator: the P1 ranks sharing my p0, column communicator: the P0 ranks
        sharing my p1) of a 2-d process mesh; created once, collectively."""
        if self._sub is None:
            P0, P1 = self.np
            p0, p1 = [int(x) for x in self.this]
            rows = [[q0 * P1 + q1 for q1 in range(P1)] for q0 in range(P0)]
            cols = [[q0 * P1 + q1 for q0 in range(P0)] for q1 in range(P1)]
            groups = self.comm.subgroups(rows + cols)
            self._sub = (groups[p0], groups[P0 + p1])
        return self._sub


class Partition(object):
    """The local blocks of the real ("i") and complex ("o") meshes
    (pfft.Partition; attributes used by pm.py:237-242, 1185-1187, 1209-1211, 1450-1453)."""

    def __init__(self, Nmesh, procmesh, transposed, is_c2c=False, itemsize=8):
        self.Nmesh = numpy.array(Nmesh, dtype='intp')
        self.ndim = len(self.Nmesh)
        self.procmesh = procmesh
        self.transposed = transposed
        nd = self.ndim
        np_ = procmesh.np
        P = procmesh.comm.size
        self.is_c2c = bool(is_c2c)
        if nd == 1 and P > 1:
            raise ValueError("Running 1d transforms on multiple ranks is not supported")
        if nd > 4 and P > 1:
            # (4-d meshes on several ranks: slabs — the local stage is one batched 3-d rocFFT plan, Plan._execute_slab;
            # beyond that the local stage itself needs more than one plan: one rank only)
            raise NotImplementedError('meshes of more than 4 dimensions are transformed on one rank only')
        self.nproc = P
        # (a 2-d process mesh [P, 1] IS the slab: axis 0 of the real field distributed, axis 1 of the transposed spectrum;
        # [1, P] distributes axes 1 and 2, as PFFT lays it out: the pencil schedule with a first group of one)
        self.pencil = len(np_) == 2 and P > 1 and np_[1] > 1
        if self.pencil:
            if nd != 3:
                raise ValueError('a 2-d process mesh needs a 3-d mesh')
            self._init_pencil(np_, itemsize)
            return
        r = procmesh.comm.rank
        Nc = self.Nmesh.copy()
        if not is_c2c:
            Nc[-1] = Nc[-1] // 2 + 1           # r2c: the non-negative half of the last axis
        self.cshape_o = Nc
        # real space: axis 0 distributed
        self.i_edges = [block_edges(self.Nmesh[0], P)] + \
                       [numpy.array([0, n], dtype='intp') for n in self.Nmesh[1:]]
        self.local_i_start = numpy.array([self.i_edges[0][r]] + [0] * (nd - 1), dtype='intp')
        self.local_i_shape = numpy.array([self.i_edges[0][r + 1] - self.i_edges[0][r]] +
                                         list(self.Nmesh[1:]), dtype='intp')
        # complex space: transposed -> axis 1 distributed; untransposed -> axis 0
        if P == 1:
            oax = None
        elif transposed:
            oax = 1
        else:
            oax = 0
        self.o_axis = oax
        self.o_edges = [numpy.array([0, n], dtype='intp') for n in Nc]
        if oax is not None:
            self.o_edges[oax] = block_edges(Nc[oax], P)
        self.local_o_start = numpy.array([e[r] if d == oax else 0 for d, e in enumerate(self.o_edges)],
                                         dtype='intp')
        self.local_o_shape = numpy.array([(e[r + 1] - e[r]) if d == oax else e[-1]
                                          for d, e in enumerate(self.o_edges)], dtype='intp')
        # memory layout, in elements
        # complex rows: N2c modes; on one rank the row pitch of 3-d meshes is rounded up to a
        # whole number of 128-byte lines (8 complex128 / 16 complex64) so that every row of
        # the array, and every tile row of the column FFT and of the window kernels, starts
        # on a line boundary.  Consumers are stride agnostic (as in the reference, pm.py:97).
        pitch_c = int(Nc[-1])
        if is_c2c:
            # complex-to-complex meshes (ParticleMesh(dtype='c16')): configuration space holds
            # complex elements too; dense C-order on both sides, strides in complex elements
            self.pitch_c = self.pitch_i = pitch_c
            self.plane_c = None
            self.i_strides = _c_strides([int(x) for x in self.local_i_shape])
            self.i_alloc = int(numpy.prod(self.local_i_shape, dtype='i8'))
            self.o_strides = _c_strides([int(x) for x in self.local_o_shape])
            self.o_alloc = int(numpy.prod(self.local_o_shape, dtype='i8'))
            n0loc = int(self.local_i_shape[0])
            mid = n0loc * int(numpy.prod(Nc[1:], dtype='i8'))
            self.alloc_reals = max(2 * self.i_alloc, 2 * self.o_alloc, 2 * mid, 2)
            return
        if P == 1 and nd == 3:
            q = 128 // (2 * itemsize)
            pitch_c = -(-pitch_c // q) * q
        self.pitch_c = pitch_c
        # real: padded in place: the last axis has 2*pitch_c reals
        padded = list(self.local_i_shape)
        padded[-1] = 2 * pitch_c
        self.i_strides = _c_strides(padded)
        self.i_alloc = int(numpy.prod(padded, dtype='i8'))
        oshape_mem = [int(x) for x in self.local_o_shape]
        if pitch_c != int(Nc[-1]):
            oshape_mem[-1] = pitch_c
        self.o_strides = _c_strides(oshape_mem)
        self.o_alloc = int(numpy.prod(oshape_mem, dtype='i8'))
        # The axis-0 pass of the transform walks lines one plane apart.  A plane of N1 * pitch_c
        # elements that is a multiple of 8 KiB (2^16 x 33 bytes at 512^3) puts all lines of a
        # column on the same few HBM channels (562 us instead of 470 us per pass at 512^3,
        # scripts/stride_probe.py): one extra 128-byte line per plane breaks the pattern.
        # Only where the LDS row/column kernels (which take the plane stride) run the transform.
        self.plane_c = None
        if P == 1 and nd == 3 and PLANE_PAD and _own_kernel_lengths(self.Nmesh, itemsize):
            plane = int(self.Nmesh[1]) * pitch_c
            if (plane * 2 * itemsize) % 8192 == 0:
                plane += 128 // (2 * itemsize)
                self.plane_c = plane
                n0 = int(self.Nmesh[0])
                self.i_strides = [2 * plane, 2 * pitch_c, 1]
                self.i_alloc = n0 * 2 * plane
                self.o_strides = [plane, pitch_c, 1]
                self.o_alloc = n0 * plane
        # Several ranks, slab, lengths of the LDS kernels, equal power-of-two ranges of axis 1 (the
        # case where the pack / unpack ride on the column pass, Plan._execute_slab): the REAL side
        # keeps its rows on 128-byte boundaries (pitch_i complex per row; 257 -> 264 at 512^3:
        # row and axis-1 passes 57 -> 50 and 75 -> 58 us per rank at P = 8, scripts/pitch_probe.py)
        # while the transposed complex side and the wire format stay dense.
        self.pitch_i = pitch_c
        if (P > 1 and nd == 3 and PLANE_PAD and _own_kernel_lengths(self.Nmesh, itemsize)):
            n1 = int(self.Nmesh[1])
            n1loc = n1 // P
            if n1loc * P == n1 and n1loc >= 1 and n1loc & (n1loc - 1) == 0:
                q = 128 // (2 * itemsize)
                self.pitch_i = -(-int(Nc[-1]) // q) * q
                padded = list(self.local_i_shape)
                padded[-1] = 2 * self.pitch_i
                self.i_strides = _c_strides(padded)
                self.i_alloc = int(numpy.prod(padded, dtype='i8'))
        # one buffer serves both views (in-place transforms)
        self.alloc_reals = max(self.i_alloc, 2 * self.o_alloc, 2)
        if P > 1:
            # the slab transpose needs the full local plane set on both sides
            n0loc = int(self.local_i_shape[0])
            mid = n0loc * int(numpy.prod(Nc[1:-1], dtype='i8')) * max(int(Nc[-1]), self.pitch_i)
            self.alloc_reals = max(self.alloc_reals, 2 * mid)


#: number of chunks of the last axis the slab transposes are pipelined over (the all-to-all of
#: a chunk runs under the column passes of its neighbours); 1: one exchange per transform
OVERLAP_CHUNKS = int(os.environ.get('PMESH_AMD_OVERLAP_CHUNKS', '2'))
#: ... but never chunks of fewer bytes than this per rank (of the local complex block): what a chunk buys is its share of
#: the wire time hidden under its neighbours' kernels, what it costs is a set of launches that fill the device badly —
#: at 512^3 on 8 ranks a chunk of 67 MB runs its column passes in 36-55 us per launch (profiles/r03_h_multirank8_*:
#: 8 % more kernel time than one launch for the whole block) against ~100 us of wire per chunk; a quarter of that is
#: where the launches cost what they hide.  (The rate of a link is not measured here: one GPU.  PMESH_AMD_OVERLAP_MIN_MB.)
OVERLAP_MIN_CHUNK_BYTES = int(float(os.environ.get('PMESH_AMD_OVERLAP_MIN_MB', '16')) * (1 << 20))


def _overlap_chunks(local_bytes):
    """how many chunks a transpose of `local_bytes` per rank is pipelined over"""
    C = int(OVERLAP_CHUNKS)
    if OVERLAP_MIN_CHUNK_BYTES > 0:
        C = min(C, int(local_bytes // OVERLAP_MIN_CHUNK_BYTES))
    return C


def _async_exchange_works(comm):
    """one tiny asynchronous all-to-all per communicator, the first time a transform wants to
    pipeline its transposes: a backend that cannot do it (raises) switches the pipeline off for
    that communicator instead of failing the transform.  Collective: every rank calls it."""
    ok = getattr(comm, '_pmx_async_ok', None)
    if ok is None:
        try:
            dev = backend.get().device
            a = torch.arange(comm.size, dtype=torch.float64, device=dev)
            b = torch.empty_like(a)
            comm.alltoall(a, b, async_op=True).wait()
            ok = True
        except Exception as ex:          # noqa
            warnings.warn('asynchronous all-to-all is not available (%r): transposes are not pipelined' % (ex,))
            ok = False
        try:
            comm._pmx_async_ok = ok
        except Exception:
            pass
    return ok

#: pad the plane stride of the one-rank 3-d layout (see Partition); False: dense planes
PLANE_PAD = True


def _col_length_ok(n, itemsize):
    """pmx_colfft_supported: powers of two 64..2048, 3 * 2^k in 192..1536 (not 1536 in fp32),
    5 * 2^k in 320..1280 (not 1280 in fp32)"""
    n = int(n)
    if n > 0 and n & (n - 1) == 0:
        return 64 <= n <= 2048
    if n % 3 == 0 and (n // 3) & (n // 3 - 1) == 0:
        return 192 <= n <= 1536 and not (n == 1536 and itemsize == 4)
    if n % 5 == 0 and (n // 5) & (n // 5 - 1) == 0:
        return 320 <= n <= 1280 and not (n == 1280 and itemsize == 4)
    return False


def _row_length_ok(n):
    """pmx_rowfft_supported: powers of two 128..2048, or 384 / 768 / 1536 / 640 / 1280"""
    n = int(n)
    if n > 0 and n & (n - 1) == 0:
        return 128 <= n <= 2048
    return n in (384, 768, 1536, 640, 1280)


def _own_kernel_lengths(Nmesh, itemsize):
    """True if csrc/pmx_colfft.hip runs every stage of a 3-d transform of this mesh (the
    arithmetic of pmx_rowfft_supported / pmx_colfft_supported, needed here without a backend)"""
    n0, n1, n2 = [int(x) for x in Nmesh]
    return _col_length_ok(n0, itemsize) and _col_length_ok(n1, itemsize) and _row_length_ok(n2) and n1 % 16 == 0


def _pencil_init(self, np_, itemsize):
    """2-d process mesh (P0, P1), rank = p0*P1 + p1 (C order).
    real    : (N0/P0, N1/P1, N2)        axes 0, 1 distributed
    complex : (N0, N1/P0, N2c/P1)       axes 1, 2 distributed ("transposed out", PFFT's layout
              for 3-d r2c on a 2-d process mesh), local block in C order."""
    P0, P1 = int(np_[0]), int(np_[1])
    p0, p1 = [int(x) for x in self.procmesh.this]
    N0, N1, N2 = [int(x) for x in self.Nmesh]
    # complex-to-complex meshes (pm.py:1270): configuration space is complex too, the whole last axis is kept
    N2c = N2 if self.is_c2c else N2 // 2 + 1
    self.P0, self.P1, self.p0, self.p1 = P0, P1, p0, p1
    self.cshape_o = numpy.array([N0, N1, N2c], dtype='intp')
    self.i_edges = [block_edges(N0, P0), block_edges(N1, P1), numpy.array([0, N2], dtype='intp')]
    self.local_i_start = numpy.array([self.i_edges[0][p0], self.i_edges[1][p1], 0], dtype='intp')
    self.local_i_shape = numpy.array([self.i_edges[0][p0 + 1] - self.i_edges[0][p0],
                                      self.i_edges[1][p1 + 1] - self.i_edges[1][p1], N2], dtype='intp')
    self.o_axis = None
    self.o_edges = [numpy.array([0, N0], dtype='intp'), block_edges(N1, P0), block_edges(N2c, P1)]
    self.local_o_start = numpy.array([0, self.o_edges[1][p0], self.o_edges[2][p1]], dtype='intp')
    self.local_o_shape = numpy.array([N0, self.o_edges[1][p0 + 1] - self.o_edges[1][p0],
                                      self.o_edges[2][p1 + 1] - self.o_edges[2][p1]], dtype='intp')
    self.pitch_c = self.pitch_i = N2c
    n0l, n1l = int(self.local_i_shape[0]), int(self.local_i_shape[1])
    if self.is_c2c:
        self.i_strides = _c_strides([n0l, n1l, N2c])           # in complex elements
        self.i_alloc = n0l * n1l * N2c
    else:
        padded = [n0l, n1l, 2 * N2c]
        self.i_strides = _c_strides(padded)
        self.i_alloc = n0l * n1l * 2 * N2c
    tshape = [int(x) for x in self.local_o_shape]          # the transposed block (N0, m1, m2)
    mid = n0l * N1 * tshape[2]                             # (n0loc, N1, n2cloc) between the transposes
    self.t_alloc = int(numpy.prod(tshape, dtype='i8'))
    if not self.transposed:
        # the untransposed complex field is distributed like the real one (PFFT without
        # PFFT_TRANSPOSED_OUT on a 2-d process mesh): (N0 / P0, N1 / P1, N2c), dense
        self.o_edges = [self.i_edges[0], self.i_edges[1], numpy.array([0, N2c], dtype='intp')]
        self.local_o_start = numpy.array([self.i_edges[0][p0], self.i_edges[1][p1], 0], dtype='intp')
        self.local_o_shape = numpy.array([n0l, n1l, N2c], dtype='intp')
    oshape = [int(x) for x in self.local_o_shape]
    self.o_strides = _c_strides(oshape)
    self.o_alloc = int(numpy.prod(oshape, dtype='i8'))
    ireals = 2 * self.i_alloc if self.is_c2c else self.i_alloc
    self.alloc_reals = max(ireals, 2 * self.o_alloc, 2 * self.t_alloc, 2 * mid, 2)


Partition._init_pencil = _pencil_init


def _c_strides(shape):
    s = [1] * len(shape)
    for d in range(len(shape) - 2, -1, -1):
        s[d] = s[d + 1] * int(shape[d + 1])
    return s


#: r2c -> [transfer] -> c2r back to back: the last pass of the forward transform (axis 0) is DEFERRED — the plan
#: leaves a note on the buffer's storage instead of running it — and an in-place c2r that finds the note runs ONE
#: kernel for both axis-0 passes (pmx_colfft_roundtrip: forward, scale, transfer, inverse with the column in LDS):
#: one sweep of the array less per cycle.  Whoever else looks at the spectrum first (Field.value, any other plan,
#: cast, apply with a callable ...) runs the deferred pass then (`settle`): the values are those of the eager
#: transform either way, bit for bit.  One rank, the LDS column kernels (Plan._execute_local_hybrid).
DEFER_LAST_PASS = os.environ.get('PMESH_AMD_DEFER_LAST_PASS', '1') not in ('0', '', 'false')
# pencil transforms: the last-axis split of the first transpose rides on the row pass (pmx_rowfft_split); 0 = the row
# pass and pmx_slab_pack as two sweeps (A/B and the parity tests)
ROW_SPLIT = os.environ.get('PMESH_AMD_ROW_SPLIT', '1') not in ('0', '', 'false')


class _Pending(object):
    """the deferred axis-0 pass of a forward transform.  kind: 'local' (one rank), 'slab' (several ranks, one
    exchange: the pass is due on the received block), 'slabpipe' (pipelined exchange: `data` holds the chunk
    buffers the all-to-alls deliver into and their handles)"""
    __slots__ = ('partition', 'run', 'fused', 'kind', 'data')

    def __init__(self, partition, run, fused, kind='local', data=None):
        self.partition, self.run, self.fused, self.kind, self.data = partition, run, fused, kind, data


def settle(storage):
    """run what is deferred on this storage now (no-op if there is nothing): the halo merge a paint left to the
    forward transform (window._HaloDebt), the last pass of a forward transform"""
    debt = getattr(storage, '_pmx_halo', None)
    if debt is not None:
        storage._pmx_halo = None
        debt.settle()
    pend = getattr(storage, '_pmx_pending', None)
    if pend is not None:
        storage._pmx_pending = None
        pend.run()


def forget(storage):
    """the storage is about to be overwritten as a whole: a deferred pass on it is moot"""
    if getattr(storage, '_pmx_pending', None) is not None:
        storage._pmx_pending = None
    debt = getattr(storage, '_pmx_halo', None)
    if debt is not None:
        storage._pmx_halo = None
        debt.drop()


class LocalBuffer(object):
    """pfft.LocalBuffer(partition, base=None): device storage with a real
    ("input") and a complex ("output") view (pm.py:226, 236, 240).  `a in b` tests
    aliasing as the reference does (pm.py:677)."""

    def __init__(self, partition, dtype, base=None):
        self.partition = partition
        self.rdtype = torch_dtype(dtype)
        if base is None:
            be = backend.get()
            self.storage = torch.zeros(partition.alloc_reals, dtype=self.rdtype, device=be.device)
        else:
            self.storage = base.storage if isinstance(base, LocalBuffer) else base
            if self.storage.numel() < partition.alloc_reals:
                raise ValueError('base buffer is too small for this partition')

    def __contains__(self, other):
        return isinstance(other, LocalBuffer) and other.storage.data_ptr() == self.storage.data_ptr()

    def settle(self):
        settle(self.storage)

    def view_input(self):
        p = self.partition
        shape = getattr(p, '_i_shape_list', None)
        if shape is None:
            shape = p._i_shape_list = [int(x) for x in p.local_i_shape]
        if getattr(p, 'is_c2c', False):
            return torch.as_strided(torch.view_as_complex(self.storage.view(-1, 2)), shape, p.i_strides)
        return torch.as_strided(self.storage, shape, p.i_strides)

    def view_output(self):
        p = self.partition
        shape = getattr(p, '_o_shape_list', None)
        if shape is None:
            shape = p._o_shape_list = [int(x) for x in p.local_o_shape]
        return torch.as_strided(torch.view_as_complex(self.storage.view(-1, 2)), shape, p.o_strides)

    def view_raw(self):
        return self.storage


class _Storage(object):
    """a bare work buffer with the one attribute Plan.execute reads from a LocalBuffer"""
    def __init__(self, storage):
        self.storage = storage


class Plan(object):
    """pfft.Plan(...).execute(bufin, bufout) for one direction."""

    _shared_gpu_told = False

    def __init__(self, partition, forward, dtype, inplace):
        self.partition = partition
        self.forward = forward
        self.inplace = inplace
        self.elsize = numpy.dtype(dtype).itemsize
        if partition.nproc > 1 and not Plan._shared_gpu_told:
            # transforms on several ranks overlap with RCCL's kernels, which hold compute units for the length of a
            # transfer: the column passes then run one workgroup per tile instead of a persistent one per CU, whose
            # fixed share of the tiles would wait for a free CU as a whole (pmx_colfft_configure)
            Plan._shared_gpu_told = True
            be = backend.get()
            if hasattr(be, 'colfft_configure'):
                be.colfft_configure(0)
        self._plans = {}
        self._work = None
        self.sibling = None       # untransposed plans: the plan of the transposed partition

    # lazily created native plans (rocFFT kernels are built on first use)
    def _native(self, key, maker):
        if key not in self._plans:
            self._plans[key] = maker()
        return self._plans[key]

    def execute(self, bufin, bufout, transfer=None):
        p = self.partition
        # a forward plan whose last pass is still deferred has its work buffers lent out (pipelined slabs)
        owed = getattr(self, '_deferred_on', None)
        if owed is not None:
            self._deferred_on = None
            settle(owed)
        debt = getattr(bufin.storage, '_pmx_halo', None)
        if debt is not None:
            # the halo merge of the paint that made this field: the forward transform of one rank adds the staged
            # halos inside its row pass (_execute_local_hybrid); anything else needs the finished mesh first
            mine = (self.forward and p.nproc == 1 and not getattr(p, 'is_c2c', False) and p.ndim == 3 and
                    not (getattr(p, 'pencil', False) and not p.transposed))
            # slab ranks: the row pass of the local stage does the same for the block of planes (_slab_row_forward)
            slab = (self.forward and p.nproc > 1 and not getattr(p, 'is_c2c', False) and p.ndim == 3 and
                    not getattr(p, 'pencil', False) and p.transposed and bufin.storage is bufout.storage)
            if not (mine or slab):
                settle(bufin.storage)
        pend = getattr(bufin.storage, '_pmx_pending', None)
        if pend is not None:
            # a deferred forward pass on the input: an in-place inverse transform of the same partition takes it
            # over (one rank: _execute_local_hybrid; slabs: _execute_slab); anything else needs the finished
            # spectrum first
            mine = (not self.forward and pend.partition is p and bufin.storage is bufout.storage and
                    not getattr(p, 'is_c2c', False) and
                    (pend.kind == 'local') == (p.nproc == 1))
            if not mine:
                settle(bufin.storage)
        if bufout.storage is not bufin.storage:
            forget(bufout.storage)
        if getattr(p, 'pencil', False) and not p.transposed:
            if transfer is not None:
                raise NotImplementedError('fused transfer needs the transposed complex field')
            return self._execute_pencil_untransposed(bufin, bufout)
        if getattr(p, 'is_c2c', False):
            if transfer is not None:
                raise NotImplementedError('fused transfer on a complex-to-complex mesh')
            if getattr(p, 'pencil', False):
                return self._execute_pencil(bufin, bufout, None)
            if p.nproc > 1 and not p.transposed:
                return self._execute_slab_untransposed(bufin, bufout)
            return self._execute_c2c(bufin, bufout)
        if p.nproc == 1:
            self._execute_local(bufin, bufout, transfer)
        else:
            if getattr(p, 'pencil', False):
                self._execute_pencil(bufin, bufout, transfer)
            else:
                self._execute_slab(bufin, bufout, transfer)

    # ---- single stages on (A, N, B) / row arrays: own LDS kernels, rocFFT otherwise ----
    def _row(self, be, buf, nrows, n, pitch, inverse):
        """in-place r2c (inverse False) / c2r of nrows rows of n reals at `pitch` complex"""
        if nrows == 0:
            return
        if COLFFT != 'never' and hasattr(be, 'rowfft') and be.rowfft_supported(n, self.elsize):
            be.rowfft(self.elsize, inverse, buf, nrows, n, pitch)
            return

        def make():
            if inverse:
                return be.fft_create(_abi.PMX_FFT_C2R, self.elsize, [n], [1], pitch, [1], 2 * pitch,
                                     nrows, 1.0, True)
            return be.fft_create(_abi.PMX_FFT_R2C, self.elsize, [n], [1], 2 * pitch, [1], pitch,
                                 nrows, 1.0, True)
        be.fft_execute(self._native(('row', inverse, nrows, n, pitch), make), buf, buf)

    def _col(self, be, buf, A, N, B, inverse, scale=1.0):
        """in-place C2C along the middle axis of the (A, N, B) complex array in buf"""
        if A == 0 or B == 0:
            return
        if COLFFT != 'never' and hasattr(be, 'colfft') and be.colfft_supported(N, self.elsize):
            be.colfft(self.elsize, inverse, buf, A, N, B, scale=scale)
            return
        kind = _abi.PMX_FFT_C2C_BWD if inverse else _abi.PMX_FFT_C2C_FWD

        def make():
            return be.fft_create(kind, self.elsize, [N], [B], 1, [B], 1, B, scale, True)
        plan = self._native(('col', inverse, N, B, scale), make)
        for a in range(A):
            sub = buf[2 * a * N * B:]
            be.fft_execute(plan, sub, sub)

    def _execute_pencil(self, bufin, bufout, transfer=None):
        """Pencil-decomposed 3-d transform: two transposes, on the row and the column
        sub-communicators of the (P0, P1) process mesh (PFFT's scheme for np=[P0, P1])."""
        be = backend.get()
        p = self.partition
        if getattr(p, '_unsupported', None):
            raise NotImplementedError(p._unsupported)
        if transfer is not None and (self.forward or not self.can_fuse()):
            raise NotImplementedError('fused transfer needs the column-FFT first stage of c2r')
        rowc, colc = p.procmesh.subcomms()
        P0, P1 = p.P0, p.P1
        N0, N1, N2 = [int(x) for x in p.Nmesh]
        c2c = bool(getattr(p, 'is_c2c', False))
        N2c = N2 if c2c else N2 // 2 + 1
        n0l, n1l = int(p.local_i_shape[0]), int(p.local_i_shape[1])
        m1, m2 = int(p.local_o_shape[1]), int(p.local_o_shape[2])      # complex local extents
        if c2c:
            # the row stage of a complex-to-complex mesh: C2C along the contiguous axis (rocFFT), in place
            def row(buf, nrows, inverse):
                if nrows == 0:
                    return
                kind = _abi.PMX_FFT_C2C_BWD if inverse else _abi.PMX_FFT_C2C_FWD
                plan = self._native(('crow', inverse, nrows, N2),
                                    lambda: be.fft_create(kind, self.elsize, [N2], [1], N2, [1], N2, nrows, 1.0, True))
                be.fft_execute(plan, buf, buf)
        else:
            def row(buf, nrows, inverse):
                self._row(be, buf, nrows, N2, N2c, inverse)
        e1i = [int(x) for x in p.i_edges[1]]      # axis 1 by P1 (real side)
        e0i = [int(x) for x in p.i_edges[0]]      # axis 0 by P0
        e1o = [int(x) for x in p.o_edges[1]]      # axis 1 by P0 (complex side)
        e2o = [int(x) for x in p.o_edges[2]]      # axis 2 by P1
        elb = 2 * self.elsize
        norm = 1.0 / float(N0) / float(N1) / float(N2)
        rdt = bufin.storage.dtype
        need = max(2 * n0l * n1l * N2c, 2 * n0l * N1 * m2, 2 * N0 * m1 * m2, 2)
        if self._work is None or self._work[0].numel() < need or self._work[0].dtype != rdt:
            self._work = [torch.empty(need, dtype=rdt, device=bufin.storage.device) for _ in range(3)]
        W0, W1, W2 = self._work
        same = bufin.storage.data_ptr() == bufout.storage.data_ptr()
        # transpose 1 (row group, P1 ranks): (n0l, n1l, N2c) <-> (n0l, N1, m2)
        s1 = [2 * n0l * n1l * (e2o[q + 1] - e2o[q]) for q in range(P1)]
        r1 = [2 * n0l * (e1i[q + 1] - e1i[q]) * m2 for q in range(P1)]
        # transpose 2 (column group, P0 ranks): (n0l, N1, m2) <-> (N0, m1, m2)
        s2 = [2 * n0l * (e1o[q + 1] - e1o[q]) * m2 for q in range(P0)]
        r2 = [2 * (e0i[q + 1] - e0i[q]) * m1 * m2 for q in range(P0)]
        # equal power-of-two ranges of axis 1 in both directions of the process mesh: the unpack
        # before and the pack after the axis-1 pass ride on the column kernel (pmx_colfft_resplit)
        fuse1 = (COLFFT != 'never' and hasattr(be, 'colfft_resplit') and be.colfft_supported(N1, self.elsize) and
                 n1l * P1 == N1 and m1 * P0 == N1 and n1l & (n1l - 1) == 0 and m1 & (m1 - 1) == 0 and
                 all(e1i[q + 1] - e1i[q] == n1l for q in range(P1)) and
                 all(e1o[q + 1] - e1o[q] == m1 for q in range(P0)))
        planes = None if c2c else self._plane_chunks(p, rowc, colc, fuse1, n0l, N0, P0)
        # the last-axis split on the row pass itself (pmx_rowfft_split: rows -> the blocks of the first transpose and
        # back, no pack sweep and no copy of an input the caller keeps)
        split = (not c2c and ROW_SPLIT and COLFFT != 'never' and hasattr(be, 'rowfft_split') and
                 be.rowfft_split_supported(N2, self.elsize, P1))
        if planes:
            return self._execute_pencil_pipelined(be, bufin, bufout, transfer, planes, same, rowc, colc,
                                                  N0, N1, N2, N2c, n0l, n1l, m1, m2, e2o, e0i, norm, W0, W1, W2, split)
        if self.forward:
            if split:
                be.rowfft_split(self.elsize, False, bufin.storage, W1, n0l * n1l, N2, N2c, e2o)
            else:
                X = bufin.storage
                if not same:
                    nreal = n0l * n1l * 2 * N2c
                    W0[:nreal].copy_(bufin.storage[:nreal])
                    X = W0
                row(X, n0l * n1l, False)
                be.slab_pack(X, W1, n0l * n1l, N2c, 1, e2o, elb)                 # split the last axis
            rowc.alltoall(W1[:sum(s1)], W2[:sum(r1)], s1, r1)
            if fuse1:
                # unpack, axis-1 pass and pack in one kernel: split by P1 in, split by P0 out
                if n0l and m2:
                    be.colfft_resplit(self.elsize, False, W2, W1, n0l, N1, m2, n1l, m1)
            else:
                Y = W0
                be.slab_pack(W2, Y, n0l, N1, m2, e1i, elb, inverse=True)    # blocks -> (n0l, N1, m2)
                self._col(be, Y, n0l, N1, m2, False)
                be.slab_pack(Y, W1, n0l, N1, m2, e1o, elb)                  # split axis 1 by P0
            out = bufout.storage
            colc.alltoall(W1[:sum(s2)], out[:sum(r2)], s2, r2)              # row ranges of (N0, m1, m2)
            if c2c:
                self._col(be, out, 1, N0, m1 * m2, False, scale=norm)
            else:
                self._last_pencil_pass(be, out, N0, m1, m2, norm)
        else:
            S = self._first_pencil_pass(be, bufin, same, transfer, W0, N0, m1, m2)
            colc.alltoall(S[:sum(r2)], W1[:sum(s2)], r2, s2)
            if fuse1:
                if n0l and m2:
                    be.colfft_resplit(self.elsize, True, W1, W2, n0l, N1, m2, m1, n1l)
                W1, W2 = W2, W1
            else:
                Y = W2
                be.slab_pack(W1, Y, n0l, N1, m2, e1o, elb, inverse=True)
                self._col(be, Y, n0l, N1, m2, True)
                be.slab_pack(Y, W1, n0l, N1, m2, e1i, elb)
            Z = W0
            rowc.alltoall(W1[:sum(r1)], Z[:sum(s1)], r1, s1)
            out = bufout.storage
            if split:
                be.rowfft_split(self.elsize, True, Z, out, n0l * n1l, N2, N2c, e2o)
            else:
                be.slab_pack(Z, out, n0l * n1l, N2c, 1, e2o, elb, inverse=True)
                row(out, n0l * n1l, True)

    def _execute_pencil_untransposed(self, bufin, bufout, mode=None):
        """The untransposed complex layout on a 2-d process mesh — (N0 / P0, N1 / P1, N2c), distributed like
        the real field: what PFFT delivers without PFFT_TRANSPOSED_OUT, and what the reference builds next
        to the transposed plans on any process mesh (pm.py:1332-1349).  It is the transposed transform
        (self.sibling) plus the two global transposes back, without the FFT stages in between:
        (N0, m1, m2) <-> column group <-> (n0l, N1, m2) <-> row group <-> (n0l, n1l, N2c).
        mode 'T->U' / 'U->T': the layout change alone (Field.cast)."""
        be = backend.get()
        p = self.partition
        t = self.sibling
        if t is None:
            raise NotImplementedError('untransposed plan without its transposed sibling')
        pt = t.partition
        rowc, colc = p.procmesh.subcomms()
        P0, P1 = p.P0, p.P1
        N0, N1 = int(p.Nmesh[0]), int(p.Nmesh[1])
        N2c = int(p.pitch_c)
        n0l, n1l = int(p.local_i_shape[0]), int(p.local_i_shape[1])
        m1, m2 = int(pt.local_o_shape[1]), int(pt.local_o_shape[2])
        e1i = [int(x) for x in pt.i_edges[1]]
        e0i = [int(x) for x in pt.i_edges[0]]
        e1o = [int(x) for x in pt.o_edges[1]]
        e2o = [int(x) for x in pt.o_edges[2]]
        elb = 2 * self.elsize
        rdt = bufin.storage.dtype
        dev = bufin.storage.device
        need = max(int(pt.alloc_reals), 2 * n0l * N1 * m2, 2 * n0l * n1l * N2c, 2 * N0 * m1 * m2, 2)
        if self._work is None or self._work[0].numel() < need or self._work[0].dtype != rdt:
            self._work = [torch.empty(need, dtype=rdt, device=dev) for _ in range(3)]
        TB, WA, WB = self._work
        tb = _Storage(TB)
        s1 = [2 * n0l * n1l * (e2o[q + 1] - e2o[q]) for q in range(P1)]       # row group, by last-axis range
        r1 = [2 * n0l * (e1i[q + 1] - e1i[q]) * m2 for q in range(P1)]        # ... by axis-1 range
        s2 = [2 * n0l * (e1o[q + 1] - e1o[q]) * m2 for q in range(P0)]        # column group, by axis-1 range
        r2 = [2 * (e0i[q + 1] - e0i[q]) * m1 * m2 for q in range(P0)]         # ... by row range (contiguous)

        def t_to_u(src, dst):
            colc.alltoall(src[:sum(r2)], WA[:sum(s2)], r2, s2)
            be.slab_pack(WA, WB, n0l, N1, m2, e1o, elb, inverse=True)          # blocks -> (n0l, N1, m2)
            be.slab_pack(WB, WA, n0l, N1, m2, e1i, elb)                        # split axis 1 by P1
            rowc.alltoall(WA[:sum(r1)], WB[:sum(s1)], r1, s1)
            be.slab_pack(WB, dst, n0l * n1l, N2c, 1, e2o, elb, inverse=True)   # blocks -> (n0l n1l, N2c)

        def u_to_t(src, dst):
            be.slab_pack(src, WA, n0l * n1l, N2c, 1, e2o, elb)
            rowc.alltoall(WA[:sum(s1)], WB[:sum(r1)], s1, r1)
            be.slab_pack(WB, WA, n0l, N1, m2, e1i, elb, inverse=True)
            be.slab_pack(WA, WB, n0l, N1, m2, e1o, elb)
            colc.alltoall(WB[:sum(s2)], dst[:sum(r2)], s2, r2)

        if mode == 'T->U':
            t_to_u(bufin.storage, bufout.storage)
        elif mode == 'U->T':
            u_to_t(bufin.storage, bufout.storage)
        elif self.forward:
            t.execute(bufin, tb)
            settle(TB)                     # (the transposed plan may have left its last pass as a note)
            t_to_u(TB, bufout.storage)
        else:
            forget(TB)
            u_to_t(bufin.storage, TB)
            t.execute(tb, bufout)

    def _plane_chunks(self, p, rowc, colc, fuse1, n0l, N0, P0):
        """[(first plane, planes)] of the local axis-0 range if both transposes of the pencil transform
        can be pipelined over chunks of planes: the fused axis-1 pass, equal plane ranges on all ranks
        (the chunk boundaries must agree) and asynchronous exchanges on both sub-communicators"""
        # (every rank must come to the same answer — the probe below and the chunked exchanges are collective: the size
        # that decides is the mean block, not this rank's own, which differs from rank to rank on uneven meshes)
        C = _overlap_chunks(2 * self.elsize * int(numpy.prod(p.cshape_o, dtype='i8')) // max(1, int(p.nproc)))
        if C < 2 or not fuse1 or n0l * P0 != N0 or n0l < 2 * C:
            return None
        if not (hasattr(rowc, 'alltoall_views') and hasattr(colc, 'alltoall_views')):
            return None
        if not (_async_exchange_works(rowc) and _async_exchange_works(colc)):
            return None
        w = -(-n0l // C)
        out, a = [], 0
        while a < n0l:
            out.append((a, min(w, n0l - a)))
            a += w
        return out if len(out) > 1 else None

    def _execute_pencil_pipelined(self, be, bufin, bufout, transfer, planes, same, rowc, colc,
                                  N0, N1, N2, N2c, n0l, n1l, m1, m2, e2o, e0i, norm, W0, W1, W2, split=False):
        """The pencil transform with BOTH global transposes cut into chunks of the local planes (axis 0 of
        the real side, which neither the row transform, nor the first transpose, nor the axis-1 pass mixes):
        chunk c is row-transformed and packed while the first transpose of chunk c - 1 is on the wire
        (RCCL's stream), its axis-1 pass (pmx_colfft_resplit: unpack, transform, pack in one kernel) runs
        under the first transpose of chunk c + 1, and its second transpose under the axis-1 pass of
        chunk c + 1.  The second transpose delivers chunk c of rank s as the planes [e0i[s] + a, e0i[s] + b)
        of the (N0, m1, m2) block: row ranges that are contiguous one by one (comm.alltoall_views), so nothing
        is copied; the axis-0 pass runs once everything has arrived.  PFFT (pm.py:1417-1434) has no such
        overlap.  c2r mirrors it.  Same numbers as the single exchanges
        (tests/mp_cases.py::case_pencil_pipelined_equals_single_exchange)."""
        es = self.elsize
        elb = 2 * es
        P0, P1 = len(e0i) - 1, len(e2o) - 1
        # per chunk: where its pieces live in the work buffers (reals)
        o1, o2, acc1, acc2 = [], [], 0, 0
        for a, n in planes:
            o1.append(acc1)
            o2.append(acc2)
            acc1 += 2 * n * n1l * N2c          # row side: (n, n1l, N2c) = the sum of its P1 last-axis blocks
            acc2 += 2 * n * N1 * m2            # between the transposes: (n, N1, m2)
        t1s = lambda n: [2 * n * n1l * (e2o[q + 1] - e2o[q]) for q in range(P1)]
        t1r = lambda n: [2 * n * n1l * m2] * P1
        out = bufout.storage
        if self.forward:
            X = bufin.storage
            if not same and not split:
                nreal = n0l * n1l * 2 * N2c
                W0[:nreal].copy_(bufin.storage[:nreal])
                X = W0
            w1 = []
            for (a, n), q1, q2 in zip(planes, o1, o2):
                rows = X[2 * a * n1l * N2c:]
                if split:
                    be.rowfft_split(es, False, rows, W1[q1:], n * n1l, N2, N2c, e2o)
                else:
                    self._row(be, rows, n * n1l, N2, N2c, False)
                    be.slab_pack(rows, W1[q1:], n * n1l, N2c, 1, e2o, elb)          # split the last axis
                w1.append(rowc.alltoall(W1[q1:q1 + 2 * n * n1l * N2c], W2[q2:q2 + 2 * n * N1 * m2], t1s(n), t1r(n),
                                        async_op=True))
            w2 = []
            # the packed output of the axis-1 pass goes to W0: the row side (X, which may be W0) is only read by
            # the packs above, and those are all enqueued — hence executed — before the first of these passes
            Y = W0
            for (a, n), q2, w in zip(planes, o2, w1):
                w.wait()
                if m2:
                    be.colfft_resplit(es, False, W2[q2:], Y[q2:], n, N1, m2, n1l, m1)
                blk = 2 * n * m1 * m2
                send = [Y[q2 + s * blk:q2 + (s + 1) * blk] for s in range(P0)]
                recv = [out[2 * (e0i[s] + a) * m1 * m2:2 * (e0i[s] + a + n) * m1 * m2] for s in range(P0)]
                w2.append(colc.alltoall_views(send, recv, async_op=True))
            for w in w2:
                w.wait()
            self._last_pencil_pass(be, out, N0, m1, m2, norm)
        else:
            S = self._first_pencil_pass(be, bufin, same, transfer, W0, N0, m1, m2)
            w2 = []
            for (a, n), q2 in zip(planes, o2):
                blk = 2 * n * m1 * m2
                send = [S[2 * (e0i[s] + a) * m1 * m2:2 * (e0i[s] + a + n) * m1 * m2] for s in range(P0)]
                recv = [W1[q2 + s * blk:q2 + (s + 1) * blk] for s in range(P0)]
                w2.append(colc.alltoall_views(send, recv, async_op=True))
            # where the first transpose (row group) delivers: W0 when the transform works in place (S is the
            # caller's buffer), else — S is W0 — a buffer of this plan's own
            if S is W0:
                if getattr(self, '_work_z', None) is None or self._work_z.numel() < acc1 or self._work_z.dtype != W0.dtype:
                    self._work_z = torch.empty(acc1, dtype=W0.dtype, device=W0.device)
                Z = self._work_z
            else:
                Z = W0
            w1 = []
            for (a, n), q1, q2, w in zip(planes, o1, o2, w2):
                w.wait()
                if m2:
                    be.colfft_resplit(es, True, W1[q2:], W2[q2:], n, N1, m2, m1, n1l)
                w1.append(rowc.alltoall(W2[q2:q2 + 2 * n * N1 * m2], Z[q1:q1 + 2 * n * n1l * N2c], t1r(n), t1s(n),
                                        async_op=True))
            # (every second transpose has been waited for by now: `out`, which is S when the transform works
            # in place, may be overwritten)
            for (a, n), q1, w in zip(planes, o1, w1):
                w.wait()
                rows = out[2 * a * n1l * N2c:]
                if split:
                    be.rowfft_split(es, True, Z[q1:], rows, n * n1l, N2, N2c, e2o)
                else:
                    be.slab_pack(Z[q1:], rows, n * n1l, N2c, 1, e2o, elb, inverse=True)
                    self._row(be, rows, n * n1l, N2, N2c, True)

    def fills_output(self):
        """True if an out-of-place execute(bufin, bufout) writes every element of `bufout` that any later reader looks
        at, reading `bufin` only in its first pass (one rank, the LDS row / column kernels with their out-of-place
        first pass): the caller may then hand in uninitialised memory and need not copy the input first"""
        p = self.partition
        if p.nproc != 1 or p.ndim != 3 or getattr(p, 'is_c2c', False) or not p.plane_c:
            return False
        if getattr(p, 'pencil', False) and not p.transposed:
            return False
        be = backend.get()
        return (hasattr(be, 'rowfft_to') and self._use_colfft(be, [int(x) for x in p.Nmesh[:2]]) and
                be.rowfft_supported(int(p.Nmesh[2]), self.elsize))

    def can_fuse(self):
        """True if execute(..., transfer=) can fold a transfer function into the transform"""
        p = self.partition
        if self.forward or p.ndim != 3 or getattr(p, 'is_c2c', False):
            return False
        if p.nproc != 1:
            be = backend.get()
            if getattr(p, 'pencil', False):
                # the transfer rides on the first stage, the axis-0 column pass
                return (not getattr(p, '_unsupported', None) and COLFFT != 'never' and hasattr(be, 'colfft') and
                        be.colfft_supported(int(p.Nmesh[0]), self.elsize))
            return p.transposed and self._slab_own(be)
        return self._use_colfft(backend.get(), [int(x) for x in p.Nmesh[:2]])

    def _slab_own(self, be):
        """True when every local stage of the slab transform runs on the LDS row/column
        kernels of csrc/pmx_colfft.hip"""
        p = self.partition
        return (p.ndim == 3 and COLFFT != 'never' and hasattr(be, 'colfft') and
                be.rowfft_supported(int(p.Nmesh[2]), self.elsize) and
                be.colfft_supported(int(p.Nmesh[1]), self.elsize) and
                be.colfft_supported(int(p.Nmesh[0]), self.elsize))

    def _use_colfft(self, be, lengths):
        if COLFFT == 'never' or not hasattr(be, 'colfft'):
            return False
        return all(be.colfft_supported(n, self.elsize) for n in lengths)

    def _execute_local(self, bufin, bufout, transfer=None):
        be = backend.get()
        p = self.partition
        n = [int(x) for x in p.Nmesh]
        norm = 1.0 / float(numpy.prod(p.Nmesh, dtype='f8'))
        inplace = bufin.storage.data_ptr() == bufout.storage.data_ptr()
        if p.ndim == 3 and self._use_colfft(be, n[:2]):
            return self._execute_local_hybrid(be, bufin, bufout, inplace, transfer)
        settle(bufin.storage)
        if transfer is not None:
            raise NotImplementedError('fused transfer needs the column-FFT path')
        if p.ndim > 3:
            return self._execute_local_nd(be, bufin, bufout, inplace)

        def make():
            if self.forward:
                return be.fft_create(_abi.PMX_FFT_R2C, self.elsize, n, p.i_strides, p.i_alloc,
                                     p.o_strides, p.o_alloc, 1, norm, inplace)
            return be.fft_create(_abi.PMX_FFT_C2R, self.elsize, n, p.o_strides, p.o_alloc,
                                 p.i_strides, p.i_alloc, 1, 1.0, inplace)
        plan = self._native(('local', inplace), make)
        be.fft_execute(plan, bufin.storage, bufout.storage)

    def _execute_local_nd(self, be, bufin, bufout, inplace):
        """Meshes of more than three dimensions on one rank (pfft transforms any number; ParticleMesh.reshape(Nmesh=
        [8, 8, 8, 8]), pmesh/tests/test_pm.py:381-384): rocFFT plans hold up to three, so the last three axes are ONE
        batched 3-d plan (real-to-complex or complex; the leading axes, dense among themselves, are its batch) and
        every leading axis a strided 1-d complex plan over the contiguous run behind it — in place in the output
        buffer (an out-of-place call copies first: a rare path, not a hot one)."""
        p = self.partition
        nd = p.ndim
        n = [int(x) for x in p.Nmesh]
        lead = nd - 3
        norm = 1.0 / float(numpy.prod(p.Nmesh, dtype='f8'))
        work = bufout.storage
        if not inplace:
            cnt = (p.i_alloc if self.forward else 2 * p.o_alloc) if not p.is_c2c else 2 * max(p.i_alloc, p.o_alloc)
            work[:cnt].copy_(bufin.storage[:cnt])
        istr = [int(x) for x in p.i_strides]
        ostr = [int(x) for x in p.o_strides]
        nbatch = 1
        for x in n[:lead]:
            nbatch *= x
        if p.is_c2c:
            k3f, k3b = _abi.PMX_FFT_C2C_FWD, _abi.PMX_FFT_C2C_BWD
        else:
            k3f, k3b = _abi.PMX_FFT_R2C, _abi.PMX_FFT_C2R

        def inner():
            if self.forward:
                plan = self._native(('nd3',), lambda: be.fft_create(k3f, self.elsize, n[lead:], istr[lead:], istr[lead - 1],
                                                                   ostr[lead:], ostr[lead - 1], nbatch, norm, True))
            else:
                plan = self._native(('nd3',), lambda: be.fft_create(k3b, self.elsize, n[lead:], ostr[lead:], ostr[lead - 1],
                                                                   istr[lead:], istr[lead - 1], nbatch, 1.0, True))
            be.fft_execute(plan, work, work)

        def leading():
            kind = _abi.PMX_FFT_C2C_FWD if self.forward else _abi.PMX_FFT_C2C_BWD
            for d in range(lead):
                run = ostr[d]                               # complex elements behind axis d: contiguous
                plan = self._native(('nd1', d), lambda: be.fft_create(kind, self.elsize, [n[d]], [run], 1, [run], 1,
                                                                      run, 1.0, True))
                outer = 1
                for x in n[:d]:
                    outer *= x
                for j in range(outer):
                    off = 2 * j * n[d] * run                # in reals: the storage is a real tensor
                    be.fft_execute(plan, work[off:], work[off:])

        if self.forward:
            inner()
            leading()
        else:
            leading()
            inner()

    def _execute_local_hybrid(self, be, bufin, bufout, inplace, transfer):
        """3-d transform on one rank: rocFFT along the contiguous axis, column FFTs along
        axes 1 and 0 (one read + one write of the array per pass)."""
        p = self.partition
        N0, N1, N2 = [int(x) for x in p.Nmesh]
        N2c = p.pitch_c                       # row pitch in complex elements (>= N2/2+1)
        norm = 1.0 / float(N0) / float(N1) / float(N2)
        rows = N0 * N1
        plane = p.plane_c                      # padded plane stride (complex elements) or None
        rpp, ppitch, sa, sn = (N1, plane, plane, plane) if plane else (0, 0, 0, 0)
        # out of place (the reference's default: r2c() / c2r() return a new field): the FIRST pass reads the input and
        # writes the output buffer (pmx_rowfft_to / pmx_colfft_to), the others run in place there — no copy of the
        # array in front of an in-place transform
        oop = bool(plane) and not inplace and hasattr(be, 'rowfft_to') and be.rowfft_supported(N2, self.elsize)
        if plane and not inplace and not oop:
            # the padded layout is only walked by the in-place kernels: transform a copy — of the COMPLETE field: a halo
            # merge that the paint left to this transform is a note on the input's storage, which the copy does not carry
            if self.forward and getattr(bufin.storage, '_pmx_halo', None) is not None:
                settle(bufin.storage)
            n = p.i_alloc if self.forward else 2 * p.o_alloc
            bufout.storage[:n].copy_(bufin.storage[:n])
            bufin = bufout
            inplace = True
        src0 = bufin.storage                    # what the first pass reads
        work = bufout.storage if oop else bufin.storage
        own_rows = (inplace or oop) and be.rowfft_supported(N2, self.elsize)
        # the halo merge of the paint that made this field, left to this transform's row pass (window._HaloDebt)
        debt = getattr(src0, '_pmx_halo', None) if self.forward else None
        if debt is not None and not (own_rows and hasattr(be, 'rowfft_halo') and debt.open and
                                     debt.canvas_ptr == src0.data_ptr() and
                                     be.lib.pmx_rowfft_halo_supported(N2, self.elsize) == 0):
            settle(src0)
            debt = None
        hrpp, hpp = (N1, plane) if plane else (N1, N1 * N2c)
        # (out of place the input field keeps its debt: its own values still lack the halos this pass has added to
        # what it wrote elsewhere; whoever reads that field, or needs the plan, runs the merge kernel then)
        keeps_debt = debt is not None and oop

        def row_forward(blk_in, blk, nrows, x0, last):
            if debt is not None:
                be.rowfft_halo(self.elsize, blk_in, nrows, N2, N2c, hrpp, hpp, debt.plan, debt.canvas_ptr, x0,
                               last and not keeps_debt, dst=blk if oop else None)
            elif oop:
                be.rowfft_to(self.elsize, False, blk_in, blk, nrows, N2, N2c, rows_per_plane=rpp, plane_pitch=ppitch)
            else:
                be.rowfft(self.elsize, False, blk, nrows, N2, N2c, rows_per_plane=rpp, plane_pitch=ppitch)

        def first_inverse(st):
            """the axis-0 pass that starts c2r (with the transfer function riding on it)"""
            if not oop and self._take_over_forward_pass(be, st, transfer, N0, N1, N2c, sn):
                return
            kw = {}
            if transfer is not None:
                t, start, nmesh, boxsize = transfer
                kw = dict(transfer=t, n1=N1, n2=N2c, start=start, nmesh=nmesh, boxsize=boxsize)
            if oop:
                be.colfft_to(self.elsize, True, src0, st, 1, N0, N1 * N2c, n_stride=sn, **kw)
            else:
                be.colfft(self.elsize, True, st, 1, N0, N1 * N2c, n_stride=sn, **kw)

        # Infinity-Cache blocking: the row pass and the axis-1 pass both work inside single
        # planes, so they can run back to back on a block of planes that fits the 256 MiB
        # last-level cache: the second touch of a block is served on die instead of from HBM
        # (3 HBM passes -> ~2).  L3_BLOCK_BYTES = 0 switches it off.
        nblk = 1
        if own_rows and plane and L3_BLOCK_BYTES > 0 and N0 * plane * 2 * self.elsize <= L3_BLOCK_MAX_ARRAY:
            plane_bytes = plane * 2 * self.elsize
            per = max(1, int(L3_BLOCK_BYTES // plane_bytes))
            if per < N0:
                nblk = (N0 + per - 1) // per
        if nblk > 1:
            st = work
            es = 2                                  # real elements per complex element
            if self.forward:
                for b in range(nblk):
                    i0, i1 = b * per, min(N0, (b + 1) * per)
                    blk = st[i0 * plane * es:]
                    row_forward(src0[i0 * plane * es:], blk, (i1 - i0) * N1, i0, b == nblk - 1)
                    be.colfft(self.elsize, False, blk, i1 - i0, N1, N2c, a_stride=sa)
                if debt is not None and not keeps_debt:
                    debt.taken()
                self._last_forward_pass(be, st, N0, N1, N2c, norm, sn)
            else:
                first_inverse(st)
                for b in range(nblk):
                    i0, i1 = b * per, min(N0, (b + 1) * per)
                    blk = st[i0 * plane * es:]
                    be.colfft(self.elsize, True, blk, i1 - i0, N1, N2c, a_stride=sa)
                    be.rowfft(self.elsize, True, blk, (i1 - i0) * N1, N2, N2c, rows_per_plane=rpp, plane_pitch=ppitch)
            return
        if self.forward:
            if own_rows:
                row_forward(src0, work, rows, 0, True)
                if debt is not None and not keeps_debt:
                    debt.taken()
            else:
                def make():
                    return be.fft_create(_abi.PMX_FFT_R2C, self.elsize, [N2], [1], 2 * N2c, [1], N2c,
                                         rows, 1.0, inplace)
                be.fft_execute(self._native(('z', inplace), make), bufin.storage, bufout.storage)
            out = bufout.storage
            be.colfft(self.elsize, False, out, N0, N1, N2c, a_stride=sa)
            self._last_forward_pass(be, out, N0, N1, N2c, norm, sn)
        else:
            # in place on the complex data (out of place: the first pass has moved it into the output buffer;
            # without the out-of-place kernels c2r(out=...) made `bufin` a copy)
            src = work
            first_inverse(src)
            be.colfft(self.elsize, True, src, N0, N1, N2c, a_stride=sa)
            if own_rows:
                be.rowfft(self.elsize, True, src, rows, N2, N2c, rows_per_plane=rpp, plane_pitch=ppitch)
            else:
                def make():
                    return be.fft_create(_abi.PMX_FFT_C2R, self.elsize, [N2], [1], N2c, [1], 2 * N2c,
                                         rows, 1.0, inplace)
                be.fft_execute(self._native(('z', inplace), make), src, bufout.storage)

    def _last_forward_pass(self, be, st, N0, N1, N2c, norm, sn):
        """the axis-0 pass that ends r2c on one rank — run now, or left as a note on the storage for the inverse
        transform that may follow at once (DEFER_LAST_PASS)"""
        es = self.elsize

        if not (DEFER_LAST_PASS and hasattr(be, 'colfft_roundtrip') and be.colfft_roundtrip_supported(N0, es)):
            return be.colfft(es, False, st, 1, N0, N1 * N2c, scale=norm, n_stride=sn)
        # (the note hangs on the storage: its closures hold the storage weakly, or a field dropped before its c2r
        # would only be freed by the cyclic collector)
        ref = weakref.ref(st)
        del st

        def run():
            st = ref()
            if st is not None:
                be.colfft(es, False, st, 1, N0, N1 * N2c, scale=norm, n_stride=sn)

        def fused(transfer):
            st = ref()
            if st is None:
                return
            if transfer is not None:
                t, start, nmesh, boxsize = transfer
                be.colfft_roundtrip(es, st, N0, N1 * N2c, scale=norm, transfer=t, n1=N1, n2=N2c, start=start,
                                    nmesh=nmesh, boxsize=boxsize, n_stride=sn)
            else:
                be.colfft_roundtrip(es, st, N0, N1 * N2c, scale=norm, n_stride=sn)
        ref()._pmx_pending = _Pending(self.partition, run, fused, 'local')

    def _last_slab_pass(self, be, out, N0, n1loc, N2c, nb):
        """the axis-0 pass on the block a slab transpose delivered (one exchange) — now, or deferred"""
        es = self.elsize

        if not (DEFER_LAST_PASS and hasattr(be, 'colfft_roundtrip') and be.colfft_roundtrip_supported(N0, es)):
            return be.colfft(es, False, out, 1, N0, nb)
        ref = weakref.ref(out)
        del out

        def run():
            out = ref()
            if out is not None:
                be.colfft(es, False, out, 1, N0, nb)

        def fused(transfer):
            out = ref()
            if out is None:
                return
            if transfer is not None:
                t, start, nmesh, boxsize = transfer
                be.colfft_roundtrip(es, out, N0, nb, transfer=t, n1=n1loc, n2=N2c, start=start, nmesh=nmesh,
                                    boxsize=boxsize)
            else:
                be.colfft_roundtrip(es, out, N0, nb)
        ref()._pmx_pending = _Pending(self.partition, run, fused, 'slab')

    def _last_pencil_pass(self, be, out, N0, m1, m2, norm):
        """the axis-0 pass on the (N0, m1, m2) block the second transpose of a pencil transform delivered (it carries
        the 1 / prod(Nmesh)) — now, or deferred like the slab one"""
        es = self.elsize
        if m1 * m2 == 0:
            return

        if not (DEFER_LAST_PASS and COLFFT != 'never' and hasattr(be, 'colfft_roundtrip') and
                be.colfft_roundtrip_supported(N0, es) and be.colfft_supported(N0, es)):
            return self._col(be, out, 1, N0, m1 * m2, False, scale=norm)
        ref = weakref.ref(out)
        del out

        def run():
            out = ref()
            if out is not None:
                self._col(be, out, 1, N0, m1 * m2, False, scale=norm)

        def fused(transfer):
            out = ref()
            if out is None:
                return
            if transfer is not None:
                t, start, nmesh, boxsize = transfer
                be.colfft_roundtrip(es, out, N0, m1 * m2, scale=norm, transfer=t, n1=m1, n2=m2, start=start,
                                    nmesh=nmesh, boxsize=boxsize)
            else:
                be.colfft_roundtrip(es, out, N0, m1 * m2, scale=norm)
        ref()._pmx_pending = _Pending(self.partition, run, fused, 'slab')

    def _first_pencil_pass(self, be, bufin, same, transfer, W0, N0, m1, m2):
        """first stage of c2r on pencils: the inverse axis-0 pass on the local (N0, m1, m2) block, with the transfer
        function riding on it — or, when the forward transform left its last pass for us, both passes and the
        transfer as one kernel.  Returns the storage the spectrum of the next stage lives in."""
        es = self.elsize
        S = bufin.storage
        pend = getattr(S, '_pmx_pending', None)
        taken = False
        if pend is not None:
            S._pmx_pending = None
            if pend.kind == 'slab' and same and pend.partition is self.partition:
                pend.fused(transfer)
                taken = True
            else:
                pend.run()
        if not same:
            ncplx = 2 * N0 * m1 * m2
            W0[:ncplx].copy_(bufin.storage[:ncplx])
            S = W0
        if taken:
            pass
        elif transfer is not None and m1 * m2:
            # the local block is (N0, m1, m2) at the global start the caller passes
            # (0, o_start[1], o_start[2]): the transfer rides on the axis-0 pass as on slabs
            t, start, nmesh, boxsize = transfer
            be.colfft(es, True, S, 1, N0, m1 * m2, transfer=t, n1=m1, n2=m2,
                      start=start, nmesh=nmesh, boxsize=boxsize)
        else:
            self._col(be, S, 1, N0, m1 * m2, True)
        return S

    def _take_over_forward_pass(self, be, st, transfer, N0, N1, N2c, sn):
        """first stage of c2r on one rank: if the forward transform left its last pass for us, both axis-0 passes
        (and the transfer) are one kernel; returns False when there is nothing to take over"""
        pend = getattr(st, '_pmx_pending', None)
        if pend is None:
            return False
        st._pmx_pending = None
        if pend.partition is not self.partition:
            pend.run()
            return False
        pend.fused(transfer)
        return True

    def _slab_row_forward(self, be, X, nrows, N1, N2, pi):
        """the forward row pass of a slab rank's block, in place on X.  A paint that left its halo merge to this
        transform (window._HaloDebt on X) is paid here: the staged halos are added to the rows as they are loaded"""
        debt = getattr(X, '_pmx_halo', None)
        if debt is not None:
            if (hasattr(be, 'rowfft_halo') and debt.open and debt.canvas_ptr == X.data_ptr() and nrows and
                    be.lib.pmx_rowfft_halo_supported(N2, self.elsize) == 0):
                be.rowfft_halo(self.elsize, X, nrows, N2, pi, N1, N1 * pi, debt.plan, debt.canvas_ptr, 0, True)
                debt.taken()
                return
            settle(X)
        if nrows:
            be.rowfft(self.elsize, False, X, nrows, N2, pi)

    def _execute_slab(self, bufin, bufout, transfer=None):
        """Slab-decomposed 3-D (or 2-D) transform with one global transpose."""
        be = backend.get()
        p = self.partition
        comm = p.procmesh.comm
        if not p.transposed:
            return self._execute_slab_untransposed(bufin, bufout)
        nd = p.ndim
        P = p.nproc
        N0 = int(p.Nmesh[0])
        n0loc = int(p.local_i_shape[0])
        N1c = int(p.cshape_o[1])                 # complex extent of axis 1 (N1, or N1/2+1 in 2-D)
        n1loc = int(p.local_o_shape[1])
        rest = [int(x) for x in p.cshape_o[2:]]  # trailing complex extents
        n2 = 1
        for x in rest:
            n2 *= x
        e0 = [int(x) for x in p.i_edges[0]]
        e1 = [int(x) for x in p.o_edges[1]]
        elb = 2 * self.elsize
        cdt = torch.complex64 if self.elsize == 4 else torch.complex128
        norm = 1.0 / float(numpy.prod(p.Nmesh, dtype='f8'))
        rdt = bufin.storage.dtype
        need = max(2 * n0loc * N1c * max(n2, int(getattr(p, 'pitch_i', 0))), 2 * n1loc * N0 * n2, 2)
        if self._work is None or self._work[0].numel() < need or self._work[0].dtype != rdt:
            self._work = [torch.empty(need, dtype=rdt, device=bufin.storage.device) for _ in range(3)]
        W0, W1, W2 = self._work
        send_splits = [2 * n0loc * (e1[r + 1] - e1[r]) * n2 for r in range(P)]
        recv_splits = [2 * (e0[s + 1] - e0[s]) * n1loc * n2 for s in range(P)]
        inner_real = [int(x) for x in p.Nmesh[1:]]           # transform lengths of the local stage
        inner_strides_r = p.i_strides[1:]
        plane_r = p.i_strides[0]
        inner_c = [int(x) for x in p.cshape_o[1:]]
        inner_strides_c = _c_strides(inner_c)
        plane_c = N1c * n2
        nsend, nrecv = sum(send_splits), sum(recv_splits)

        same = bufin.storage.data_ptr() == bufout.storage.data_ptr()
        own = self._slab_own(be)
        if not (own and self.forward and same):
            settle_halo = getattr(bufin.storage, '_pmx_halo', None)
            if settle_halo is not None:
                settle(bufin.storage)           # only the in-place forward row pass of the LDS kernels pays it itself
        if transfer is not None and (not own or self.forward):
            raise NotImplementedError('fused transfer needs the column-FFT path of c2r')
        if own:
            # the LDS-resident row / column kernels of csrc/pmx_colfft.hip, all in place
            N1, N2 = int(p.Nmesh[1]), int(p.Nmesh[2])
            N2c = n2
            nb = n1loc * N2c
            pi = int(getattr(p, 'pitch_i', N2c))                # complex elements per real-side row
            chunks = self._chunks(be, p, P, N0, N1, N2c, n0loc, n1loc, e0, e1)
            pend = getattr(bufin.storage, '_pmx_pending', None)
            if pend is not None and not (same and ((chunks and pend.kind == 'slabpipe' and pend.data['chunks'] == chunks) or
                                                    (not chunks and pend.kind == 'slab'))):
                settle(bufin.storage)                          # a deferred pass this transform cannot take over
            if chunks:
                return self._execute_slab_pipelined(be, comm, bufin, bufout, transfer, chunks, same,
                                                    N0, N1, N2, N2c, n0loc, n1loc, pi, norm, W0, W1, W2)
            if self.forward:
                X = bufin.storage
                if not same:
                    nreal = n0loc * N1 * 2 * pi
                    W0[:nreal].copy_(bufin.storage[:nreal])     # r2c preserves its input
                    X = W0
                # equal power-of-two ranges of axis 1: the pack rides on the column pass
                fuse_pack = (hasattr(be, 'colfft_split') and n1loc * P == N1 and
                             n1loc & (n1loc - 1) == 0 and all(e1[r + 1] - e1[r] == n1loc for r in range(P)))
                assert fuse_pack or pi == N2c               # padded rows only with the fused pack
                self._slab_row_forward(be, X, n0loc * N1, N1, N2, pi)
                if n0loc:
                    if fuse_pack:
                        be.colfft_split(self.elsize, False, X, W1, n0loc, N1, N2c, n1loc, scale=norm,
                                        plain_pitch=pi)
                    else:
                        be.colfft(self.elsize, False, X, n0loc, N1, N2c, scale=norm)
                if not (fuse_pack and n0loc):
                    be.slab_pack(X, W1, n0loc, N1c, n2, e1, elb)
                out = bufout.storage
                comm.alltoall(W1[:nsend], out[:nrecv], send_splits, recv_splits)
                if nb:
                    self._last_slab_pass(be, out, N0, n1loc, N2c, nb)
            else:
                S = bufin.storage
                pend = getattr(S, '_pmx_pending', None)
                if pend is not None:
                    S._pmx_pending = None
                    if pend.kind == 'slab' and same:
                        pend.fused(transfer)                    # both axis-0 passes and the transfer: one kernel
                    else:
                        pend.run()
                        pend = None
                if not same:
                    ncplx = 2 * n1loc * N0 * n2
                    W0[:ncplx].copy_(bufin.storage[:ncplx])     # c2r preserves its input
                    S = W0
                if pend is not None:
                    pass
                elif nb and transfer is not None:
                    # the local block is (N0, n1loc, N2c) at global start (0, o_start[1], 0)
                    t, start, nmesh, boxsize = transfer
                    be.colfft(self.elsize, True, S, 1, N0, nb, transfer=t, n1=n1loc, n2=N2c,
                              start=start, nmesh=nmesh, boxsize=boxsize)
                elif nb:
                    be.colfft(self.elsize, True, S, 1, N0, nb)
                comm.alltoall(S[:nrecv], W1[:nsend], recv_splits, send_splits)
                Y = bufout.storage
                fuse_pack = (hasattr(be, 'colfft_split') and n1loc * P == N1 and
                             n1loc & (n1loc - 1) == 0 and all(e1[r + 1] - e1[r] == n1loc for r in range(P)))
                assert fuse_pack or pi == N2c
                if fuse_pack and n0loc:
                    be.colfft_split(self.elsize, True, W1, Y, n0loc, N1, N2c, n1loc, plain_pitch=pi)
                    be.rowfft(self.elsize, True, Y, n0loc * N1, N2, pi)
                else:
                    be.slab_pack(W1, Y, n0loc, N1c, n2, e1, elb, inverse=True)
                    if n0loc:
                        be.colfft(self.elsize, True, Y, n0loc, N1, N2c)
                        be.rowfft(self.elsize, True, Y, n0loc * N1, N2, N2c)
            return

        if self.forward:
            # 1. local (nd-1)-D R2C over the n0loc planes: real (padded) -> W0 (n0loc, N1c, n2)
            def make1():
                return be.fft_create(_abi.PMX_FFT_R2C, self.elsize, inner_real, inner_strides_r,
                                     plane_r, inner_strides_c, plane_c, n0loc, norm, False)
            if n0loc:
                be.fft_execute(self._native('stage1', make1), bufin.storage, W0)
            # 2. pack by destination rank; 3. all-to-all straight into the output buffer: the
            #    block from rank s is rows [e0[s], e0[s+1]) of the local (N0, n1loc, n2) array
            be.slab_pack(W0, W1, n0loc, N1c, n2, e1, elb)
            out = bufout.storage
            comm.alltoall(W1[:nsend], out[:nrecv], send_splits, recv_splits)
            # 4. one strided, batched 1-D C2C along axis 0 (stride n1loc*n2, batch n1loc*n2)
            nb = n1loc * n2

            def make2():
                return be.fft_create(_abi.PMX_FFT_C2C_FWD, self.elsize, [N0], [nb], 1, [nb], 1, nb,
                                     1.0, True)
            if nb:
                be.fft_execute(self._native('stage2', make2), out, out)
        else:
            # the backward pass works on a copy so that c2r preserves its input
            ncplx = 2 * n1loc * N0 * n2
            W0[:ncplx].copy_(bufin.storage[:ncplx])
            nb = n1loc * n2

            def make2():
                return be.fft_create(_abi.PMX_FFT_C2C_BWD, self.elsize, [N0], [nb], 1, [nb], 1, nb,
                                     1.0, True)
            if nb:
                be.fft_execute(self._native('stage2', make2), W0, W0)
            # row ranges go back to their owners; blocks -> (n0loc, N1c, n2)
            comm.alltoall(W0[:nrecv], W1[:nsend], recv_splits, send_splits)
            be.slab_pack(W1, W2, n0loc, N1c, n2, e1, elb, inverse=True)

            def make1():
                try:
                    return ('padded', be.fft_create(_abi.PMX_FFT_C2R, self.elsize, inner_real,
                                                    inner_strides_c, plane_c, inner_strides_r, plane_r,
                                                    n0loc, 1.0, False))
                except backend.PmxError:
                    # rocFFT 7.2 rejects out-of-place C2R into a padded real layout for 2-d
                    # lengths <= 64 (its single-kernel 2-d path); meshes that small are test
                    # cases only: transform into a dense buffer and copy the rows over
                    dense = _c_strides(inner_real)
                    nreal = int(numpy.prod(inner_real, dtype='i8'))
                    return ('dense', be.fft_create(_abi.PMX_FFT_C2R, self.elsize, inner_real,
                                                   inner_strides_c, plane_c, dense, nreal, n0loc, 1.0,
                                                   False))
            if n0loc:
                mode, plan1 = self._native('stage1', make1)
                if mode == 'padded':
                    be.fft_execute(plan1, W2, bufout.storage)
                else:
                    be.fft_execute(plan1, W2, W1)
                    shape = [n0loc] + inner_real
                    dst = torch.as_strided(bufout.storage, shape, [plane_r] + list(inner_strides_r))
                    dst.copy_(W1[:int(numpy.prod(shape, dtype='i8'))].view(shape))

    def _execute_c2c(self, bufin, bufout):
        """complex-to-complex transform (forward: fftn / prod(N), backward: ifftn * prod(N)) with
        rocFFT: one n-d plan on one rank; on a slab decomposition the (n-1)-d transform of the
        local planes, the global transpose, and the strided 1-d transform along axis 0"""
        be = backend.get()
        p = self.partition
        n = [int(x) for x in p.Nmesh]
        norm = 1.0 / float(numpy.prod(p.Nmesh, dtype='f8'))
        kind = _abi.PMX_FFT_C2C_FWD if self.forward else _abi.PMX_FFT_C2C_BWD
        scale = norm if self.forward else 1.0
        same = bufin.storage.data_ptr() == bufout.storage.data_ptr()
        if p.nproc == 1 and p.ndim > 3:
            settle(bufin.storage)
            return self._execute_local_nd(be, bufin, bufout, same)
        if p.nproc == 1:
            def make():
                return be.fft_create(kind, self.elsize, n, p.i_strides, p.i_alloc, p.o_strides, p.o_alloc, 1,
                                     scale, same)
            be.fft_execute(self._native(('c2c', same), make), bufin.storage, bufout.storage)
            return
        comm = p.procmesh.comm
        P = p.nproc
        N0 = n[0]
        n0loc = int(p.local_i_shape[0])
        N1 = n[1]
        n1loc = int(p.local_o_shape[1])
        n2 = 1
        for x in n[2:]:
            n2 *= x
        e0 = [int(x) for x in p.i_edges[0]]
        e1 = [int(x) for x in p.o_edges[1]]
        elb = 2 * self.elsize
        rdt = bufin.storage.dtype
        need = max(2 * n0loc * N1 * n2, 2 * n1loc * N0 * n2, 2)
        if self._work is None or self._work[0].numel() < need or self._work[0].dtype != rdt:
            self._work = [torch.empty(need, dtype=rdt, device=bufin.storage.device) for _ in range(3)]
        W0, W1, W2 = self._work
        send_splits = [2 * n0loc * (e1[r + 1] - e1[r]) * n2 for r in range(P)]
        recv_splits = [2 * (e0[s + 1] - e0[s]) * n1loc * n2 for s in range(P)]
        inner = n[1:]
        inner_strides = _c_strides(inner)
        plane = N1 * n2
        nb = n1loc * n2

        def make1():
            return be.fft_create(kind, self.elsize, inner, inner_strides, plane, inner_strides, plane, n0loc,
                                 scale, False)

        def make2():
            return be.fft_create(kind, self.elsize, [N0], [nb], 1, [nb], 1, nb, 1.0, True)
        if self.forward:
            if n0loc:
                be.fft_execute(self._native('c2c1', make1), bufin.storage, W0)
            be.slab_pack(W0, W1, n0loc, N1, n2, e1, elb)
            out = bufout.storage
            comm.alltoall(W1[:sum(send_splits)], out[:sum(recv_splits)], send_splits, recv_splits)
            if nb:
                be.fft_execute(self._native('c2c2', make2), out, out)
        else:
            ncplx = 2 * n1loc * N0 * n2
            W0[:ncplx].copy_(bufin.storage[:ncplx])            # the input is preserved
            if nb:
                be.fft_execute(self._native('c2c2', make2), W0, W0)
            comm.alltoall(W0[:sum(recv_splits)], W1[:sum(send_splits)], recv_splits, send_splits)
            be.slab_pack(W1, W2, n0loc, N1, n2, e1, elb, inverse=True)
            if n0loc:
                be.fft_execute(self._native('c2c1', make1), W2, bufout.storage)

    def _chunks(self, be, p, P, N0, N1, N2c, n0loc, n1loc, e0, e1):
        """[(first column, width)] of the last axis if the transposes can be pipelined: equal
        power-of-two blocks on both sides and enough columns; widths are multiples of 8 columns
        (128-byte lines of complex128) except for the last chunk"""
        # Every rank must come to the same answer (the probe and the chunked exchanges are collective): first what all
        # ranks see alike — equal blocks on both sides — then the size, which is then the same everywhere.  (The size
        # of a rank's OWN block came first until round 5: on uneven slabs near the threshold some ranks ran the probe
        # and the others did not — scripts/halo_fuzz_slabs.py, 3 ranks on a 192 x 128 x 512 mesh.)
        if any(e1[r + 1] - e1[r] != e1[1] - e1[0] for r in range(P)) or any(e0[r + 1] - e0[r] != e0[1] - e0[0] for r in range(P)):
            return None
        if n1loc * P != N1 or n1loc & (n1loc - 1) or n0loc * P != N0 or n0loc == 0:
            return None
        C = _overlap_chunks(2 * self.elsize * N0 * n1loc * N2c)
        if C < 2 or not hasattr(be, 'colfft_chunk') or N2c < 64:
            return None
        if not _async_exchange_works(p.procmesh.comm):
            return None
        w = -(-N2c // C)
        w = -(-w // 8) * 8
        out, b0 = [], 0
        while b0 < N2c:
            cw = min(w, N2c - b0)
            out.append((b0, cw))
            b0 += cw
        return out if len(out) > 1 else None

    def _execute_slab_pipelined(self, be, comm, bufin, bufout, transfer, chunks, same,
                                N0, N1, N2, N2c, n0loc, n1loc, pi, norm, W0, W1, W2):
        """The slab transform with its global transpose cut into chunks of the last axis: every
        stage but the row transform works chunk by chunk, so the all-to-all of chunk c (on RCCL's
        stream) runs under the column pass of chunk c+1 before it and of chunk c-1 after it.
        The fused pack / unpack and the axis-0 pass address their chunk inside the standard
        layouts (pmx_colfft_split with an offset base, pmx_colfft_chunk), nothing is copied."""
        es = self.elsize
        offs, o = [], 0
        for b0, cw in chunks:
            offs.append(o)
            o += 2 * n0loc * N1 * cw                  # reals of a chunk (= 2 * N0 * n1loc * cw)
        works = []
        if self.forward:
            X = bufin.storage
            if not same:
                nreal = n0loc * N1 * 2 * pi
                W0[:nreal].copy_(bufin.storage[:nreal])     # r2c preserves its input
                X = W0
            self._slab_row_forward(be, X, n0loc * N1, N1, N2, pi)
            for (b0, cw), o in zip(chunks, offs):
                n = 2 * n0loc * N1 * cw
                be.colfft_split(es, False, X[2 * b0:], W1[o:o + n], n0loc, N1, cw, n1loc, scale=norm, plain_pitch=pi)
                works.append(comm.alltoall(W1[o:o + n], W2[o:o + n], async_op=True))
            out = bufout.storage
            oref = weakref.ref(out)            # (the note hangs on `out`: no cycle through its closure)

            def run():
                dst = oref()
                for (b0, cw), o, w in zip(chunks, offs, works):
                    w.wait()
                    if dst is not None:
                        be.colfft_chunk(es, False, W2[o:], dst, N0, n1loc, cw, N2c, b0, True)
            if DEFER_LAST_PASS and hasattr(be, 'colfft_roundtrip') and be.colfft_roundtrip_supported(N0, es):
                # the chunks stay where the all-to-alls deliver them: an in-place c2r that follows at once runs
                # both axis-0 passes and the transfer on them as one kernel and sends them straight back
                out._pmx_pending = _Pending(self.partition, run, None, 'slabpipe',
                                            dict(chunks=chunks, offs=offs, works=works, W2=W2))
                self._deferred_on = out
            else:
                run()
        else:
            S = bufin.storage                              # read only: c2r preserves its input
            t = transfer
            pend = getattr(S, '_pmx_pending', None)
            if pend is not None:
                # (checked by _execute_slab: same chunks, in place) the forward transform left its chunks for us
                S._pmx_pending = None
                F2 = pend.data['W2']
                for (b0, cw), o, fw in zip(chunks, offs, pend.data['works']):
                    n = 2 * N0 * n1loc * cw
                    fw.wait()
                    if t is not None:
                        st = [int(t[1][0]), int(t[1][1]), int(t[1][2]) + b0]
                        be.colfft_roundtrip(es, F2[o:], N0, n1loc * cw, transfer=t[0], n1=n1loc, n2=cw, start=st,
                                            nmesh=t[2], boxsize=t[3])
                    else:
                        be.colfft_roundtrip(es, F2[o:], N0, n1loc * cw)
                    works.append(comm.alltoall(F2[o:o + n], W1[o:o + n], async_op=True))
            for (b0, cw), o in zip(chunks, offs):
                if pend is not None:
                    break
                n = 2 * N0 * n1loc * cw
                if t is not None:
                    be.colfft_chunk(es, True, W2[o:], S, N0, n1loc, cw, N2c, b0, False, transfer=t[0],
                                    start=t[1], nmesh=t[2], boxsize=t[3])
                else:
                    be.colfft_chunk(es, True, W2[o:], S, N0, n1loc, cw, N2c, b0, False)
                works.append(comm.alltoall(W2[o:o + n], W1[o:o + n], async_op=True))
            Y = bufout.storage
            for (b0, cw), o, w in zip(chunks, offs, works):
                w.wait()
                be.colfft_split(es, True, W1[o:], Y[2 * b0:], n0loc, N1, cw, n1loc, plain_pitch=pi)
            be.rowfft(es, True, Y, n0loc * N1, N2, pi)

    def _execute_slab_untransposed(self, bufin, bufout, mode=None):
        """The untransposed complex layout (n0_local, N1, N2c) on several ranks: the transposed
        transform (self.sibling, the plan of the transposed partition) plus the second global
        transpose that PFFT makes when PFFT_TRANSPOSED_OUT is not asked for."""
        be = backend.get()
        p = self.partition
        t = self.sibling
        if t is None:
            raise NotImplementedError('untransposed plan without its transposed sibling')
        pt = t.partition
        comm = p.procmesh.comm
        P = p.nproc
        N0 = int(p.Nmesh[0])
        n0loc = int(p.local_i_shape[0])
        N1c = int(pt.cshape_o[1])
        n1loc = int(pt.local_o_shape[1])
        n2 = 1
        for x in pt.cshape_o[2:]:
            n2 *= int(x)
        e0 = [int(x) for x in pt.i_edges[0]]
        e1 = [int(x) for x in pt.o_edges[1]]
        elb = 2 * self.elsize
        rdt = bufin.storage.dtype
        dev = bufin.storage.device
        need = max(int(pt.alloc_reals), 2 * n0loc * N1c * n2, 2)
        if self._work is None or self._work[0].numel() < need or self._work[0].dtype != rdt:
            self._work = [torch.empty(need, dtype=rdt, device=dev) for _ in range(2)]
        TB, W = self._work
        tb = _Storage(TB)
        # blocks of the transposed array by row range (contiguous) <-> blocks by axis-1 range
        rows_splits = [2 * (e0[s + 1] - e0[s]) * n1loc * n2 for s in range(P)]
        col_splits = [2 * n0loc * (e1[r + 1] - e1[r]) * n2 for r in range(P)]
        if mode == 'T->U':            # cast of a transposed complex field (bufin) to the untransposed layout
            comm.alltoall(bufin.storage[:sum(rows_splits)], W[:sum(col_splits)], rows_splits, col_splits)
            be.slab_pack(W, bufout.storage, n0loc, N1c, n2, e1, elb, inverse=True)
        elif mode == 'U->T':
            be.slab_pack(bufin.storage, W, n0loc, N1c, n2, e1, elb)
            comm.alltoall(W[:sum(col_splits)], bufout.storage[:sum(rows_splits)], col_splits, rows_splits)
        elif self.forward:
            t.execute(bufin, tb)                                        # -> (N0, n1loc, n2) in TB
            settle(TB)                                                  # (the sibling may have deferred its last pass)
            comm.alltoall(TB[:sum(rows_splits)], W[:sum(col_splits)], rows_splits, col_splits)
            be.slab_pack(W, bufout.storage, n0loc, N1c, n2, e1, elb, inverse=True)
        else:
            be.slab_pack(bufin.storage, W, n0loc, N1c, n2, e1, elb)
            comm.alltoall(W[:sum(col_splits)], TB[:sum(rows_splits)], col_splits, rows_splits)
            t.execute(tb, bufout)

    def retranspose(self, bufin, bufout, to_untransposed):
        """(N0, n1_local, N2c) <-> (n0_local, N1, N2c): the layout change between the transposed
        and the untransposed complex field of a slab decomposition (Field.cast); out of place"""
        p = self.partition
        if p.nproc == 1 or p.transposed:
            raise NotImplementedError('retranspose is a method of the untransposed plans of several ranks')
        settle(bufin.storage)
        forget(bufout.storage)
        if getattr(p, 'pencil', False):
            return self._execute_pencil_untransposed(bufin, bufout, mode='T->U' if to_untransposed else 'U->T')
        self._execute_slab_untransposed(bufin, bufout, mode='T->U' if to_untransposed else 'U->T')

    def destroy(self):
        try:
            be = backend.get()
        except Exception:
            return
        for plan in self._plans.values():
            if isinstance(plan, tuple):
                plan = plan[1]
            try:
                be.fft_destroy(plan)
            except Exception:
                pass
        self._plans = {}

    def __del__(self):
        self.destroy()
