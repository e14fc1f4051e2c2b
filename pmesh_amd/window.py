"""Resampling windows: the host-side mirror of ``pmesh/window.py``.

Same names, arguments and error behaviour as the reference module
(pmesh/window.py:18-263): ``Affine``, ``ResampleWindow`` with ``paint`` /
``readout`` / ``get_fwindow`` / ``get_compensation`` / ``resize``,
``FindResampler`` and the ``windows`` registry.  The arithmetic is NOT here: it
runs in the HIP kernels of csrc/pmx_window.hip (direct) and csrc/pmx_binned.hip
(LDS-tiled) behind ``pmx_paint`` / ``pmx_readout`` (include/pmesh_amd.h), which
replace the per-particle C API of pmesh/_window_imp.h:76-86 and the Cython loop
of pmesh/_window.pyx:128-205.
"""
import ctypes as C
import os
import sys

import threading
import weakref

import numpy
import torch

from . import _abi, backend
try:
    from . import _pmx          # the Cython shim (csrc/gen_pyx.py); without it the structs are filled field by field
except ImportError:             # (not a compute path: the backend itself refuses to run without the shim)
    _pmx = None
from ._arrays import to_device, vec, vec_ref, real_view, is_tensor, touched, upload, to_numpy, version_of


def _mkarr(var, shape, dtype):
    # window.py:6-16
    var = numpy.asarray(var, dtype=dtype)
    if numpy.isscalar(shape):
        shape = (int(shape),)
    r = numpy.empty(shape, dtype)
    r[...] = var
    return r


class Affine(object):
    """ Defines an affine Transformation, used by ResampleWindow (window.py:18-55).

        mesh coordinate = position x scale + translate, wrapped by period where period > 0:
            scale : factor (or one per axis) that takes positions to mesh units;
            translate, period : in integer mesh units, scalars or one per axis.
    """
    def __init__(self, ndim, scale=None, translate=None, period=None):
        # the identity unless told otherwise; scalars are repeated along the axes
        self.ndim = ndim
        self.scale = _mkarr(1.0 if scale is None else scale, ndim, 'f8')
        self.translate = _mkarr(0 if translate is None else translate, ndim, 'f8')
        self.period = _mkarr(0 if period is None else period, ndim, 'intp')

    def rescale(self, amount):
        """ Returns a new Affine where the scale is multipled by amount. """
        return Affine(self.ndim, self.scale * amount, self.translate, self.period)

    def shift(self, amount):
        """ Returns a new Affine where the translate is shifted by amount
            (integer mesh units, as translate). """
        return Affine(self.ndim, self.scale, self.translate + amount, self.period)


# ---- tile-binned fast path (csrc/pmx_binned.hip) --------------------------------
# 'auto'  : use the LDS-tiled kernels when the batch is large, 3-d, on the device and the
#           window is a tuned one at native support; otherwise the direct kernels.
# 'never' : always the direct (global-atomic) kernels.  'always': binned whenever legal.
BINNED = 'auto'
BINNED_MIN_PARTICLES = 1 << 17
# 'auto' also asks for a minimum mean density (particles per cell of the local block): the
# binned kernels pay a fixed cost per mesh tile, the direct ones per (particle x window
# point).  Measured break-even on a 64 x 512 x 512 block (scripts/thresh_probe.py):
# CIC 2^20 particles, PCS 2^17; thin ghost bands at a slab face sit well below both.
BINNED_MIN_DENSITY = {1: 0.06, 2: 0.06, 3: 0.02, 4: 0.008}
# Which form of the binned kernels: 'auto' / 'never': the tile kernels (csrc/pmx_binned.hip); 'chunks': the
# tile kernels with the chunk form of the single-pass rebuild (a test hook: what plans with a tile-ordered
# copy use).  ('always' used to select the walk kernels of rounds 2-3, removed: PmxError.)
WALK = os.environ.get('PMESH_AMD_WALK', 'auto')
_FORMS = {'auto': -1, 'never': 0, 'always': 1, 'chunks': 2}
_SORTS = {'auto': -1, 'never': 0, 'always': 1}
# A tile-ordered copy of the positions inside the plan, for rows without spatial coherence
# (include/pmesh_amd.h: pmx_binplan_sorted): 'auto' (measured by the first build), 'never', 'always'.
SORTED = os.environ.get('PMESH_AMD_SORTED', 'auto')
# Deterministic paint: True makes every paint that the tile kernels can serve (3-d, tuned window, native support,
# no hsml) bit-reproducible — independent of the order of arrival of anything and of the order of the rows: all
# sums are 64-bit integers in units of 2^-f (include/pmesh_amd.h: pmx_binplan_deterministic), rounded once into the
# canvas.  The reference's scatter is a serial loop (pmesh/_window.pyx:157-165) and reproducible for that reason.
# Costs one more sweep over the block (measured at 512^3: bench.py --deterministic 1).  Batches the tile kernels cannot
# take (2-d meshes, hsml, tables) still use floating-point atomics.
DETERMINISTIC = os.environ.get('PMESH_AMD_DETERMINISTIC', '0') not in ('0', '', 'false', 'False')
# Arithmetic of the tile-binned READOUT (include/pmesh_amd.h: pmx_binplan_exact).  The cell indices are the reference's
# bit for bit either way.  False (default): weights and sums from fused multiply-adds in the canvas' precision — a third
# of the instructions, results within 1e-14 (f8) / 1e-6 (f4) of the reference's relative to sum |weight x cell|.  True:
# every product and sum formed as the reference forms it (_window_generics.h:213-242): bit-identical to the CPU
# reference and to the direct kernels (what rounds 1-3 always did).
EXACT = os.environ.get('PMESH_AMD_EXACT', '0') not in ('0', '', 'false', 'False')


# The halo merge of a tile-binned paint left to the forward transform that usually follows at once
# (include/pmesh_amd.h: pmx_paint_binned_defer).  ParticleMesh.paint asks for it when it paints a field of its own
# making (pm.py: HALO_DEFER); the merge is then a note (`_HaloDebt`) on the field's storage, and whoever reads the
# field first pays it: RealField.r2c on one rank adds the staged halos inside its first pass (no sweep, no atomics),
# everyone else (Field.value, a readout, another paint into the field) runs the merge kernel then.  The staged values
# live in the bin plan: ONE debt per host thread may be outstanding, and every other use of a plan settles it first.
_debt_tls = threading.local()


class _HaloDebt(object):
    __slots__ = ('be', 'plan', 'painter', 'canvas_ptr', 'storage', 'open', '__weakref__')

    def __init__(self, be, plan, painter, canvas_ptr, storage):
        self.be, self.plan, self.painter, self.canvas_ptr = be, plan, painter, canvas_ptr
        self.storage = weakref.ref(storage)
        self.open = True

    def _close(self):
        self.open = False
        st = self.storage()
        if st is not None and getattr(st, '_pmx_halo', None) is self:
            st._pmx_halo = None
        if getattr(_debt_tls, 'debt', None) is self:
            _debt_tls.debt = None

    def settle(self):
        """add the staged halos to the canvas now (the merge the paint left out)"""
        if not self.open:
            return
        if self.storage() is None:
            return self.drop()
        self.be.call('halo_merge', self.plan, C.byref(self.painter), C.c_void_p(self.canvas_ptr), self.be.stream())
        self._close()

    def drop(self):
        """the canvas is gone or about to be overwritten as a whole: the staged halos are moot"""
        if not self.open:
            return
        halo, S, nt = C.c_void_p(), C.c_int32(), (C.c_int32 * 4)()
        self.be.call('binplan_halo_source', self.plan, C.c_void_p(self.canvas_ptr), int(self.painter.canvas_elsize),
                     C.byref(halo), C.byref(S), nt, 1)
        self._close()

    def taken(self):
        """a forward row pass has added the halos (pmx_rowfft_halo with last = 1 released the plan)"""
        self._close()


def settle_halo_debt():
    """pay this thread's outstanding halo debt, if any (before a plan is rebuilt, reused or destroyed)"""
    debt = getattr(_debt_tls, 'debt', None)
    if debt is not None:
        debt.settle()


class _BinCache(object):
    """A bin plan (particles ordered by mesh tile) depends only on the positions, the
    window and the affine/block geometry; the PM cycle paints and reads out at the same
    positions, so the plan of the last batches is kept and found again by the identity
    and version counter of the position tensor.  Plans are pooled: no allocation in
    steady state."""
    SLOTS = 2

    def __init__(self):
        self.entries = []     # [key, plan handle, pos tensor (kept alive), built, clock, shape]
        self.clock = 0
        self._told = {}       # id(entry) -> [(deterministic, exact), (form, sorted)] as last told to the library

    def _key(self, pos, painter):
        return (pos.data_ptr(), version_of(pos), tuple(pos.shape), pos.stride(), pos.dtype, WALK, SORTED,
                painter.kind, tuple(painter.scale), tuple(painter.translate),
                tuple(painter.period), tuple(painter.size))

    def lookup(self, be, pos, painter, pv, n):
        settle_halo_debt()
        key = self._key(pos, painter)
        for e in self.entries:
            if e[0] == key and e[3]:
                e[4] = self._tick()
                if self._warn_if_stale(be, e):
                    break                        # (built again below)
                self._options(be, e)
                return e[1]
        # a plan that last served the same geometry and about as many particles rebuilds in a single
        # pass over the positions (csrc/pmx_binned.hip: slot ranges of the previous build; "about": within an eighth —
        # on several ranks the particles migrate and a rank's count changes a little with every step)
        shape = (key[2][1:],) + key[3:] + (n,)
        # free: invalidated entries and those built for an older version of this very tensor
        free = [e for e in self.entries if not e[3] or (e[0] is not None and e[0][0] == key[0])]
        same = [e for e in self.entries if e[5] is not None and e[5][:-1] == shape[:-1] and
                abs(e[5][-1] - n) * 8 <= e[5][-1]]
        best = lambda cands: max(cands, key=lambda q: (-abs(q[5][-1] - n), q[4]))      # the closest count, then the one used last
        like = [e for e in same if any(e is q for q in free)]
        # entries of the same shape whose tensor nobody else holds any more: a time-stepping caller that
        # makes a new position tensor every step and dropped the old one.  The new tensor takes over that
        # plan and its history (single-pass rebuild) rather than opening a second one from nothing.  A
        # tensor that is still alive elsewhere — a second particle set of equal size (two species; probes
        # at as many points as there are particles) — keeps its plan while a slot is free.
        dead = [e for e in same if e[2] is not None and sys.getrefcount(e[2]) <= 2]
        if like:
            # the one used last: its lists are the closest to these positions
            e = best(like)
        elif dead:
            e = best(dead)
        elif len(self.entries) < self.SLOTS:
            plan = C.c_void_p()
            be.call('binplan_create', C.byref(plan))
            e = [None, plan, None, False, 0, None, 0]
            self.entries.append(e)
        elif same and not free:
            # every slot is taken by a live tensor: the plan of the same shape used last has the closest lists
            e = best(same)
        else:
            e = min(free or self.entries, key=lambda q: q[4])
        e[0], e[2], e[3], e[5] = key, pos, False, shape
        self._options(be, e, build=True)
        be.call('binplan_build', e[1], C.byref(painter), C.byref(pv), n, be.stream())
        e[3] = True
        # what the plan's counter of skipped rows shows NOW belongs to the positions it served before (the counter is
        # per plan object and never reset): the next lookup compares against this, not against what another tensor left
        c = C.c_uint32(0)
        be.call('binplan_stale', e[1], C.byref(c))
        if len(e) > 6:
            e[6] = int(c.value)
        e[4] = self._tick()
        return e[1]

    def _options(self, be, e, build=False):
        """the module-level switches a plan object follows (arithmetic and summation mode always, the form of its
        kernels and the tile-ordered copy when it is built): told to the library when they CHANGE — four calls per
        lookup otherwise, for values that a run sets once"""
        want = (int(bool(DETERMINISTIC)), int(bool(EXACT)))
        opts = self._told.setdefault(id(e), [None, None])
        if opts[0] != want:
            be.call('binplan_deterministic', e[1], want[0])
            be.call('binplan_exact', e[1], want[1])
            opts[0] = want
        if build:
            form = (_FORMS[WALK], _SORTS[SORTED])
            if opts[1] != form:
                be.call('binplan_configure', e[1], form[0])
                be.call('binplan_sorted', e[1], form[1], None)
                opts[1] = form

    def _tick(self):
        self.clock += 1
        return self.clock

    def sorted_plans(self, be):
        """how many of the built plans carry the tile-ordered copy of their positions"""
        n = 0
        for e in self.entries:
            if e[3]:
                c = C.c_int32(0)
                be.call('binplan_sorted', e[1], -2, C.byref(c))
                n += int(c.value)
        return n

    def _warn_if_stale(self, be, e):
        """A plan is reused for a tensor at the same address and version (torch counts in-place operations).  Rows
        rewritten behind that — a kernel of the caller's own through data_ptr(), a numpy view of shared memory — leave
        a plan built for OTHER positions: the tile kernels skip (and count) the particles they find outside the
        region their list entry names.  The count is read without waiting for the device: what a previous use of
        the plan has found makes this use rebuild it, and warns.  (The reference keeps no state between calls:
        pm.py:1795-1869.)"""
        c = C.c_uint32(0)
        be.call('binplan_stale', e[1], C.byref(c))
        if len(e) > 6 and int(c.value) != e[6]:
            lost = int(c.value) - e[6]
            e[6] = int(c.value)
            e[3] = False                         # rebuilt by this lookup
            import warnings
            warnings.warn('pmesh_amd: %d particles were skipped by an earlier paint / readout because their positions '
                          'had been changed in place without torch noticing (a stale bin plan); the plan is rebuilt '
                          'now.  Call pmesh_amd.window.clear_bin_cache() after such an update.' % lost, RuntimeWarning)
            return True
        return False

    def overflows(self, be):
        """single-pass rebuilds of the pooled plans that had to be repaired by the two-pass
        build so far (synchronise the stream first for an exact count)"""
        total = 0
        for e in self.entries:
            c = C.c_uint32(0)
            be.call('binplan_overflows', e[1], C.byref(c))
            total += int(c.value)
        return total

    def clear(self):
        """forget which batches are binned (the pooled device buffers are kept)"""
        for e in self.entries:
            e[0], e[2], e[3] = None, None, False

    def destroy(self, be):
        try:
            settle_halo_debt()
        except Exception:
            pass
        for e in self.entries:
            try:
                be.call('binplan_destroy', e[1])
            except Exception:
                pass
        self.entries = []
        self._told = {}


_bin_tls = threading.local()


def bin_cache():
    """the calling thread's plan cache (a plan is mutable device state: one cache per host
    thread keeps concurrent callers from rebuilding each other's plans)"""
    c = getattr(_bin_tls, 'cache', None)
    if c is None:
        c = _bin_tls.cache = _BinCache()
    return c


def clear_bin_cache():
    """Invalidate cached bin plans, e.g. at the start of a time step when positions
    were rewritten in place through a foreign pointer."""
    bin_cache().clear()


# [key, weak reference to the mass tensor, its statistics (4 doubles on the device)]: per host thread, like the plan
# cache (ranks that run as threads of one process would evict each other's entries: every paint recomputed its statistics)
_mass_stats_tls = threading.local()


def _mass_stats(be, m, mv):
    """What the fixed-point paint needs to know about a per-particle mass array (largest and smallest |mass|,
    all finite: include/pmesh_amd.h, pmx_mass_stats), computed ON THE DEVICE once per tensor and version — a
    time-stepping caller's masses do not change — with no temporary and no host synchronisation.  The memo
    holds the tensor by weak reference: nothing keeps a caller's masses alive, and an address that has been
    reused by another tensor is never mistaken for the old one."""
    key = (m.data_ptr(), version_of(m), tuple(m.shape), m.stride(), m.dtype)
    _mass_stats_memo = getattr(_mass_stats_tls, 'memo', None)
    if _mass_stats_memo is None:
        _mass_stats_memo = _mass_stats_tls.memo = []
    for e in _mass_stats_memo:
        if e[0] == key and e[1]() is m:
            return e[2]
    stats = torch.empty(4, dtype=torch.float64, device=m.device)
    be.call('mass_stats', C.byref(mv), m.shape[0], stats.data_ptr(), be.stream())
    try:
        ref = weakref.ref(m)
    except TypeError:
        return stats
    _mass_stats_memo[:] = [e for e in _mass_stats_memo if e[1]() is not None][-3:] + [[key, ref, stats]]
    return stats


def _binned_ok(be, painter, pos, n, hs):
    if BINNED == 'never' or be.name != 'hip' or hs is not None or isinstance(painter, _abi.PainterND):
        return False
    if BINNED == 'auto' and not DETERMINISTIC:
        if n < BINNED_MIN_PARTICLES:
            return False
        cells = 1
        for d in range(painter.ndim):
            cells *= int(painter.size[d])
        if n < BINNED_MIN_DENSITY.get(int(painter.support), 0.008) * cells:
            return False
    return be.lib.pmx_binplan_supported(C.byref(painter), n) == 0


# every kind of the reference registry is built; the table-driven ones (lanczos, acg and the
# wavelet scaling functions db / sym) regenerate their tables in pmesh_amd/_tables.py
_UNBUILT = []
_TABLES_SENT = set()


class ResampleWindow(object):
    """A resampling window (window.py:57-221; _window.pyx:67-126).

    Attributes ``kind``, ``support`` (effective integer support) and
    ``nativesupport`` as in the reference.
    """

    def __init__(self, kind, support=-1):
        self.kind = kind
        if kind in _UNBUILT:
            self._k = None
            self.nativesupport = -1
            self.support = support
            return
        if kind not in _abi.KINDS:
            raise ValueError('unknown window kind %r' % (kind,))
        self._k = _abi.KINDS[kind]
        if kind in _abi.TABLE_KINDS:
            from . import _tables
            native = _tables.table(kind)[2]
        else:
            native = {0: 1, 4: 1, 1: 2, 5: 2, 2: 3, 6: 3, 3: 4, 7: 4}[self._k]
        # pmesh_window_info_init (_window_imp.c:24-47): support <= 0 means native
        self.nativesupport = native
        self.support = native if support <= 0 else int(support)
        self._support_arg = int(support)

    def _require_built(self):
        if self._k is None:
            raise NotImplementedError(
                "window kind %r is not built" % (self.kind,))
        if self.kind in _abi.TABLE_KINDS:
            be = backend.get()
            key = (id(be), self._k)
            if key not in _TABLES_SENT:
                from . import _tables
                values, step, _ = _tables.table(self.kind)
                values = numpy.ascontiguousarray(values, dtype='f8')
                be.call('window_set_table', self._k, values.ctypes.data_as(C.POINTER(C.c_double)),
                        len(values), float(step))
                _TABLES_SENT.add(key)

    def resize(self, support):
        """ Change the support of the window, returning a new window. """
        return ResampleWindow(self.kind, support)

    def get_compensation(self):
        """ Return a function that compensates the resampling window by deconvolving in
            Fourier space; usable as an argument of ComplexField.apply with kind='circular'
            (window.py:65-80). """
        def deconvolve(w, v):
            # the window's transform is separable: one factor per axis
            for wi in w:
                v = v / self.get_fwindow(wi)
            return v
        return deconvolve

    def get_fwindow(self, w):
        """ 1d fourier space window T(w) at circular frequencies w (window.py:82-104);
            1 if the window has no analytic transform. Evaluated on the host: it feeds
            init-time compensation tables, not the particle path. """
        self._require_built()
        dev = w.device if is_tensor(w) else None
        wh = w.detach().cpu().numpy() if is_tensor(w) else w
        w1d = numpy.ascontiguousarray(numpy.reshape(wh, -1).astype('float64'))
        T = numpy.zeros_like(w1d)
        be = backend.get()
        be.call('fwindow', self._k, self.support,
                w1d.ctypes.data_as(C.POINTER(C.c_double)), len(w1d),
                T.ctypes.data_as(C.POINTER(C.c_double)))
        T = T.reshape(numpy.shape(wh))
        if dev is not None:
            return torch.from_numpy(T).to(dev)
        return T

    # ------------------------------------------------------------------
    def _painter(self, real, order, transform):
        # (meshes of more than three dimensions: pmx_painter_nd, served by pmx_paint_nd / pmx_readout_nd)
        if _pmx is not None and real.dim() <= _abi.PMX_MAXDIM:
            # (the struct filled by one typed call of the Cython shim instead of ~30 ctypes attribute stores)
            p = _abi.Painter()
            _pmx.fill_painter(p, self._k, self.support, real.element_size(), real.shape, real.stride(), order,
                              transform.scale, transform.translate, transform.period)
            return p
        p = _abi.Painter() if real.dim() <= _abi.PMX_MAXDIM else _abi.PainterND()
        p.kind = self._k
        p.support = self.support
        p.ndim = real.dim()
        p.canvas_elsize = real.element_size()
        es = real.element_size()
        for d in range(real.dim()):
            p.order[d] = int(order[d])
            p.scale[d] = float(transform.scale[d])
            p.translate[d] = float(transform.translate[d])
            p.period[d] = int(transform.period[d])
            p.size[d] = real.shape[d]
            p.strides[d] = real.stride(d) * es
        return p

    @staticmethod
    def _canvas(real, be):
        """-> (device float view, writeback callable or None)"""
        if is_tensor(real):
            if real.device != be.device:
                raise ValueError('canvas tensor is on %s, backend on %s' % (real.device, be.device))
            view = real_view(real)
            if view.dtype not in (torch.float32, torch.float64):
                raise AssertionError("real.dtype.kind == 'f'")  # _window.pyx:135
            return view, None
        host = real  # numpy array: must be written back in place
        if not isinstance(host, numpy.ndarray):
            raise TypeError('canvas must be a numpy array or a torch tensor')
        if numpy.iscomplexobj(host):
            host = host.real
        assert host.dtype.kind == 'f' and host.dtype.itemsize in (4, 8)
        dev = upload(torch.from_numpy(numpy.ascontiguousarray(host)), be.device)

        def writeback():
            to_numpy(dev, out=host)
        return dev, writeback

    def _particles(self, pos, hsml, be):
        pos, _ = to_device(pos, be.device, 'pos')
        if pos.dim() != 2:
            raise ValueError('pos must be 2 dimensional (npart, ndim)')
        n = pos.shape[0]
        hs = None
        if hsml is not None:
            hs, _ = to_device(hsml, be.device, 'hsml')
            if hs.dim() > 1 or (hs.dim() == 1 and hs.shape[0] not in (1, n)):
                raise ValueError('hsml must be a scalar or have one entry per particle')
            if hs.dim() == 1 and hs.shape[0] == 1 and n != 1:
                hs = hs[0]
        return pos, hs, n

    def prebin(self, real, pos, transform=None):
        """Bin a particle batch by mesh tile ahead of paint/readout (device tensors only).
        Optional: paint and readout bin on demand and share the plan; calling this
        separately only makes the cost of the binning visible on its own.  Returns True if
        the tile-binned kernels will be used for this batch."""
        self._require_built()
        be = backend.get()
        if not (is_tensor(real) and is_tensor(pos)):
            return False
        canvas = real_view(real)
        if transform is None:
            transform = Affine(canvas.dim())
        p = self._painter(canvas, numpy.zeros(canvas.dim(), dtype=int), transform)
        n = pos.shape[0]
        if not (n and _binned_ok(be, p, pos, n, None)):
            return False
        bin_cache().lookup(be, pos, p, vec(pos), n)
        return True

    def paint(self, real, pos, hsml=None, mass=None, diffdir=None, transform=None, _overwrite=False, _defer_to=None,
              _direct=False):
        """
            paint to a field (window.py:106-163).

            Parameters
            ----------
            real : array_like (numpy array or device tensor); original values are preserved.
            pos : array_like (npart, ndim)
            mass : array_like or None; None for 1
            hsml: array_like or None; dimensionless scaling of the kernel
            diffdir: int or None; direction for the differentiation kernel
            transform: Affine; from position to grid units.
        """
        self._require_built()
        be = backend.get()
        canvas, writeback = self._canvas(real, be)
        if transform is None:
            transform = Affine(canvas.dim())
        assert isinstance(transform, Affine)
        order = numpy.zeros(canvas.dim(), dtype=int)
        if diffdir is not None:
            order[diffdir] = 1
        pos, hs, n = self._particles(pos, hsml, be)
        if pos.shape[1] < canvas.dim():
            raise ValueError('pos has fewer columns than the canvas has dimensions')
        mass_scalar = 1.0
        mv = None
        if mass is not None:
            if numpy.isscalar(mass) or (hasattr(mass, 'ndim') and mass.ndim == 0):
                mass_scalar = float(mass)
            else:
                m, _ = to_device(mass, be.device, 'mass')
                if m.dim() != 1 or m.shape[0] not in (1, n):
                    raise ValueError('mass must be a scalar or have one entry per particle')
                if m.shape[0] == 1 and n != 1:
                    mass_scalar = float(m[0])
                else:
                    mv = vec(m)
        p = self._painter(canvas, order, transform)
        pv = vec(pos)
        hv = vec(hs) if hs is not None else None
        if n and not _direct and _binned_ok(be, p, pos, n, hs):
            plan = bin_cache().lookup(be, pos, p, pv, n)
            # a caller's mass TENSOR: its statistics are found once per tensor and version, so the kernels need
            # no pass over the masses in front of every paint (numpy masses make a new device copy per call: the
            # paint looks at those itself)
            stats = _mass_stats(be, m, mv) if (mv is not None and is_tensor(mass)) else None
            be.call('binplan_mass_stats', plan, stats.data_ptr() if stats is not None else None)
            if _defer_to is not None and writeback is None and hasattr(be.lib, 'pmx_paint_binned_defer'):
                # _defer_to: the storage tensor behind `real` (ParticleMesh.paint, a field of its own making): the
                # halo merge becomes a note on it where the library can leave it to the forward transform
                deferred = C.c_int32(0)
                be.call('paint_binned_defer', plan, C.byref(p), canvas.data_ptr(), C.byref(pv), vec_ref(mv),
                        mass_scalar, int(bool(_overwrite)), C.byref(deferred), be.stream())
                if deferred.value:
                    debt = _HaloDebt(be, plan, p, canvas.data_ptr(), _defer_to)
                    _defer_to._pmx_halo = debt
                    _debt_tls.debt = debt
            else:
                be.call('paint_binned', plan, C.byref(p), canvas.data_ptr(), C.byref(pv), vec_ref(mv),
                        mass_scalar, int(bool(_overwrite)), be.stream())
        else:
            if _overwrite:
                canvas.zero_()
            be.call('paint' if canvas.dim() <= _abi.PMX_MAXDIM else 'paint_nd', C.byref(p), canvas.data_ptr(), C.byref(pv), vec_ref(mv), mass_scalar,
                    vec_ref(hv), n, be.stream())
        touched(canvas)
        if writeback is not None:
            writeback()

    def readout(self, real, pos, hsml=None, out=None, diffdir=None, transform=None):
        """
            readout from a field (window.py:165-221).  Returns `out`; by default a new
            float64 array with one value per particle, living where `pos` lives.
        """
        self._require_built()
        be = backend.get()
        if is_tensor(real):
            canvas = real_view(real)
        else:
            host = numpy.asarray(real)
            if numpy.iscomplexobj(host):
                host = host.real
            canvas = upload(torch.from_numpy(numpy.ascontiguousarray(host)), be.device)
        if canvas.dtype not in (torch.float32, torch.float64):
            raise AssertionError("real.dtype.kind == 'f'")
        if transform is None:
            transform = Affine(canvas.dim())
        assert isinstance(transform, Affine)
        order = numpy.zeros(canvas.dim(), dtype=int)
        if diffdir is not None:
            order[diffdir] = 1
        pos_in = pos
        pos, hs, n = self._particles(pos, hsml, be)
        host_out = None
        if out is None:
            dout = torch.empty(n, dtype=torch.float64, device=be.device)
            ret_host = not is_tensor(pos_in)
        elif is_tensor(out):
            dout = out
            ret_host = False
            if dout.device != be.device:
                raise ValueError('out tensor is on the wrong device')
        else:
            host_out = out
            if not isinstance(host_out, numpy.ndarray) or host_out.dtype.kind != 'f':
                raise TypeError('out must be a float32/float64 numpy array or tensor')
            dout = torch.empty(host_out.shape, dtype=torch.float32 if host_out.dtype.itemsize == 4
                               else torch.float64, device=be.device)
            ret_host = True
        if dout.dim() != 1 or dout.shape[0] != n:
            raise ValueError('out must have one entry per particle')
        p = self._painter(canvas, order, transform)
        pv = vec(pos)
        hv = vec(hs) if hs is not None else None
        ov = vec(dout)
        if n and _binned_ok(be, p, pos, n, hs):
            plan = bin_cache().lookup(be, pos, p, pv, n)
            be.call('readout_binned', plan, C.byref(p), canvas.data_ptr(), C.byref(pv), C.byref(ov),
                    be.stream())
        else:
            be.call('readout' if canvas.dim() <= _abi.PMX_MAXDIM else 'readout_nd', C.byref(p), canvas.data_ptr(), C.byref(pv), vec_ref(hv), C.byref(ov), n,
                    be.stream())
        touched(dout)
        if host_out is not None:
            return to_numpy(dout, out=host_out)
        if ret_host:
            return to_numpy(dout)
        return dout


    def readout_many(self, reals, pos, out=None, diffdir=None, transform=None):
        """ Readout of several fields of one shape at the same positions (an extension; the reference's callers fill
            a column at a time, examples/nbody.py:214-216): out[i, f] = reals[f] at pos[i].  `out`: a (n, len(reals))
            device tensor (any strides) or None for a new float64 one; returned.  On the GPU the tile kernels read
            every tile's positions once and write the components of a row together (pmx_readout_binned_multi); what
            that entry does not serve is read field by field into the columns of `out`. """
        self._require_built()
        be = backend.get()
        reals = list(reals)
        if not reals:
            raise ValueError('readout_many needs at least one field')
        dpos, _ = to_device(pos, be.device, 'pos')
        n = dpos.shape[0]
        if out is None:
            out = torch.empty((n, len(reals)), dtype=torch.float64, device=be.device)
        if not (is_tensor(out) and out.dim() == 2 and tuple(out.shape) == (n, len(reals))):
            raise ValueError('out must be a device tensor of shape (npart, nfields)')

        def one_by_one():
            for f, real in enumerate(reals):
                self.readout(real, dpos, out=out[:, f], diffdir=diffdir, transform=transform)
            return out
        fused = (be.name == 'hip' and hasattr(be.lib, 'pmx_readout_binned_multi') and 1 < len(reals) <= _abi.PMX_MAXFIELDS
                 and all(is_tensor(r) and r.device == be.device for r in reals)
                 and out.device == be.device and out.dtype in (torch.float32, torch.float64))
        if fused:
            canvases = [real_view(r) for r in reals]
            c0 = canvases[0]
            fused = (c0.dtype in (torch.float32, torch.float64) and c0.dim() == 3 and
                     all(c.dtype == c0.dtype and c.shape == c0.shape and c.stride() == c0.stride() for c in canvases))
        if not fused:
            return one_by_one()
        if transform is None:
            transform = Affine(c0.dim())
        order = numpy.zeros(c0.dim(), dtype=int)
        if diffdir is not None:
            order[diffdir] = 1
        p = self._painter(c0, order, transform)
        pv = vec(dpos)
        if not (n and dpos.dim() == 2 and _binned_ok(be, p, dpos, n, None)):
            return one_by_one()
        plan = bin_cache().lookup(be, dpos, p, pv, n)
        ptrs = (C.c_void_p * len(canvases))(*[c.data_ptr() for c in canvases])
        ov = vec(out)
        rc = be.lib.pmx_readout_binned_multi(plan, C.byref(p), ptrs, len(canvases), C.byref(pv), C.byref(ov), be.stream())
        if rc == _abi.PMX_EUNSUPPORTED:
            return one_by_one()          # (exact arithmetic, a plan with the tile-ordered copy, positions with a pitch)
        if rc != 0:
            raise backend.PmxError('pmx_readout_binned_multi', rc, be.lib.pmx_last_error().decode())
        touched(out)
        return out


def FindResampler(window):
    if isinstance(window, str) and window in windows:
        window = windows[window]
    if not isinstance(window, ResampleWindow):
        raise TypeError("argument is not a ResampleWindow name or a ResampleWindow object")
    return window


#: the registry (window.py:231-263): every kind under its upper-case name, its lower-case name, and as a module
#: attribute (`window.CIC`); the tuned kinds go by their short names
_REGISTRY_NAMES = dict(
    [(k.upper(), k) for k in ('nearest', 'linear', 'quadratic', 'cubic')]
    + [(k.upper(), 'tuned' + k) for k in ('nnb', 'cic', 'tsc', 'pcs')]
    + [('%s%d' % (fam.upper(), n), '%s%d' % (fam, n)) for fam, orders in
       (('lanczos', (2, 3, 4, 5, 6)), ('acg', (2, 3, 4, 5, 6)), ('db', (6, 12, 20)), ('sym', (6, 12, 20))) for n in orders]
    + [(k.upper(), k) for k in _UNBUILT])
windows = {}
for _name, _kind in _REGISTRY_NAMES.items():
    windows[_name] = windows[_name.lower()] = globals()[_name] = ResampleWindow(kind=_kind)
del _name, _kind

# compatible.
methods = windows
