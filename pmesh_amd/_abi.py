"""The structs of the C ABI in include/pmesh_amd.h as ctypes mirrors, and its prototypes as a ctypes table.

The product binds the library through the Cython shim ``pmesh_amd._pmx`` (generated from the header,
csrc/gen_pyx.py; backend.load_library): its wrappers take these struct mirrors (or their byref()) as pointer
arguments.  The prototype table binds a library under a symbol prefix with ctypes alone: ``pmo_`` for the CPU oracle
(host pointers; tests only) and, with PMESH_AMD_BINDING=ctypes, ``pmx_`` for the product library (a debugging aid).
Nothing here computes anything.
"""
import ctypes as C

PMX_MAXDIM = 3
PMX_MAXRANKS = 64
PMX_MAXFIELDS = 4
PMX_MAXSEG = 16

PMX_OK, PMX_EINVAL, PMX_EUNSUPPORTED, PMX_EHIP, PMX_EFFT, PMX_ENOMEM = range(6)
STATUS_NAMES = {0: 'PMX_OK', 1: 'PMX_EINVAL', 2: 'PMX_EUNSUPPORTED', 3: 'PMX_EHIP',
                4: 'PMX_EFFT', 5: 'PMX_ENOMEM'}

# pmx_window_kind
KINDS = {
    'nearest': 0, 'linear': 1, 'quadratic': 2, 'cubic': 3,
    'tunednnb': 4, 'tunedcic': 5, 'tunedtsc': 6, 'tunedpcs': 7,
    'lanczos2': 8, 'lanczos3': 9, 'lanczos4': 10, 'lanczos5': 11, 'lanczos6': 12,
    'acg2': 13, 'acg3': 14, 'acg4': 15, 'acg5': 16, 'acg6': 17,
    'db6': 18, 'db12': 19, 'db20': 20, 'sym6': 21, 'sym12': 22, 'sym20': 23,
}
TABLE_KINDS = [k for k, v in KINDS.items() if v >= 8]

PMX_FFT_R2C, PMX_FFT_C2R, PMX_FFT_C2C_FWD, PMX_FFT_C2C_BWD = range(4)


class Painter(C.Structure):
    _fields_ = [
        ('kind', C.c_int32), ('support', C.c_int32), ('ndim', C.c_int32),
        ('canvas_elsize', C.c_int32),
        ('order', C.c_int32 * PMX_MAXDIM), ('_pad', C.c_int32),
        ('scale', C.c_double * PMX_MAXDIM), ('translate', C.c_double * PMX_MAXDIM),
        ('period', C.c_int64 * PMX_MAXDIM), ('size', C.c_int64 * PMX_MAXDIM),
        ('strides', C.c_int64 * PMX_MAXDIM),
    ]


PMX_MAXDIM_ND = 8


class PainterND(C.Structure):
    """pmx_painter_nd (include/pmesh_amd.h): meshes of 4 .. PMX_MAXDIM_ND dimensions, generic kernels only"""
    _fields_ = [
        ('kind', C.c_int32), ('support', C.c_int32), ('ndim', C.c_int32),
        ('canvas_elsize', C.c_int32),
        ('order', C.c_int32 * PMX_MAXDIM_ND),
        ('scale', C.c_double * PMX_MAXDIM_ND), ('translate', C.c_double * PMX_MAXDIM_ND),
        ('period', C.c_int64 * PMX_MAXDIM_ND), ('size', C.c_int64 * PMX_MAXDIM_ND),
        ('strides', C.c_int64 * PMX_MAXDIM_ND),
    ]


class Vec(C.Structure):
    _fields_ = [
        ('data', C.c_void_p), ('elsize', C.c_int32), ('ncol', C.c_int32),
        ('stride0', C.c_int64), ('stride1', C.c_int64),
    ]


class Grid(C.Structure):
    _fields_ = [
        ('ndim', C.c_int32), ('periodic', C.c_int32), ('nranks', C.c_int32),
        ('shape', C.c_int32 * PMX_MAXDIM),
        ('edges', C.c_void_p * PMX_MAXDIM),
        ('assign', C.c_void_p), ('degenerate', C.c_void_p),
    ]


class Transfer(C.Structure):
    _fields_ = [
        ('amplitude', C.c_double), ('laplace_pow', C.c_int32), ('grad_dir', C.c_int32),
        ('grad_kind', C.c_int32), ('deconv_pow', C.c_int32), ('gauss_r', C.c_double),
    ]


_P = C.POINTER
_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double

# name -> (restype, argtypes); names without the pmx_/pmo_ prefix
PROTOTYPES = {
    'window_info': (C.c_int, [_i32, _i32, _P(_i32), _P(_i32)]),
    'fwindow': (C.c_int, [_i32, _i32, _P(_f64), _i64, _P(_f64)]),
    'paint': (C.c_int, [_P(Painter), _vp, _P(Vec), _P(Vec), _f64, _P(Vec), _i64, _vp]),
    'readout': (C.c_int, [_P(Painter), _vp, _P(Vec), _P(Vec), _P(Vec), _i64, _vp]),
    'paint_nd': (C.c_int, [_P(PainterND), _vp, _P(Vec), _P(Vec), _f64, _P(Vec), _i64, _vp]),
    'readout_nd': (C.c_int, [_P(PainterND), _vp, _P(Vec), _P(Vec), _P(Vec), _i64, _vp]),
    'decompose_count': (C.c_int, [_P(Grid), _P(Vec), _P(_f64), _P(_f64), _i64, _vp, _vp, _vp]),
    'decompose_fill': (C.c_int, [_i32, _vp, _i64, _vp, _vp, _i32, _vp]),
    'take_rows': (C.c_int, [_vp, _i64, _i64, _vp, _i32, _i64, _vp, _vp]),
    'pack_rows': (C.c_int, [_vp, _i64, _i64, _vp, _i32, _i64, _vp, _i64, _vp]),
    'scatter_add': (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i64, _vp, _i64, _vp]),
    'apply_transfer': (C.c_int, [_P(Transfer), _i32, _i32, _vp, _P(_i64), _vp, _P(_i64),
                                 _P(_i64), _P(_i64), _P(_i64), _P(_f64), _vp]),
    'whitenoise': (C.c_int, [C.c_uint32, _i32, _P(_i64), _P(_i64), _P(_i64), _P(_i64), _i32, _vp, _vp]),
    'synth_uniform': (C.c_int, [_P(Vec), _i64, _f64, C.c_uint64, _i64, _i64, _vp]),
    'synth_clustered': (C.c_int, [_P(Vec), _i64, _f64, _P(_f64), _i32, _f64, _i64, _i64, _vp]),
}

# entry points that only the device library has
DEVICE_ONLY = {
    'last_error': (C.c_char_p, []),
    'version': (C.c_int, []),
    'build_flags': (C.c_char_p, []),
    'device_count': (C.c_int, []),
    'window_set_table': (C.c_int, [_i32, _P(_f64), _i32, _f64]),
    'whitenoise_master': (C.c_int, [_i32]),
    'binplan_create': (C.c_int, [_P(_vp)]),
    'binplan_destroy': (C.c_int, [_vp]),
    'binplan_configure': (C.c_int, [_vp, _i32]),
    'binplan_exact': (C.c_int, [_vp, _i32]),
    'binplan_deterministic': (C.c_int, [_vp, _i32]),
    'mass_stats': (C.c_int, [_P(Vec), _i64, _vp, _vp]),
    'binplan_mass_stats': (C.c_int, [_vp, _vp]),
    'binplan_overflows': (C.c_int, [_vp, _P(C.c_uint32)]),
    'binplan_stale': (C.c_int, [_vp, _P(C.c_uint32)]),
    'binplan_builds': (C.c_int, [_vp, _P(C.c_uint32), _P(C.c_uint32)]),
    'binplan_sorted': (C.c_int, [_vp, _i32, _P(_i32)]),
    'binplan_order': (C.c_int, [_vp, _vp, _vp]),
    'binplan_supported': (C.c_int, [_P(Painter), _i64]),
    'binplan_build': (C.c_int, [_vp, _P(Painter), _P(Vec), _i64, _vp]),
    'paint_binned': (C.c_int, [_vp, _P(Painter), _vp, _P(Vec), _P(Vec), _f64, _i32, _vp]),
    'readout_binned': (C.c_int, [_vp, _P(Painter), _vp, _P(Vec), _P(Vec), _vp]),
    'readout_binned_multi': (C.c_int, [_vp, _P(Painter), _P(_vp), _i32, _P(Vec), _P(Vec), _vp]),
    'paint_binned_defer': (C.c_int, [_vp, _P(Painter), _vp, _P(Vec), _P(Vec), _f64, _i32, _P(_i32), _vp]),
    'halo_merge': (C.c_int, [_vp, _P(Painter), _vp, _vp]),
    'binplan_halo_source': (C.c_int, [_vp, _vp, _i32, _P(_vp), _P(_i32), _P(_i32), _i32]),
    'rowfft_halo_supported': (C.c_int, [_i64, _i32]),
    'rowfft_halo': (C.c_int, [_i32, _vp, _vp, _i64, _i64, _i64, _f64, _i64, _i64, _vp, _vp, _i64, _i32, _vp]),
    'rowfft_to': (C.c_int, [_i32, _i32, _vp, _vp, _i64, _i64, _i64, _f64, _i64, _i64, _vp]),
    'rowfft_split_supported': (C.c_int, [_i64, _i32, _i32]),
    'rowfft_split': (C.c_int, [_i32, _i32, _vp, _vp, _i64, _i64, _i64, _f64, _P(_i64), _i32, _vp]),
    'colfft_to': (C.c_int, [_i32, _i32, _vp, _vp, _i64, _i64, _i64, _f64, _P(Transfer), _i64, _i64, _P(_i64),
                            _P(_i64), _P(_f64), _i64, _i64, _vp]),
    'fft_create': (C.c_int, [_P(_vp), _i32, _i32, _i32, _P(_i64), _P(_i64), _i64, _P(_i64), _i64,
                             _i64, _f64, _i32]),
    'fft_execute': (C.c_int, [_vp, _vp, _vp, _vp]),
    'fft_destroy': (C.c_int, [_vp]),
    'colfft_supported': (C.c_int, [_i64, _i32]),
    'colfft': (C.c_int, [_i32, _i32, _vp, _i64, _i64, _i64, _f64, _P(Transfer), _i64, _i64, _P(_i64),
                         _P(_i64), _P(_f64), _i64, _i64, _vp]),
    'colfft_roundtrip_supported': (C.c_int, [_i64, _i32]),
    'colfft_configure': (C.c_int, [_i32]),
    'colfft_roundtrip': (C.c_int, [_i32, _vp, _i64, _i64, _f64, _P(Transfer), _i64, _i64, _P(_i64), _P(_i64), _P(_f64),
                                   _i64, _vp]),
    'colfft_split': (C.c_int, [_i32, _i32, _vp, _vp, _i64, _i64, _i64, _i64, _f64, _i64, _vp]),
    'colfft_resplit': (C.c_int, [_i32, _i32, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _f64, _vp]),
    'colfft_chunk': (C.c_int, [_i32, _i32, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _f64, _P(Transfer),
                               _P(_i64), _P(_i64), _P(_f64), _vp]),
    'rowfft_supported': (C.c_int, [_i64, _i32]),
    'rowfft': (C.c_int, [_i32, _i32, _vp, _i64, _i64, _i64, _f64, _i64, _i64, _vp]),
    'slab_pack': (C.c_int, [_vp, _vp, _i64, _i64, _i64, _P(_i64), _i32, _i32, _vp]),
    'slab_unpack': (C.c_int, [_vp, _vp, _i64, _i64, _i64, _P(_i64), _i32, _i32, _vp]),
}


def declare(lib, prefix, table):
    """Attach restype/argtypes for every symbol of `table` found under `prefix`.
    Returns the list of names that are missing from the library."""
    missing = []
    for name, (res, args) in table.items():
        try:
            fn = getattr(lib, prefix + name)
        except AttributeError:
            missing.append(prefix + name)
            continue
        fn.restype = res
        fn.argtypes = args
    return missing


_ARRAY_MEMO = {}


def _memo_array(ctype, conv, seq, n):
    """small constant argument arrays (mesh shapes, starts, box sizes) are asked for again on every launch: one ctypes
    array per distinct content, built once (the library only reads them)"""
    key = (ctype, tuple(seq), n)
    arr = _ARRAY_MEMO.get(key)
    if arr is None:
        vals = [conv(x) for x in seq]
        if n is not None:
            vals = vals + [conv(0)] * (n - len(vals))
        if len(_ARRAY_MEMO) > 4096:
            _ARRAY_MEMO.clear()
        arr = _ARRAY_MEMO[key] = (ctype * len(vals))(*vals)
    return arr


def i64arr(seq, n=None):
    return _memo_array(C.c_int64, int, seq, n)


def f64arr(seq, n=None):
    return _memo_array(C.c_double, float, seq, n)
