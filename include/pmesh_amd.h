/*
 * pmesh_amd.h — C ABI of the MI355X-native particle-mesh hot path.
 *
 * This is the drop-in boundary for the PM cycle
 *     decompose -> paint -> r2c -> apply-transfer -> c2r -> readout
 * of MP-Gadget/pmesh.  Every entry point replaces one native interface of the
 * reference (cited per function as file:line relative to the reference tree).
 * The reference's native boundary is per particle (pmesh/_window_imp.h:76-86:
 * pmesh_painter_paint(painter, pos[], weight, hsml) called from a Python-level
 * loop, pmesh/_window.pyx:157-165); a GPU needs the whole particle batch, so
 * the entry points here are the batched form of the same contract.
 *
 * Conventions
 *  - plain C: pointers + sizes, no torch / numpy types.
 *  - all `void*` data pointers are DEVICE pointers (HBM) for the pmx_* library
 *    (libpmesh_amd.so).  The test oracle (oracle/liboracle.so) exports the
 *    same signatures with the prefix pmo_ and HOST pointers.
 *  - strides are in BYTES (as numpy strides; pmesh/_window.pyx:152-154).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *    are asynchronous with respect to the host unless stated otherwise.
 *  - every function returns a pmx_status; pmx_last_error() gives the message.
 */
#ifndef PMESH_AMD_H
#define PMESH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMX_MAXDIM 3
#define PMX_MAXRANKS 64      /* decomposition targets are kept as a 64-bit mask */
#define PMX_MAXSUPPORT 32    /* same cap as the reference (_window_generics.h:23) */

typedef enum pmx_status {
    PMX_OK = 0,
    PMX_EINVAL = 1,       /* bad argument */
    PMX_EUNSUPPORTED = 2, /* valid in the reference but not built here (yet) */
    PMX_EHIP = 3,         /* HIP runtime error */
    PMX_EFFT = 4,         /* rocFFT error */
    PMX_ENOMEM = 5
} pmx_status;

/* Window kinds on the path (pmesh/_window_imp.h:4-28; window.py:230-255): every kind of the
 * reference's registry, the table-driven ones (lanczos, acg, db / sym wavelets) after
 * pmx_window_set_table. */
typedef enum pmx_window_kind {
    PMX_NEAREST = 0,   /* PMESH_PAINTER_NEAREST   */
    PMX_LINEAR = 1,    /* PMESH_PAINTER_LINEAR    */
    PMX_QUADRATIC = 2, /* PMESH_PAINTER_QUADRATIC */
    PMX_CUBIC = 3,     /* PMESH_PAINTER_CUBIC     */
    PMX_TUNED_NNB = 4, /* PMESH_PAINTER_TUNED_NNB */
    PMX_TUNED_CIC = 5, /* PMESH_PAINTER_TUNED_CIC */
    PMX_TUNED_TSC = 6, /* PMESH_PAINTER_TUNED_TSC */
    PMX_TUNED_PCS = 7, /* PMESH_PAINTER_TUNED_PCS */
    /* table driven (generic path only; need pmx_window_set_table first) */
    PMX_LANCZOS2 = 8, PMX_LANCZOS3 = 9, PMX_LANCZOS4 = 10, PMX_LANCZOS5 = 11, PMX_LANCZOS6 = 12,
    PMX_ACG2 = 13, PMX_ACG3 = 14, PMX_ACG4 = 15, PMX_ACG5 = 16, PMX_ACG6 = 17,
    /* scaling functions of orthonormal wavelets, tabulated on [0, support) (_window_wavelets.h) */
    PMX_DB6 = 18, PMX_DB12 = 19, PMX_DB20 = 20, PMX_SYM6 = 21, PMX_SYM12 = 22, PMX_SYM20 = 23
} pmx_window_kind;

/* The geometric part of `struct PMeshPainter` (pmesh/_window_imp.h:48-62):
 * window, affine transform, periodicity and the canvas block it addresses. */
typedef struct pmx_painter {
    int32_t kind;          /* pmx_window_kind */
    int32_t support;       /* <= 0: native support (window_info_init, _window_imp.c:24-47) */
    int32_t ndim;          /* 1..3 */
    int32_t canvas_elsize; /* 4 (float) or 8 (double) */
    int32_t order[PMX_MAXDIM];   /* 0 = window, 1 = derivative along that axis */
    int32_t _pad;
    double scale[PMX_MAXDIM];     /* grid = pos * scale + translate (no FMA) */
    double translate[PMX_MAXDIM];
    int64_t period[PMX_MAXDIM];   /* Nmesh; 0 = non periodic */
    int64_t size[PMX_MAXDIM];     /* extent of the local canvas block */
    int64_t strides[PMX_MAXDIM];  /* canvas strides in bytes */
} pmx_painter;

/* The same for meshes of 4 to PMX_MAXDIM_ND dimensions (the reference's painter takes up to 32, _window_imp.h:50-60;
 * ParticleMesh.reshape(Nmesh=[8, 8, 8, 8]), pmesh/tests/test_pm.py:381-384): a struct of its own, because pmx_painter
 * travels by value to the hot kernels of the 3-d path and every element costs them scalar registers.  Served by the
 * generic per-particle kernels only (pmx_paint_nd / pmx_readout_nd; no tuned, no tile-binned path). */
#define PMX_MAXDIM_ND 8
typedef struct pmx_painter_nd {
    int32_t kind;          /* pmx_window_kind */
    int32_t support;       /* <= 0: native support */
    int32_t ndim;          /* 1..PMX_MAXDIM_ND */
    int32_t canvas_elsize; /* 4 (float) or 8 (double) */
    int32_t order[PMX_MAXDIM_ND];
    double scale[PMX_MAXDIM_ND];
    double translate[PMX_MAXDIM_ND];
    int64_t period[PMX_MAXDIM_ND];
    int64_t size[PMX_MAXDIM_ND];
    int64_t strides[PMX_MAXDIM_ND];
} pmx_painter_nd;

/* A strided per-particle column set (numpy view semantics): element (i, c) is
 * at data + i*stride0 + c*stride1 and is a float (elsize 4) or double (8).
 * Mirrors the fused postype/masstype/hsmltype arguments of _window.pyx:6-16. */
typedef struct pmx_vec {
    void *data;      /* NULL = absent */
    int32_t elsize;  /* 4 or 8 */
    int32_t ncol;
    int64_t stride0; /* bytes between particles (0 broadcasts one row) */
    int64_t stride1; /* bytes between columns */
} pmx_vec;

const char *pmx_last_error(void);
int pmx_version(void);
/* the compiler line libpmesh_amd.so was built with (csrc/Makefile); a build that contains wrong-result timing
 * experiments carries -DPMX_EXPERIMENT there and bench.py refuses to report numbers for it */
const char *pmx_build_flags(void);
/* number of visible HIP devices (0 if none); never throws */
int pmx_device_count(void);

/* ---- window metadata ---------------------------------------------------- */
/* pmesh_painter_init + pmesh_window_info_init (_window_imp.c:24-47, 246-459):
 * native support and effective integer support of (kind, support). */
int pmx_window_info(int32_t kind, int32_t support, int32_t *nativesupport, int32_t *eff_support);
/* Register the lookup table of a table-driven kind on the current device (the reference
 * compiles them in: pmesh/_window_lanczos.h, _window_acg.h; `_<name>_kernel/_diff` there):
 * n values on [0, (n-1)*step], HOST array, linear interpolation.  Once per device and kind. */
int pmx_window_set_table(int32_t kind, const double *values, int32_t n, double step);
/* pmesh_painter_get_fwindow (_window_imp.c:473-485) for n circular
 * frequencies; HOST arrays (tiny, init-time). */
int pmx_fwindow(int32_t kind, int32_t support, const double *w, int64_t n, double *out);

/* ---- paint / readout ---------------------------------------------------- */
/* Batched form of pmesh_painter_paint over the Cython loop _window.pyx:128-165:
 *   for i < npart: canvas[cells of window at pos[i]] += mass[i] * W(...)
 * mass == NULL or mass->data == NULL means every particle has `mass_scalar`
 * (the 0-stride broadcast of window.py:6-16,146).  hsml NULL = 1.0.
 * Cells outside the local block after periodic wrapping are dropped
 * (_window_generics.h:144-167). Accumulates into the canvas (window.py:113). */
int pmx_paint(const pmx_painter *p, void *canvas, const pmx_vec *pos, const pmx_vec *mass,
              double mass_scalar, const pmx_vec *hsml, int64_t npart, void *stream);

/* Batched form of pmesh_painter_readout (_window.pyx:167-205):
 *   out[i] = sum over window cells canvas[cell] * W(...), in the reference's
 * lexicographic cell order, stored as float or double per out->elsize. */
int pmx_readout(const pmx_painter *p, const void *canvas, const pmx_vec *pos, const pmx_vec *hsml,
                const pmx_vec *out, int64_t npart, void *stream);
/* pmx_paint / pmx_readout for meshes of up to PMX_MAXDIM_ND dimensions (_generic_paint / _generic_readout,
 * _window_generics.h:4-142, with _fill_k, _window_imp.c:50-83: any kind, any integer support <= PMX_MAXSUPPORT,
 * per-particle hsml).  For ndim <= 3 the same numbers as the generic path of pmx_paint / pmx_readout. */
int pmx_paint_nd(const pmx_painter_nd *p, void *canvas, const pmx_vec *pos, const pmx_vec *mass,
                 double mass_scalar, const pmx_vec *hsml, int64_t npart, void *stream);
int pmx_readout_nd(const pmx_painter_nd *p, const void *canvas, const pmx_vec *pos, const pmx_vec *hsml,
                   const pmx_vec *out, int64_t npart, void *stream);

/* ---- tile-binned paint / readout (device-side acceleration structure) ---- */
/* A bin plan orders the particles of one batch by mesh tile (an index list per
 * tile; positions are not copied) so that paint accumulates each tile in LDS
 * and writes it with plain stores, and readout gathers from an LDS-staged tile.
 * 3-d meshes, tuned windows (NNB/CIC/TSC/PCS) at native support, no hsml.
 * Results: readout is bit-identical to pmx_readout; paint equals pmx_paint up
 * to the order of floating-point additions into a cell.  One plan serves any
 * number of paint/readout calls on the same positions and geometry (the PM
 * cycle paints and reads out at the same positions). */
typedef struct pmx_binplan pmx_binplan;
int pmx_binplan_create(pmx_binplan **plan);
int pmx_binplan_destroy(pmx_binplan *plan);
/* Which kernels the next builds of this plan serve: 0 / -1 (default) = tile form (one workgroup
 * accumulates a tile of 8 x 16 x 32 cells in LDS), 2 = tiles with the chunk form of the single-pass
 * rebuild (what plans with a tile-ordered copy use; a test hook).  1 was the walk form of rounds
 * 2-3 (a measured alternative that was no faster on MI355X; removed): PMX_EUNSUPPORTED.
 * Same results in every form (readout bit-identical, paint up to the order of the additions into
 * a cell). */
int pmx_binplan_configure(pmx_binplan *plan, int32_t form);
/* Arithmetic of pmx_readout_binned.  The cell indices of a particle are always the reference's bit for bit
 * (floor(pos * scale + translate) in double precision without FMA, _window_tuned_*.h).  on = 1: the weights and
 * the sum are also formed operation by operation as the reference does (_window_generics.h:213-242): results
 * bit-identical to pmx_readout and to the CPU reference.  on = 0 (default): the weights are the same polynomials
 * evaluated in one offset with fused multiply-adds, and the S^3 products are summed as nested FMAs in the type of
 * the canvas — a third of the instructions; results within 1e-14 (double canvas) / 1e-6 (float canvas) of the
 * exact form relative to the sum of |weight x cell|, inside the tolerance the parity tests allow for values. */
int pmx_binplan_exact(pmx_binplan *plan, int32_t on);
/* Deterministic paint (the reference's scatter is a serial loop, pmesh/_window.pyx:157-165: the same call gives
 * the same bits).  on = 1: pmx_paint_binned accumulates every cell as a 64-bit integer in units of 2^-f — the
 * LDS regions, the halos between tiles (integer atomics on a dense int64 copy of the block) and the pieces of
 * crowded tiles — with one f for the batch (from the largest tile population and the largest |mass|), and
 * rounds once into the caller's canvas: independent of the order in which anything arrives, run to run and
 * whatever the order of the rows.  Within 2^-f (<= 2^-50 x the largest |mass| x tile population / 2^11) per
 * contribution of the reference's sum.  Default 0: S >= 3 windows still accumulate their LDS regions in fixed
 * point (it is the faster form), the halos are merged with floating-point atomics. */
int pmx_binplan_deterministic(pmx_binplan *plan, int32_t on);
/* The fixed-point regions of pmx_paint_binned (S >= 3 windows, deterministic paint) take one scale 2^-f per
 * z segment from the largest |mass| of a per-particle mass array, and must know that every mass is finite and
 * that the masses do not span more than 2^20 (a batch that does — or that holds a NaN / Inf — is painted by the
 * floating-point form of the same kernels, exactly as in round 2; see INTEGRATION.md section 1 for the error
 * model).  By default a reduction kernel in front of every paint finds out (one read of the masses).
 * pmx_mass_stats runs that reduction on its own: stats = 4 doubles of device memory (contents opaque);
 * pmx_binplan_mass_stats hands them to the NEXT pmx_paint_binned of the plan, which then skips its own pass
 * (a time-stepping caller computes them once per mass array).  stats = NULL: back to the default. */
int pmx_mass_stats(const pmx_vec *mass, int64_t n, double *stats, void *stream);
int pmx_binplan_mass_stats(pmx_binplan *plan, const double *stats);
/* Rows without spatial coherence (catalogues in file order, shuffled sets) make every access
 * through the index list a sector of its own.  A plan can instead carry a copy of the positions
 * in tile order (one gather per build): paint and readout stream it, readout writes its results
 * in tile order and pulls them back through the inverse list.  pref: -1 (default) = decided by
 * every build of a geometry from the measured coherence of the row order: a plan without the
 * copy takes it above PMX_SORTED_TAKE_BREAKS changes of tile per 64 consecutive rows, a plan
 * with it gives it up below PMX_SORTED_DROP_BREAKS (two thresholds, so that position sets on
 * either side of one do not restart the plan every step; lattice order shows a handful of
 * breaks, random order 63), 0 = never, 1 = always, -2 = leave unchanged.
 * is_sorted (optional): whether the plan as built carries the copy.  Results do not depend on it
 * (readout bit-identical, paint up to the order of the additions). */
#define PMX_SORTED_TAKE_BREAKS 61.5
#define PMX_SORTED_DROP_BREAKS 58.0
int pmx_binplan_sorted(pmx_binplan *plan, int32_t pref, int32_t *is_sorted);
/* How many builds of this plan so far found the slot ranges of their previous build too small
 * (particles moved a lot) and fell back to the exact two-pass build on the device.  Host
 * counter, written by the device: exact once the stream has been synchronised. */
int pmx_binplan_overflows(pmx_binplan *plan, uint32_t *count);
/* How many particles the tile kernels of this plan have skipped so far because their position no longer lay in
 * the region of the tile their list entry names: the plan was built for other positions — rows rewritten in
 * place without the caller's cache noticing (the reference has no such state: it re-reads every position on
 * every call, pm.py:1795-1869).  Blocks that are the whole periodic mesh cannot tell (a particle's cell modulo
 * the tile is always inside: its mass lands in the wrong cell of the right tile).  Host counter written by the
 * device: exact once the stream has been synchronised; 0 for every correct use. */
int pmx_binplan_stale(pmx_binplan *plan, uint32_t *count);
/* [r5] builds of this plan so far with npart > 0: in ONE pass into the slot ranges of the build before (same geometry,
 * a particle count within an eighth of the previous one: a time-stepping caller, also one whose particles migrate
 * between ranks) / in two passes (the first build, another geometry or count, the back-off after an overflow). */
int pmx_binplan_builds(pmx_binplan *plan, uint32_t *single_pass, uint32_t *two_pass);
/* [r6] The order of the rows that a built plan holds, for the caller: order[k] (npart int64 of device memory) = the row that
 * stands k-th when the rows are taken tile by tile — inside a tile in the order of the rows themselves — and the rows
 * that touch no local cell last.  A time-stepping caller re-sorts its particle arrays with it every few steps
 * (ParticleMesh.tile_order; the reference has no counterpart): position gathers and result stores of the tile kernels
 * then touch whole lines again, whatever the flow has done to the order the particles were made in. */
int pmx_binplan_order(pmx_binplan *plan, int64_t *order, void *stream);
/* PMX_OK if (painter, npart) can use the binned kernels */
int pmx_binplan_supported(const pmx_painter *p, int64_t npart);
/* bin the batch: tile id + slot per particle, per-tile counts, scan, index lists */
int pmx_binplan_build(pmx_binplan *plan, const pmx_painter *p, const pmx_vec *pos, int64_t npart,
                      void *stream);
/* overwrite != 0: the canvas content is ignored and every cell of the block is
 * written (paint with hold=False without a separate zero fill, pm.py:1852-1853) */
int pmx_paint_binned(pmx_binplan *plan, const pmx_painter *p, void *canvas, const pmx_vec *pos,
                     const pmx_vec *mass, double mass_scalar, int32_t overwrite, void *stream);
int pmx_readout_binned(pmx_binplan *plan, const pmx_painter *p, const void *canvas,
                       const pmx_vec *pos, const pmx_vec *out, void *stream);
/* [r6] The readout of up to PMX_MAXFIELDS canvases of one block geometry (same painter) at the same positions, the results
 * side by side in the rows of `out`: out(i, f) = canvas f at x_i (`out`: ncol >= ncanvas, any stride0 / stride1) — the
 * three force components of a PM step written once per row (the reference's caller fills F[..., d] column by column,
 * examples/nbody.py:214-216; a column at a time every 64-byte piece of F goes to memory and back three times).  Serves
 * what the default path of pmx_readout_binned serves (relaxed arithmetic, plans without the tile-ordered copy, dense rows
 * of three positions); PMX_EUNSUPPORTED otherwise — the caller then reads the canvases one by one. */
#define PMX_MAXFIELDS 4
int pmx_readout_binned_multi(pmx_binplan *plan, const pmx_painter *p, const void *const *canvases, int32_t ncanvas,
                             const pmx_vec *pos, const pmx_vec *out, void *stream);
/* [r4] The halo merge of a paint left to its consumer.  pmx_paint_binned ends with a pass that adds the staged
 * halos of all tiles (the cells of a tile's region beyond its own box) to their owners with atomics: a
 * read-modify-write of a quarter (CIC) to two thirds (PCS) of the mesh on top of the paint itself.  In the PM cycle
 * the next reader of the mesh is the forward row pass of r2c (pm.py:1795-1869 -> pm.py:655-694), which can add the
 * staged values while it loads the rows: no pass of their own, no atomics.
 * pmx_paint_binned_defer : as pmx_paint_binned; *deferred = 1 if the merge was left out (axes 1 and 2 whole and
 *                          periodic; axis 0 the same — one rank's mesh — or [r5] a block of planes of a larger
 *                          period, a slab rank of domain.py:561-652; overwrite != 0, not deterministic, a row
 *                          length pmx_rowfft_halo gathers for), else 0
 *                          and the call is pmx_paint_binned.  While a plan holds staged halos it refuses to build or
 *                          paint (PMX_EINVAL): one of the next two calls comes first.
 * pmx_halo_merge         : the merge pmx_paint_binned would have run (no-op if nothing is staged).
 * pmx_rowfft_halo        : pmx_rowfft (forward) on rows [x0 * rows_per_plane, ...) of the painted canvas, the staged
 *                          halos added to every row as it is loaded; last != 0 releases the plan (the caller has
 *                          transformed every plane).  Values equal pmx_halo_merge + pmx_rowfft up to the order of
 *                          the additions into a cell. */
int pmx_paint_binned_defer(pmx_binplan *plan, const pmx_painter *p, void *canvas, const pmx_vec *pos,
                           const pmx_vec *mass, double mass_scalar, int32_t overwrite, int32_t *deferred,
                           void *stream);
int pmx_halo_merge(pmx_binplan *plan, const pmx_painter *p, void *canvas, void *stream);
/* what pmx_rowfft_halo reads: the staging buffer of the plan's last deferred paint (elements of the canvas type,
 * ntiles x the halo cells of a tile region in the compact numbering of csrc/pmx_binplan.h), the window support S and
 * nt[4]: the tiles per axis, then the plane of tile space the block starts at along axis 0 (0: one rank's whole mesh;
 * S - 1: a slab rank's block of planes); PMX_EINVAL unless `canvas` / `elsize` are those of that paint.  consume != 0 releases the
 * plan (the staged values are moot: the canvas is gone or about to be overwritten as a whole). */
int pmx_binplan_halo_source(pmx_binplan *plan, const void *canvas, int32_t elsize, const void **halo,
                            int32_t *S, int32_t *nt, int32_t consume);
int pmx_rowfft_halo_supported(int64_t n, int32_t elsize);
int pmx_rowfft_halo(int32_t elsize, void *data, void *dst, int64_t nrows, int64_t n, int64_t pitch, double scale,
                    int64_t rows_per_plane, int64_t plane_pitch, pmx_binplan *plan, const void *canvas,
                    int64_t x0, int32_t last, void *stream);      /* dst: NULL or data = in place, else as pmx_rowfft_to */

/* ---- domain decomposition (pmesh/domain.py:561-652 + _domain.pyx:9-122) -- */
typedef struct pmx_grid {
    int32_t ndim;
    int32_t periodic;
    int32_t nranks;                 /* size of the communicator */
    int32_t shape[PMX_MAXDIM];      /* domains per axis */
    const double *edges[PMX_MAXDIM];/* shape[d]+1 doubles each */
    const int32_t *assign;          /* DomainAssign[prod(shape)]        */
    const int16_t *degenerate;      /* DomainDegenerate[prod(shape)]    */
} pmx_grid;

/* Pass 1 (domain.py:605-636, gridnd_fill mode 0): per particle the set of
 * target ranks within +-smoothing of scale*pos, as a bit mask, and the number
 * of particles per rank.  masks: npart uint64 (out); counts: nranks int64 (out). */
int pmx_decompose_count(const pmx_grid *g, const pmx_vec *pos, const double *scale,
                        const double *smoothing, int64_t npart, uint64_t *masks,
                        int64_t *counts, void *stream);
/* Pass 2 (gridnd_fill mode 1, _domain.pyx:45-51,120-121): rank-major, stable
 * list of particle indices.  offsets: nranks int64 exclusive prefix of counts;
 * indices: sum(counts) integers of index_elsize (4 or 8) bytes. */
int pmx_decompose_fill(int32_t nranks, const uint64_t *masks, int64_t npart,
                       const int64_t *offsets, void *indices, int32_t index_elsize,
                       void *stream);

/* Layout.exchange pack (domain.py:188 `data.take(indices)`): dst row j = src row indices[j]. */
int pmx_take_rows(const void *src, int64_t src_stride0, int64_t row_bytes, const void *indices,
                  int32_t index_elsize, int64_t nrows, void *dst, void *stream);
/* The same into rows `dst_stride` bytes apart: one column of a row that packs several arrays side by side — what
 * Layout.exchange(pack=True) ships (domain.py:161-166, pack_arrays 59-80) — written where it travels from, without
 * gathering the columns one by one and concatenating them.  indices = NULL: dst row j = src row j (a column taken
 * out of packed rows on the receiving side). */
int pmx_pack_rows(const void *src, int64_t src_stride0, int64_t row_bytes, const void *indices,
                  int32_t index_elsize, int64_t nrows, void *dst, int64_t dst_stride, void *stream);
/* Layout.gather mode='sum' (domain.py:294-295, bincountv 26-48):
 * out[i*ncol + c] = sum over j with indices[j] == i of values[j*ncol + c], for all
 * i < nout (rows that receive nothing become 0, as numpy.bincount does).
 * nout = 0: `out` is not cleared first — the rows are added into what it already holds. */
int pmx_scatter_add(const void *values, int32_t elsize, int32_t ncol, const void *indices,
                    int32_t index_elsize, int64_t nrows, void *out, int64_t nout, void *stream);

/* closed-form transfer functions T(k); see pmx_apply_transfer below */
typedef struct pmx_transfer {
    double amplitude;     /* real prefactor */
    int32_t laplace_pow;  /* multiply by (k^2)^laplace_pow, k^2(0) := 1 (nbody.py:156-157); 0 = off */
    int32_t grad_dir;     /* -1 = off; else multiply by i * D(k_dir) */
    int32_t grad_kind;    /* 0: D = k (dx1_transfer nbody.py:154-160);
                             1: D = (8 sin w - sin 2w)/(6 C), w = k C, C = L/N (force_transfer 162-171) */
    int32_t deconv_pow;   /* divide by prod_d sinc(w_d/2)^deconv_pow (window.py:65-80); 0 = off */
    double gauss_r;       /* multiply by exp(-0.5 k^2 r^2) (lowpass_transfer nbody.py:177-181); 0 = off */
} pmx_transfer;

/* ---- FFT (replaces pfft.Plan / plan.execute, pm.py:1429-1434, 689, 1017) -- */
typedef enum pmx_fft_kind { PMX_FFT_R2C = 0, PMX_FFT_C2R = 1, PMX_FFT_C2C_FWD = 2, PMX_FFT_C2C_BWD = 3 } pmx_fft_kind;
typedef struct pmx_fft pmx_fft;
/* A batched strided transform of rank `ndim` over lengths n[] (C order, last
 * axis fastest; for R2C/C2R the real lengths).  Strides/dists in ELEMENTS of
 * the respective side (real elements on the real side, complex on the complex
 * side).  `scale` multiplies the output (r2c carries 1/prod(Nmesh), pm.py:692). */
int pmx_fft_create(pmx_fft **plan, int32_t kind, int32_t elsize, int32_t ndim, const int64_t *n,
                   const int64_t *istride, int64_t idist, const int64_t *ostride, int64_t odist,
                   int64_t batch, double scale, int32_t inplace);
int pmx_fft_execute(pmx_fft *plan, void *in, void *out, void *stream);
int pmx_fft_destroy(pmx_fft *plan);

/* Batched strided ("column") complex FFT, in place, lengths 2^k in 64..2048, 3 * 2^k in 192..1536 and 5 * 2^k in 320..1280, with the
 * columns resident in LDS (csrc/pmx_colfft.hip): the passes of a 3-d transform along the
 * non-contiguous axes.  `data` is an (A, N, B) complex array in C order; the transform runs
 * along the middle axis.  inverse = 0: exp(-i k x); 1: exp(+i k x); unnormalised, the result
 * is multiplied by `scale`.  transfer != NULL fuses ComplexField.apply (pm.py:1047-1070)
 * into the load of the axis-0 pass: A must be 1, B = n1*n2, element (i0, i1, i2) of the
 * local block starting at global index start[] is multiplied by T(k) first (closed forms
 * without transcendentals only: laplace_pow in -1..1, spectral gradient).
 * a_stride / n_stride (complex elements, 0 = dense): stride between successive a (>= N*B) and,
 * for A == 1, between successive lines n (>= B) — the padded plane stride of the one-rank
 * complex layout. */
int pmx_colfft_supported(int64_t n, int32_t elsize);
int pmx_colfft(int32_t elsize, int32_t inverse, void *data, int64_t A, int64_t N, int64_t B,
               double scale, const pmx_transfer *transfer, int64_t n1, int64_t n2, const int64_t *start,
               const int64_t *nmesh, const double *boxsize, int64_t a_stride, int64_t n_stride,
               void *stream);

/* The axis-1 column pass of a slab-decomposed transform fused with the pack / unpack that
 * brackets PFFT's global transpose (what pmx_slab_pack does, for equal power-of-two ranges):
 * inverse = 0: src plain (A, N, B) -> dst "split": block r = lines [r*nsplit, (r+1)*nsplit)
 * as one contiguous (A, nsplit, B) array, i.e. the all-to-all send buffer; inverse = 1: src
 * split (the receive buffer) -> dst plain.  Out of place; unnormalised, times `scale`.
 * plain_pitch: elements per line of the plain side (>= B; 0 = B): the slab layout keeps its
 * real-side rows on 128-byte boundaries while the wire format stays dense. */
int pmx_colfft_split(int32_t elsize, int32_t inverse, const void *src, void *dst, int64_t A, int64_t N,
                     int64_t B, int64_t nsplit, double scale, int64_t plain_pitch, void *stream);

/* r2c -> transfer -> c2r back to back (what a PM force step does with every density field): the last forward
 * pass and the first inverse pass run along the same axis and are one kernel — forward column transform, times
 * `scale` (the forward normalisation), times the transfer function (t != NULL, as in pmx_colfft), inverse column
 * transform, the column resident in LDS — one sweep of the array instead of two, bit-identical to
 * pmx_colfft(inverse = 0, scale) followed by pmx_colfft(inverse = 1, t).  In place on the (N, B) block at
 * n_stride elements per line (0 = B).  PFFT has no such fusion (pm.py:689, 1017 are two separate executes). */
int pmx_colfft_roundtrip_supported(int64_t n, int32_t elsize);
int pmx_colfft_roundtrip(int32_t elsize, void *data, int64_t N, int64_t B, double scale,
                         const pmx_transfer *transfer, int64_t n1, int64_t n2, const int64_t *start,
                         const int64_t *nmesh, const double *boxsize, int64_t n_stride, void *stream);

/* Scheduling of the column passes whose tile fills a compute unit (N = 1024 in both precisions, 768 / 640 in double).
 * persistent = 1 (default): one workgroup per CU walks a fixed share of the tiles and prefetches its next one —
 * the faster form when the GPU is the kernel's alone.  persistent = 0: one workgroup per tile, placed by the hardware
 * as CUs become free — what a process should choose whose transforms overlap with collectives (RCCL's kernels hold
 * CUs for the length of a transfer; a persistent workgroup that cannot be placed beside them starts when another has
 * finished its whole share).  Process-wide; the host side selects 0 for plans on more than one rank.  No PFFT
 * counterpart (pfft.Plan has no such knob). */
int pmx_colfft_configure(int32_t persistent);

/* The axis-1 pass of a pencil transform (PFFT's 2-d process mesh, pm.py:1417-1434) between its two
 * global transposes: src is the (A, N, B) array cut into ranges of nsplit_in lines (the receive
 * buffer of one all-to-all), dst the same array cut into ranges of nsplit_out lines (the send
 * buffer of the other); 0 = plain dense.  Replaces two pmx_slab_pack sweeps around pmx_colfft.
 * nsplit: 0 or a power of two dividing N.  src and dst must not overlap. */
int pmx_colfft_resplit(int32_t elsize, int32_t inverse, const void *src, void *dst, int64_t A, int64_t N,
                       int64_t B, int64_t nsplit_in, int64_t nsplit_out, double scale, void *stream);

/* The axis-0 pass on one chunk [coff, coff + cw) of the last axis of the (N, n1, pitch) block
 * `full` (pipelined slab transposes: the all-to-all of one chunk overlaps the passes of its
 * neighbours; PFFT has no such overlap).  `chunk` is the dense (N, n1, cw) buffer the
 * all-to-all delivers or takes.  to_full = 1: transform `chunk`, scatter into `full`;
 * to_full = 0: gather the chunk's columns from `full` (optionally times the transfer function,
 * as in pmx_colfft; start[] = global start of `full`), transform, write `chunk`. */
int pmx_colfft_chunk(int32_t elsize, int32_t inverse, void *chunk, void *full, int64_t N, int64_t n1,
                     int64_t cw, int64_t pitch, int64_t coff, int32_t to_full, double scale,
                     const pmx_transfer *transfer, const int64_t *start, const int64_t *nmesh,
                     const double *boxsize, void *stream);

/* Real <-> half-complex transform along the contiguous axis, in place, with the rows
 * resident in LDS (csrc/pmx_colfft.hip): `nrows` rows of n reals (n a power of two in
 * 128..2048, or 384 / 768 / 1536 / 640 / 1280) at a pitch of `pitch` complex elements <-> n/2+1 modes.  inverse = 0: r2c,
 * 1: c2r; unnormalised, times `scale`.  rows_per_plane > 0: row r starts at
 * (r / rows_per_plane) * plane_pitch + (r % rows_per_plane) * pitch complex elements (padded
 * plane stride; rows_per_plane a multiple of 8 (f8) / 16 (f4)); 0: r * pitch. */
int pmx_rowfft_supported(int64_t n, int32_t elsize);
int pmx_rowfft(int32_t elsize, int32_t inverse, void *data, int64_t nrows, int64_t n, int64_t pitch,
               double scale, int64_t rows_per_plane, int64_t plane_pitch, void *stream);
/* [r4] The same passes from `src` into `dst` (same layout, distinct buffers): the first pass of a transform whose
 * caller keeps its input — r2c() / c2r() with out=None, the reference's default (pm.py:655-694, 987-1019: PFFT plans
 * built out of place) — reads the input and writes the result buffer; the remaining passes run in place there.
 * Replaces a copy of the whole array in front of an in-place transform (and, for c2r(transfer=...), a separate
 * transfer kernel: the transfer rides on this pass as in pmx_colfft). */
int pmx_rowfft_to(int32_t elsize, int32_t inverse, const void *src, void *dst, int64_t nrows, int64_t n,
                  int64_t pitch, double scale, int64_t rows_per_plane, int64_t plane_pitch, void *stream);
int pmx_colfft_to(int32_t elsize, int32_t inverse, const void *src, void *dst, int64_t A, int64_t N, int64_t B,
                  double scale, const pmx_transfer *transfer, int64_t n1, int64_t n2, const int64_t *start,
                  const int64_t *nmesh, const double *boxsize, int64_t a_stride, int64_t n_stride, void *stream);

/* [r6] The row pass of a pencil transform (PFFT's 2-d process mesh, pm.py:1417-1434) with the last-axis split of its
 * first global transpose on it: inverse = 0: src = nrows rows of n reals (row pitch `pitch` complex elements) -> dst =
 * the n/2 + 1 modes of every row in nparts blocks, block q = the modes [offsets[q], offsets[q + 1]) of all rows as one
 * dense (nrows, offsets[q + 1] - offsets[q]) array at element nrows * offsets[q] — the send buffer of the all-to-all
 * over the row group; inverse = 1: src = those blocks (the receive buffer) -> dst = rows of n reals.  Out of place.
 * Replaces the pmx_slab_pack / pmx_slab_unpack sweep (n0 = nrows, n1 = n/2 + 1, n2 = 1) next to pmx_rowfft; same
 * values.  Built for every length of pmx_rowfft and nparts <= PMX_MAXSEG (offsets: host, nparts + 1 entries from 0
 * to n/2 + 1, not decreasing — empty blocks are allowed). */
#define PMX_MAXSEG 16
int pmx_rowfft_split_supported(int64_t n, int32_t elsize, int32_t nparts);
int pmx_rowfft_split(int32_t elsize, int32_t inverse, const void *src, void *dst, int64_t nrows, int64_t n,
                     int64_t pitch, double scale, const int64_t *offsets, int32_t nparts, void *stream);

/* Local transpose next to the all-to-all of a distributed FFT (PFFT's global transpose).
 * pmx_slab_pack  : src (n0, n1, n2) C order -> nparts contiguous blocks, block r = (n0,
 *                  n1 range [n1_offsets[r], n1_offsets[r+1]), n2): the send buffer of an
 *                  all-to-all that redistributes axis 1.
 * pmx_slab_unpack: the inverse (blocks -> (n0, n1, n2)): what the reverse all-to-all delivers.
 * The blocks an all-to-all delivers for axis 0 are row ranges of the destination array
 * and need no kernel.  Elements are `elbytes` wide (8 = complex64, 16 = complex128). */
int pmx_slab_pack(const void *src, void *dst, int64_t n0, int64_t n1, int64_t n2,
                  const int64_t *n1_offsets /* host, nparts+1 */, int32_t nparts, int32_t elbytes,
                  void *stream);
int pmx_slab_unpack(const void *src, void *dst, int64_t n0, int64_t n1, int64_t n2,
                    const int64_t *n1_offsets /* host, nparts+1 */, int32_t nparts, int32_t elbytes,
                    void *stream);

/* ---- apply-transfer (Field.apply, pm.py:617-648, with the transfer functions
 * of examples/nbody.py:154-181 and pmesh/transfer.py fused) ---------------- */

/* out[m] = T(k(m)) * in[m] over a local complex block of logical shape
 * shape[0..ndim) starting at global index start[], with byte strides; k_d =
 * 2 pi / L_d * (i - N_d [i >= N_d/2]) (pm.py:1200-1226: Nyquist negative).
 * in may equal out.  elsize = 4 (complex64) or 8 (complex128) per component. */
int pmx_apply_transfer(const pmx_transfer *t, int32_t ndim, int32_t elsize, const void *in,
                       const int64_t *in_strides, void *out, const int64_t *out_strides,
                       const int64_t *shape, const int64_t *start, const int64_t *nmesh,
                       const double *boxsize, void *stream);

/* Where the master seed stream of pmx_whitenoise runs (pmesh/_whitenoise_generics.h:73-93: one RANLUX stream walked in
 * rings over the (i, j) plane, one seed per column): 0 (default) one host core + a copy of 8 bytes per local column;
 * 1 one device thread (no copy, no wait; a sequential chain: ~35 x slower than the host core).  Same tables bit for bit. */
int pmx_whitenoise_master(int32_t on_device);
/* ---- white noise (the step before the cycle: initial conditions) -------------------------
 * pmesh.whitenoise.generate for 3-d meshes (pmesh/_whitenoise.pyx:25-45,
 * _whitenoise_imp.c:75-105, _whitenoise_generics.h:29-238; pm.py:1656-1696): fills the local
 * block [start, start+size) of the spectrum with the Gadget / N-GenIC compatible Gaussian (unitary = 0) or fixed-amplitude
 * (unitary = 1) Hermitian white noise of `seed`: every (i, j) column has its own RANLUX stream
 * seeded from a master stream walked in N-GenIC's ring order, so the result is independent
 * of the decomposition and the large scales do not change with the mesh size.
 * canvas: complex64 (elsize 8) or complex128 (elsize 16), byte strides.  The (i, j) seed table
 * is built on the host (a sequential stream of N0*N1 draws) and the columns are filled on the
 * device, one thread per column and generator.  A block that reaches beyond the Nyquist plane
 * (start[2] + size[2] > nmesh[2]/2 + 1: complex-to-complex meshes) gets the full spectrum. */
int pmx_whitenoise(uint32_t seed, int32_t unitary, const int64_t *nmesh, const int64_t *start,
                   const int64_t *size, const int64_t *strides, int32_t elsize, void *canvas,
                   void *stream);

/* ---- synthetic inputs for bench.py ------------------------ */
/* lattice + hashed jitter; writes pos (npart,3) for lattice ids [g0, g0+npart) */
int pmx_synth_uniform(const pmx_vec *pos, int64_t nlat, double boxsize, uint64_t seed, int64_t g0,
                      int64_t npart, void *stream);
/* lattice + plane-wave Zel'dovich displacement; modes: nmodes x 8 doubles on the
 * HOST (nx, ny, nz, dirx, diry, dirz, amplitude, phase) */
int pmx_synth_clustered(const pmx_vec *pos, int64_t nlat, double boxsize, const double *modes,
                        int32_t nmodes, double shift, int64_t g0, int64_t npart, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PMESH_AMD_H */
