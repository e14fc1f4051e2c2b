// pmx_binned.hip — tile-binned paint / readout: the LDS-tiled form of the window
// kernels for 3-d meshes (the hot path of the PM cycle).
//
// Why: the direct scatter (pmx_window.hip) is bound by the chip-wide float-atomic
// rate (~1.3 TB/s of added bytes, MI355X_MICROARCH.md "Global float atomics"):
// CIC at 512^3 adds 8 x 8 B per particle = 8.6 GB -> 6.9 ms measured.  Here the
// mesh is cut into tiles of T^3 cells; the particles are binned by the tile of
// their window's base cell (an index list per tile: positions are NOT copied);
// one workgroup per tile accumulates its (T+S-1)^3 region in LDS with ds_add and
// flushes it once: cells that only this tile can touch (the deep interior, 82 % of
// a CIC tile) are written with plain coalesced stores, cells that a neighbouring
// tile's region also covers (the S-1 deep faces) with global atomics.  Readout
// stages the tile region in LDS the same way and gathers from there.
//
// Arithmetic is the same as the direct kernels (same Tuned<KIND>::axis, same
// left-to-right products): readout is bit-identical to pmx_readout (the
// per-particle sum keeps the reference's lexicographic order), paint equals
// pmx_paint up to the order of floating-point additions into a cell.
//
// Index bookkeeping (per axis d, S = support, T = tile size):
//   X   = pos*scale + translate;  I0 = first (unwrapped) stencil index
//   w   = I0 mod period (period > 0) or I0
//   I0w = w            if w < size                      (stencil starts in the block)
//       = w - period   if period > 0 and w >= period-(S-1)   (wraps into the block)
//       = dropped      otherwise                        (touches no local cell)
//   c   = I0w + o,  o = S-1 if the block is not the full periodic mesh else 0
//   tile t = c / T, local base lb = c - t*T in [0, T)
// The region of tile t covers unwrapped cells l in [t*T - o, t*T - o + T+S-1); cell l
// maps to g = l mod period (or l); g outside [0,size) is dropped; the tile OWNS the
// cells with l in [max(0, t*T-o), min(size, (t+1)*T-o)), everything else valid is halo.
#include <hip/hip_runtime.h>
#include <math.h>

#include "pmx_common.h"
#include "pmx_window_dev.h"

namespace pmx {

constexpr int TILE = 16;
constexpr int TBLOCK = 256;

struct BinGeom {
    int32_t kind, S;
    int32_t nt[3];        // tiles per axis
    int32_t o[3];         // tile-space offset per axis
    int32_t R;            // region extent per axis = TILE + S - 1
    int32_t wrapcover[3]; // full periodic axes: cells [0, wrapcover) are also covered by the last tile
    int64_t ntiles;
};

}  // namespace pmx

struct pmx_binplan {
    pmx::BinGeom g;
    pmx_painter painter;        // geometry the plan was built for
    int64_t npart = 0;
    bool built = false;
    // device arrays
    int32_t *tid = nullptr;     // tile id per particle (-1 = dropped)
    uint32_t *slot = nullptr;   // rank of the particle inside its tile
    uint32_t *list = nullptr;   // particle indices, tile major
    size_t cap_part = 0;
    uint32_t *counts = nullptr; // particles per tile
    int64_t *offsets = nullptr; // exclusive prefix (ntiles + 1)
    size_t cap_tiles = 0;
    uint32_t *flags = nullptr;  // [0] != 0: some particle touches no local cell (tid == -1)
};

namespace pmx {

template <int KIND>
__device__ __forceinline__ bool base_cell(const pmx_painter &p, const BinGeom &g, const double *x,
                                          int *c /* tile-space coords */)
{
    constexpr int S = Tuned<KIND>::S;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        double X = x[d] * p.scale[d] + p.translate[d];
        if (!(fabs(X) < 1073741824.0)) return false;
        int I[S];
        double V[S];
        Tuned<KIND>::axis(X, 0, 1.0, I, V);
        int w = wrap1(I[0], p.period[d]);
        int i0w;
        if (p.period[d] > 0) {
            if (w < p.size[d]) i0w = w;
            else if (w >= p.period[d] - (S - 1)) i0w = w - (int)p.period[d];
            else return false;
        } else {
            if (w < -(S - 1) || w >= p.size[d]) return false;
            i0w = w;
        }
        c[d] = i0w + g.o[d];
        if (c[d] < 0) return false;
    }
    return true;
}

template <int KIND>
__global__ void __launch_bounds__(TBLOCK) bin_count_kernel(pmx_painter p, BinGeom g, DVec pos, int64_t n,
                                                           int32_t *tid, uint32_t *slot, uint32_t *counts,
                                                           uint32_t *flags)
{
    const int lane = threadIdx.x & 63;
    for (int64_t base = blockIdx.x * (int64_t)TBLOCK; base < n; base += (int64_t)gridDim.x * TBLOCK) {
        int64_t i = base + threadIdx.x;
        int t = -1;
        if (i < n) {
            double x[3] = {pos.get(i, 0), pos.get(i, 1), pos.get(i, 2)};
            int c[3];
            if (base_cell<KIND>(p, g, x, c))
                t = ((c[0] / TILE) * g.nt[1] + (c[1] / TILE)) * g.nt[2] + (c[2] / TILE);
        }
        // wave-aggregated counting: find the lanes that share my tile (ballots only), then
        // ONE atomicAdd instruction for the whole wave (the first lane of every group adds the
        // group's population), i.e. one memory round trip per wave
        unsigned long long same = 0;
        unsigned long long active = __ballot(t >= 0);
        while (active) {
            int leader = __ffsll((long long)active) - 1;
            int lt = __shfl(t, leader);
            unsigned long long m = __ballot(t == lt) & active;
            if (t == lt) same = m;
            active &= ~m;
        }
        uint32_t myslot = 0;
        if (t >= 0) {
            int leader = __ffsll((long long)same) - 1;
            uint32_t b = 0;
            if (lane == leader) b = atomicAdd(&counts[t], (uint32_t)__popcll(same));
            b = __shfl(b, leader);
            myslot = b + (uint32_t)__popcll(same & (((unsigned long long)1 << lane) - 1));
        } else if (i < n) {
            atomicOr(&flags[0], 1u);
        }
        if (i < n) {
            tid[i] = t;
            slot[i] = myslot;
        }
    }
}

// exclusive scan of counts -> offsets[ntiles+1]; one workgroup
__global__ void __launch_bounds__(1024) bin_scan_kernel(const uint32_t *counts, int64_t ntiles, int64_t *offsets)
{
    __shared__ int64_t sh[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < ntiles; base += 1024) {
        int64_t i = base + threadIdx.x;
        int64_t v = i < ntiles ? counts[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int64_t t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        int64_t incl = sh[threadIdx.x];
        if (i < ntiles) offsets[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[ntiles] = carry;
}

__global__ void __launch_bounds__(TBLOCK) bin_scatter_kernel(const int32_t *tid, const uint32_t *slot,
                                                             const int64_t *offsets, int64_t n, uint32_t *list)
{
    for (int64_t i = blockIdx.x * (int64_t)TBLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * TBLOCK) {
        int t = tid[i];
        if (t >= 0) list[offsets[t] + slot[i]] = (uint32_t)i;
    }
}

// wrap an index that is at most one period outside [0, period) (guaranteed by
// pmx_binplan_supported: every axis spans at least one tile region)
__device__ __forceinline__ int wrap_near(int l, int64_t period)
{
    if (period > 0) {
        if (l < 0) l += (int)period;
        else if (l >= period) l -= (int)period;
    }
    return l;
}

// classify one region cell of a tile: 0 = dropped (outside the local block),
// 1 = exclusive (no other tile's region covers it: plain store), 2 = shared (atomic add);
// *goff = byte offset in the canvas
__device__ __forceinline__ int region_cell(const pmx_painter &p, const BinGeom &g, const int *t, int a, int b, int c,
                                           int64_t *goff)
{
    int loc[3] = {a, b, c};
    bool owned = true, shared = false;
    int64_t off = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        int l = t[d] * TILE - g.o[d] + loc[d];
        int gidx = wrap_near(l, p.period[d]);
        if (gidx < 0 || gidx >= p.size[d]) return 0;
        owned = owned && (loc[d] < TILE) && (l >= 0) && (l < p.size[d]);
        // covered by the lower neighbour's halo, or by the periodic wrap of the last tile
        shared = shared || (loc[d] < g.S - 1) || (l < g.wrapcover[d]);
        off += gidx * p.strides[d];
    }
    *goff = off;
    return (owned && !shared) ? 1 : 2;
}

template <int KIND, typename T>
__global__ void __launch_bounds__(TBLOCK) paint_tile_kernel(pmx_painter p, BinGeom g, char *canvas, DVec pos,
                                                            DVec mass, double mass_scalar,
                                                            const uint32_t *list, const int64_t *offsets,
                                                            int overwrite)
{
    constexpr int S = Tuned<KIND>::S;
    constexpr int R = TILE + S - 1;
    __shared__ T lds[R * R * R];
    for (int64_t tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        int t[3];
        {
            int64_t r = tile;
            t[2] = (int)(r % g.nt[2]); r /= g.nt[2];
            t[1] = (int)(r % g.nt[1]); r /= g.nt[1];
            t[0] = (int)r;
        }
        const int64_t start = offsets[tile];
        const int count = (int)(offsets[tile + 1] - start);
        if (count == 0) continue;   // nothing to add (overwrite: the canvas is already zero)
        for (int q = threadIdx.x; q < R * R * R; q += TBLOCK) lds[q] = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < count; j += TBLOCK) {
            int64_t i = list[start + j];
            double x[3] = {pos.get(i, 0), pos.get(i, 1), pos.get(i, 2)};
            double m = mass.data ? mass.get(i, 0) : mass_scalar;
            int lb[3];
            double V[3][S];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                double X = x[d] * p.scale[d] + p.translate[d];
                int I[S];
                Tuned<KIND>::axis(X, p.order[d], p.scale[d], I, V[d]);
                int w = wrap1(I[0], p.period[d]);
                int i0w = w;
                if (p.period[d] > 0 && w >= p.size[d]) i0w = w - (int)p.period[d];
                lb[d] = i0w + g.o[d] - t[d] * TILE;
            }
#pragma unroll
            for (int a = 0; a < S; a++) V[0][a] *= m;
#pragma unroll
            for (int a = 0; a < S; a++)
#pragma unroll
                for (int b = 0; b < S; b++) {
                    double fb = V[0][a] * V[1][b];
                    int rowoff = ((lb[0] + a) * R + (lb[1] + b)) * R + lb[2];
#pragma unroll
                    for (int c = 0; c < S; c++) unsafeAtomicAdd(&lds[rowoff + c], (T)(fb * V[2][c]));
                }
        }
        __syncthreads();
        // flush: exclusive cells -> plain stores (rows along the last axis), shared cells ->
        // global atomics.  With `overwrite` the canvas was zero-filled before this kernel.
        for (int q = threadIdx.x; q < R * R * R; q += TBLOCK) {
            int c = q % R, r = q / R;
            int b = r % R, a = r / R;
            int64_t goff;
            int cls = region_cell(p, g, t, a, b, c, &goff);
            T v = lds[q];
            if (cls == 1) {
                T *dst = (T *)(canvas + goff);
                if (overwrite) *dst = v;
                else *dst += v;
            } else if (cls == 2) {
                if (v != (T)0) unsafeAtomicAdd((T *)(canvas + goff), v);
            }
        }
        __syncthreads();
    }
}

// entries of `out` for particles that are in no tile (they touch no local cell) read 0
__global__ void __launch_bounds__(TBLOCK) zero_dropped_kernel(const uint32_t *flags, const int32_t *tid, int64_t n,
                                                              DVec out)
{
    if (flags[0] == 0) return;   // the common case: nothing was dropped
    for (int64_t i = blockIdx.x * (int64_t)TBLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * TBLOCK)
        if (tid[i] < 0) out.set(i, 0, 0.0);
}

template <int KIND, typename T>
__global__ void __launch_bounds__(TBLOCK) readout_tile_kernel(pmx_painter p, BinGeom g, const char *canvas,
                                                              DVec pos, DVec out, const uint32_t *list,
                                                              const int64_t *offsets)
{
    constexpr int S = Tuned<KIND>::S;
    constexpr int R = TILE + S - 1;
    __shared__ T lds[R * R * R];
    for (int64_t tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        const int64_t start = offsets[tile];
        const int count = (int)(offsets[tile + 1] - start);
        if (count == 0) continue;
        int t[3];
        {
            int64_t r = tile;
            t[2] = (int)(r % g.nt[2]); r /= g.nt[2];
            t[1] = (int)(r % g.nt[1]); r /= g.nt[1];
            t[0] = (int)r;
        }
        for (int q = threadIdx.x; q < R * R * R; q += TBLOCK) {
            int c = q % R, r = q / R;
            int b = r % R, a = r / R;
            int64_t goff;
            int cls = region_cell(p, g, t, a, b, c, &goff);
            lds[q] = cls ? *(const T *)(canvas + goff) : (T)0;   // outside the block reads as 0
        }
        __syncthreads();
        for (int j = threadIdx.x; j < count; j += TBLOCK) {
            int64_t i = list[start + j];
            double x[3] = {pos.get(i, 0), pos.get(i, 1), pos.get(i, 2)};
            int lb[3];
            double V[3][S];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                double X = x[d] * p.scale[d] + p.translate[d];
                int I[S];
                Tuned<KIND>::axis(X, p.order[d], p.scale[d], I, V[d]);
                int w = wrap1(I[0], p.period[d]);
                int i0w = w;
                if (p.period[d] > 0 && w >= p.size[d]) i0w = w - (int)p.period[d];
                lb[d] = i0w + g.o[d] - t[d] * TILE;
            }
            double value = 0;
#pragma unroll
            for (int a = 0; a < S; a++)
#pragma unroll
                for (int b = 0; b < S; b++) {
                    double fb = V[0][a] * V[1][b];
                    int rowoff = ((lb[0] + a) * R + (lb[1] + b)) * R + lb[2];
#pragma unroll
                    for (int c = 0; c < S; c++) value += (double)lds[rowoff + c] * (fb * V[2][c]);
                }
            out.set(i, 0, value);
        }
        __syncthreads();
    }
}

static int ensure(void **ptr, size_t *cap, size_t need)
{
    if (need <= *cap) return PMX_OK;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr;
    *cap = 0;
    PMX_HIP_CHECK(hipMalloc(ptr, need));
    *cap = need;
    return PMX_OK;
}

static bool same_geometry(const pmx_painter &a, const pmx_painter &b)
{
    if (a.kind != b.kind || a.ndim != b.ndim) return false;
    for (int d = 0; d < 3; d++)
        if (a.scale[d] != b.scale[d] || a.translate[d] != b.translate[d] || a.period[d] != b.period[d] ||
            a.size[d] != b.size[d])
            return false;
    return true;
}

}  // namespace pmx

using namespace pmx;

extern "C" int pmx_binplan_create(pmx_binplan **plan)
{
    PMX_REQUIRE(plan != nullptr, PMX_EINVAL, "plan pointer is NULL");
    *plan = new pmx_binplan();
    return PMX_OK;
}

extern "C" int pmx_binplan_destroy(pmx_binplan *pl)
{
    if (!pl) return PMX_OK;
    if (pl->tid) (void)hipFree(pl->tid);
    if (pl->slot) (void)hipFree(pl->slot);
    if (pl->list) (void)hipFree(pl->list);
    if (pl->counts) (void)hipFree(pl->counts);
    if (pl->offsets) (void)hipFree(pl->offsets);
    if (pl->flags) (void)hipFree(pl->flags);
    delete pl;
    return PMX_OK;
}

// 0 if (p, npart) can use the binned kernels, else a status explaining why not
extern "C" int pmx_binplan_supported(const pmx_painter *p, int64_t npart)
{
    PMX_REQUIRE(p != nullptr, PMX_EINVAL, "painter is NULL");
    PMX_REQUIRE(p->ndim == 3, PMX_EUNSUPPORTED, "binned kernels are 3-d only");
    PMX_REQUIRE(p->kind >= PMX_TUNED_NNB && p->kind <= PMX_TUNED_PCS, PMX_EUNSUPPORTED, "tuned windows only");
    PMX_REQUIRE(p->support <= 0 || p->support == native_support(p->kind), PMX_EUNSUPPORTED, "native support only");
    PMX_REQUIRE(npart < (int64_t)4294967295ll, PMX_EUNSUPPORTED, "more than 2^32 particles per rank");
    int S = native_support(p->kind);
    for (int d = 0; d < 3; d++) {
        // every region cell must map to a distinct canvas cell
        int64_t span = p->period[d] > 0 ? p->period[d] : p->size[d];
        PMX_REQUIRE(span >= TILE + S - 1, PMX_EUNSUPPORTED, "mesh smaller than a tile region");
        PMX_REQUIRE(p->size[d] >= 1, PMX_EUNSUPPORTED, "empty block");
        PMX_REQUIRE(p->period[d] == 0 || p->size[d] <= p->period[d], PMX_EUNSUPPORTED, "block larger than period");
        // a block that is almost (but not exactly) the whole period would let a stencil wrap
        // from below onto cells another tile stores exclusively
        PMX_REQUIRE(p->period[d] == 0 || p->size[d] == p->period[d] || p->size[d] <= p->period[d] - (S - 1),
                    PMX_EUNSUPPORTED, "block within S-1 cells of the full period");
    }
    return PMX_OK;
}

extern "C" int pmx_binplan_build(pmx_binplan *pl, const pmx_painter *p_, const pmx_vec *pos, int64_t npart,
                                 void *stream)
{
    PMX_REQUIRE(pl != nullptr, PMX_EINVAL, "plan is NULL");
    int rc = pmx_binplan_supported(p_, npart);
    if (rc) return rc;
    PMX_REQUIRE(npart == 0 || (vec_ok(pos) && pos->ncol >= 3), PMX_EINVAL, "pos must be (n, >=3) f4/f8");
    hipStream_t st = (hipStream_t)stream;
    pmx_painter p = *p_;
    BinGeom g;
    g.kind = p.kind;
    g.S = native_support(p.kind);
    g.R = TILE + g.S - 1;
    g.ntiles = 1;
    for (int d = 0; d < 3; d++) {
        bool full = p.period[d] > 0 && p.size[d] == p.period[d];
        g.o[d] = full ? 0 : g.S - 1;
        g.nt[d] = (int32_t)((p.size[d] + g.o[d] + TILE - 1) / TILE);
        g.wrapcover[d] = full ? (int32_t)(g.nt[d] * TILE + g.S - 1 - p.size[d]) : 0;
        g.ntiles *= g.nt[d];
    }
    pl->g = g;
    pl->painter = p;
    pl->npart = npart;
    pl->built = false;
    size_t np1 = (size_t)(npart > 0 ? npart : 1);
    size_t cp = pl->cap_part;
    if (np1 * 4 > cp) {
        size_t c1 = 0, c2 = 0, c3 = 0;
        if (pl->tid) (void)hipFree(pl->tid);
        if (pl->slot) (void)hipFree(pl->slot);
        if (pl->list) (void)hipFree(pl->list);
        pl->tid = nullptr; pl->slot = nullptr; pl->list = nullptr; pl->cap_part = 0;
        rc = ensure((void **)&pl->tid, &c1, np1 * 4); if (rc) return rc;
        rc = ensure((void **)&pl->slot, &c2, np1 * 4); if (rc) return rc;
        rc = ensure((void **)&pl->list, &c3, np1 * 4); if (rc) return rc;
        pl->cap_part = np1 * 4;
    }
    size_t ct = pl->cap_tiles;
    if ((size_t)(g.ntiles + 1) > ct) {
        size_t c1 = 0, c2 = 0;
        if (pl->counts) (void)hipFree(pl->counts);
        if (pl->offsets) (void)hipFree(pl->offsets);
        pl->counts = nullptr; pl->offsets = nullptr; pl->cap_tiles = 0;
        rc = ensure((void **)&pl->counts, &c1, (size_t)(g.ntiles + 1) * 4); if (rc) return rc;
        rc = ensure((void **)&pl->offsets, &c2, (size_t)(g.ntiles + 1) * 8); if (rc) return rc;
        pl->cap_tiles = (size_t)(g.ntiles + 1);
    }
    if (!pl->flags) PMX_HIP_CHECK(hipMalloc((void **)&pl->flags, 16));
    PMX_HIP_CHECK(hipMemsetAsync(pl->flags, 0, 16, st));
    PMX_HIP_CHECK(hipMemsetAsync(pl->counts, 0, (size_t)(g.ntiles + 1) * 4, st));
    DVec dpos = dvec(pos);
    if (npart > 0) {
        unsigned grid = grid_for(npart, TBLOCK);
        switch (p.kind) {
        case PMX_TUNED_NNB: bin_count_kernel<PMX_TUNED_NNB><<<grid, TBLOCK, 0, st>>>(p, g, dpos, npart, pl->tid, pl->slot, pl->counts, pl->flags); break;
        case PMX_TUNED_CIC: bin_count_kernel<PMX_TUNED_CIC><<<grid, TBLOCK, 0, st>>>(p, g, dpos, npart, pl->tid, pl->slot, pl->counts, pl->flags); break;
        case PMX_TUNED_TSC: bin_count_kernel<PMX_TUNED_TSC><<<grid, TBLOCK, 0, st>>>(p, g, dpos, npart, pl->tid, pl->slot, pl->counts, pl->flags); break;
        default: bin_count_kernel<PMX_TUNED_PCS><<<grid, TBLOCK, 0, st>>>(p, g, dpos, npart, pl->tid, pl->slot, pl->counts, pl->flags); break;
        }
    }
    bin_scan_kernel<<<1, 1024, 0, st>>>(pl->counts, g.ntiles, pl->offsets);
    if (npart > 0)
        bin_scatter_kernel<<<grid_for(npart, TBLOCK), TBLOCK, 0, st>>>(pl->tid, pl->slot, pl->offsets, npart, pl->list);
    PMX_HIP_CHECK(hipGetLastError());
    pl->built = true;
    return PMX_OK;
}

template <typename T>
static int paint_binned_t(pmx_binplan *pl, const pmx_painter &p, void *canvas, DVec pos, DVec mass, double ms,
                          int overwrite, hipStream_t st)
{
    const BinGeom &g = pl->g;
    if (overwrite) {
        // shared cells are accumulated with atomics: start them (and everything else) from zero
        bool contiguous = p.strides[2] == (int64_t)sizeof(T) && p.strides[1] == p.size[2] * p.strides[2] &&
                          p.strides[0] == p.size[1] * p.strides[1];
        if (contiguous) {
            PMX_HIP_CHECK(hipMemsetAsync(canvas, 0, (size_t)(p.size[0] * p.strides[0]), st));
        } else {
            // padded / strided block: zero row by row (2-d memset over the last axis)
            for (int64_t i = 0; i < p.size[0]; i++)
                PMX_HIP_CHECK(hipMemset2DAsync((char *)canvas + i * p.strides[0], (size_t)p.strides[1], 0,
                                               (size_t)(p.size[2] * sizeof(T)), (size_t)p.size[1], st));
            PMX_REQUIRE(p.strides[2] == (int64_t)sizeof(T), PMX_EUNSUPPORTED,
                        "overwrite needs a unit-stride last axis");
        }
    }
    unsigned grid = (unsigned)(g.ntiles < 65535 * 8 ? g.ntiles : 65535 * 8);
#define PT(K) paint_tile_kernel<K, T><<<grid, TBLOCK, 0, st>>>(p, g, (char *)canvas, pos, mass, ms, pl->list, pl->offsets, overwrite)
    switch (p.kind) {
    case PMX_TUNED_NNB: PT(PMX_TUNED_NNB); break;
    case PMX_TUNED_CIC: PT(PMX_TUNED_CIC); break;
    case PMX_TUNED_TSC: PT(PMX_TUNED_TSC); break;
    default: PT(PMX_TUNED_PCS); break;
    }
#undef PT
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_paint_binned(pmx_binplan *pl, const pmx_painter *p_, void *canvas, const pmx_vec *pos,
                                const pmx_vec *mass, double mass_scalar, int32_t overwrite, void *stream)
{
    PMX_REQUIRE(pl && pl->built, PMX_EINVAL, "bin plan is not built");
    PMX_REQUIRE(p_ && same_geometry(*p_, pl->painter), PMX_EINVAL, "painter differs from the one the plan was built for");
    PMX_REQUIRE(canvas != nullptr, PMX_EINVAL, "canvas is NULL");
    PMX_REQUIRE(p_->canvas_elsize == 4 || p_->canvas_elsize == 8, PMX_EINVAL, "canvas must be float or double");
    PMX_REQUIRE(pl->npart == 0 || vec_ok(pos), PMX_EINVAL, "pos");
    pmx_painter p = *p_;
    hipStream_t st = (hipStream_t)stream;
    if (p.canvas_elsize == 8) return paint_binned_t<double>(pl, p, canvas, dvec(pos), dvec(mass), mass_scalar, overwrite, st);
    return paint_binned_t<float>(pl, p, canvas, dvec(pos), dvec(mass), mass_scalar, overwrite, st);
}

extern "C" int pmx_readout_binned(pmx_binplan *pl, const pmx_painter *p_, const void *canvas, const pmx_vec *pos,
                                  const pmx_vec *out, void *stream)
{
    PMX_REQUIRE(pl && pl->built, PMX_EINVAL, "bin plan is not built");
    PMX_REQUIRE(p_ && same_geometry(*p_, pl->painter), PMX_EINVAL, "painter differs from the one the plan was built for");
    PMX_REQUIRE(canvas != nullptr, PMX_EINVAL, "canvas is NULL");
    PMX_REQUIRE(vec_ok(out), PMX_EINVAL, "out must be f4/f8");
    if (pl->npart == 0) return PMX_OK;
    PMX_REQUIRE(vec_ok(pos), PMX_EINVAL, "pos");
    pmx_painter p = *p_;
    const BinGeom &g = pl->g;
    hipStream_t st = (hipStream_t)stream;
    DVec dout = dvec(out), dpos = dvec(pos);
    // particles that touch no local cell are in no tile: they read 0
    zero_dropped_kernel<<<grid_for(pl->npart, TBLOCK, 1024), TBLOCK, 0, st>>>(pl->flags, pl->tid, pl->npart, dout);
    unsigned grid = (unsigned)(g.ntiles < 65535 * 8 ? g.ntiles : 65535 * 8);
#define RT(K, T) readout_tile_kernel<K, T><<<grid, TBLOCK, 0, st>>>(p, g, (const char *)canvas, dpos, dout, pl->list, pl->offsets)
    if (p.canvas_elsize == 8) {
        switch (p.kind) {
        case PMX_TUNED_NNB: RT(PMX_TUNED_NNB, double); break;
        case PMX_TUNED_CIC: RT(PMX_TUNED_CIC, double); break;
        case PMX_TUNED_TSC: RT(PMX_TUNED_TSC, double); break;
        default: RT(PMX_TUNED_PCS, double); break;
        }
    } else {
        switch (p.kind) {
        case PMX_TUNED_NNB: RT(PMX_TUNED_NNB, float); break;
        case PMX_TUNED_CIC: RT(PMX_TUNED_CIC, float); break;
        case PMX_TUNED_TSC: RT(PMX_TUNED_TSC, float); break;
        default: RT(PMX_TUNED_PCS, float); break;
        }
    }
#undef RT
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}
