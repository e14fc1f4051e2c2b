// pmx_binplan.h — the bin plan (particles ordered by mesh tile) of the tile kernels (pmx_binned.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "pmx_common.h"
#include "pmx_window_dev.h"

namespace pmx {

#ifndef PMX_T0
#define PMX_T0 8
#endif
#ifndef PMX_T1
#define PMX_T1 16
#endif
#ifndef PMX_T2
#define PMX_T2 32
#endif
constexpr int T0 = PMX_T0, T1 = PMX_T1, T2 = PMX_T2;   // tile extents (cells) along axes 0, 1, 2
constexpr int TCELLS = T0 * T1 * T2;
constexpr int TBLOCK = 256;
#ifndef PMX_ZSEG
#define PMX_ZSEG 4
#endif
constexpr int ZSEG = PMX_ZSEG;        // tiles per segment of paint_tile_kernel
// The axis along which paint_tile_kernel walks its segments and carries the halo face in LDS: z (the face of T0 x T1
// rows of S - 1 cells: few cells, but every row a piece of its own for halo_merge's atomics) or x (the face of
// (S - 1) R1 rows of R2 cells: most of the halo's cells, in long rows).  What halo_merge pays for is pieces, not cells:
// measured at 512^3 in double, z-walk against x-walk, CIC 247 / 317 us, PCS 634 / 517 us (paint_tile_kernel itself
// the same either way), TSC within 1-2 % both ways.  So (round 3): x for PCS, z below.  PMX_WALK_AXIS = 0 / 2 forces one.
#ifndef PMX_WALK_AXIS
#define PMX_WALK_AXIS -1
#endif
// [r4, second session] With the halo merge gathered by the forward row pass (pmx_rowfft_halo) pieces no longer cost
// atomics, and what counts is the bytes staged: the x-walk stages 40 % fewer for TSC and CIC too.  Same box, z-walk
// against x-walk under the deferred merge: r2c of TSC f4 0.530 -> 0.483 ms, TSC f8 0.948 -> 0.890, CIC f8 0.838 -> 0.827
// (paint_tile_kernel +0.02 ms for TSC, unchanged for CIC).  TSC, whose merge KERNEL never cared, walks along x now; CIC
// stays on z for the callers that still run the merge kernel (several ranks, a caller's own field: 247 against 317 us).
constexpr bool walk_x(int S) { return PMX_WALK_AXIS == 0 || (PMX_WALK_AXIS < 0 && S >= 3); }
#ifndef PMX_TILE_THREADS
#define PMX_TILE_THREADS 512
#endif
#ifndef PMX_TILE_THREADS_RF4
#define PMX_TILE_THREADS_RF4 256
#endif
#ifndef PMX_UNROLL
#define PMX_UNROLL 2
#endif
constexpr int UNROLL = PMX_UNROLL;    // particles in flight per lane in the tile kernels
#ifndef PMX_ONEPASS_U
#define PMX_ONEPASS_U 2
#endif

struct BinGeom {
    int32_t kind, S;
    int32_t nt[3];        // tiles per axis
    int32_t o[3];         // tile-space offset per axis (S-1 unless the axis is the full period)
    int64_t ntiles;       // tiles of T0 x T1 x T2 cells
    int32_t chunk;        // list entries of a tile that the tile kernels take themselves
    int32_t deal_min;     // [r6] tiles of at least this many entries deal them to the lanes of the one-lane deposit loops (tile_deposit)
    int32_t quad_min;     // [r6] PCS on fixed-point regions: a tile (or piece) of at least this many entries is deposited four lanes per particle (tile_deposit_quadz)
    uint32_t *stale;      // host-visible counter of list entries found outside their tile's region (a stale plan), or NULL
};

template <int S> struct Region {
    static constexpr int R0 = T0 + S - 1, R1 = T1 + S - 1, R2 = T2 + S - 1;
    static constexpr int CELLS = R0 * R1 * R2;
    // Row pitch of the region in LDS (cells).  For S >= 3 rows are 48 cells = 384 bytes = three whole
    // bank rows instead of T2 + S - 1 = 34 / 35 cells: a particle whose base cell differs from its
    // neighbour's along x or y (half of them on the benchmark's jittered lattice under TSC, whose rounding
    // boundary the lattice straddles; any real distribution) then still hits the banks its z cell names, and
    // lanes that walk along z stay conflict free: scripts/ldsatomic_patterns.hip 27 -> 19 clocks per ds_add_f64
    // instruction.  It costs TSC its third workgroup per CU (69 KB regions), PCS nothing (80 KB: still two).
#ifndef PMX_ROW_PITCH3
#define PMX_ROW_PITCH3 48
#endif
    static constexpr int P2 = (S >= 3 && PMX_ROW_PITCH3 > R2) ? PMX_ROW_PITCH3 : R2;
    static constexpr int LDS = R0 * R1 * P2;      // elements to allocate
    // CIC / TSC paint: the split layout.  Rows of exactly T2 = 32 cells (two bank rows: every row starts on bank 0) and
    // the S - 1 halo columns of all rows in an array of their own behind them: bank-neutral like the 48-cell
    // rows, but in the 49 KB of the dense TSC region — three workgroups per CU instead of two (TSC paint 2.64 -> 2.53 ms);
    // for CIC the same 40 KB as before and nothing on the lattice, where its base cells never leave their row, but
    // -3.5 % on the clustered set (paint 1.395 -> 1.345 ms).
#ifndef PMX_SPLIT_TSC
#define PMX_SPLIT_TSC 1
#endif
    static constexpr bool SPLIT = PMX_SPLIT_TSC && (S == 3 || S == 2);
    static constexpr int DMAIN = R0 * R1 * T2;
    static constexpr int DLDS = SPLIT ? DMAIN + R0 * R1 * (S - 1) : LDS;     // elements of the deposit region
    // element of row `row` (= a * R1 + b), column c of the deposit region
    __device__ static __forceinline__ int dat(int row, int c)
    {
        if (SPLIT) return c < T2 ? row * T2 + c : DMAIN + row * (S - 1) + (c - T2);
        return row * P2 + c;
    }
    // The readout's copy of the region: the same pitch where it costs no workgroup (PCS: paint 4.36 -> 3.90,
    // readout 2.59 -> 2.44 ms at 512^3), the dense one for TSC, whose readout loses more by running two
    // workgroups per CU instead of three than its LDS reads gain (1.64 -> 1.89 ms; paint 3.26 -> 2.75)
    // (and only for 8-byte regions: 48 floats are 192 bytes, no multiple of a bank row — PCS fp32 readout 2.48 -> 2.60)
    template <typename T> static constexpr int gpitch() { return (S >= 4 && sizeof(T) == 8) ? P2 : R2; }
    template <typename T> static constexpr int glds() { return R0 * R1 * gpitch<T>(); }
    // compact numbering of the halo (region minus the T0 x T1 x T2 box)
    static constexpr int NA = (S - 1) * R1 * R2;     // a >= T0
    static constexpr int NB = T0 * (S - 1) * R2;     // a < T0, b >= T1
    static constexpr int NC = T0 * T1 * (S - 1);     // a < T0, b < T1, c >= T2
    static constexpr int HALO = NA + NB + NC;
    __device__ static __forceinline__ int halo_index(int a, int b, int c)
    {
        if (a >= T0) return ((a - T0) * R1 + b) * R2 + c;
        if (b >= T1) return NA + (a * (S - 1) + (b - T1)) * R2 + c;
        return NA + NB + (a * T1 + b) * (S - 1) + (c - T2);
    }
    __device__ static __forceinline__ void halo_decode(int h, int *a, int *b, int *c)
    {
        if (h < NA) {
            *c = h % R2; int r = h / R2; *b = r % R1; *a = T0 + r / R1;
        } else if (h < NA + NB) {
            h -= NA;
            *c = h % R2; int r = h / R2; *b = T1 + r % (S - 1 > 0 ? S - 1 : 1); *a = r / (S - 1 > 0 ? S - 1 : 1);
        } else {
            h -= NA + NB;
            *c = T2 + h % (S - 1 > 0 ? S - 1 : 1); int r = h / (S - 1 > 0 ? S - 1 : 1); *b = r % T1; *a = r / T1;
        }
    }
};

}  // namespace pmx

struct pmx_binplan {
    pmx::BinGeom g;
    pmx_painter painter;        // geometry the plan was built for
    int64_t npart = 0;
    bool built = false;
    // device arrays
    int32_t *tid = nullptr;     // tile id per particle (-1 = touches no local cell)
    uint32_t *list = nullptr;   // particle indices, tile major
    size_t cap_part = 0;
    size_t cap_list = 0;        // entries of `list`: npart + slack (see slot_capacity)
    uint32_t *ctl = nullptr;    // ONE allocation for what every build clears: [flags: 4 words][nheavy: 4 words][counts ...] (one memset per build instead of three)
    uint32_t *counts = nullptr; // particles per tile; entry [ntiles] = particles that touch no local cell (= ctl + 8)
    int64_t *offsets = nullptr; // first list slot of every tile (ntiles + 2 entries): tile t owns
                                // slots [offsets[t], offsets[t+1]), of which counts[t] are used
    unsigned long long *cursor = nullptr;   // next free slot per tile while scattering
    size_t cap_tiles = 0;
    uint32_t *flags = nullptr;  // [0] != 0: the single-pass build ran out of slots in some tile
    uint32_t *host_flag = nullptr;          // pinned, device-visible: overflows seen so far
    void *halo = nullptr;       // staging of the halo cells: ntiles * Region<S>::HALO elements
    size_t cap_halo = 0;
    int form = -1;              // -1 / 0: tile kernels; 2: tile kernels with the chunk form of the single-pass rebuild
    // Rows whose order has no spatial coherence (catalogues in file order, shuffled sets): the
    // index list then sends every position gather, and every result store of readout, to a
    // sector of its own.  The plan keeps a copy of the positions in TILE ORDER instead (one
    // gather per build), paint and readout stream it, readout writes its results in tile order
    // and a last pass pulls them back through the inverse list.
    int sort_pref = -1;         // -1: decided from the coherence measured by the first build; 0 / 1: forced
    bool sorted = false;        // this plan carries the tile-ordered copy
    void *pos_copy = nullptr;   // cap_list rows of 3 position elements (the element type of `pos`)
    size_t cap_copy = 0;
    uint32_t *inv = nullptr;    // slot of every particle in the list (the inverse of `list`)
    size_t cap_inv = 0;
    double *out_sorted = nullptr;   // results of readout in list order
    size_t cap_out = 0;
    uint32_t *host_groups = nullptr;   // pinned: coherence counter of the count pass
    bool have_measure = false;         // host_groups holds the coherence of some earlier build of this plan
    int copy_elsize = 0;               // element size of the rows in pos_copy
    // Crowded tiles (halos, blobs): the tile kernels take the first `chunk` list entries of a tile;
    // what lies behind is cut into work items (tile, piece) for a second kernel, one workgroup each,
    // so that a tile with 100 x the mean population does not keep one workgroup busy for milliseconds
    uint64_t *heavy_items = nullptr;   // (tile << 20) | piece, piece >= 1
    size_t cap_heavy = 0;
    uint32_t *nheavy = nullptr;        // device: number of items of this build
    double *mstats = nullptr;          // device: [0] max |m| of the finite per-particle masses, [1] != 0: the floating-point kernels serve the batch
    const double *mass_stats_ext = nullptr;   // the same four words computed by the caller (pmx_mass_stats) for the masses of the next paint
    int exact = 0;                     // readout in the reference's arithmetic, operation by operation (bit-identical to pmx_readout)
    int deterministic = 0;             // paint through a dense int64 copy of the block: bit-reproducible
    // [r4] The halo merge of the last paint left to its consumer: the staged halos of every tile are still in `halo`
    // and belong to the canvas at `halo_canvas`; pmx_halo_merge adds them with atomics (what the paint itself does
    // otherwise), pmx_rowfft_halo adds them while the forward row pass of r2c loads the canvas (no sweep of their own).
    int halo_pending = 0;
    int halo_elsize = 0;               // element size of the staged values (that of the canvas)
    const void *halo_canvas = nullptr;
    void *dscratch = nullptr;          // that copy (+ the batch's exponent behind it)
    size_t cap_dscratch = 0;
    void *dhalo = nullptr;             // its halo staging (8 bytes per cell whatever the canvas type)
    size_t cap_dhalo = 0;
    int32_t chunk = 1 << 30;           // list entries per piece
    // history for the single-pass build: the slot ranges of the previous build of the same
    // geometry and particle count are reused (particles move little between time steps)
    bool have_history = false;
    uint32_t seen_overflows = 0;
    int distrust = 0, skip = 0;  // back-off after an overflow
    int slack = 0;               // [r6] how much room the slot ranges carry (slot_capacity): raised when a single-pass rebuild has overflowed
    bool last_reuse = false;     // the previous build was a single-pass (history) build
    uint32_t builds[2] = {0, 0}; // builds of this plan so far: single pass into the previous ranges / two passes (pmx_binplan_builds)
};

namespace pmx {

__device__ __forceinline__ int tile_ext(int d) { return d == 0 ? T0 : (d == 1 ? T1 : T2); }

// true modulo with a fast path for indices within one period of the box
__device__ __forceinline__ int wrap_fast(int i, int n)
{
    if (n <= 0) return i;
    int m = n;
    if (i < 0) { i += m; if (i < 0) { i %= m; if (i < 0) i += m; } }
    else if (i >= m) { i -= m; if (i >= m) i %= m; }
    return i;
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

// first stencil index of a particle along axis d in the local frame (see header);
// returns false if the particle touches no local cell along this axis
// (period and size as 32-bit values: every mesh axis is far below 2^31 cells, and the 64-bit
// compares of the painter's own fields cost several instructions each in the hot loops)
template <int KIND>
__device__ __forceinline__ bool local_base32(int period, int size, int I0, int *i0w)
{
    constexpr int S = Tuned<KIND>::S;
    int w = I0;
    if (period > 0) {
        // true modulo.  Indices within one period of the box — all but stray particles — take two selects; the
        // division sits behind ONE branch that a wave only enters if some lane is further out (the nested
        // branches this replaces cost the bin pass a dozen exec-mask instructions per axis whether taken or not)
        w += (w < 0) ? period : 0;
        w -= (w >= period) ? period : 0;
        if ((unsigned)w >= (unsigned)period) { w %= period; if (w < 0) w += period; }
        const bool in = w < size;
        *i0w = in ? w : w - period;
        return in || w >= period - (S - 1);
    }
    *i0w = w;
    return !(w < -(S - 1) || w >= size);
}

template <int KIND>
__device__ __forceinline__ bool local_base(const pmx_painter &p, int d, int I0, int *i0w)
{
    return local_base32<KIND>((int)p.period[d], (int)p.size[d], I0, i0w);
}

int plan_ensure(void **ptr, size_t *cap, size_t need);

}  // namespace pmx
