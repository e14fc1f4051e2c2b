// pmx_binned.hip — tile-binned paint / readout: the LDS-tiled form of the window
// kernels for 3-d meshes (the hot path of the PM cycle).
//
// Why: the direct scatter (pmx_window.hip) is bound by the chip-wide float-atomic
// rate (~1.3 TB/s of added bytes, MI355X_MICROARCH.md "Global float atomics"):
// CIC at 512^3 adds 8 x 8 B per particle = 8.6 GB -> 6.9 ms measured.  Here the
// mesh is cut into tiles of T^3 cells; the particles are binned by the tile of
// their window's base cell (an index list per tile: positions are NOT copied);
// one workgroup per tile accumulates its region (tile + S-1 halo cells on the high
// side of every axis) in LDS with ds_add and writes the cells it owns with plain
// coalesced stores (256-byte rows); the halo cells (23 % for CIC) are parked in a
// compact staging buffer and added to their owners by a second kernel with global
// atomics shaped as contiguous rows.  Tiles are 8 x 16 x 32 cells: long along the
// contiguous axis so that almost all halo traffic is whole rows.  Readout stages
// the tile region in LDS the same way and gathers from there.
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
#include <stdlib.h>
#include <type_traits>

#include "pmx_binplan.h"

// This file compiles as one unit (PMX_BINNED_PART undefined or 0: scripts/build_variant.sh, the resource test) or as
// four — the plan and its bin kernels (1), paint on double / float canvases (2 / 4: pmx_binned_paint.hip,
// pmx_binned_paint_f4.hip), readout (3: pmx_binned_readout.hip) —
// which the Makefile builds side by side: the tile kernels come in a dozen template forms each and took 100 s of the
// 105 s a clean build needs.  The parts differ only in which host launchers (and with them which kernel
// instantiations) they contain; kernels that are not templates are `static`, every part that launches one has its own.
#ifndef PMX_BINNED_PART
#define PMX_BINNED_PART 0
#endif
#define PMX_PART_PLAN (PMX_BINNED_PART == 0 || PMX_BINNED_PART == 1)
#define PMX_PART_PAINT (PMX_BINNED_PART == 0 || PMX_BINNED_PART == 2 || PMX_BINNED_PART == 4)    // 4: the float canvases
#define PMX_PART_READOUT (PMX_BINNED_PART == 0 || PMX_BINNED_PART == 3)

namespace pmx {

// DENSE: positions are a contiguous (n, 3) array.  A lane-per-particle load of 3 elements
// at a 24-byte stride touches three times the cache lines per instruction that a dense
// load does (the kernel is address-path bound, not bandwidth bound), so the block copies
// its 256 rows with 16-byte-per-lane loads into LDS and every lane picks its row there.
// slots reserved for a tile that held c particles: a quarter more plus a constant, so that the
// next build of slowly moving particles can reuse the ranges (single pass, see bin_onepass)
// slots a tile of c particles reserves for the single-pass rebuilds of the steps that follow.  level 0: a quarter + 64 of
// slack; [r6] a plan whose rebuilds have overflowed reserves more (level 1: half + 256, level 2: as many again + 1024):
// in a strongly clustered, fast-moving set SOME tile changes its population by more than a quarter in nearly every
// step, and every such step pays the repair — a second pass over the rows (scripts/nbody_long.py, 200 steps: 9 builds
// in 10 repaired late in the run; with the levels forced, steps 191-200 take 39.1 / 37.1 / 36.4 ms).  The list grows
// from 1.25 to 2 entries of 4 bytes per particle; only the used entries are ever read.
__host__ __device__ __forceinline__ int64_t slot_capacity(int64_t c, int level)
{
    return level <= 0 ? c + (c >> 2) + 64 : (level == 1 ? c + (c >> 1) + 256 : 2 * c + 1024);
}

// bucket of a particle at x: its tile (tiles in C order), or g.ntiles if it touches no local cell
template <int KIND>
__device__ __forceinline__ int64_t particle_bucket(const pmx_painter &p, const BinGeom &g, const double *x)
{
    constexpr int S = Tuned<KIND>::S;
    bool ok = true;
    int tt[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        double X = x[d] * p.scale[d] + p.translate[d];
        ok = ok && (fabs(X) < 1073741824.0);   // NaN / out of int range: dropped
        int I[S];
        double V[S];
        Tuned<KIND>::axis(ok ? X : 0.0, 0, 1.0, I, V);
        int i0w = 0;
        ok = ok && local_base<KIND>(p, d, I[0], &i0w);
        tt[d] = (int)((unsigned)(i0w + g.o[d]) / (unsigned)tile_ext(d));      // (never negative for a particle that counts)
    }
    // (32-bit: pmx_binplan_build refuses more than 2^31 buckets; the 64-bit multiplies cost four issue slots each)
    const int tb = (tt[0] * g.nt[1] + tt[1]) * g.nt[2] + tt[2];
    return ok ? (int64_t)tb : g.ntiles;
}

template <int NT>
__device__ __forceinline__ void scan_ranges(const uint32_t *counts, int64_t ntiles, int64_t *offsets, unsigned long long *cursor,
                                            uint32_t *zero, int level);

// MODE 0: count pass of the two-pass build: tid[i] = tile, counts[tile]++.
// MODE 1: single-pass build into the slot ranges of the previous build: the wave-aggregated
//         atomic returns the first free slot of the group; particle i goes to
//         list[offsets[tile] + slot] unless the tile's range is full (-> flags[0], host_flag).
// MODE 3: the repair after a single pass that overflowed: every single-pass form adds to counts[] before it looks at
//         the range, so the counts are exact; a gated bin_scan_kernel in front of this launch lays the new ranges out
//         from them (offsets, cursor), then every group of rows takes its slots from cursor[tile].
// Particles that touch no local cell go to bucket `ntiles`.  gate != NULL: do nothing unless
// *gate != 0 (the repair after a single-pass build is always enqueued and only
// runs if it overflowed — no host synchronisation).
template <int KIND, bool DENSE, int MODE, bool SORTP>
__global__ void __launch_bounds__(TBLOCK) bin_count_kernel(pmx_painter p, BinGeom g, DVec pos, int64_t n,
                                                           int32_t *tid, uint32_t *counts, uint32_t *flags,
                                                           const int64_t *offsets, uint32_t *list,
                                                           uint32_t *host_flag, const uint32_t *gate, uint32_t *inv_,
                                                           void *copy_, unsigned long long *cursor)
{
    // SORTP: the plan keeps a tile-ordered copy of the positions: the inverse list is recorded, and
    constexpr bool noagg = SORTP && MODE == 1;
    uint32_t *const inv = SORTP ? inv_ : nullptr;
    // noagg: the rows are known to be in no spatial order (the plan keeps a tile-ordered copy):
    // every lane then has a tile of its own and the search for equal tiles, up to 64 rounds of
    // ballots per chunk, finds nothing to merge — every lane adds for itself
    // particle chunks per trip: U independent load -> atomic chains per wave.  The single-pass
    // mode waits for its atomics to return: fewer registers / less LDS per block, more waves
    constexpr int U = MODE == 1 ? PMX_ONEPASS_U : 4;
    const int lane = threadIdx.x & 63;
    if (gate != nullptr && *gate == 0) return;
    // (MODE 3: the ranges were laid out from the exact counts by the gated bin_scan_kernel in front of this launch)
    __shared__ __align__(16) unsigned char stage[DENSE ? U * TBLOCK * 24 : 16];
    // Workgroups that run at the same time take chunks that are far apart in the array: rows in
    // lattice order put neighbouring chunks into the same few tiles, and the ~7000 resident
    // waves would queue on a few hundred tile counters (the smaller the mesh the fewer: the
    // rebuild ran at 15 ps/particle at 256^3 against 7 at 512^3).  Logical chunk c -> physical
    // chunk (c mod G) * Q + c / G: G = 256 interleaved streams.
    const int64_t nchunks = (n + TBLOCK * U - 1) / (TBLOCK * U);
    const int64_t G = nchunks < 256 ? nchunks : 256, Q = (nchunks + G - 1) / G;
    uint32_t nbreaks = 0, nsampled = 0;   // see below: one add per wave at the end, not one per chunk
    for (int64_t c = blockIdx.x; c < G * Q; c += gridDim.x) {
        const int64_t chunk = (c % G) * Q + c / G;
        if (chunk >= nchunks) continue;
        const int64_t base = chunk * (TBLOCK * U);
        double xin[U][3];
        if (DENSE) {
            const int rowb = 3 * pos.elsize;
            const int64_t left = n - base;
            const int nbytes = (int)((left < TBLOCK * U ? left : TBLOCK * U) * rowb);
            const char *src = pos.data + base * rowb;
            __syncthreads();
            const int n16 = nbytes & ~15;
            for (int off = threadIdx.x * 16; off < n16; off += TBLOCK * 16)
                *(uint4 *)(stage + off) = *(const uint4 *)(src + off);
            for (int off = n16 + threadIdx.x * 4; off < nbytes; off += TBLOCK * 4)
                *(uint32_t *)(stage + off) = *(const uint32_t *)(src + off);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < U; u++) {
                int row = u * TBLOCK + threadIdx.x;
                if (base + row < n) {
                    if (pos.elsize == 8) {
                        const double *r = (const double *)stage + 3 * row;
                        xin[u][0] = r[0]; xin[u][1] = r[1]; xin[u][2] = r[2];
                    } else {
                        const float *r = (const float *)stage + 3 * row;
                        xin[u][0] = r[0]; xin[u][1] = r[1]; xin[u][2] = r[2];
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                int64_t i = base + u * TBLOCK + threadIdx.x;
                if (i < n) { xin[u][0] = pos.get(i, 0); xin[u][1] = pos.get(i, 1); xin[u][2] = pos.get(i, 2); }
            }
        }
        int t[U];
        unsigned long long same[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int64_t i = base + u * TBLOCK + threadIdx.x;
            t[u] = -1;
            if (i < n) t[u] = (int)particle_bucket<KIND>(p, g, xin[u]);
            // wave-aggregated counting: find the lanes that share my tile (ballots only)
            same[u] = 0;
            {
                // noagg: at most NOAGG_ROUNDS groups are looked for, the lanes left over add for themselves.  (None
                // at all was the first form: 15.5 ms instead of ~1 for rows that are only PARTLY out of order — a
                // lattice with 2 cells of jitter, which the coherence measure already calls incoherent: up to 64
                // lanes of a wave, and every wave of the neighbourhood, queue on the same few tile counters.)
                constexpr int NOAGG_ROUNDS = 8;
                unsigned long long active = __ballot(t[u] >= 0);
                int rounds = noagg ? NOAGG_ROUNDS : 64;
                while (active && rounds-- > 0) {
                    int leader = __ffsll((long long)active) - 1;
                    int lt = __shfl(t[u], leader);
                    unsigned long long m = __ballot(t[u] == lt) & active;
                    if (t[u] == lt) same[u] = m;
                    active &= ~m;
                }
                if (noagg && t[u] >= 0 && same[u] == 0) same[u] = 1ull << lane;
            }
        }
        // coherence of the row order (flags[1]), sampled on one chunk in 32 (pseudo-randomly chosen): lanes whose tile differs
        // from their neighbour's — a few per 64 rows in lattice order, ~63 for rows in random order
        if ((((uint32_t)chunk * 2654435761u) >> 27) == 0) {      // (hashed: a regular stride would alias with the lattice)
            const int tprev = __shfl_up(t[0], 1);
            nbreaks += (uint32_t)__popcll(__ballot(lane > 0 && t[0] >= 0 && t[0] != tprev));
            nsampled += (uint32_t)__popcll(__ballot(t[0] >= 0));
        }

        // ONE atomicAdd instruction per chunk for the whole wave (the first lane of every
        // group adds the group's population)
        if (MODE == 0) {
            // no return value needed: the slots are handed out by bin_scatter_kernel
#pragma unroll
            for (int u = 0; u < U; u++) {
                int64_t i = base + u * TBLOCK + threadIdx.x;
                const int leader = t[u] >= 0 ? __ffsll((long long)same[u]) - 1 : lane;
                if (t[u] >= 0 && lane == leader) atomicAdd(&counts[t[u]], (uint32_t)__popcll(same[u]));
                if (i < n) tid[i] = t[u];
            }
        } else if (MODE == 3) {
            unsigned long long b[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int leader = t[u] >= 0 ? __ffsll((long long)same[u]) - 1 : lane;
                b[u] = 0;
                if (t[u] >= 0 && lane == leader) b[u] = atomicAdd(&cursor[t[u]], (unsigned long long)__popcll(same[u]));
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = base + u * TBLOCK + threadIdx.x;
                const int leader = t[u] >= 0 ? __ffsll((long long)same[u]) - 1 : lane;
                const unsigned long long bb = __shfl(b[u], leader);
                if (t[u] >= 0) {
                    const unsigned long long slot = bb + (unsigned long long)__popcll(same[u] & (((unsigned long long)1 << lane) - 1));
                    list[slot] = (uint32_t)i;
                    if (inv) inv[i] = (uint32_t)slot;
                }
            }
        } else {
            uint32_t b[U];
            int64_t o0[U], o1[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int leader = t[u] >= 0 ? __ffsll((long long)same[u]) - 1 : lane;
                b[u] = 0; o0[u] = 0; o1[u] = 0;
                if (t[u] >= 0 && lane == leader) {
                    b[u] = atomicAdd(&counts[t[u]], (uint32_t)__popcll(same[u]));
                    o0[u] = offsets[t[u]];
                    o1[u] = offsets[t[u] + 1];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                int64_t i = base + u * TBLOCK + threadIdx.x;
                const int leader = t[u] >= 0 ? __ffsll((long long)same[u]) - 1 : lane;
                const uint32_t bb = __shfl(b[u], leader);
                const int64_t start = __shfl(o0[u], leader), end = __shfl(o1[u], leader);
                if (t[u] >= 0) {
                    int64_t slot = start + bb + __popcll(same[u] & (((unsigned long long)1 << lane) - 1));
                    if (slot < end) {
                        list[slot] = (uint32_t)i;
                        if (inv) inv[i] = (uint32_t)slot;
                        if (SORTP && copy_) {
                            // the tile-ordered copy of the positions, written where the row lands (the
                            // values are in registers: no second, gathered pass over the positions)
                            if (pos.elsize == 8) {
                                double *c = (double *)copy_ + 3 * slot;
                                c[0] = xin[u][0]; c[1] = xin[u][1]; c[2] = xin[u][2];
                            } else {
                                float *c = (float *)copy_ + 3 * slot;
                                c[0] = (float)xin[u][0]; c[1] = (float)xin[u][1]; c[2] = (float)xin[u][2];
                            }
                        }
                    } else if (atomicOr(&flags[0], 1u) == 0) atomicAdd_system(host_flag, 1u);
                }
            }
        }
    }
    if (lane == 0 && nsampled) {
        atomicAdd(&flags[1], nbreaks);
        atomicAdd(&flags[2], nsampled);
    }
}

// Single-pass build for rows in a coherent order (the common case: lattice order, tile order, the
// order of the previous step), tile form without the tile-ordered copy.  bin_count_kernel<MODE 1> asks
// the tile's global counter once per wave and chunk: a returning global atomic for every ~32 rows, and
// a list piece of ~128 bytes written wherever the counter happened to stand — unaligned pieces from
// different CUs cost the memory side twice what aligned ones do (scripts/partial_line_bench.hip).
// Here a workgroup takes a BLOCK of BLOCK_ROWS (4096) consecutive rows, counts them per tile in an LDS table
// (a few dozen distinct tiles at most when the order is coherent), asks every tile's global counter
// ONCE, and then writes its rows of a tile as one contiguous piece (256 entries per tile in lattice
// order at 512^3), from one CU, so that the pieces meet in its L2 as whole lines.
// Tiles that find no room in the table are handled per wave as in bin_count_kernel.
#ifndef PMX_BLOCK_ITERS
#define PMX_BLOCK_ITERS 8
#endif
constexpr int BLOCK_ITERS = PMX_BLOCK_ITERS;                     // trips of TBLOCK * PMX_ONEPASS_U rows
constexpr int BLOCK_ROWS = TBLOCK * PMX_ONEPASS_U * BLOCK_ITERS;
constexpr int BLOCK_HT = 128;                       // entries of the LDS table (a power of two)
#if !defined(PMX_EXPERIMENT) || !defined(PMX_EXP_BINFLOOR)
#undef PMX_EXP_BINFLOOR
#define PMX_EXP_BINFLOOR 0                          // timing experiment (-DPMX_EXPERIMENT builds only), see bin_block_kernel
#endif
// 16-byte pieces of the TBLOCK * U dense rows from `base` on, one per thread and q: -> bytes requested
template <int NPRE, int U>
__device__ __forceinline__ int request_rows(const DVec &pos, int64_t n, int64_t base, uint4 (&pre)[NPRE])
{
    const int rowb = 3 * pos.elsize;
    const int64_t left = n - base;
    const int bytes = left <= 0 ? 0 : ((int)((left < TBLOCK * U ? left : TBLOCK * U) * rowb) & ~15);
    const char *src = pos.data + base * rowb;
#pragma unroll
    for (int q = 0; q < NPRE; q++) {
        const int off = (threadIdx.x + q * TBLOCK) * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (off < bytes) v = *(const uint4 *)(src + off);
        pre[q] = v;
    }
    return bytes;
}

template <int KIND, bool DENSE>
__global__ void __launch_bounds__(TBLOCK) bin_block_kernel(pmx_painter p, BinGeom g, DVec pos, int64_t n,
                                                           uint32_t *counts, uint32_t *flags, const int64_t *offsets,
                                                           uint32_t *list, uint32_t *host_flag)
{
    constexpr int U = PMX_ONEPASS_U;
    constexpr uint32_t EMPTY = 0xFFFFFFFFu, DIRECT = 0xFFu;
    __shared__ __align__(16) unsigned char stage[DENSE ? U * TBLOCK * 24 : 16];
    __shared__ uint32_t keys[BLOCK_HT], cnt[BLOCK_HT];
    __shared__ int64_t first[BLOCK_HT], last[BLOCK_HT];
    // where row (it, u) of thread tid goes: table entry (8 bits) | rank in the workgroup's piece (24 bits)
    // (in LDS, and the trips as a real loop: unrolled, the kernel was 80 KB of code with its scalar
    // registers spilled into vector lanes)
    __shared__ uint32_t where[BLOCK_ITERS * U * TBLOCK];
    const int lane = threadIdx.x & 63;
    uint32_t nbreaks = 0, nsampled = 0;
    for (int64_t blk = blockIdx.x; blk * BLOCK_ROWS < n; blk += gridDim.x) {
        const int64_t row0 = blk * BLOCK_ROWS;
        __syncthreads();
        for (int s = threadIdx.x; s < BLOCK_HT; s += TBLOCK) { keys[s] = EMPTY; cnt[s] = 0; }
        __syncthreads();
        // DENSE: 16-byte pieces of the next trip's rows, in flight while this trip computes
        constexpr int NPRE = (U * TBLOCK * 24 + TBLOCK * 16 - 1) / (TBLOCK * 16);
        uint4 pre[NPRE];
        int pre_bytes = 0;
        if (DENSE) pre_bytes = request_rows<NPRE, U>(pos, n, row0, pre);
#pragma unroll 1
        for (int it = 0; it < BLOCK_ITERS; it++) {
            const int64_t base = row0 + (int64_t)it * (TBLOCK * U);
            double xin[U][3];
            if (base < n) {
                if (DENSE) {
                    // the rows of this trip were requested during the previous one (pre[]): they go to LDS,
                    // the next trip's rows are requested, and only then does this trip compute
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < NPRE; q++) {
                        const int off = (threadIdx.x + q * TBLOCK) * 16;
                        if (off < pre_bytes) *(uint4 *)(stage + off) = pre[q];
                    }
                    {
                        // (the last bytes of an array whose size is not a multiple of 16)
                        const int rowb = 3 * pos.elsize;
                        const int64_t left = n - base;
                        const int nbytes = (int)((left < TBLOCK * U ? left : TBLOCK * U) * rowb);
                        const char *src = pos.data + base * rowb;
                        for (int off = (nbytes & ~15) + threadIdx.x * 4; off < nbytes; off += TBLOCK * 4)
                            *(uint32_t *)(stage + off) = *(const uint32_t *)(src + off);
                    }
                    __syncthreads();
                    if (it + 1 < BLOCK_ITERS) pre_bytes = request_rows<NPRE, U>(pos, n, base + TBLOCK * U, pre);
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        int row = u * TBLOCK + threadIdx.x;
                        if (base + row < n) {
                            if (pos.elsize == 8) {
                                const double *r = (const double *)stage + 3 * row;
                                xin[u][0] = r[0]; xin[u][1] = r[1]; xin[u][2] = r[2];
                            } else {
                                const float *r = (const float *)stage + 3 * row;
                                xin[u][0] = r[0]; xin[u][1] = r[1]; xin[u][2] = r[2];
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        int64_t i = base + u * TBLOCK + threadIdx.x;
                        if (i < n) { xin[u][0] = pos.get(i, 0); xin[u][1] = pos.get(i, 1); xin[u][2] = pos.get(i, 2); }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = base + u * TBLOCK + threadIdx.x;
                const int t = i < n ? (int)particle_bucket<KIND>(p, g, xin[u]) : -1;
                // the lanes that share my tile (ballots only)
                unsigned long long same = 0, active = __ballot(t >= 0);
                while (active) {
                    int leader = __ffsll((long long)active) - 1;
                    int lt = __shfl(t, leader);
                    unsigned long long m = __ballot(t == lt) & active;
                    if (t == lt) same = m;
                    active &= ~m;
                }
                if (u == 0 && ((((uint32_t)(base / (TBLOCK * U))) * 2654435761u) >> 27) == 0) {   // coherence sample, as in bin_count_kernel
                    const int tprev = __shfl_up(t, 1);
                    nbreaks += (uint32_t)__popcll(__ballot(lane > 0 && t >= 0 && t != tprev));
                    nsampled += (uint32_t)__popcll(__ballot(t >= 0));
                }
                const int leader = t >= 0 ? __ffsll((long long)same) - 1 : lane;
                uint32_t w = 0, bd = 0;
#if PMX_EXP_BINFLOOR
                // timing experiment (wrong lists): no table, no counters — what positions -> tile ids -> list costs
                where[(it * U + u) * TBLOCK + threadIdx.x] = t >= 0 ? (uint32_t)(t & 0xFFFFFF) : EMPTY;
                if (PMX_EXP_BINFLOOR > 1) continue;
#endif
                if (t >= 0 && lane == leader) {
                    // the group's entry of the table and its first rank there
                    uint32_t h = ((uint32_t)t * 2654435761u) >> 25;               // 7 bits: BLOCK_HT = 128
                    uint32_t e = DIRECT;
                    for (int probe = 0; probe < 8; probe++) {
                        const uint32_t k = atomicCAS(&keys[h], EMPTY, (uint32_t)t);
                        if (k == EMPTY || k == (uint32_t)t) { e = h; break; }
                        h = (h + 1) & (BLOCK_HT - 1);
                    }
                    if (e != DIRECT) w = (e << 24) | atomicAdd(&cnt[e], (uint32_t)__popcll(same));
                    else {
                        // no room in the table: this group asks the global counter itself
                        w = DIRECT << 24;
                        bd = atomicAdd(&counts[t], (uint32_t)__popcll(same));
                    }
                }
                w = __shfl(w, leader);
                const uint32_t rank = (uint32_t)__popcll(same & (((unsigned long long)1 << lane) - 1));
                if (t < 0) where[(it * U + u) * TBLOCK + threadIdx.x] = EMPTY;
                else if ((w >> 24) == DIRECT) {
                    // written at once (rare): list[offsets[t] + b + rank]
                    bd = __shfl(bd, leader);
                    const int64_t slot = offsets[t] + (int64_t)bd + rank;
                    if (slot < offsets[t + 1]) list[slot] = (uint32_t)i;
                    else if (atomicOr(&flags[0], 1u) == 0) atomicAdd_system(host_flag, 1u);
                    where[(it * U + u) * TBLOCK + threadIdx.x] = EMPTY;
                } else where[(it * U + u) * TBLOCK + threadIdx.x] = w + rank;       // (rank < 64, the count below 2^24: no carry into the entry bits)
            }
        }
        __syncthreads();
        // one request per tile of the block to its global counter
        for (int s = threadIdx.x; s < BLOCK_HT; s += TBLOCK) {
            if (keys[s] != EMPTY) {
                const uint32_t t = keys[s];
                const uint32_t b = atomicAdd(&counts[t], cnt[s]);
                first[s] = offsets[t] + b;
                last[s] = offsets[t + 1];
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int it = 0; it < BLOCK_ITERS; it++) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t w = where[(it * U + u) * TBLOCK + threadIdx.x];
                if (w == EMPTY) continue;
                const int64_t i = row0 + (int64_t)it * (TBLOCK * U) + u * TBLOCK + threadIdx.x;
#if PMX_EXP_BINFLOOR > 1
                list[i] = (uint32_t)i + (w & 1);      // sequential: the floor of reading rows and writing a list
                continue;
#endif
                const uint32_t e = w >> 24;
                const int64_t slot = first[e] + (w & 0xFFFFFFu);
                if (slot < last[e]) list[slot] = (uint32_t)i;
                else if (atomicOr(&flags[0], 1u) == 0) atomicAdd_system(host_flag, 1u);
            }
        }
    }
    if (lane == 0 && nsampled) {
        atomicAdd(&flags[1], nbreaks);
        atomicAdd(&flags[2], nsampled);
    }
}

// a position row in registers: PE = 4 / 8: dense rows of three floats / doubles; 0: any strides and element size
template <int PE> struct PosRow { double x[3]; };
template <> struct PosRow<4> { float x[3]; };
template <int PE> __device__ __forceinline__ PosRow<PE> pos_row(const DVec &pos, int64_t i)
{
    PosRow<PE> r;
    if constexpr (PE == 4) {
        // (three dwords in one instruction: rows are 4-byte aligned)
        const float *q = (const float *)(pos.data + i * 12);
        r.x[0] = q[0]; r.x[1] = q[1]; r.x[2] = q[2];
    } else if constexpr (PE == 8) {
        const double *q = (const double *)(pos.data + i * 24);
        r.x[0] = q[0]; r.x[1] = q[1]; r.x[2] = q[2];
    } else {
        r.x[0] = pos.get(i, 0); r.x[1] = pos.get(i, 1); r.x[2] = pos.get(i, 2);
    }
    return r;
}

// ---- [r5] the block form of the single-pass rebuild for dense rows, as a loop of its own ----------------------------
// bin_block_kernel above is bound by instruction issue (~250 vector + scalar instructions per 64 rows; the 3.8 GB it
// moves would take 0.68 ms, it takes 0.95-1.0, and 0.9 for 12-byte rows that move 2.1 GB).  Most of those instructions
// are not the binning: rows staged through LDS in 16-byte pieces behind two workgroup barriers per trip, bounds tests
// on every piece, the element size of the rows looked up per component, the general wrap of the block on every axis.
// Here every lane loads its own rows (PE = 4: one 12-byte load per row; PE = 8: a 16- and an 8-byte one — the 64 rows
// of a wave are one contiguous piece of memory either way), the next trip's rows are in flight while this trip's are
// binned, the trips need no barrier, and WHOLE (the block is the whole periodic mesh: the one-rank case) is the
// launcher's to know.  Same table, same slots, same flags: the list is a permutation of the other form's within each tile.
template <int PE> __device__ __forceinline__ PosRow<PE> dense_row(const char *data, int64_t i)
{
    PosRow<PE> r;
    if constexpr (PE == 4) {
        const float *q = (const float *)(data + i * 12);
        r.x[0] = q[0]; r.x[1] = q[1]; r.x[2] = q[2];
    } else {
        const double *q = (const double *)(data + i * 24);
        r.x[0] = q[0]; r.x[1] = q[1]; r.x[2] = q[2];
    }
    return r;
}

// bucket of a row: its tile, or g.ntiles if it touches no local cell (particle_bucket); WHOLE: every axis is the whole periodic mesh, a multiple of the tile
template <int KIND, bool WHOLE, typename R>
__device__ __forceinline__ int row_tile(const pmx_painter &p, const BinGeom &g, const R &row)
{
    if constexpr (!WHOLE) {
        // blocks of any shape (slab / pencil ranks): particle_bucket without the weights it computes on the way — the
        // first cell of the stencil is all a tile id needs (12-byte rows: 0.69 -> ... ms per 512^3 rows, the whole-mesh
        // form 0.59)
        bool ok = true;
        int tt[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double X = (double)row.x[d] * p.scale[d] + p.translate[d];
            ok = ok && (fabs(X) < 1073741824.0);               // NaN / out of int range: dropped
            const int I0 = Tuned<KIND>::first(ok ? X : 0.0);
            const int period = (int)p.period[d];
            if (g.o[d] == 0 && period == (int)p.size[d]) {
                // (uniform: an axis that IS the whole periodic mesh — two of three on a slab rank — wraps as in the
                // whole-mesh form and cannot miss the block)
                int w = I0;
                w += (w < 0) ? period : 0;
                w -= (w >= period) ? period : 0;
                if ((unsigned)w >= (unsigned)period) { w %= period; if (w < 0) w += period; }
                tt[d] = (int)((unsigned)w / (unsigned)tile_ext(d));
                continue;
            }
            int i0w = 0;
            ok = ok && local_base<KIND>(p, d, I0, &i0w);
            tt[d] = (int)((unsigned)(i0w + g.o[d]) / (unsigned)tile_ext(d));
        }
        const int tb = (tt[0] * g.nt[1] + tt[1]) * g.nt[2] + tt[2];
        return ok ? tb : (int)g.ntiles;
    } else {
        bool ok = true;
        int tt[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double X = (double)row.x[d] * p.scale[d] + p.translate[d];
            ok = ok && (fabs(X) < 1073741824.0);               // NaN / out of int range: dropped
            int w = Tuned<KIND>::first(ok ? X : 0.0);
            const int period = (int)p.period[d];
            w += (w < 0) ? period : 0;
            w -= (w >= period) ? period : 0;
            if ((unsigned)w >= (unsigned)period) { w %= period; if (w < 0) w += period; }
            tt[d] = (int)((unsigned)w / (unsigned)tile_ext(d));
        }
        const int tb = (tt[0] * g.nt[1] + tt[1]) * g.nt[2] + tt[2];
        return ok ? tb : (int)g.ntiles;
    }
}

#ifndef PMX_LEAN_U
#define PMX_LEAN_U 4
#endif
template <int KIND, int PE, bool WHOLE>
__device__ __forceinline__ void lean_blocks(const pmx_painter &p, const BinGeom &g, const DVec &pos, int64_t n,
                                            uint32_t *counts, uint32_t *flags, const int64_t *offsets,
                                            uint32_t *list, uint32_t *host_flag)
{
    constexpr int U = PMX_LEAN_U, BLOCK_ITERS = BLOCK_ROWS / (TBLOCK * U);
    static_assert(BLOCK_ITERS * TBLOCK * U == BLOCK_ROWS, "rows of a block");
    constexpr uint32_t EMPTY = 0xFFFFFFFFu, DIRECT = 0xFFu;
    __shared__ uint32_t keys[BLOCK_HT], cnt[BLOCK_HT];
    __shared__ int64_t first[BLOCK_HT], last[BLOCK_HT];
    __shared__ uint32_t where[BLOCK_ITERS * U * TBLOCK];
    const int lane = threadIdx.x & 63;
    // row (it, u) of this lane within a trip: the U chunks of 64 rows a wave takes follow each other in memory, and its
    // U requests reach a tile's counter back to back: the list keeps the rows of a wave in their order (with the chunks
    // of a wave TBLOCK rows apart the tile kernels read rows 0-63, 256-319, 512-575, ... of a trip in turn: PCS readout
    // 1.63 -> 1.84 ms at U = 4)
    const int wave0 = (int)(threadIdx.x >> 6) * (U * 64) + lane;
    auto lrow = [&](int u) { return wave0 + u * 64; };
    uint32_t nbreaks = 0, nsampled = 0;
    const char *data = pos.data;
    for (int64_t blk = blockIdx.x; blk * BLOCK_ROWS < n; blk += gridDim.x) {
        const int64_t row0 = blk * BLOCK_ROWS;
        __syncthreads();
        for (int s = threadIdx.x; s < BLOCK_HT; s += TBLOCK) { keys[s] = EMPTY; cnt[s] = 0; }
        __syncthreads();
        PosRow<PE> next[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = row0 + lrow(u);
            next[u] = dense_row<PE>(data, i < n ? i : n - 1);           // (a row beyond the end reads the last one: no load behind a branch)
        }
#pragma unroll 1
        for (int it = 0; it < BLOCK_ITERS; it++) {
            const int64_t base = row0 + (int64_t)it * (TBLOCK * U);
            PosRow<PE> row[U];
#pragma unroll
            for (int u = 0; u < U; u++) row[u] = next[u];
            if (it + 1 < BLOCK_ITERS) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int64_t i = base + TBLOCK * U + lrow(u);
                    next[u] = dense_row<PE>(data, i < n ? i : n - 1);
                }
            }
            // the U rows of a lane side by side: their table requests travel together (two LDS round trips per trip —
            // the claim of an entry, the add to its counter — instead of two per row: with the rows loaded leanly the
            // chain of dependent LDS operations behind the leaders is what the trip waits for)
            int t[U], leader[U];
            unsigned long long same[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = base + lrow(u);
                t[u] = i < n ? row_tile<KIND, WHOLE>(p, g, row[u]) : -1;
#if defined(PMX_EXPERIMENT) && defined(PMX_EXP_LEANBIN)
                // timing experiments (wrong lists): 1: no match loop, no table: the list written sequentially;
                // 2: no list either (the tile ids summed into the coherence counter); 3: the rows summed, no tile
                if (PMX_EXP_LEANBIN == 1) { where[(it * U + u) * TBLOCK + threadIdx.x] = t[u] >= 0 ? (uint32_t)(t[u] & 1) : EMPTY; continue; }
                if (PMX_EXP_LEANBIN == 2) { nbreaks += (uint32_t)t[u]; where[(it * U + u) * TBLOCK + threadIdx.x] = EMPTY; continue; }
                if (PMX_EXP_LEANBIN == 3) { nbreaks += (uint32_t)(int)((double)row[u].x[0] + (double)row[u].x[1] + (double)row[u].x[2]); where[(it * U + u) * TBLOCK + threadIdx.x] = EMPTY; continue; }
#endif
                // the lanes that share my tile (ballots only)
                unsigned long long sm = 0, active = __ballot(t[u] >= 0);
                while (active) {
                    const int ld = __ffsll((long long)active) - 1;
                    const int lt = __shfl(t[u], ld);
                    const unsigned long long m = __ballot(t[u] == lt) & active;
                    if (t[u] == lt) sm = m;
                    active &= ~m;
                }
                same[u] = sm;
                leader[u] = t[u] >= 0 ? __ffsll((long long)sm) - 1 : lane;
                if (u == 0 && ((((uint32_t)(base / (TBLOCK * U))) * 2654435761u) >> 27) == 0) {   // coherence sample, as in bin_count_kernel
                    const int tprev = __shfl_up(t[u], 1);
                    nbreaks += (uint32_t)__popcll(__ballot(lane > 0 && t[u] >= 0 && t[u] != tprev));
                    nsampled += (uint32_t)__popcll(__ballot(t[u] >= 0));
                }
            }
#if defined(PMX_EXPERIMENT) && defined(PMX_EXP_LEANBIN)
            continue;
#endif
            bool lead[U];
            uint32_t h[U], k[U], e[U], w[U], bd[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                lead[u] = t[u] >= 0 && lane == leader[u];
                h[u] = ((uint32_t)t[u] * 2654435761u) >> 25;               // 7 bits: BLOCK_HT = 128
                k[u] = 0; w[u] = 0; bd[u] = 0; e[u] = DIRECT;
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (lead[u]) k[u] = atomicCAS(&keys[h[u]], EMPTY, (uint32_t)t[u]);
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (lead[u]) {
                    if (k[u] == EMPTY || k[u] == (uint32_t)t[u]) e[u] = h[u];
                    else {
                        // (the entry belongs to another tile: the next ones, as many as seven)
                        uint32_t hh = h[u];
                        for (int probe = 1; probe < 8; probe++) {
                            hh = (hh + 1) & (BLOCK_HT - 1);
                            const uint32_t kk = atomicCAS(&keys[hh], EMPTY, (uint32_t)t[u]);
                            if (kk == EMPTY || kk == (uint32_t)t[u]) { e[u] = hh; break; }
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (lead[u]) {
                    if (e[u] != DIRECT) w[u] = (e[u] << 24) | atomicAdd(&cnt[e[u]], (uint32_t)__popcll(same[u]));
                    else {
                        // no room in the table: this group asks the global counter itself
                        w[u] = DIRECT << 24;
                        bd[u] = atomicAdd(&counts[t[u]], (uint32_t)__popcll(same[u]));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) w[u] = __shfl(w[u], leader[u]);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = base + lrow(u);
                const uint32_t rank = (uint32_t)__popcll(same[u] & (((unsigned long long)1 << lane) - 1));
                uint32_t wh = EMPTY;
                if (t[u] >= 0) {
                    if ((w[u] >> 24) == DIRECT) {
                        // (rare) written at once: list[offsets[t] + b + rank]
                        const uint32_t b0 = __shfl(bd[u], leader[u]);
                        if (list != nullptr) {
                            const int64_t slot = offsets[t[u]] + (int64_t)b0 + rank;
                            if (slot < offsets[t[u] + 1]) list[slot] = (uint32_t)i;
                            else if (atomicOr(&flags[0], 1u) == 0) atomicAdd_system(host_flag, 1u);
                        }
                    } else wh = w[u] + rank;
                }
                where[(it * U + u) * TBLOCK + threadIdx.x] = wh;
            }
        }
        __syncthreads();
        // one request per tile of the block to its global counter
        for (int s = threadIdx.x; s < BLOCK_HT; s += TBLOCK) {
            if (keys[s] != EMPTY) {
                const uint32_t t = keys[s];
                const uint32_t b = atomicAdd(&counts[t], cnt[s]);
                if (list != nullptr) {
                    first[s] = offsets[t] + b;
                    last[s] = offsets[t + 1];
                }
            }
        }
        if (list == nullptr) continue;        // (the count pass of a two-pass build: no ranges yet, nothing to write)
        __syncthreads();
#pragma unroll 2
        for (int it = 0; it < BLOCK_ITERS; it++) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t w = where[(it * U + u) * TBLOCK + threadIdx.x];
                if (w == EMPTY) continue;
                const int64_t i = row0 + (int64_t)it * (TBLOCK * U) + lrow(u);
#if defined(PMX_EXPERIMENT) && defined(PMX_EXP_LEANBIN)
                list[i] = (uint32_t)i + (w & 1);
                continue;
#endif
                const uint32_t e = w >> 24;
                const int64_t slot = first[e] + (w & 0xFFFFFFu);
                if (slot < last[e]) list[slot] = (uint32_t)i;
                else if (atomicOr(&flags[0], 1u) == 0) atomicAdd_system(host_flag, 1u);
            }
        }
    }
    if (lane == 0 && (nsampled || nbreaks == 0xFFFFFFFFu)) {
        atomicAdd(&flags[1], nbreaks);
        atomicAdd(&flags[2], nsampled);
    }
}

// [r6] waves per SIMD the kernel is held to.  Positions in float need 83 registers unbounded — five waves where 80
// allow six; held there (no spill) config 3's bin pass goes 0.61 -> 0.56 ms, CIC f4 0.57 -> 0.54.  Positions in double
// need 93, and held to 80 they spill: 0.76 -> 0.85-0.90 (to 64: 1.3).  `profiles/r06_lean_waves_ab.txt`.
#ifndef PMX_LEAN_WAVES
#define PMX_LEAN_WAVES 0
#endif
template <int KIND, int PE, bool WHOLE>
__global__ void __launch_bounds__(TBLOCK, (PMX_LEAN_WAVES ? PMX_LEAN_WAVES : (PE == 4 ? 6 : 1))) bin_lean_kernel(pmx_painter p, BinGeom g, DVec pos, int64_t n,
                                                          uint32_t *counts, uint32_t *flags, const int64_t *offsets,
                                                          uint32_t *list, uint32_t *host_flag)
{
    lean_blocks<KIND, PE, WHOLE>(p, g, pos, n, counts, flags, offsets, list, host_flag);
}

template <int NT>
__device__ __forceinline__ void scan_ranges(const uint32_t *counts, int64_t ntiles, int64_t *offsets, unsigned long long *cursor,
                                            uint32_t *zero, int level);

// [r5] The repair of a single pass that overflowed, for the rows the lean form takes.  The counts are exact (every
// single-pass form adds before it looks at the range): a gated bin_scan_kernel lays the new ranges out from them and
// zeroes the counters, then this gated launch fills the ranges block by block like any rebuild.  (bin_count_kernel<MODE 3>, which does the same
// with one cursor request per wave and tile, took 9.7 ms for the 512^3 rows where this takes 2.6: scripts/overflow_probe.py.)
template <int KIND, int PE, bool WHOLE>
__global__ void __launch_bounds__(TBLOCK) bin_repair_lean_kernel(pmx_painter p, BinGeom g, DVec pos, int64_t n,
                                                                 uint32_t *counts, uint32_t *flags, int64_t *offsets,
                                                                 unsigned long long *cursor, uint32_t *list,
                                                                 uint32_t *host_flag, const uint32_t *gate)
{
    // ([r6] the new ranges are laid out — and the counters zeroed — by a gated bin_scan_kernel launched in front of
    // this one: until round 5 workgroup 0 did it here while the others spun on a flag, which is only safe if workgroup 0
    // is resident whenever they are; nothing guarantees that on a device shared with other streams or ranks)
    if (*gate == 0) return;
    lean_blocks<KIND, PE, WHOLE>(p, g, pos, n, counts, flags, offsets, list, host_flag);
}

// exclusive scan of slot_capacity(counts) -> offsets[nbuckets+1]; one workgroup of NT threads
template <int NT>
__device__ __forceinline__ void scan_ranges(const uint32_t *counts, int64_t ntiles, int64_t *offsets, unsigned long long *cursor,
                                            uint32_t *zero, int level)
{
    __shared__ int64_t sh[NT];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < ntiles; base += NT) {
        int64_t i = base + threadIdx.x;
        int64_t v = i < ntiles ? slot_capacity(counts[i], level) : 0;
        if (zero != nullptr && i < ntiles) zero[i] = 0;       // (the counters start the pass that fills the ranges from 0)
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < NT; off <<= 1) {
            int64_t t = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        int64_t incl = sh[threadIdx.x];
        if (i < ntiles) {
            offsets[i] = carry + incl - v;
            cursor[i] = (unsigned long long)(carry + incl - v);
        }
        __syncthreads();
        if (threadIdx.x == NT - 1) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[ntiles] = carry;
}

static __global__ void __launch_bounds__(1024) bin_scan_kernel(const uint32_t *counts, int64_t ntiles, int64_t *offsets,
                                                        unsigned long long *cursor, const uint32_t *gate, uint32_t *zero, int level)
{
    if (gate != nullptr && *gate == 0) return;
    scan_ranges<1024>(counts, ntiles, offsets, cursor, zero, level);
}

static __global__ void __launch_bounds__(TBLOCK) bin_scatter_kernel(const int32_t *tid, unsigned long long *cursor, int64_t n,
                                                             uint32_t *list, const uint32_t *gate, uint32_t *inv)
{
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    if (gate != nullptr && *gate == 0) return;
    for (int64_t base = blockIdx.x * (int64_t)(TBLOCK * U); base < n; base += (int64_t)gridDim.x * TBLOCK * U) {
        int t[U];
        unsigned long long same[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int64_t i = base + u * TBLOCK + threadIdx.x;
            t[u] = i < n ? tid[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            same[u] = 0;
            unsigned long long active = __ballot(t[u] >= 0);
            while (active) {
                int leader = __ffsll((long long)active) - 1;
                int lt = __shfl(t[u], leader);
                unsigned long long m = __ballot(t[u] == lt) & active;
                if (t[u] == lt) same[u] = m;
                active &= ~m;
            }
            b[u] = 0;
            const int leader = t[u] >= 0 ? __ffsll((long long)same[u]) - 1 : lane;
            if (t[u] >= 0 && lane == leader) b[u] = atomicAdd(&cursor[t[u]], (unsigned long long)__popcll(same[u]));
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            int64_t i = base + u * TBLOCK + threadIdx.x;
            const int leader = t[u] >= 0 ? __ffsll((long long)same[u]) - 1 : lane;
            unsigned long long bb = __shfl(b[u], leader);
            if (t[u] >= 0) {
                const unsigned long long slot = bb + (unsigned long long)__popcll(same[u] & (((unsigned long long)1 << lane) - 1));
                list[slot] = (uint32_t)i;
                if (inv) inv[i] = (uint32_t)slot;
            }
        }
    }
}

// positions in list (tile) order: copy[slot] = pos[list[slot]] for the used slots of every bucket.
// One lane per 4-byte word of the output: the words of a row are read by adjacent lanes, the output
// is written in order.  WPR: words per row (3 floats: 3, 3 doubles: 6).
template <int WPR>
__global__ void __launch_bounds__(TBLOCK) sort_copy_kernel(const uint32_t *list, const int64_t *offsets,
                                                           const uint32_t *counts, int64_t nbuckets, DVec pos, uint32_t *copy,
                                                           const uint32_t *gate)
{
    constexpr int WPE = WPR / 3;                 // words per element
    if (gate != nullptr && *gate == 0) return;   // the single-pass rebuild wrote the copy itself
    for (int64_t b = blockIdx.x; b < nbuckets; b += gridDim.x) {
        const int64_t start = offsets[b];
        const int nwords = (int)counts[b] * WPR;
        // four independent list -> row gathers in flight per lane (the copy is bound by their latency)
        for (int w0 = threadIdx.x; w0 < nwords; w0 += 4 * TBLOCK) {
            uint32_t v[4];
            int64_t i[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int w = w0 + u * TBLOCK;
                i[u] = w < nwords ? (int64_t)list[start + w / WPR] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int w = w0 + u * TBLOCK;
                const int k = w % WPR;
                if (i[u] >= 0) v[u] = *(const uint32_t *)(pos.data + i[u] * pos.stride0 + (k / WPE) * pos.stride1 + (k % WPE) * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int w = w0 + u * TBLOCK;
                if (i[u] >= 0) copy[start * WPR + w] = v[u];
            }
        }
    }
}

// out[i] = sorted[inv[i]]: the results of a readout in list order back into row order
static __global__ void __launch_bounds__(TBLOCK) unsort_kernel(const double *sorted, const uint32_t *inv, int64_t n, DVec out)
{
    const int64_t stride = (int64_t)gridDim.x * TBLOCK;
    for (int64_t i0 = blockIdx.x * (int64_t)TBLOCK + threadIdx.x; i0 < n; i0 += 4 * stride) {
        uint32_t s[4];
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) s[u] = i0 + u * stride < n ? inv[i0 + u * stride] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = sorted[s[u]];            // four independent gathers in flight
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (i0 + u * stride < n) out.set(i0 + u * stride, 0, v[u]);
    }
}

__device__ __forceinline__ void tile_coords(const BinGeom &g, int64_t tile, int *t)
{
    int64_t r = tile;
    t[2] = (int)(r % g.nt[2]); r /= g.nt[2];
    t[1] = (int)(r % g.nt[1]); r /= g.nt[1];
    t[0] = (int)r;
}

// region cell (a, b, c) of tile t -> canvas byte offset; false if it lies outside the block
__device__ __forceinline__ bool region_cell(const pmx_painter &p, const BinGeom &g, const int *t, int a, int b, int c,
                                            int64_t *goff)
{
    int loc[3] = {a, b, c};
    int64_t off = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        int l = t[d] * tile_ext(d) - g.o[d] + loc[d];
        int gidx = wrap_near(l, p.period[d]);
        if (gidx < 0 || gidx >= p.size[d]) return false;
        off += gidx * p.strides[d];
    }
    *goff = off;
    return true;
}

// The same per axis, once per tile: tab[d0 + loc] = byte offset of region coordinate loc along axis d
// (region_cell's gidx * stride), or -1 where it lies outside the block; a cell's offset is then the sum
// of three table entries, valid iff none of them is negative.  (region_cell costs ~40 instructions per
// cell — three wraps, six compares, three 64-bit multiplies — and the staging / flush loops touch 1.2-1.5
// cells per particle: a fifth of the instructions of the CIC kernels.)  OWNED: unwrapped coordinates
// outside [0, size) count as outside too (the cells of the owned box that exist only on non-periodic or
// slab axes).  R0 + R1 + R2 entries of LDS; all threads of the workgroup call it, then synchronise.
// (Used by the readout: CIC 1.20 -> 1.165 ms, TSC 1.73 -> 1.64, PCS 2.89 -> 2.65, f4 1.37 -> 1.28.  The same in
// the paint flush and halo_merge measured no gain for CIC / PCS and +6 % on TSC paint: not used there.)
template <int S, bool OWNED>
__device__ __forceinline__ void region_tables(const pmx_painter &p, const BinGeom &g, const int *t, int64_t *tab, int nthreads)
{
    using Rg = Region<S>;
    constexpr int R0 = Rg::R0, R1 = Rg::R1, R2 = Rg::R2;
    for (int i = threadIdx.x; i < R0 + R1 + R2; i += nthreads) {
        const int d = i < R0 ? 0 : (i < R0 + R1 ? 1 : 2);
        const int loc = i - (d == 0 ? 0 : (d == 1 ? R0 : R0 + R1));
        const int l = t[d] * tile_ext(d) - g.o[d] + loc;
        const int gidx = wrap_near(l, p.period[d]);
        bool ok = !(gidx < 0 || gidx >= p.size[d]);
        if (OWNED) ok = ok && l >= 0 && l < p.size[d];
        tab[i] = ok ? (int64_t)gidx * p.strides[d] : (int64_t)-1;
    }
}

// per-particle setup shared by paint and readout: weights and local base of the stencil
// all three axes are the whole periodic mesh (the one-rank case): see particle_setup
__device__ __forceinline__ bool whole_mesh(const pmx_painter &p, const BinGeom &g)
{
    bool w = true;
#pragma unroll
    for (int d = 0; d < 3; d++) w = w && g.o[d] == 0 && (int)p.period[d] == (int)p.size[d];
    return w;
}

// a list entry whose particle no longer lies in the tile's region: the plan was built for other positions (rows rewritten
// in place behind the cache's back).  The particle is skipped — never an access outside the region — and counted where
// the host can see it (pmx_binplan_stale; pmesh_amd.window warns).  Only blocks that are not the whole periodic mesh can
// tell: there the local base of a particle is its cell modulo the tile (see particle_setup), always inside.
__device__ __forceinline__ void stale_row(const BinGeom &g)
{
    if (g.stale) atomicAdd_system(g.stale, 1u);
}

template <int KIND, bool WHOLE = false>
__device__ __forceinline__ void particle_setup(const pmx_painter &p, const BinGeom &g, const int *t,
                                               const double *x, double (*V)[Tuned<KIND>::S], int *lb)
{
    constexpr int S = Tuned<KIND>::S;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        double X = x[d] * p.scale[d] + p.translate[d];
        int I[S];
        Tuned<KIND>::axis(X, p.order[d], p.scale[d], I, V[d]);
        const int per = (int)p.period[d], siz = (int)p.size[d];      // (32-bit compares: see local_base32)
        // WHOLE: the caller has checked the condition for all three axes (whole_mesh()): no branch, and the
        // general path is not even in the instruction stream of the loop
        if (WHOLE || (g.o[d] == 0 && per == siz)) {
            // the axis is the whole periodic mesh, a multiple of the (power of two) tile extent
            // (pmx_binplan_supported): the base cell relative to the particle's tile is I0 mod T, whatever
            // period the coordinate is in — two instructions instead of the wrap, the shift into the block
            // and the subtraction of the tile origin.  (The list says which tile that is; a plan that no
            // longer matches the positions then deposits into the wrong cells of the region, never outside it.)
            lb[d] = I[0] & (tile_ext(d) - 1);
            continue;
        }
        int w = wrap_fast(I[0], per);
        int i0w = w;
        if (per > 0 && w >= siz) i0w = w - per;
        lb[d] = i0w + g.o[d] - t[d] * tile_ext(d);
    }
}

// The same with the weights of the RELAXED forms (pmx_window_dev.h, Fast<KIND, F>): the first cell is the
// reference's bit for bit (Tuned<KIND>::first, double precision, no FMA); the weights are polynomials in the one
// offset d = X - I_ref, formed in double, evaluated in F.
template <int KIND, bool WHOLE, typename F, bool ORDER0 = false>
__device__ __forceinline__ void particle_setup_fast(const pmx_painter &p, const BinGeom &g, const int *t,
                                                    const double *x, F (*V)[Tuned<KIND>::S], int *lb)
{
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const double X = x[d] * p.scale[d] + p.translate[d];
        const int I0 = Tuned<KIND>::first(X);
        const F off = (F)(X - (double)(I0 + Fast<KIND, F>::REF));
        Fast<KIND, F>::axis(off, ORDER0 ? 0 : p.order[d], (F)p.scale[d], V[d]);      // (ORDER0: the caller knows that no axis is differentiated)
        const int per = (int)p.period[d], siz = (int)p.size[d];
        if (WHOLE || (g.o[d] == 0 && per == siz)) {
            lb[d] = I0 & (tile_ext(d) - 1);
            continue;
        }
        int w = wrap_fast(I0, per);
        int i0w = w;
        if (per > 0 && w >= siz) i0w = w - per;
        lb[d] = i0w + g.o[d] - t[d] * tile_ext(d);
    }
}

// timing experiments (wrong results; -DPMX_EXPERIMENT builds only, profiled with PMESH_AMD_BENCH_NOCHECK=1, which prints no bench line): where does the deposit spend its time?
//   PMX_EXP_NOATOM: the weights are computed and folded into one register, nothing goes to LDS
//   PMX_EXP_NOWEIGHT: the LDS atomics with a constant instead of the weight products
#ifndef PMX_FIXED_POINT
#define PMX_FIXED_POINT 1
#endif
#ifndef PMX_FIXED_MIN_S
#define PMX_FIXED_MIN_S 3
#endif
#if !defined(PMX_EXPERIMENT) || !defined(PMX_EXP_NOATOM)
#undef PMX_EXP_NOATOM
#define PMX_EXP_NOATOM 0
#endif
#if !defined(PMX_EXPERIMENT) || !defined(PMX_EXP_NOWEIGHT)
#undef PMX_EXP_NOWEIGHT
#define PMX_EXP_NOWEIGHT 0
#endif

// ---- fixed-point accumulation (FIXED) -------------------------------------------------------------------
// The region is accumulated as 64-bit integers: every contribution v is rounded once to a multiple of 2^-f
// (v 2^f + 1.5 2^52 in ONE fma, whose mantissa then holds the integer) and added with ds_add_u64.  Why: on
// gfx950 a ds_add_f64 instruction whose lanes meet on a cell or a bank costs far more than an integer one
// (scripts/deposit_model.hip, the benchmark's jittered lattice under TSC: 19.0 against 11.5 clocks per wave
// instruction; a perfect lattice 11.9 against 10.5) and the S >= 3 paint kernels are bound by exactly that
// (with the atomics compiled out TSC paints in 1.10 instead of 2.10 ms, PCS in 1.78 instead of 3.49).  The sum of
// a region no longer depends on the order of arrival: it is bit-reproducible.  f comes from what can meet in a
// cell: at most n particles (those of the z segment's tiles) of mass <= mb with a weight product <= wb:
// every contribution below 2^50 units, every sum below 2^61.  The reference adds doubles (relative error
// 2^-53 per add); here the absolute error per add is 2^-f-1 with 2^f >= 2^50 / (mb wb) unless a segment holds
// more than 2^11 particles per unit... in numbers: a uniform 512^3 set (16384 particles per segment) is
// accumulated in steps of 2^-47 = 7e-15 of the particle mass.
constexpr double FIXED_MAGIC = 6755399441055744.0;                    // 1.5 * 2^52
constexpr long long FIXED_MAGIC_BITS = 0x4338000000000000ll;          // its bit pattern (low word 0)

// scale 2^f for a region that at most n particles of |mass| <= mb deposit into (weights of painter p)
__device__ __forceinline__ int fixed_exponent(const pmx_painter &p, double mb, int64_t n)
{
    double wb = 1.0;
#pragma unroll
    for (int d = 0; d < 3; d++)
        if (p.order[d]) wb *= 2.0 * fabs(p.scale[d]) + 2.0;           // bound of the derivative weights
    // mb wb < 2^e, from the two exponents (their product may not be representable)
    const int e = ((mb > 0) ? ilogb(mb) + 1 : -1000) + (wb > 1.0 ? ilogb(wb) + 1 : 0);
    int lg = 0;
    while (((int64_t)1 << lg) < n && lg < 40) lg++;                    // n <= 2^lg
    int f = 50 - e;
    if (61 - e - lg < f) f = 61 - e - lg;
    return f < -1020 ? -1020 : (f > 1020 ? 1020 : f);                  // (2^f and 2^-f stay normal doubles)
}
__device__ __forceinline__ double pow2(int f) { return __longlong_as_double((long long)(1023 + f) << 52); }
// a wave-uniform double (loaded through a vector register) moved to scalar registers
__device__ __forceinline__ double uniform_double(double x)
{
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---- [r6] PCS on fixed-point regions: FOUR LANES PER PARTICLE ----------------------------------------------------------
// One lane per particle puts 64 DIFFERENT particles into every ds_add_u64 instruction of the deposit: neighbours of the
// list — neighbours in space — which meet on cells and banks (scripts/deposit_quad_model.hip, the loop with its
// arithmetic, clocks per particle and CU: a perfect lattice 8.8, the benchmark's jittered lattice 10.7, rows in random
// order 12.7, a clustered set — half of a tile's particles in 27 cells — 23.2; the 10.7 IS the 2.3 ms of the PCS
// paint at 512^3).  Here the four lanes of a quad serve ONE particle: lane q owns the stencil index q along z (the four
// cells of a quad are 32 contiguous bytes) and walks the 16 (x, y) rows; an instruction then holds 16 particles, a
// 16-lane group of the LDS 4.  The quad's particles reach its lanes through DPP quad_perm moves (no LDS, no
// readlane): every lane prepares ITS particle as before (cell, x and y weights times the mass, the z offset), then in
// four sub-trips the quad takes the particle of its lane 0, 1, 2, 3; a lane evaluates only the z weight of its own
// index, from per-lane polynomial coefficients.  Model, same patterns: 9.1 / 9.8 / 11.1 / 13.5 clocks — the conflicts of
// a clustered set cost 4 instead of 14 clocks per particle, the vector work 6.2 instead of 5.0.  Same cells (the index
// arithmetic is untouched), same fixed-point sums up to the z weight's rounding (a cubic in Horner form instead of the
// factored one: absolute difference <= 2^-52 of the particle's mass per cell, far inside the contract's 1e-12).
#ifndef PMX_QUAD_PCS
#define PMX_QUAD_PCS 1
#endif
#ifndef PMX_DEAL_CROWDED
#define PMX_DEAL_CROWDED 0      // (an experiment: see tile_deposit)
#endif
#ifndef PMX_DEAL_SAME_OF_64
#define PMX_DEAL_SAME_OF_64 24       // neighbouring entries of the sample that share their first cell, from which on the rows count as cell-ordered
#endif
#ifndef PMX_DEAL_MIN_DEFAULT
#define PMX_DEAL_MIN_DEFAULT (2 * TCELLS)
#endif
#ifndef PMX_QUAD_STRIDE8
#define PMX_QUAD_STRIDE8 1
#endif
#ifndef PMX_QUAD_MIN_DEFAULT
#define PMX_QUAD_MIN_DEFAULT (9 * TCELLS / 8)
#endif
template <int CTRL> __device__ __forceinline__ int quad_take(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ double quad_take(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = quad_take<CTRL>((int)b), hi = quad_take<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int TTHREADS, bool SORTED, int PE, bool WHOLE, typename WF>
__device__ __forceinline__ void tile_deposit_quadz(const pmx_painter &p, const BinGeom &g, const int *t, const DVec &pos,
                                                   const DVec &mass, double mass_scalar, const uint32_t *list,
                                                   int64_t start, int count, double *lds, double scale)
{
    constexpr int KIND = PMX_TUNED_PCS, S = 4;
    using Rg = Region<S>;
    static_assert(!Rg::SPLIT, "dense rows: the four z cells of a quad are one address + its lane");
    constexpr int R1 = Rg::R1, P2 = Rg::P2;
    const int q = threadIdx.x & 3;
    // W_q(d) = k0 + d (k1 + d (k2 + d k3)): Fast<PCS>::axis expanded in d (d = X - the second cell, in [0, 1)); the
    // derivative form (order 1) carries no scale factor (quirk Q1)
    double k0, k1, k2, k3;
    if (p.order[2] == 0) {
        k0 = q == 0 ? 1.0 / 6.0 : (q == 1 ? 2.0 / 3.0 : (q == 2 ? 1.0 / 6.0 : 0.0));
        k1 = q == 0 ? -0.5 : (q == 2 ? 0.5 : 0.0);
        k2 = q == 0 ? 0.5 : (q == 1 ? -1.0 : (q == 2 ? 0.5 : 0.0));
        k3 = q == 0 ? -1.0 / 6.0 : (q == 1 ? 0.5 : (q == 2 ? -0.5 : 1.0 / 6.0));
    } else {
        k0 = q == 0 ? -0.5 : (q == 2 ? 0.5 : 0.0);
        k1 = q == 0 ? 1.0 : (q == 1 ? -2.0 : (q == 2 ? 1.0 : 0.0));
        k2 = q == 0 ? -0.5 : (q == 1 ? 1.5 : (q == 2 ? -1.5 : 0.5));
        k3 = 0.0;
    }
    const uint32_t *tl = list + start;
    // (a uniform trip count: the lanes of a quad serve each other's particles, so all of them stay in the loop)
    for (int jb = 0; jb < count; jb += TTHREADS * UNROLL) {
        int64_t idx[UNROLL];
        double x[UNROLL][3], m[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#if PMX_QUAD_STRIDE8
            // a wave holds 128 CONSECUTIVE entries, the quad k of it entries 8k .. 8k + 7 (lane q: 8k + q and 8k + 4 + q):
            // the 16 particles of an instruction are then 8 entries apart — neighbouring quads' z runs of four cells
            // do not overlap up to two particles per cell (4 apart they did on a lattice of 2 per cell: 5.5 against
            // 4.4 ms), and the loads still cover whole lines
            static_assert(UNROLL == 2, "two entries per lane");
            const int j = jb + ((int)threadIdx.x >> 6) * 128 + 2 * ((int)threadIdx.x & 63) - q + 4 * u;
#else
            const int j = jb + u * TTHREADS + (int)threadIdx.x;
#endif
            idx[u] = j < count ? (SORTED ? start + j : (int64_t)tl[j]) : -1;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (idx[u] >= 0) {
                x[u][0] = pos_get<PE>(pos, idx[u], 0); x[u][1] = pos_get<PE>(pos, idx[u], 1); x[u][2] = pos_get<PE>(pos, idx[u], 2);
                m[u] = mass.data ? mass.get(SORTED ? (int64_t)list[idx[u]] : idx[u], 0) : mass_scalar;
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            // this lane's particle: first cell in the region, x weights (times mass and 2^f) and y weights, z offset
            int base = -1;
            double Wx[S], Wy[S], dz = 0;
            if (idx[u] >= 0) {
                int lb[3];
                double off[3];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const double X = x[u][d] * p.scale[d] + p.translate[d];
                    const int I0 = Tuned<KIND>::first(X);
                    off[d] = X - (double)(I0 + Fast<KIND, double>::REF);
                    const int per = (int)p.period[d], siz = (int)p.size[d];
                    if (WHOLE || (g.o[d] == 0 && per == siz)) { lb[d] = I0 & (tile_ext(d) - 1); continue; }
                    const int w = wrap_fast(I0, per);
                    lb[d] = ((per > 0 && w >= siz) ? w - per : w) + g.o[d] - t[d] * tile_ext(d);
                }
                WF wx[S], wy[S];
                Fast<KIND, WF>::axis((WF)off[0], p.order[0], (WF)p.scale[0], wx);
                Fast<KIND, WF>::axis((WF)off[1], p.order[1], (WF)p.scale[1], wy);
                const double mu = m[u] * scale;
#pragma unroll
                for (int a = 0; a < S; a++) { Wx[a] = (double)wx[a] * mu; Wy[a] = (double)wy[a]; }
                dz = off[2];
                // a plan that no longer matches the positions must not index outside the LDS region
                if ((unsigned)lb[0] >= (unsigned)T0 || (unsigned)lb[1] >= (unsigned)T1 || (unsigned)lb[2] >= (unsigned)T2) stale_row(g);
                else base = (lb[0] * R1 + lb[1]) * P2 + lb[2];
            }
#define PMX_QUAD_SUBTRIP(CT) {                                                                                   \
                const int qb = quad_take<CT>(base);                                                              \
                const double zd = quad_take<CT>(dz);                                                             \
                double ax[S], ay[S];                                                                             \
                _Pragma("unroll") for (int a = 0; a < S; a++) { ax[a] = quad_take<CT>(Wx[a]); ay[a] = quad_take<CT>(Wy[a]); } \
                if (qb >= 0) {                                                                                   \
                    const double wz = __builtin_fma(__builtin_fma(__builtin_fma(k3, zd, k2), zd, k1), zd, k0);   \
                    double *cellp = lds + qb + q;                                                                \
                    _Pragma("unroll") for (int b = 0; b < S; b++) {                                              \
                        const double fz = ay[b] * wz;                                                            \
                        _Pragma("unroll") for (int a = 0; a < S; a++) {                                          \
                            const double r = __builtin_fma(ax[a], fz, FIXED_MAGIC);                              \
                            atomicAdd((unsigned long long *)(cellp + (a * R1 + b) * P2),                         \
                                      (unsigned long long)(__double_as_longlong(r) - FIXED_MAGIC_BITS));         \
                        }                                                                                        \
                    }                                                                                            \
                }                                                                                                \
            }
            PMX_QUAD_SUBTRIP(0x00) PMX_QUAD_SUBTRIP(0x55) PMX_QUAD_SUBTRIP(0xaa) PMX_QUAD_SUBTRIP(0xff)
#undef PMX_QUAD_SUBTRIP
        }
    }
}

// The particles [start, start + count) of a tile's list are deposited into its LDS region.
// WF: double / float = the weights of the RELAXED form in that precision (fixed-point regions of the S >= 3 windows:
// the sum is rounded to 2^-f anyway, and to the canvas type once more for float canvases); void = the reference's
// arithmetic, operation by operation (NNB / CIC, whose sums are doubles and reproduce the reference bit for bit on
// exactly summable inputs, the floating-point twin and the deterministic mode)
template <int KIND, int TTHREADS, bool SORTED, bool FIXED = false, int PE = 0, bool WHOLE = false, typename WF = void>
__device__ __forceinline__ void tile_deposit(const pmx_painter &p, const BinGeom &g, const int *t, const DVec &pos,
                                             const DVec &mass, double mass_scalar, const uint32_t *list,
                                             int64_t start, int count, double *lds, double scale = 1.0)
{
    if constexpr (PMX_QUAD_PCS && KIND == PMX_TUNED_PCS && FIXED && !std::is_same<WF, void>::value) {
        // (uniform per workgroup: `count` is the tile's)
        if (SORTED || count >= g.quad_min) {
            tile_deposit_quadz<TTHREADS, SORTED, PE, WHOLE, WF>(p, g, t, pos, mass, mass_scalar, list, start, count, lds, scale);
            return;
        }
    }
    constexpr bool sorted = SORTED;
    constexpr int S = Tuned<KIND>::S;
    using Rg = Region<S>;
    constexpr int R1 = Rg::R1;
    // UNROLL particles per thread and trip: all index and position loads are issued before
    // the first use, so several dependent gathers are in flight per lane
    // SWAP (TSC, PCS): odd lanes deposit their second particle first, see below
    constexpr bool SWAP = S >= 3 && UNROLL == 2;
    // ([r4] measured and dropped: dealing the 64 list entries of a wave to its lanes so that neighbouring entries — the
    // ones a jittered lattice puts on one cell — sit in different 16-lane groups of the LDS.  scripts/ldsatomic_groups.hip:
    // a 64-bit LDS atomic is served in groups of 16 consecutive lanes, two lanes of a group on one address cost 12.3
    // instead of 6.5 clocks per instruction, on one bank 8.3, lanes of different groups meet for free; the deal took the
    // benchmark's pattern from 12.1 to 8.4 clocks there — and changed nothing in this kernel: TSC f4 paint 1.65 / 1.65 ms,
    // PCS 2.77 / 2.90, with the odd-lane swap on top 1.78 / 3.22.  The kernel is not waiting for those conflicts.)
    double sink = 0;
    const uint32_t *tl = list + start;       // (a wave-uniform base + the lane's entry: nothing per thread for the compiler to keep across tiles)
#if PMX_DEAL_CROWDED
    // [r6] (experiment builds only, -DPMX_DEAL_CROWDED=1) Crowded tiles deal their entries: slot s of a trip of W takes
    // entry (65 s) mod W.  Rows that arrive sorted by cell — a caller that keeps its particles in Peano-Hilbert or cell
    // order, as tree codes do — put the particles of a crowded cell into neighbouring lanes, every lane of an instruction
    // on ONE address: an evolved 512^3 state (scripts/clustered_state_probe.py) painted in 4.2 ms cell-sorted against 1.4
    // in random order, 2.2 with the deal.  Whether a crowded tile's rows ARE in cell order is looked up on 64 entries from
    // the middle of its list (rows that merely arrive in the order the particles were made in lose the locality of
    // their gathers for nothing when dealt: 2.30 -> 3.18 ms).  NOT in the product: the sample and the slots cost the CIC
    // kernel 14 registers — 69 instead of 55, the fourth workgroup of a CU — and the benchmark's paint 7-12 %
    // (scripts/r06/deal_ab.sh); pm.tile_order, which sorts by tile and NOT by cell, is the remedy a caller has.
    constexpr int W = TTHREADS * UNROLL;
    constexpr bool CAN_DEAL = !FIXED && !sorted && (W & (W - 1)) == 0;
    bool deal = false;
    if (CAN_DEAL && count >= g.deal_min) {      // (uniform per workgroup)
        __shared__ int deal_flag;
        if (threadIdx.x < 64) {
            const int64_t row = (int64_t)tl[count / 2 + (int)threadIdx.x];
            int key = 0;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const double X = (double)pos_get<PE>(pos, row, d) * p.scale[d] + p.translate[d];
                key = key * 1021 + Tuned<KIND>::first(X);
            }
            const int prev = __shfl_up(key, 1);
            const unsigned long long same = __ballot(threadIdx.x > 0 && key == prev);
            if (threadIdx.x == 0) deal_flag = __popcll(same) >= PMX_DEAL_SAME_OF_64;
        }
        __syncthreads();
        deal = deal_flag != 0;
    }
    int slot[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        const int s = u * TTHREADS + (int)threadIdx.x;
        slot[u] = (CAN_DEAL && deal) ? ((s * 65) & (W - 1)) : s;
    }
    for (int j0 = 0; j0 < count; j0 += W) {
        int64_t idx[UNROLL];
        double x[UNROLL][3], m[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int j = j0 + slot[u];
            idx[u] = j < count ? (sorted ? start + j : (int64_t)tl[j]) : -1;
        }
#else
    for (int j0 = threadIdx.x; j0 < count; j0 += TTHREADS * UNROLL) {
        int64_t idx[UNROLL];
        double x[UNROLL][3], m[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            int j = j0 + u * TTHREADS;
            idx[u] = j < count ? (sorted ? start + j : (int64_t)tl[j]) : -1;
        }
#endif
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (idx[u] >= 0) {
                x[u][0] = pos_get<PE>(pos, idx[u], 0); x[u][1] = pos_get<PE>(pos, idx[u], 1); x[u][2] = pos_get<PE>(pos, idx[u], 2);
                // (sorted: idx is the list slot; a per-particle mass lives at the row the list names)
                m[u] = mass.data ? mass.get(sorted ? (int64_t)list[idx[u]] : idx[u], 0) : mass_scalar;
            }
        }
        if (SWAP && (threadIdx.x & 1)) {
            // odd lanes deposit their second particle first: neighbouring list entries — neighbours along z, every
            // fourth pair on the same cell under TSC — then sit in different instructions, while the loads stay dense
            // and every instruction still reaches all banks (even cells of one run, odd cells of another).  Same box,
            // with the bank-neutral rows: TSC paint 2.70 -> 2.53 ms on the jittered lattice, 1.53 -> 1.56 on a perfect
            // one.  (A lane taking two CONSECUTIVE entries instead did the same for the jittered lattice, 2.58, but its
            // stride-2 loads cost the perfect one 0.7 ms.)  PCS: the clustered set gains — config 5's per-GPU load 98 -> 87 ms of
            // paint, 256^3 1.12 -> 0.96 — the lattice, on which PCS base cells have no jitter to begin with, pays 6 %
            // (3.92 -> 4.16 ms).  CIC: +3 % either way, not used.
#pragma unroll
            for (int d = 0; d < 3; d++) { const double tmp = x[0][d]; x[0][d] = x[1][d]; x[1][d] = tmp; }
            const double tm = m[0]; m[0] = m[1]; m[1] = tm;
            const int64_t ti = idx[0]; idx[0] = idx[1]; idx[1] = ti;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (idx[u] < 0) continue;
            int lb[3];
            double V[3][S];
            constexpr bool FAST = !std::is_same<WF, void>::value;
            if constexpr (FAST) {
                using WT = typename std::conditional<FAST, WF, double>::type;
                WT W[3][S];
                particle_setup_fast<KIND, WHOLE, WT>(p, g, t, x[u], W, lb);
#pragma unroll
                for (int d = 0; d < 3; d++)
#pragma unroll
                    for (int a = 0; a < S; a++) V[d][a] = (double)W[d][a];
            } else particle_setup<KIND, WHOLE>(p, g, t, x[u], V, lb);
            // a plan that no longer matches the positions (rewritten behind the cache's back)
            // must not index outside the LDS region
            if ((unsigned)lb[0] >= (unsigned)T0 || (unsigned)lb[1] >= (unsigned)T1 || (unsigned)lb[2] >= (unsigned)T2) { stale_row(g); continue; }
            // (FIXED: the mass carries the 2^f — a power of two, so the products are those of the reference times 2^f)
            const double mu = FIXED ? m[u] * scale : m[u];
#pragma unroll
            for (int a = 0; a < S; a++) V[0][a] *= mu;
            // Split layout: whether stencil point c of this particle lies in the halo columns depends on its z cell
            // alone, so the choice between the two bases of a row — its 32 cells and its halo columns — is one select
            // per point with the masks of the particle, and c rides on the instruction's immediate offset (written
            // out by hand: left to itself the compiler compares and selects per atomic, a quarter of the kernel's
            // vector instructions)
            bool inhalo[S];
#pragma unroll
            for (int c = 0; c < S; c++) inhalo[c] = Rg::SPLIT && c > 0 && lb[2] + c >= T2;
#pragma unroll
            for (int a = 0; a < S; a++)
#pragma unroll
                for (int b = 0; b < S; b++) {
                    double fb = V[0][a] * V[1][b];
                    const int row = (lb[0] + a) * R1 + (lb[1] + b);
                    const int imain = Rg::SPLIT ? row * T2 + lb[2] : row * Rg::P2 + lb[2];
                    const int ihalo = Rg::DMAIN + row * (S - 1) + lb[2] - T2;
#pragma unroll
                    for (int c = 0; c < S; c++) {
                        const int cell = (inhalo[c] ? ihalo : imain) + c;
                        if (PMX_EXP_NOATOM) sink += fb * V[2][c];
                        else if (PMX_EXP_NOWEIGHT) unsafeAtomicAdd(&lds[cell], m[u]);
                        else if (FIXED) {
                            const double r = __builtin_fma(fb, V[2][c], FIXED_MAGIC);
                            atomicAdd((unsigned long long *)&lds[cell],
                                      (unsigned long long)(__double_as_longlong(r) - FIXED_MAGIC_BITS));
                        } else unsafeAtomicAdd(&lds[cell], fb * V[2][c]);
                    }
                }
        }
    }
    if (PMX_EXP_NOATOM && sink == 12345.678) lds[0] = sink;
}

// (the form of the loop chosen once per launch: see tile_gather_any)
template <int KIND, int TTHREADS, bool SORTED, bool FIXED, typename WF = void>
__device__ __forceinline__ void tile_deposit_any(bool whole, const pmx_painter &p, const BinGeom &g, const int *t,
                                                 const DVec &pos, const DVec &mass, double mass_scalar,
                                                 const uint32_t *list, int64_t start, int count, double *lds, double scale)
{
    if (whole) {
        if (pos.elsize == 8) tile_deposit<KIND, TTHREADS, SORTED, FIXED, 8, true, WF>(p, g, t, pos, mass, mass_scalar, list, start, count, lds, scale);
        else tile_deposit<KIND, TTHREADS, SORTED, FIXED, 4, true, WF>(p, g, t, pos, mass, mass_scalar, list, start, count, lds, scale);
    } else {
        if (pos.elsize == 8) tile_deposit<KIND, TTHREADS, SORTED, FIXED, 8, false, WF>(p, g, t, pos, mass, mass_scalar, list, start, count, lds, scale);
        else tile_deposit<KIND, TTHREADS, SORTED, FIXED, 4, false, WF>(p, g, t, pos, mass, mass_scalar, list, start, count, lds, scale);
    }
}

// The particles [start, start + count) of a tile's list read their values from its LDS region.
// RELAX: the weights of pmx_window_dev.h's Fast<KIND, F> and the sum as nested fused multiply-adds — sum_a Wx[a]
// (sum_b Wy[b] (sum_c Wz[c] cell)) — in F = the canvas type: S^3 + S^2 + S FMAs instead of the 3 S^3 + S^2
// products and adds of the reference's order (which the other form keeps bit for bit: window.EXACT).  Within
// 1e-14 (double) / 1e-6 (float) of the exact form relative to sum |w cell|.
template <int KIND, typename T, int TTHREADS, bool SORTED, int PE = 0, bool WHOLE = false, bool RELAX = false>
__device__ __forceinline__ void tile_gather(const pmx_painter &p, const BinGeom &g, const int *t, const DVec &pos,
                                            const DVec &out, const uint32_t *list, int64_t start, int count,
                                            const T *lds)
{
    constexpr bool sorted = SORTED;
    constexpr int S = Tuned<KIND>::S;
    using Rg = Region<S>;
    constexpr int R1 = Rg::R1;
    const uint32_t *tl = list + start;
    for (int j0 = threadIdx.x; j0 < count; j0 += TTHREADS * UNROLL) {
        int64_t idx[UNROLL];
        double x[UNROLL][3];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            int j = j0 + u * TTHREADS;
            idx[u] = j < count ? (sorted ? start + j : (int64_t)tl[j]) : -1;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (idx[u] >= 0) {
                x[u][0] = pos_get<PE>(pos, idx[u], 0); x[u][1] = pos_get<PE>(pos, idx[u], 1); x[u][2] = pos_get<PE>(pos, idx[u], 2);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (idx[u] < 0) continue;
            int lb[3];
            if constexpr (RELAX) {
                T W[3][S];
                particle_setup_fast<KIND, WHOLE, T>(p, g, t, x[u], W, lb);
                // (a row the plan no longer matches reads 0, not what was in `out`: it gathers at the tile's first cell and
                // stores 0 — one store path, no second set of addresses)
                const bool stale = (unsigned)lb[0] >= (unsigned)T0 || (unsigned)lb[1] >= (unsigned)T1 || (unsigned)lb[2] >= (unsigned)T2;
                if (stale) { stale_row(g); lb[0] = lb[1] = lb[2] = 0; }
                T acc = 0;
#pragma unroll
                for (int a = 0; a < S; a++) {
                    T plane = 0;
#pragma unroll
                    for (int b = 0; b < S; b++) {
                        const int rowoff = ((lb[0] + a) * R1 + (lb[1] + b)) * Rg::template gpitch<T>() + lb[2];
                        T row = 0;
#pragma unroll
                        for (int c = 0; c < S; c++) row = fma_(lds[rowoff + c], W[2][c], row);
                        plane = fma_(W[1][b], row, plane);
                    }
                    acc = fma_(W[0][a], plane, acc);
                }
                out.set(idx[u], 0, stale ? 0.0 : (double)acc);
                continue;
            }
            double V[3][S];
            particle_setup<KIND, WHOLE>(p, g, t, x[u], V, lb);
            const bool stale = (unsigned)lb[0] >= (unsigned)T0 || (unsigned)lb[1] >= (unsigned)T1 || (unsigned)lb[2] >= (unsigned)T2;
            if (stale) { stale_row(g); lb[0] = lb[1] = lb[2] = 0; }
            double value = 0;
#pragma unroll
            for (int a = 0; a < S; a++)
#pragma unroll
                for (int b = 0; b < S; b++) {
                    double fb = V[0][a] * V[1][b];
                    int rowoff = ((lb[0] + a) * R1 + (lb[1] + b)) * Rg::template gpitch<T>() + lb[2];
#pragma unroll
                    for (int c = 0; c < S; c++) value += (double)lds[rowoff + c] * (fb * V[2][c]);
                }
            out.set(idx[u], 0, stale ? 0.0 : value);
        }
    }
}

// One choice per launch instead of one per position component and axis: the element size of the positions and
// "every axis is the whole periodic mesh" (`whole`, whole_mesh()) select a form of the loop in which the other
// cases do not appear at all.  Left in the loop they were 30 wave-uniform branches per trip of two particles:
// config 3 (fp32 positions) readout 1.54 -> 1.25 ms, paint 1.81 -> 1.68; fp64 positions 2-4 %.
template <int KIND, typename T, int TTHREADS, bool SORTED, bool RELAX = false>
__device__ __forceinline__ void tile_gather_any(bool whole, const pmx_painter &p, const BinGeom &g, const int *t,
                                                const DVec &pos, const DVec &out, const uint32_t *list, int64_t start,
                                                int count, const T *lds)
{
    if (whole) {
        if (pos.elsize == 8) tile_gather<KIND, T, TTHREADS, SORTED, 8, true, RELAX>(p, g, t, pos, out, list, start, count, lds);
        else tile_gather<KIND, T, TTHREADS, SORTED, 4, true, RELAX>(p, g, t, pos, out, list, start, count, lds);
    } else {
        if (pos.elsize == 8) tile_gather<KIND, T, TTHREADS, SORTED, 8, false, RELAX>(p, g, t, pos, out, list, start, count, lds);
        else tile_gather<KIND, T, TTHREADS, SORTED, 4, false, RELAX>(p, g, t, pos, out, list, start, count, lds);
    }
}

// [r4] The tile kernels are bound by a chain of dependent loads per trip — list entry -> position row — at the
// bytes a CU keeps in flight (scripts/kernel_split.sh: with its atomics compiled out TSC f4 paint still takes 0.96 of its
// 1.36 ms, PCS f8 1.56 of 2.35, at ~4 TB/s of real traffic).  The tile's slice of the index list is contiguous and
// known when the workgroup arrives: one lane per 128-byte line asks for it BEFORE the region is zeroed / staged, so
// that the list loads of every trip find their line in the L2 instead of in HBM.  The value itself is not used; the
// caller keeps it alive until behind its next barrier (list_touch_done), where the compiler waits for it for free.
// Measured (same box, off / on): PCS paint 2.76 -> 2.60 ms, clustered 3.55 -> 3.38; TSC paint 1.62 -> 1.65, CIC the
// same; the readouts LOSE (one more live register: PCS on 768 threads 1.68 -> 2.46 ms, TSC f4 1.08 -> 1.16): on for
// PCS paint only.
#ifndef PMX_LIST_PREFETCH
#define PMX_LIST_PREFETCH 1
#endif
__device__ __forceinline__ uint32_t list_touch(const uint32_t *list, int64_t start, int count)
{
    uint32_t v = 0;
    if (PMX_LIST_PREFETCH && (int)threadIdx.x * 32 < count) v = list[start + (int)threadIdx.x * 32];
    return v;
}
__device__ __forceinline__ void list_touch_done(uint32_t v)
{
    if (PMX_LIST_PREFETCH) asm volatile("" ::"v"(v));
}

// Threads of a tile workgroup.  The LDS region fixes the workgroups per CU (4 / 3 / 2 for CIC /
// TSC / PCS in double); with 256 threads that left 16 / 12 / 8 waves per CU on kernels that spend
// 60 % of their cycles parked on memory.  512 threads double the waves at the same LDS and still
// fit the registers (59-83 VGPRs): readout 1.46 -> 1.26 (CIC), 2.39 -> 1.78 (TSC), 3.83 -> 2.67 ms
// (PCS), PCS paint 3.77 -> 3.17 ms at 512^3.  The float readout already runs 7-8 workgroups of
// 256 per CU on its half-size region and keeps them.
// (PCS with 768 threads — two workgroups per CU by its 58 KB region, six waves per SIMD by its
// registers — measured 3 % faster before the z-halo carry and 25 % slower in paint with it: 512.)
#ifndef PMX_TILE_THREADS_PCS
#define PMX_TILE_THREADS_PCS 512
#endif
// [r4] the PCS readout of a double canvas: its 81 KB region lets two workgroups share a CU, its 75-81 VGPRs six waves a
// SIMD — 768 threads fill what 512 leave empty (24 instead of 16 waves per CU): 2.03 -> 1.77 ms at 512^3, clustered
// 2.61 -> 2.48, config 5's per-GPU load 35.3 -> 34.3 (1024 threads: 2.38 / 3.20 / 44.8, one workgroup per CU)
#ifndef PMX_TILE_THREADS_PCS_READOUT
#define PMX_TILE_THREADS_PCS_READOUT 768
#endif
template <int KIND, typename T> struct TileThreads {
    static constexpr int paint = KIND == PMX_TUNED_PCS ? PMX_TILE_THREADS_PCS : PMX_TILE_THREADS;
    static constexpr int readout = KIND == PMX_TUNED_PCS ? PMX_TILE_THREADS_PCS_READOUT : PMX_TILE_THREADS;
};
template <int KIND> struct TileThreads<KIND, float> {
    static constexpr int paint = KIND == PMX_TUNED_PCS ? PMX_TILE_THREADS_PCS : PMX_TILE_THREADS;
    static constexpr int readout = PMX_TILE_THREADS_RF4;
};

// MODE of the paint kernels: 0 = the region accumulates doubles; 1 = 64-bit fixed point, one scale per z segment,
// converted when the region is flushed; 2 = DETERMINISTIC: fixed point with ONE scale for the whole batch
// (*dexp), and the region is flushed AS INTEGERS into a dense int64 copy of the block (`canvas` is that copy, the
// painter describes its dense layout): owned cells stored, halos and crowded pieces added with integer atomics —
// exact, so the result does not depend on any order — and det_finish_kernel converts once into the caller's canvas.
// what a cell of the region holds as a double (fixed point: the integer sum times 2^-f)
template <int MODE> __device__ __forceinline__ double cell_value(double raw, double inv)
{
    return MODE == 1 ? (double)__double_as_longlong(raw) * inv : raw;      // (MODE 2: the raw integer travels on)
}
// mstats (per-particle masses only): [0] = max |m| over the finite masses, [1] = number of non-finite ones.
// The FIXED kernel serves a batch whose masses are all finite, its floating-point twin (launched behind it,
// want_odd = 1) the others: each returns at once when the batch is not its own.
__device__ __forceinline__ bool batch_is_mine(const double *mstats, int want_odd)
{
    const bool odd = mstats != nullptr && mstats[1] != 0.0;
    return odd == (want_odd != 0);
}

// which weights the deposit of a paint kernel forms: the RELAXED ones in the canvas' precision for the fixed-point
// regions of the S >= 3 windows (MODE 1), the reference's for everything else (see tile_deposit)
#ifndef PMX_FAST_DEPOSIT
#define PMX_FAST_DEPOSIT 1
#endif
template <int KIND, typename T, int MODE> struct DepositWeights {
    using type = typename std::conditional<(PMX_FAST_DEPOSIT && MODE == 1 && Tuned<KIND>::S >= 3), T, void>::type;
};

// waves per SIMD the compiler must leave room for (the register budget): the regions allow 4 / 3 / 2 workgroups of
// 512 threads per CU for CIC / TSC / PCS, i.e. 8 / 6 / 4 waves per SIMD; the fixed-point TSC kernel came out at 84-90
// VGPRs, one over the 80 that six waves per SIMD leave: two workgroups per CU instead of three
#ifndef PMX_PAINT_WAVES_TSC
#define PMX_PAINT_WAVES_TSC 6
#endif
#ifndef PMX_PAINT_WAVES_CIC
#define PMX_PAINT_WAVES_CIC 1
#endif
// (the variants on the tile-ordered copy would spill a few bytes under that budget: they keep the default)
#ifndef PMX_PAINT_WAVES_PCS
#define PMX_PAINT_WAVES_PCS 4
#endif
// (MODE 2, the deterministic paint: an opt-in path with a second painter live and the larger carry of the x-walk; it
// spilled 12 bytes per lane under the TSC budget and keeps the compiler's default instead)
template <int KIND, bool SORTED, int MODE = 0> constexpr int paint_min_waves()
{
    return MODE == 2 ? 1 : (KIND == PMX_TUNED_TSC && !SORTED) ? PMX_PAINT_WAVES_TSC
         : (KIND == PMX_TUNED_CIC ? PMX_PAINT_WAVES_CIC : ((KIND == PMX_TUNED_PCS && !SORTED) ? PMX_PAINT_WAVES_PCS : 1));
}

// [r5] WHOLE (the block is the whole periodic mesh) and PE (bytes of a position element) are the launcher's to know: one
// form of the deposit loop per kernel, with its own registers (with all four in one kernel the fixed-point TSC kernel on a
// double canvas kept per-thread addresses of every form alive across the tile loop and spilled 28 bytes per lane)
template <int KIND, typename T, int TTHREADS, bool SORTED, int MODE, bool WHOLE, int PE>
__global__ void __launch_bounds__(TTHREADS, (paint_min_waves<KIND, SORTED, MODE>())) paint_tile_kernel(pmx_painter p, BinGeom g, char *canvas, DVec pos,
                                                            DVec mass, double mass_scalar,
                                                            const uint32_t *list, const int64_t *offsets,
                                                            const uint32_t *counts, T *halo, int overwrite,
                                                            const double *mstats, int want_odd, const int32_t *dexp,
                                                            pmx_painter pw)
{
    // pw: the painter of the particles (weights, scale bound); p: the layout that is written (MODE 2: the dense
    // integer copy of the block, else the same as pw)
    constexpr bool FIXED = MODE != 0;
    if (!batch_is_mine(mstats, want_odd)) return;
    // (only the deterministic mode has two layouts; elsewhere the second painter is never read, which keeps its 38
    // scalar registers out of the kernel — with both live the scalar file spilled into vector lanes)
    const pmx_painter &pwr = MODE == 2 ? pw : p;
    // SORTED: `pos` is the plan's copy of the positions in list order (row = list slot);
    // the list itself is then only read for a per-particle mass
    constexpr int S = Tuned<KIND>::S;
    using Rg = Region<S>;
    constexpr int R1 = Rg::R1;
    // The tile is accumulated in double whatever the canvas type: ds_add_f32 measured ~5x
    // slower than ds_add_f64 on gfx950 (CIC f4 paint 5.3 ms vs 1.0 ms at 512^3), and the sum
    // is rounded to the canvas type once, at the flush.
    __shared__ double lds[Rg::DLDS];
    // A workgroup walks a SEGMENT of up to ZSEG tiles that follow each other along z (the tile
    // index runs fastest along z) and keeps the z-halo — the planes c >= T2 of the region — in
    // LDS, where it becomes the first S-1 planes of the next tile's region.  Only the last tile of
    // a segment stages its z-face: those cells are single 8-byte atomics in separate sectors for
    // halo_merge, 13 % of the halo cells but half of its time (404 -> 248 us at 512^3 with 3 of 4
    // z-faces gone).
    // [r3] WALK_X (PCS, see walk_x()): the segments run along x instead and carry the x-face — the planes a >= T0,
    // (S - 1) R1 R2 cells, 62 % of the halo of a PCS tile where the z-face is 12 %.  The carry is a block of whole
    // region rows moved to the front of the region.
    constexpr bool WALK_X = walk_x(S);
    constexpr int NCARRY = WALK_X ? (S - 1) * R1 * Rg::R2 : Rg::R0 * R1 * (S - 1);
    constexpr int CPT = (NCARRY + TTHREADS - 1) / TTHREADS > 0 ? (NCARRY + TTHREADS - 1) / TTHREADS : 1;
    const int ntw = WALK_X ? g.nt[0] : g.nt[2];                   // tiles along the walk axis
    const int64_t tstride = WALK_X ? (int64_t)g.nt[1] * g.nt[2] : 1;      // tile index step along it
    const int nseg = (ntw + ZSEG - 1) / ZSEG;
    const int64_t ncolumn = g.ntiles / ntw;
    const int64_t nwork = ncolumn * nseg;
    for (int64_t w = blockIdx.x; w < nwork; w += gridDim.x) {
      const int64_t column = w / nseg;
      const int seg = (int)(w - column * nseg);
      const int t2a = seg * ZSEG, t2b = (t2a + ZSEG < ntw) ? t2a + ZSEG : ntw;
      // first tile of the column: (0, t1, t2) = `column` itself under WALK_X, (t0, t1, 0) else
      const int64_t tile0 = WALK_X ? column : column * ntw;
      bool live = false;                        // the region holds the carried face of the previous tile
      double scale = 1.0, inv = 1.0;
      if (FIXED) {
          // one scale for the segment: the face carried from tile to tile keeps its meaning
          int64_t nseg_part = 0;
          for (int t2 = t2a; t2 < t2b; t2++) {
              const uint32_t c = counts[tile0 + t2 * tstride];
              nseg_part += c < (uint32_t)g.chunk ? c : (uint32_t)g.chunk;
          }
          // (the mass bound as a scalar: as a vector register it lived — and was spilled — across the whole kernel)
          const int f = MODE == 2 ? *dexp : fixed_exponent(pwr, uniform_double(mstats ? mstats[0] : fabs(mass_scalar)), nseg_part);
          scale = pow2(f);
          inv = pow2(-f);
      }
      for (int t2 = t2a; t2 < t2b; t2++) {
        const int64_t tile = tile0 + t2 * tstride;
        const bool last = (t2 == t2b - 1);
        int t[3];
        tile_coords(g, tile, t);
        const int64_t start = offsets[tile];
        // (what a crowded tile holds beyond g.chunk entries is painted by paint_heavy_kernel)
        const int count = counts[tile] < (uint32_t)g.chunk ? (int)counts[tile] : g.chunk;
        if (count == 0 && !overwrite && !live) continue;   // nothing to add; uniform per workgroup
        constexpr bool TOUCH = !SORTED && S >= 4;
        uint32_t touched = 0;
        if constexpr (TOUCH) touched = list_touch(list, start, count);
        double carry[CPT];
        if (S > 1 && live) {
#pragma unroll
            for (int u = 0; u < CPT; u++) {
                int q = threadIdx.x + u * TTHREADS;
                asm volatile("" : "+v"(q));      // (opaque: the divisions below stay inside the tile loop instead of living in registers — and spilling — across the deposit)
                if (q < NCARRY) {
                    if (WALK_X) {
                        const int c = q % Rg::R2, r = q / Rg::R2;             // r = a R1 + b of the destination, a < S - 1
                        carry[u] = lds[Rg::dat(T0 * R1 + r, c)];
                    } else {
                        int c = q % (S - 1 > 0 ? S - 1 : 1), r = q / (S - 1 > 0 ? S - 1 : 1);
                        carry[u] = lds[Rg::dat(r, T2 + c)];
                    }
                }
            }
            __syncthreads();
        }
        for (int q = threadIdx.x; q < Rg::DLDS; q += TTHREADS) lds[q] = 0;
        __syncthreads();
        if (S > 1 && live) {
#pragma unroll
            for (int u = 0; u < CPT; u++) {
                int q = threadIdx.x + u * TTHREADS;
                asm volatile("" : "+v"(q));      // (opaque: the divisions below stay inside the tile loop instead of living in registers — and spilling — across the deposit)
                if (q < NCARRY) {
                    if (WALK_X) {
                        const int c = q % Rg::R2, r = q / Rg::R2;
                        lds[Rg::dat(r, c)] = carry[u];
                    } else {
                        int c = q % (S - 1 > 0 ? S - 1 : 1), r = q / (S - 1 > 0 ? S - 1 : 1);
                        lds[Rg::dat(r, c)] = carry[u];
                    }
                }
            }
            __syncthreads();
        }
        if constexpr (TOUCH) list_touch_done(touched);
        tile_deposit<KIND, TTHREADS, SORTED, FIXED, PE, WHOLE, typename DepositWeights<KIND, T, MODE>::type>(pwr, g, t, pos, mass, mass_scalar, list, start, count, lds, scale);
        __syncthreads();
        // owned box -> canvas, plain stores in rows of T2 cells
        if constexpr (WHOLE && MODE != 2) {
            // [r5] the block is the whole periodic mesh, a multiple of the tile on every axis: no wraps, no bounds
            char *tbase = canvas + (int64_t)t[0] * T0 * p.strides[0] + (int64_t)t[1] * T1 * p.strides[1] + (int64_t)t[2] * T2 * p.strides[2];
            const int s0 = (int)p.strides[0], s1 = (int)p.strides[1], s2 = (int)p.strides[2];
            for (int q = threadIdx.x; q < TCELLS; q += TTHREADS) {
                const int c = q % T2, r = q / T2;
                const int b = r % T1, a = r / T1;
                const T v = (T)cell_value<MODE>(lds[Rg::dat(a * R1 + b, c)], inv);
                T *dst = (T *)(tbase + ((int64_t)a * s0 + b * s1 + c * s2));
                if (overwrite) *dst = v;
                else *dst += v;
            }
        } else
        for (int q = threadIdx.x; q < TCELLS; q += TTHREADS) {
            int c = q % T2, r = q / T2;
            int b = r % T1, a = r / T1;
            int64_t goff;
            // cells of the box below 0 / beyond the block exist only on non-periodic or slab
            // axes and are dropped there (no wrap reaches them: pmx_binplan_supported)
            bool in = true;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                int l = t[d] * tile_ext(d) - g.o[d] + (d == 0 ? a : (d == 1 ? b : c));
                in = in && l >= 0 && l < p.size[d];
            }
            if (in && region_cell(p, g, t, a, b, c, &goff)) {
                T v = (T)cell_value<MODE>(lds[Rg::dat(a * R1 + b, c)], inv);
                T *dst = (T *)(canvas + goff);
                if (overwrite) *dst = v;
                else *dst += v;
            }
        }
        // halo -> staging (compact numbering, contiguous writes)
        if (S > 1) {
            T *hbase = halo + tile * (int64_t)Rg::HALO;
            int h0 = threadIdx.x;
            asm volatile("" : "+v"(h0));      // (opaque: no per-thread staging address kept — and spilled — across the tile loop)
            for (int h = h0; h < Rg::HALO; h += TTHREADS) {
                int a, b, c;
                Rg::halo_decode(h, &a, &b, &c);
                if (!last && (WALK_X ? a >= T0 : c >= T2)) continue;           // carried to the next tile instead
                hbase[h] = (T)cell_value<MODE>(lds[Rg::dat(a * R1 + b, c)], inv);
            }
        }
        live = !last;
        __syncthreads();
      }
    }
}

// ---- [r5] 32-bit fixed-point regions for FLOAT canvases (S >= 3) ------------------------------------------------
// A float canvas keeps 24 bits of a cell; the 64-bit fixed-point region above spends an 8-byte LDS atomic, a bias
// subtraction and a per-row address selection (split layout) on every stencil point for it.  Here the region holds
// 32-bit integers in units of 2^-f: half the LDS, `ds_add_rtn_u32` instead of `ds_add_u64` (scripts/deposit32_model.hip:
// 8.7 against 13.9 clocks per wave instruction on the benchmark's jittered lattice, 6.2 with rows of 64 cells; the
// returning form costs nothing), all S^3 cells of a particle at immediate offsets of ONE base address (dense rows), and
// the contribution is the LOW WORD of fma(w, 2^f m, 1.5 2^52) as it stands (two's complement) — one vector instruction
// per stencil point.
// 32 bits cannot hold a rigorous worst case (every particle of the segment in one cell) at a useful resolution, so the
// scale is OPTIMISTIC — room for 16 times the mean density of the segment's tiles — and every add is CHECKED: the
// atomics return the cell's previous value, the lanes OR them together, and a workgroup that has seen a value at or
// beyond 2^30 in magnitude (contributions stay below 2^29: no sum can have wrapped unseen) deposits the tile again IN
// TWO PARTS at the same scale: the high parts v >> sh of all contributions into one zeroed region, flushed, then the
// low parts v & (2^sh - 1) into another, added on top.  (A first version retried in 32 times coarser units: thousands
// of IDENTICAL contributions on one cell — their roundings do not cancel — then left 5e-5 of the cell's sum, outside
// the tolerance; tests/test_binned.py::test_float_canvas_regions_retry_on_overflow.)  Resolution, always: 2^-f =
// 32 nu m_max 2^-30 <= 3e-8 of the largest mass at nu = 1 particle per cell.  Every contribution is rounded once
// (absolute error 2^-f-1), the region's sum is exact in integers, the float canvas gets it rounded once more (twice
// for a tile in two parts): inside |d| <= 2e-6 max(1, max |cell|), the tolerance of a float canvas.
// The face carried from tile to tile of a segment is kept in 64 bits: a tile in two parts hands on more than 32.
#ifndef PMX_REGION32
#define PMX_REGION32 1
#endif
#ifndef PMX_PITCH32_TSC
#define PMX_PITCH32_TSC 64
#endif
#ifndef PMX_PITCH32_PCS
#define PMX_PITCH32_PCS 48
#endif
#ifndef PMX_TILE32_THREADS_TSC
#define PMX_TILE32_THREADS_TSC 512
#endif
#ifndef PMX_TILE32_THREADS_PCS
#define PMX_TILE32_THREADS_PCS 512
#endif
#ifndef PMX_PAINT32_WAVES_TSC
#define PMX_PAINT32_WAVES_TSC 6
#endif
#ifndef PMX_PAINT32_WAVES_PCS
#define PMX_PAINT32_WAVES_PCS 4
#endif
#ifndef PMX_HEADROOM32
#define PMX_HEADROOM32 4          // log2 of the room above the mean density of the segment's tiles
#endif
#ifndef PMX_RETRY32
#define PMX_RETRY32 5             // log2 of the coarsening of the scale itself, should even the high parts overflow
#endif
template <int KIND> struct Tile32 {
    static constexpr int S = Tuned<KIND>::S;
    using Rg = Region<S>;
    static constexpr int P = S == 3 ? PMX_PITCH32_TSC : PMX_PITCH32_PCS;       // row pitch (cells) >= R2
    static_assert(P >= Rg::R2, "row pitch of the 32-bit region");
    static constexpr int CELLS = Rg::R0 * Rg::R1 * P;
    static constexpr int threads = S == 3 ? PMX_TILE32_THREADS_TSC : PMX_TILE32_THREADS_PCS;
    static constexpr int waves = S == 3 ? PMX_PAINT32_WAVES_TSC : PMX_PAINT32_WAVES_PCS;
};

// the optimistic scale of a segment: 2^f (mass bound) (weight bound) 2^HEADROOM (mean particles per cell, at least 1) <= 2^30
__device__ __forceinline__ int fixed_exponent32(const pmx_painter &p, double mb, int64_t n, int ntiles_seg)
{
    double wb = 1.0;
#pragma unroll
    for (int d = 0; d < 3; d++)
        if (p.order[d]) wb *= 2.0 * fabs(p.scale[d]) + 2.0;
    const int e = ((mb > 0) ? ilogb(mb) + 1 : -1000) + (wb > 1.0 ? ilogb(wb) + 1 : 0);
    const int64_t cells = (int64_t)ntiles_seg * TCELLS;
    int lg = 0;
    while ((cells << lg) < n && lg < 40) lg++;                        // mean density <= 2^lg
    int f = 30 - e - PMX_HEADROOM32 - lg;
    return f < -1020 ? -1020 : (f > 1020 ? 1020 : f);
}

// The particles [start, start + count) of a tile's list deposited into its 32-bit region; returns the OR of what the
// cells held before each add (SIGNED: shifted by 2^30, so that bit 31 says "at or beyond 2^30 in magnitude").
// PE = 4 / 8: the positions are dense rows of three floats / doubles (one 12-byte load, or a 16- and an 8-byte one, per
// particle); PE = 0: any strides and element size.  !SIGNED implies a scalar mass and no differentiated axis.
// PART (a tile that overflowed its optimistic scale is deposited in two parts, see paint_tile32_kernel): 0 = the
// contribution v as it is; 1 = its high part v >> sh (arithmetic: floor); 2 = its low part v & (2^sh - 1)
template <int KIND, int TTHREADS, bool SORTED, int PE, bool WHOLE, bool SIGNED, int PART = 0>
__device__ __forceinline__ uint32_t tile_deposit32(const pmx_painter &p, const BinGeom &g, const int *t, const DVec &pos,
                                                   const DVec &mass, double mass_scalar, const uint32_t *list,
                                                   int64_t start, int count, uint32_t *lds, double scale, int sh = 0)
{
    constexpr int S = Tuned<KIND>::S;
    using Rg = Region<S>;
    constexpr int R1 = Rg::R1, P = Tile32<KIND>::P;
#ifndef PMX_SWAP32
#define PMX_SWAP32 1
#endif
    constexpr bool SWAP = PMX_SWAP32 && UNROLL == 2;
    // stencil points in groups of G atomics; the values a group returns are folded into the guard behind the NEXT
    // group's atomics: G registers of returns in flight instead of S^3 (left alone the compiler issues all S^3 first)
    constexpr int G = S == 3 ? 9 : 8, NG = S * S * S / G;
    uint32_t guard = 0;
    const uint32_t *tl = list + start;
    // ([r5] measured and dropped once more, now that this loop is bound by the LDS: the 64 list entries of a wave dealt to
    // its lanes so that the 16 lanes the LDS serves together hold every second entry — 0.96 -> 1.07 ms on config 3, PCS
    // 1.74 -> 1.83: the permuted 12-byte row loads cost more than the conflicts they avoid.)
    for (int j0 = threadIdx.x; j0 < count; j0 += TTHREADS * UNROLL) {
        // both list entries, then both rows: no load waits behind a branch (an entry beyond the end reads the last one)
        bool ok[UNROLL];
        uint32_t id[UNROLL];
        PosRow<PE> row[UNROLL];
        double m[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int j = j0 + u * TTHREADS;
            ok[u] = j < count;
            const int jc = ok[u] ? j : count - 1;
            id[u] = SORTED ? (uint32_t)jc : tl[jc];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            row[u] = pos_row<PE>(pos, SORTED ? start + id[u] : (int64_t)id[u]);
            // (sorted: `id` is the list slot; a per-particle mass lives at the row the list names)
            if (SIGNED) m[u] = mass.data ? mass.get(SORTED ? (int64_t)tl[id[u]] : (int64_t)id[u], 0) : mass_scalar;
        }
        if (SWAP && (threadIdx.x & 1)) {
            // (odd lanes take their second particle first: see tile_deposit)
            const PosRow<PE> tr = row[0]; row[0] = row[1]; row[1] = tr;
            const bool to = ok[0]; ok[0] = ok[1]; ok[1] = to;
            if (SIGNED) { const double tm = m[0]; m[0] = m[1]; m[1] = tm; }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (!ok[u]) continue;
            int lb[3];
            double V[3][S];
            const double x[3] = {(double)row[u].x[0], (double)row[u].x[1], (double)row[u].x[2]};
            particle_setup_fast<KIND, WHOLE, double, !SIGNED>(p, g, t, x, V, lb);
            // (a plan that no longer matches the positions must not index outside the region)
            if ((unsigned)lb[0] >= (unsigned)T0 || (unsigned)lb[1] >= (unsigned)T1 || (unsigned)lb[2] >= (unsigned)T2) { stale_row(g); continue; }
            const double mu = (SIGNED ? m[u] : mass_scalar) * scale;
#pragma unroll
            for (int a = 0; a < S; a++) V[0][a] *= mu;
            uint32_t *base = lds + (lb[0] * R1 + lb[1]) * P + lb[2];
            uint32_t old[2][G];
            double fb = 0;
#pragma unroll
            for (int gi = 0; gi < NG; gi++) {
#pragma unroll
                for (int k = 0; k < G; k++) {
                    const int q = gi * G + k, a = q / (S * S), b = (q / S) % S, c = q % S;
                    if (c == 0) fb = V[0][a] * V[1][b];
                    const double r = __builtin_fma(fb, V[2][c], FIXED_MAGIC);
                    uint32_t v = (uint32_t)__double_as_longlong(r);
                    if (PART == 1) v = (uint32_t)((int)v >> sh);
                    if (PART == 2) v &= (1u << sh) - 1u;
                    old[gi & 1][k] = atomicAdd(base + (a * R1 + b) * P + c, v);
                }
                if (gi > 0) {
#pragma unroll
                    for (int k = 0; k < G; k++) guard |= SIGNED ? old[(gi - 1) & 1][k] + 0x40000000u : old[(gi - 1) & 1][k];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < G; k++) guard |= SIGNED ? old[(NG - 1) & 1][k] + 0x40000000u : old[(NG - 1) & 1][k];
        }
    }
    return SIGNED ? guard & 0x80000000u : guard & 0xC0000000u;
}

// paint_tile_kernel for a float canvas with the 32-bit region: the same walk (segments of ZSEG tiles along x, the
// x-face carried in LDS), the same owned-box stores and halo staging (halo_merge_kernel / pmx_rowfft_halo read them
// the same way).  SIGNED: contributions of either sign (derivative weights, per-particle masses).  WHOLE (the block is the
// whole periodic mesh: one rank) and PE (bytes of a position element) are the launcher's to know: every form of the
// deposit loop is a kernel of its own with its own registers.
template <int KIND, int TTHREADS, bool SORTED, bool SIGNED, bool WHOLE, int PE>
__global__ void __launch_bounds__(TTHREADS, (Tile32<KIND>::waves)) paint_tile32_kernel(pmx_painter p, BinGeom g, char *canvas, DVec pos,
                                                            DVec mass, double mass_scalar,
                                                            const uint32_t *list, const int64_t *offsets,
                                                            const uint32_t *counts, float *halo, int overwrite,
                                                            const double *mstats, int want_odd)
{
    if (!batch_is_mine(mstats, want_odd)) return;
    constexpr int S = Tuned<KIND>::S;
    using Rg = Region<S>;
    constexpr int R1 = Rg::R1, P = Tile32<KIND>::P, CELLS = Tile32<KIND>::CELLS;
    __shared__ uint32_t lds[CELLS];
    __shared__ uint32_t flag[2];
    constexpr bool WALK_X = walk_x(S);
    constexpr int NCARRY = WALK_X ? (S - 1) * R1 * Rg::R2 : Rg::R0 * R1 * (S - 1);
    constexpr int CPT = (NCARRY + TTHREADS - 1) / TTHREADS;
    const int ntw = WALK_X ? g.nt[0] : g.nt[2];
    const int64_t tstride = WALK_X ? (int64_t)g.nt[1] * g.nt[2] : 1;
    const int nseg = (ntw + ZSEG - 1) / ZSEG;
    const int64_t ncolumn = g.ntiles / ntw;
    const int64_t nwork = ncolumn * nseg;
    if (threadIdx.x < 2) flag[threadIdx.x] = 0;
    int trial = 0;                              // flag[trial & 1] is the overflow word of the next deposit
    for (int64_t w = blockIdx.x; w < nwork; w += gridDim.x) {
      const int64_t column = w / nseg;
      const int seg = (int)(w - column * nseg);
      const int t2a = seg * ZSEG, t2b = (t2a + ZSEG < ntw) ? t2a + ZSEG : ntw;
      const int64_t tile0 = WALK_X ? column : column * ntw;
      bool live = false;
      long long carry[CPT];                    // the face carried from tile to tile of the segment (see below)
#pragma unroll
      for (int u = 0; u < CPT; u++) carry[u] = 0;
      int64_t nseg_part = 0;
      for (int t2 = t2a; t2 < t2b; t2++) {
          const uint32_t c = counts[tile0 + t2 * tstride];
          nseg_part += c < (uint32_t)g.chunk ? c : (uint32_t)g.chunk;
      }
      int f = fixed_exponent32(p, uniform_double(mstats ? mstats[0] : fabs(mass_scalar)), nseg_part, t2b - t2a);
      for (int t2 = t2a; t2 < t2b; t2++) {
        const int64_t tile = tile0 + t2 * tstride;
        const bool last = (t2 == t2b - 1);
        int t[3];
        tile_coords(g, tile, t);
        const int64_t start = offsets[tile];
        const int count = counts[tile] < (uint32_t)g.chunk ? (int)counts[tile] : g.chunk;
        if (count == 0 && !overwrite && !live) continue;
        constexpr bool TOUCH = !SORTED && S >= 4;
        uint32_t touched = 0;
        if constexpr (TOUCH) touched = list_touch(list, start, count);
        // `carry` (registers, 64 bits: a tile deposited in two parts hands on more than 32) was captured when the previous
        // tile of the segment was flushed: its face cells, in units of 2^-f
        auto zero_region = [&]() __attribute__((always_inline)) {
            // (16 bytes per store: the region is a multiple of four cells)
            static_assert(CELLS % 4 == 0, "32-bit region in 16-byte pieces");
            int z0 = threadIdx.x;
            asm volatile("" : "+v"(z0));
            for (int q = z0; q < CELLS / 4; q += TTHREADS) ((uint4 *)lds)[q] = make_uint4(0, 0, 0, 0);
            __syncthreads();
        };
        // the carried face as the first cells of the region: part 0 = the value, 1 = value >> sh, 2 = value & (2^sh - 1)
        auto place_carry = [&](int part, int sh) __attribute__((always_inline)) {
            if (!live) return;
#pragma unroll
            for (int u = 0; u < CPT; u++) {
                int q = threadIdx.x + u * TTHREADS;
                asm volatile("" : "+v"(q));      // (opaque: see paint_tile_kernel)
                if (q < NCARRY) {
                    const long long cv = carry[u];
                    const uint32_t v = part == 0 ? (uint32_t)cv : (part == 1 ? (uint32_t)(cv >> sh) : (uint32_t)(cv & ((1ll << sh) - 1)));
                    if (WALK_X) { const int c = q % Rg::R2, r = q / Rg::R2; lds[r * P + c] = v; }
                    else { const int c = q % (S - 1), r = q / (S - 1); lds[r * P + c] = v; }
                }
            }
            __syncthreads();
        };
        // region -> canvas and halo staging: every cell times `unit`; add: on top of what a first part has written
        auto flush = [&](double unit, bool add) __attribute__((always_inline)) {
            const bool ow = overwrite && !add;
            // owned box -> canvas, plain stores in rows of T2 cells
            if constexpr (WHOLE) {
                // the block is the whole periodic mesh, a multiple of the tile on every axis (pmx_binplan_supported): the
                // owned box lies inside it, cell (a, b, c) at a fixed offset from the tile's first — no wraps, no bounds
                char *tbase = canvas + (int64_t)t[0] * T0 * p.strides[0] + (int64_t)t[1] * T1 * p.strides[1] + (int64_t)t[2] * T2 * p.strides[2];
                const int s0 = (int)p.strides[0], s1 = (int)p.strides[1], s2 = (int)p.strides[2];
                int q0 = threadIdx.x;
                asm volatile("" : "+v"(q0));      // (opaque, likewise)
                for (int q = q0; q < TCELLS; q += TTHREADS) {
                    const int c = q % T2, r = q / T2;
                    const int b = r % T1, a = r / T1;
                    const float v = (float)((double)(int)lds[(a * R1 + b) * P + c] * unit);
                    float *dst = (float *)(tbase + ((int64_t)a * s0 + b * s1 + c * s2));
                    if (ow) *dst = v;
                    else *dst += v;
                }
            } else {
                for (int q = threadIdx.x; q < TCELLS; q += TTHREADS) {
                    const int c = q % T2, r = q / T2;
                    const int b = r % T1, a = r / T1;
                    int64_t goff;
                    bool in = true;
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        const int l = t[d] * tile_ext(d) - g.o[d] + (d == 0 ? a : (d == 1 ? b : c));
                        in = in && l >= 0 && l < p.size[d];
                    }
                    if (in && region_cell(p, g, t, a, b, c, &goff)) {
                        const float v = (float)((double)(int)lds[(a * R1 + b) * P + c] * unit);
                        float *dst = (float *)(canvas + goff);
                        if (ow) *dst = v;
                        else *dst += v;
                    }
                }
            }
            // halo -> staging (compact numbering, contiguous writes)
            float *hbase = halo + tile * (int64_t)Rg::HALO;
            int h0 = threadIdx.x;
            asm volatile("" : "+v"(h0));      // (opaque: no per-thread staging address kept — and spilled — across the tile loop)
            for (int h = h0; h < Rg::HALO; h += TTHREADS) {
                int a, b, c;
                Rg::halo_decode(h, &a, &b, &c);
                if (!last && (WALK_X ? a >= T0 : c >= T2)) continue;           // carried to the next tile instead
                const float v = (float)((double)(int)lds[(a * R1 + b) * P + c] * unit);
                if (add) hbase[h] += v;
                else hbase[h] = v;
            }
        };
        // the face the next tile of the segment starts from, out of the region: carry = (add ? carry : 0) + cell << sh
        auto capture = [&](int sh, bool add) __attribute__((always_inline)) {
            if (last) return;
#pragma unroll
            for (int u = 0; u < CPT; u++) {
                int q = threadIdx.x + u * TTHREADS;
                asm volatile("" : "+v"(q));
                if (q < NCARRY) {
                    int cell;
                    if (WALK_X) { const int c = q % Rg::R2, r = q / Rg::R2; cell = (int)lds[(T0 * R1 + r) * P + c]; }
                    else { const int c = q % (S - 1), r = q / (S - 1); cell = (int)lds[r * P + T2 + c]; }
                    carry[u] = (add ? carry[u] : 0ll) + ((long long)cell << sh);
                }
            }
        };
        // one deposit, and whether some cell came within a factor 2 of the 32 bits (uniform over the workgroup)
        auto overflowed = [&](uint32_t over) __attribute__((always_inline)) {
            if (over) flag[trial & 1] = 1;
            __syncthreads();
            const bool yes = flag[trial & 1] != 0;
            trial++;
            if (threadIdx.x == 0) flag[trial & 1] = 0;            // (the other word: read next behind two more barriers)
            return yes;
        };
        // A face that does not fit 32 bits (handed on by a tile in two parts) puts this tile in two parts from the start
        bool two = false;
        if (live) {
            uint32_t big = 0;
#pragma unroll
            for (int u = 0; u < CPT; u++)
                if ((int)(threadIdx.x + u * TTHREADS) < NCARRY && (carry[u] >= (1ll << 30) || carry[u] <= -(1ll << 30))) big = 1;
            two = overflowed(big);
        }
        if constexpr (TOUCH) list_touch_done(touched);
        if (!two) {
            zero_region();
            place_carry(0, 0);
#if defined(PMX_EXPERIMENT) && defined(PMX_EXP_NODEPOSIT32)
            const uint32_t over = 0;        // timing experiment: everything but the deposit loop
#else
            const uint32_t over = tile_deposit32<KIND, TTHREADS, SORTED, PE, WHOLE, SIGNED, 0>(p, g, t, pos, mass, mass_scalar, list, start, count, lds, pow2(f));
#endif
            two = overflowed(over);
            if (!two) {
                flush(pow2(-f), false);
                capture(0, false);
            }
        }
        if (two) {
            // Some cell came within a factor 2 of the 32 bits at the optimistic scale (a crowded tile).  The same
            // contributions, at the SAME scale, in two parts: v >> sh into one zeroed region, flushed, then v & (2^sh - 1)
            // into another, added on top — every contribution still rounded once to 2^-f, the sum exact.  sh: the low
            // parts of `count` particles (and of the carried face) cannot reach 2^30.  Should the high parts overflow
            // (more than 2^16 of the largest contributions on one cell) the scale itself is coarsened, before
            // anything is written.
            int lg = 0;
            while ((1 << lg) <= count + 1) lg++;                       // count + 1 < 2^lg
            const int sh = 29 - lg < 12 ? (29 - lg > 0 ? 29 - lg : 0) : 12;
            for (;;) {
                zero_region();
                place_carry(1, sh);
                const uint32_t over = tile_deposit32<KIND, TTHREADS, SORTED, PE, WHOLE, SIGNED, 1>(p, g, t, pos, mass, mass_scalar, list, start, count, lds, pow2(f), sh);
                if (!overflowed(over) || f <= -1000) break;
                f -= PMX_RETRY32;
#pragma unroll
                for (int u = 0; u < CPT; u++) carry[u] = (carry[u] + (1ll << (PMX_RETRY32 - 1))) >> PMX_RETRY32;
            }
            const long long keep_dummy = 0; (void)keep_dummy;
            long long carry_in[CPT];
#pragma unroll
            for (int u = 0; u < CPT; u++) carry_in[u] = carry[u];
            flush(pow2(sh - f), false);
            capture(sh, false);
            __syncthreads();
            // the low parts (the carried face's low part from what came in, not from what was just captured)
#pragma unroll
            for (int u = 0; u < CPT; u++) { const long long tmp = carry[u]; carry[u] = carry_in[u]; carry_in[u] = tmp; }
            zero_region();
            place_carry(2, sh);
            (void)tile_deposit32<KIND, TTHREADS, SORTED, PE, WHOLE, SIGNED, 2>(p, g, t, pos, mass, mass_scalar, list, start, count, lds, pow2(f), sh);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < CPT; u++) carry[u] = carry_in[u];
            flush(pow2(-f), true);
            capture(0, true);
        }
        live = !last;
        __syncthreads();
      }
    }
}

// second pass: add every tile's halo cells to their owners.  Runs after ALL owned boxes
// are stored (kernel boundary), so the atomics never race with a plain store.
// (Measured alternative: the owner tile pulling its neighbours' halos with plain
// loads/stores — deterministic, no atomics — took 690 us against 398 us for this kernel
// at 512^3: the strided single-cell faces cost a read-modify-write of a whole sector each.)
__device__ __forceinline__ bool batch_is_mine(const double *mstats, int want_odd);
template <int S, typename T, bool INTEGER = false>
__global__ void __launch_bounds__(TBLOCK) halo_merge_kernel(pmx_painter p, BinGeom g, char *canvas, const T *halo,
                                                            const uint32_t *counts, int overwrite,
                                                            const double *mstats = nullptr, int want_odd = 0)
{
    if (!batch_is_mine(mstats, want_odd)) return;      // (only the deterministic path has two staging buffers)
    using Rg = Region<S>;
    constexpr bool WALK_X = walk_x(S);
    for (int64_t tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        int t[3];
        tile_coords(g, tile, t);
        // paint_tile_kernel staged this tile iff it has particles, or an earlier tile of its z
        // segment has (the z-halo carried through it), or every tile was written (overwrite)
        const int tw = WALK_X ? t[0] : t[2];
        const int64_t tstride = WALK_X ? (int64_t)g.nt[1] * g.nt[2] : 1;
        const int t2a = (tw / ZSEG) * ZSEG;
        bool staged = overwrite != 0;
        for (int k = t2a; k <= tw && !staged; k++) staged = counts[tile - (tw - k) * tstride] != 0;
        if (!staged) continue;
        const bool last = (tw == (WALK_X ? g.nt[0] : g.nt[2]) - 1) || (tw % ZSEG == ZSEG - 1);
        const T *hbase = halo + tile * (int64_t)Rg::HALO;
        for (int h = threadIdx.x; h < Rg::HALO; h += TBLOCK) {
            int a, b, c;
            Rg::halo_decode(h, &a, &b, &c);
            if (!last && (WALK_X ? a >= T0 : c >= T2)) continue;     // not staged: carried to the next tile in LDS
            T v = hbase[h];
            int64_t goff;
            if (INTEGER) {
                // (deterministic paint: T is double, its bits a 64-bit integer: an exact, order-independent add)
                const unsigned long long bits = (unsigned long long)__double_as_longlong((double)v);
                if (bits != 0 && region_cell(p, g, t, a, b, c, &goff)) atomicAdd((unsigned long long *)(canvas + goff), bits);
                continue;
            }
            if (v == (T)0) continue;
            // ([r3] measured: the cells of the x- and y-faces that get this one contribution and no other while the
            // kernel runs — those S - 1 cells or more inside their owner along both other axes, 82 % of the faces for
            // CIC — as plain read-modify-writes in rows instead of atomics at the L2: 248 -> 653 us at 512^3.)
            if (region_cell(p, g, t, a, b, c, &goff)) unsafeAtomicAdd((T *)(canvas + goff), v);
        }
    }
}

// entries of `out` for particles that are in no tile (they touch no local cell) read 0
static __global__ void __launch_bounds__(TBLOCK) zero_dropped_kernel(const uint32_t *list, const int64_t *offsets,
                                                              const uint32_t *counts, int64_t ntiles, DVec out, int sorted)
{
    const int64_t n = counts[ntiles];   // the common case: nothing was dropped
    const int64_t start = offsets[ntiles];
    for (int64_t j = blockIdx.x * (int64_t)TBLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * TBLOCK)
        out.set(sorted ? start + j : (int64_t)list[start + j], 0, 0.0);
}

#ifndef PMX_READOUT768_WAVES
#define PMX_READOUT768_WAVES 6      // (waves per SIMD the 768-thread readouts are held to: two workgroups per CU; see readout_tile_lean_kernel)
#endif
template <int KIND, typename T, int TTHREADS, bool SORTED, bool RELAX>
__global__ void __launch_bounds__(TTHREADS, (TTHREADS == 768 ? PMX_READOUT768_WAVES : 1)) readout_tile_kernel(pmx_painter p, BinGeom g, const char *canvas,
                                                              DVec pos, DVec out, const uint32_t *list,
                                                              const int64_t *offsets, const uint32_t *counts)
{
    // SORTED: `pos` is the plan's copy of the positions in list order and `out` its buffer of
    // results in list order (unsort_kernel pulls them back): nothing is gathered or scattered
    constexpr int S = Tuned<KIND>::S;
    using Rg = Region<S>;
    constexpr int R1 = Rg::R1, R2 = Rg::R2;
    __shared__ T lds[Rg::template glds<T>()];
    __shared__ int64_t tab[Rg::R0 + Rg::R1 + Rg::R2];
    const bool whole = whole_mesh(p, g);
    for (int64_t tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        const int64_t start = offsets[tile];
        // (what a crowded tile holds beyond g.chunk entries is read out by readout_heavy_kernel)
        const int count = counts[tile] < (uint32_t)g.chunk ? (int)counts[tile] : g.chunk;
        if (count == 0) continue;
        int t[3];
        tile_coords(g, tile, t);
        region_tables<S, false>(p, g, t, tab, TTHREADS);
        __syncthreads();
#pragma unroll 4
        for (int q = threadIdx.x; q < Rg::CELLS; q += TTHREADS) {
            int c = q % R2, r = q / R2;
            int b = r % R1, a = r / R1;
            const int64_t o0 = tab[a], o1 = tab[Rg::R0 + b], o2 = tab[Rg::R0 + R1 + c];
            lds[r * Rg::template gpitch<T>() + c] = (o0 | o1 | o2) >= 0 ? *(const T *)(canvas + (o0 + o1 + o2)) : (T)0;   // outside the block reads as 0
        }
        __syncthreads();
        tile_gather_any<KIND, T, TTHREADS, SORTED, RELAX>(whole, p, g, t, pos, out, list, start, count, lds);
        __syncthreads();
    }
}

// ---- [r5] the readout of the common case as a loop of its own -----------------------------------------------------
// Relaxed arithmetic, positions in dense rows of three (PE = 4 / 8 bytes per element), results in a dense vector (OE),
// the index list (no tile-ordered copy); WHOLE (the block is the whole periodic mesh) known to the launcher.  What the
// general loop pays per trip and this one does not: a wait behind a branch between the two list entries, three
// position loads with 64-bit strides each (one 12-byte load, or a 16- and an 8-byte one), the element size of the
// results looked up per store.
template <int KIND, typename T, int TTHREADS, int PE, int OE, bool WHOLE>
__device__ __forceinline__ void tile_gather_lean(const pmx_painter &p, const BinGeom &g, const int *t, const DVec &pos,
                                                 char *out, const uint32_t *tl, int count, const T *lds, int ostride)
{
    constexpr int S = Tuned<KIND>::S;
    using Rg = Region<S>;
    constexpr int R1 = Rg::R1, GP = Rg::template gpitch<T>();
    for (int j0 = threadIdx.x; j0 < count; j0 += TTHREADS * UNROLL) {
        bool ok[UNROLL];
        uint32_t id[UNROLL];
        PosRow<PE> row[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int j = j0 + u * TTHREADS;
            ok[u] = j < count;
            id[u] = tl[ok[u] ? j : count - 1];            // (an entry beyond the end reads the last one: no load behind a branch)
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) row[u] = pos_row<PE>(pos, (int64_t)id[u]);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (!ok[u]) continue;
            int lb[3];
            T W[3][S];
            const double x[3] = {(double)row[u].x[0], (double)row[u].x[1], (double)row[u].x[2]};
            particle_setup_fast<KIND, WHOLE, T>(p, g, t, x, W, lb);
            if ((unsigned)lb[0] >= (unsigned)T0 || (unsigned)lb[1] >= (unsigned)T1 || (unsigned)lb[2] >= (unsigned)T2) {
                stale_row(g);                          // (a row the plan no longer matches reads 0, not what was in `out`)
                if (OE == 8) *(double *)(out + (int64_t)id[u] * ostride) = 0.0;
                else *(float *)(out + (int64_t)id[u] * ostride) = 0.0f;
                continue;
            }
            const T *base = lds + (lb[0] * R1 + lb[1]) * GP + lb[2];
            T acc = 0;
#pragma unroll
            for (int a = 0; a < S; a++) {
                T plane = 0;
#pragma unroll
                for (int b = 0; b < S; b++) {
                    T r = 0;
#pragma unroll
                    for (int c = 0; c < S; c++) r = fma_(base[(a * R1 + b) * GP + c], W[2][c], r);
                    plane = fma_(W[1][b], r, plane);
                }
                acc = fma_(W[0][a], plane, acc);
            }
            // (ostride: bytes from one result to the next — OE for a dense vector, more for a column of the caller's array)
            if (OE == 8) *(double *)(out + (int64_t)id[u] * ostride) = (double)acc;
            else *(float *)(out + (int64_t)id[u] * ostride) = (float)acc;
        }
    }
}

// The kernel also does what the general readout leaves to two launches of their own: the particles that touch no local
// cell read 0 (zero_dropped_kernel), and the pieces of crowded tiles beyond g.chunk entries (readout_heavy_kernel) are
// work units behind the tiles — a readout is read-only on the canvas, nothing orders them.  (Two launches of ~4.5 us per
// readout: nothing at 512^3 on one GPU, 2 % of a rank's particle kernels at 8 ranks.)
// (768 threads — PCS on double canvases, an 81 KB region: TWO workgroups per CU need 6 waves per SIMD, i.e. at most 80
// VGPRs.  The form for blocks of any shape came to 83 and ran ONE: 2.62 ms against 1.75 at 512^3 — every pencil rank
// of config 5 runs that form.  The bound makes the compiler hold it.)
// [r6] PMX_READOUT_XCD: tiles in XCD order (xcd_tile): the z- and y-neighbours of a tile, which fetch the face lines of
// its region as rows of their own, run at the same time on the same L2.  Measured between two builds of this file that
// differ in nothing else (scripts/r06/lib_ab2.sh, profiles/r06_rxcd/): PCS gains (512^3 1.75-1.77 -> 1.66-1.69 ms,
// config 5's per-GPU load 33.1-33.4 -> 32.5-32.6: its region is 11 x 19 x 35 cells for a box of 8 x 16 x 32, the faces are
// half of what it fetches), CIC and TSC LOSE (1.12-1.16 -> 1.19-1.24, 1.46 -> 1.49-1.50; rows that have drifted 4 cells
// 2.03 -> 2.63): eight far-apart streams through the particle rows, which are most of their traffic, instead of one.
// 1 (default): PCS only; 2: every window (the measurement build); 0: none.
#ifndef PMX_READOUT_XCD
#define PMX_READOUT_XCD 1
#endif
template <int KIND, typename T, int TTHREADS, int PE, int OE, bool WHOLE>
__global__ void __launch_bounds__(TTHREADS, (TTHREADS == 768 ? PMX_READOUT768_WAVES : 1)) readout_tile_lean_kernel(pmx_painter p, BinGeom g, const char *canvas,
                                                                   DVec pos, char *out, const uint32_t *list,
                                                                   const int64_t *offsets, const uint32_t *counts,
                                                                   const uint64_t *items, const uint32_t *nitems, uint32_t cap,
                                                                   int ostride)
{
    constexpr int S = Tuned<KIND>::S;
    using Rg = Region<S>;
    constexpr int R1 = Rg::R1, R2 = Rg::R2;
    __shared__ T lds[Rg::template glds<T>()];
    __shared__ int64_t tab[Rg::R0 + Rg::R1 + Rg::R2];
    {
        // entries of `out` for particles that are in no tile (the common case: none)
        const int64_t nd = counts[g.ntiles];
        const uint32_t *dl = list + offsets[g.ntiles];
        for (int64_t j = blockIdx.x * (int64_t)TTHREADS + threadIdx.x; j < nd; j += (int64_t)gridDim.x * TTHREADS) {
            if (OE == 8) *(double *)(out + (int64_t)dl[j] * ostride) = 0.0;
            else *(float *)(out + (int64_t)dl[j] * ostride) = 0.0f;
        }
    }
    const int64_t nh = *nitems < cap ? *nitems : cap;
    for (int64_t unit = blockIdx.x; unit < g.ntiles + nh; unit += gridDim.x) {
        int64_t tile = unit, first = 0;
        if ((PMX_READOUT_XCD >= 2 || (PMX_READOUT_XCD == 1 && S >= 4)) && unit < g.ntiles) tile = xcd_tile(unit, g.ntiles);
        if (unit >= g.ntiles) {
            const uint64_t it = items[unit - g.ntiles];
            tile = (int64_t)(it >> 20);
            first = (int64_t)(it & 0xFFFFF) * g.chunk;
        }
        const int64_t start = offsets[tile] + first;
        const int64_t left = (int64_t)counts[tile] - first;
        const int count = left < g.chunk ? (int)left : g.chunk;
        if (count <= 0) continue;
        int t[3];
        tile_coords(g, tile, t);
        region_tables<S, false>(p, g, t, tab, TTHREADS);
        __syncthreads();
#pragma unroll 4
        for (int q = threadIdx.x; q < Rg::CELLS; q += TTHREADS) {
            int c = q % R2, r = q / R2;
            int b = r % R1, a = r / R1;
            const int64_t o0 = tab[a], o1 = tab[Rg::R0 + b], o2 = tab[Rg::R0 + R1 + c];
            lds[r * Rg::template gpitch<T>() + c] = (o0 | o1 | o2) >= 0 ? *(const T *)(canvas + (o0 + o1 + o2)) : (T)0;
        }
        __syncthreads();
        tile_gather_lean<KIND, T, TTHREADS, PE, OE, WHOLE>(p, g, t, pos, out, list + start, count, lds, ostride);
        __syncthreads();
    }
}

// [r6] The same for up to PMX_MAXFIELDS canvases of ONE block geometry read at the same positions, the results side by
// side in the rows of the caller's array (the three force components of a PM step: F[i, d] = field_d at x_i).  One
// workgroup serves a tile for every canvas in turn: the tile's positions and list come from HBM once (the later turns
// find them in the L2), and the components of a result row are stored within microseconds of each other — the L2 holds
// the line until all of it is written.  A column at a time (readout(out=F[:, d]), three launches) every 64-byte piece
// of F goes to memory and back three times: 2.54 ms per component against 1.12 into a dense vector at 512^3.
struct CanvasSet { const char *ptr[PMX_MAXFIELDS]; int32_t n; int32_t ostride1; };
template <int KIND, typename T, int TTHREADS, int PE, int OE, bool WHOLE>
__global__ void __launch_bounds__(TTHREADS, (TTHREADS == 768 ? PMX_READOUT768_WAVES : 1)) readout_tile_multi_kernel(pmx_painter p, BinGeom g, CanvasSet cs,
                                                                   DVec pos, char *out, const uint32_t *list,
                                                                   const int64_t *offsets, const uint32_t *counts,
                                                                   const uint64_t *items, const uint32_t *nitems, uint32_t cap,
                                                                   int ostride)
{
    constexpr int S = Tuned<KIND>::S;
    using Rg = Region<S>;
    constexpr int R1 = Rg::R1, R2 = Rg::R2;
    __shared__ T lds[Rg::template glds<T>()];
    __shared__ int64_t tab[Rg::R0 + Rg::R1 + Rg::R2];
    {
        // rows of `out` for particles that are in no tile
        const int64_t nd = counts[g.ntiles];
        const uint32_t *dl = list + offsets[g.ntiles];
        for (int64_t j = blockIdx.x * (int64_t)TTHREADS + threadIdx.x; j < nd * cs.n; j += (int64_t)gridDim.x * TTHREADS) {
            char *o = out + (int64_t)dl[j / cs.n] * ostride + (j % cs.n) * cs.ostride1;
            if (OE == 8) *(double *)o = 0.0;
            else *(float *)o = 0.0f;
        }
    }
    const int64_t nh = *nitems < cap ? *nitems : cap;
    for (int64_t unit = blockIdx.x; unit < g.ntiles + nh; unit += gridDim.x) {
        int64_t tile = unit, first = 0;
        if ((PMX_READOUT_XCD >= 2 || (PMX_READOUT_XCD == 1 && S >= 4)) && unit < g.ntiles) tile = xcd_tile(unit, g.ntiles);
        if (unit >= g.ntiles) {
            const uint64_t it = items[unit - g.ntiles];
            tile = (int64_t)(it >> 20);
            first = (int64_t)(it & 0xFFFFF) * g.chunk;
        }
        const int64_t start = offsets[tile] + first;
        const int64_t left = (int64_t)counts[tile] - first;
        const int count = left < g.chunk ? (int)left : g.chunk;
        if (count <= 0) continue;
        int t[3];
        tile_coords(g, tile, t);
        region_tables<S, false>(p, g, t, tab, TTHREADS);
        __syncthreads();
        for (int f = 0; f < cs.n; f++) {
            const char *canvas = cs.ptr[f];
#pragma unroll 4
            for (int q = threadIdx.x; q < Rg::CELLS; q += TTHREADS) {
                int c = q % R2, r = q / R2;
                int b = r % R1, a = r / R1;
                const int64_t o0 = tab[a], o1 = tab[Rg::R0 + b], o2 = tab[Rg::R0 + R1 + c];
                lds[r * Rg::template gpitch<T>() + c] = (o0 | o1 | o2) >= 0 ? *(const T *)(canvas + (o0 + o1 + o2)) : (T)0;
            }
            __syncthreads();
            tile_gather_lean<KIND, T, TTHREADS, PE, OE, WHOLE>(p, g, t, pos, out + f * cs.ostride1, list + start, count, lds, ostride);
            __syncthreads();
        }
    }
}

// ---- crowded tiles ------------------------------------------------------------------------
// work items (tile, piece >= 1) for what the tiles hold beyond `chunk` list entries; the last kernel of every build,
// so it also hands the build's measurement of the row order (flags[1..2]) to the host's mapped slot (no copy node)
static __global__ void __launch_bounds__(TBLOCK) heavy_items_kernel(const uint32_t *counts, int64_t ntiles, int chunk,
                                                             uint64_t *items, uint32_t *nitems, uint32_t cap,
                                                             const uint32_t *flags, uint32_t *host_measure)
{
    if (host_measure != nullptr && blockIdx.x == 0 && threadIdx.x < 2)
        __hip_atomic_store(&host_measure[threadIdx.x], flags[1 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int64_t tile = blockIdx.x * (int64_t)TBLOCK + threadIdx.x; tile < ntiles; tile += (int64_t)gridDim.x * TBLOCK) {
        const uint32_t c = counts[tile];
        if (c > (uint32_t)chunk) {
            const uint32_t extra = (c - 1) / (uint32_t)chunk;
            const uint32_t base = atomicAdd(nitems, extra);
            for (uint32_t k = 1; k <= extra; k++)
                if (base + k - 1 < cap) items[base + k - 1] = ((uint64_t)tile << 20) | k;
        }
    }
}

// one workgroup per work item: the piece is accumulated in LDS like a tile of its own and the whole
// region, box and halo, is added to the canvas with atomics (after the tile kernel and halo_merge)
template <int KIND, typename T, int TTHREADS, bool SORTED, int MODE>
__global__ void __launch_bounds__(TTHREADS) paint_heavy_kernel(pmx_painter p, BinGeom g, char *canvas, DVec pos,
                                                             DVec mass, double mass_scalar, const uint32_t *list,
                                                             const int64_t *offsets, const uint32_t *counts,
                                                             const uint64_t *items, const uint32_t *nitems, uint32_t cap,
                                                             const double *mstats, int want_odd, const int32_t *dexp,
                                                             pmx_painter pw)
{
    constexpr bool FIXED = MODE != 0;
    if (!batch_is_mine(mstats, want_odd)) return;
    const pmx_painter &pwr = MODE == 2 ? pw : p;
    constexpr int S = Tuned<KIND>::S;
    using Rg = Region<S>;
    constexpr int R1 = Rg::R1, R2 = Rg::R2;
    __shared__ double lds[Rg::DLDS];
    const uint32_t n = *nitems < cap ? *nitems : cap;
    for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
        const int64_t tile = (int64_t)(items[item] >> 20);
        const int64_t piece = (int64_t)(items[item] & 0xFFFFF);
        int t[3];
        tile_coords(g, tile, t);
        const int64_t first = piece * g.chunk;
        const int64_t left = (int64_t)counts[tile] - first;
        const int count = left < g.chunk ? (int)left : g.chunk;
        for (int q = threadIdx.x; q < Rg::DLDS; q += TTHREADS) lds[q] = 0;
        __syncthreads();
        double scale = 1.0, inv = 1.0;
        if (FIXED) {
            const int f = MODE == 2 ? *dexp : fixed_exponent(pwr, mstats ? mstats[0] : fabs(mass_scalar), count);
            scale = pow2(f);
            inv = pow2(-f);
        }
        tile_deposit_any<KIND, TTHREADS, SORTED, FIXED, typename DepositWeights<KIND, T, MODE>::type>(whole_mesh(pwr, g), pwr, g, t, pos, mass, mass_scalar, list, offsets[tile] + first, count, lds, scale);
        __syncthreads();
        for (int q = threadIdx.x; q < Rg::CELLS; q += TTHREADS) {
            const int c = q % R2, r = q / R2;
            int64_t goff;
            if (MODE == 2) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(lds[Rg::dat(r, c)]);
                if (bits != 0 && region_cell(p, g, t, r / R1, r % R1, c, &goff)) atomicAdd((unsigned long long *)(canvas + goff), bits);
                continue;
            }
            const double v = cell_value<MODE>(lds[Rg::dat(r, c)], inv);
            if (v == 0) continue;
            if (region_cell(p, g, t, r / R1, r % R1, c, &goff)) unsafeAtomicAdd((T *)(canvas + goff), (T)v);
        }
        __syncthreads();
    }
}

template <int KIND, typename T, int TTHREADS, bool SORTED, bool RELAX>
__global__ void __launch_bounds__(TTHREADS) readout_heavy_kernel(pmx_painter p, BinGeom g, const char *canvas, DVec pos,
                                                               DVec out, const uint32_t *list, const int64_t *offsets,
                                                               const uint32_t *counts, const uint64_t *items,
                                                               const uint32_t *nitems, uint32_t cap)
{
    constexpr int S = Tuned<KIND>::S;
    using Rg = Region<S>;
    constexpr int R1 = Rg::R1, R2 = Rg::R2;
    __shared__ T lds[Rg::template glds<T>()];
    const uint32_t n = *nitems < cap ? *nitems : cap;
    for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
        const int64_t tile = (int64_t)(items[item] >> 20);
        const int64_t piece = (int64_t)(items[item] & 0xFFFFF);
        int t[3];
        tile_coords(g, tile, t);
        const int64_t first = piece * g.chunk;
        const int64_t left = (int64_t)counts[tile] - first;
        const int count = left < g.chunk ? (int)left : g.chunk;
        for (int q = threadIdx.x; q < Rg::CELLS; q += TTHREADS) {
            const int c = q % R2, r = q / R2;
            int64_t goff;
            const bool in = region_cell(p, g, t, r / R1, r % R1, c, &goff);
            lds[r * Rg::template gpitch<T>() + c] = in ? *(const T *)(canvas + goff) : (T)0;
        }
        __syncthreads();
        tile_gather_any<KIND, T, TTHREADS, SORTED, RELAX>(whole_mesh(p, g), p, g, t, pos, out, list, offsets[tile] + first, count, lds);
        __syncthreads();
    }
}

// The masses of a batch as the fixed-point kernels need to know them (stats: 4 x 8 bytes, zeroed first):
//   [0] max |m| over the finite masses (the bit pattern of a non-negative double orders like an integer),
//   [2] how many are not finite, [3] the complement of the bits of the smallest non-zero |m| (0: none seen; the
//   complement so that a zeroed word and atomicMax do).
// mass_stats_finish_kernel then sets [1] (double): 0 = the fixed-point kernels serve the batch, != 0 = its
// floating-point twin does: a NaN / Inf mass, or masses spread over more than FIXED_MASS_RANGE_LOG2 binary
// orders of magnitude.  The one scale of a segment comes from the LARGEST |mass|; a contribution of a mass 2^-R
// of that is rounded to 2^-50 + R of itself, where the reference's floating adds keep 2^-53 of the cell's sum:
// beyond R = 20 (two species at 1 : 1e-6) the light species would lose more than the 1e-9 relative that
// INTEGRATION.md section 1 promises for cells that only light particles reach.
constexpr int FIXED_MASS_RANGE_LOG2 = 20;
static __global__ void __launch_bounds__(TBLOCK) mass_stats_kernel(DVec mass, int64_t n, unsigned long long *stats)
{
    double mx = 0, mn = 1.7976931348623157e308;
    unsigned long long odd = 0;
    for (int64_t i = blockIdx.x * (int64_t)TBLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * TBLOCK) {
        const double m = fabs(mass.get(i, 0));
        if (m <= 1.7e308) {
            mx = m > mx ? m : mx;
            mn = (m > 0 && m < mn) ? m : mn;
        } else odd++;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(mx, off), q = __shfl_down(mn, off);
        mx = o > mx ? o : mx;
        mn = q < mn ? q : mn;
        odd += __shfl_down(odd, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&stats[0], (unsigned long long)__double_as_longlong(mx));
        if (mn < 1.7976931348623157e308) atomicMax(&stats[3], ~(unsigned long long)__double_as_longlong(mn));
        if (odd) atomicAdd(&stats[2], odd);
    }
}
static __global__ void mass_stats_finish_kernel(unsigned long long *stats)
{
    const double mx = __longlong_as_double((long long)stats[0]);
    bool floating = stats[2] != 0;
    if (stats[3] != 0) {
        const double mn = __longlong_as_double((long long)~stats[3]);
        if (ilogb(mx) - ilogb(mn) > FIXED_MASS_RANGE_LOG2) floating = true;
    }
    ((double *)stats)[1] = floating ? 1.0 : 0.0;
}

// deterministic paint: the one scale of the batch.  A cell can receive from the particles of the (at most) 8 tiles
// whose regions contain it: n = 8 max(counts).
static __global__ void __launch_bounds__(1024) det_scale_kernel(pmx_painter p, const uint32_t *counts, int64_t ntiles,
                                                         const double *mstats, double mass_scalar, int32_t *dexp)
{
    __shared__ uint32_t mx[1024];
    uint32_t m = 0;
    for (int64_t i = threadIdx.x; i < ntiles; i += 1024) m = counts[i] > m ? counts[i] : m;
    mx[threadIdx.x] = m;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off && mx[threadIdx.x + off] > mx[threadIdx.x]) mx[threadIdx.x] = mx[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) dexp[0] = fixed_exponent(p, mstats ? mstats[0] : fabs(mass_scalar), 8 * (int64_t)mx[0] + 1);
}

// canvas (+)= scratch 2^-f, cell by cell of the block: the one rounding of a deterministic paint
template <typename T>
__global__ void __launch_bounds__(TBLOCK) det_finish_kernel(pmx_painter p, char *canvas, const long long *scratch,
                                                            const int32_t *dexp, int overwrite, const double *mstats)
{
    if (mstats != nullptr && mstats[1] != 0.0) return;       // the floating-point kernels served this batch
    const double inv = pow2(-dexp[0]);
    const int64_t n1 = p.size[1], n2 = p.size[2], total = p.size[0] * n1 * n2;
    for (int64_t q = blockIdx.x * (int64_t)TBLOCK + threadIdx.x; q < total; q += (int64_t)gridDim.x * TBLOCK) {
        const int64_t i2 = q % n2, r = q / n2, i1 = r % n1, i0 = r / n1;
        T *dst = (T *)(canvas + i0 * p.strides[0] + i1 * p.strides[1] + i2 * p.strides[2]);
        const T v = (T)((double)scratch[q] * inv);
        if (overwrite) *dst = v;
        else *dst += v;
    }
}

#if PMX_PART_PLAN
int plan_ensure(void **ptr, size_t *cap, size_t need)
{
    if (need <= *cap) return PMX_OK;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr;
    *cap = 0;
    PMX_HIP_CHECK(hipMalloc(ptr, need));
    *cap = need;
    return PMX_OK;
}
#endif

static bool same_geometry(const pmx_painter &a, const pmx_painter &b)
{
    if (a.kind != b.kind || a.ndim != b.ndim) return false;
    for (int d = 0; d < 3; d++)
        if (a.scale[d] != b.scale[d] || a.translate[d] != b.translate[d] || a.period[d] != b.period[d] ||
            a.size[d] != b.size[d])
            return false;
    return true;
}

static int halo_cells(int S)
{
    int R0 = T0 + S - 1, R1 = T1 + S - 1, R2 = T2 + S - 1;
    return R0 * R1 * R2 - TCELLS;
}

}  // namespace pmx

using namespace pmx;

#if PMX_PART_PLAN
extern "C" int pmx_binplan_create(pmx_binplan **plan)
{
    PMX_REQUIRE(plan != nullptr, PMX_EINVAL, "plan pointer is NULL");
    *plan = new pmx_binplan();
    return PMX_OK;
}

extern "C" int pmx_binplan_configure(pmx_binplan *pl, int32_t form)
{
    PMX_REQUIRE(pl != nullptr, PMX_EINVAL, "plan is NULL");
    PMX_REQUIRE(form >= -1 && form <= 2, PMX_EINVAL, "form must be -1 (auto), 0 (tiles) or 2 (tiles, chunk rebuild)");
    PMX_REQUIRE(form != 1, PMX_EUNSUPPORTED, "form 1 (the walk kernels of rounds 2-3) is no longer part of the library");
    if (pl->form != form) pl->have_history = false;
    pl->form = form;
    return PMX_OK;
}

extern "C" int pmx_binplan_exact(pmx_binplan *pl, int32_t on)
{
    PMX_REQUIRE(pl != nullptr, PMX_EINVAL, "plan is NULL");
    pl->exact = on ? 1 : 0;
    return PMX_OK;
}

extern "C" int pmx_binplan_deterministic(pmx_binplan *pl, int32_t on)
{
    PMX_REQUIRE(pl != nullptr, PMX_EINVAL, "plan is NULL");
    pl->deterministic = on ? 1 : 0;
    return PMX_OK;
}

extern "C" int pmx_binplan_mass_stats(pmx_binplan *pl, const double *stats)
{
    PMX_REQUIRE(pl != nullptr, PMX_EINVAL, "plan is NULL");
    pl->mass_stats_ext = stats;      // NULL: found by the next paint itself
    return PMX_OK;
}

extern "C" int pmx_binplan_sorted(pmx_binplan *pl, int32_t pref, int32_t *is_sorted)
{
    PMX_REQUIRE(pl != nullptr, PMX_EINVAL, "plan is NULL");
    PMX_REQUIRE(pref >= -2 && pref <= 1, PMX_EINVAL, "pref must be -1 (auto), 0 (never), 1 (always) or -2 (query only)");
    if (pref >= -1 && pl->sort_pref != pref) {
        pl->sort_pref = pref;
        pl->have_history = false;
    }
    if (is_sorted) *is_sorted = pl->built && pl->sorted ? 1 : 0;
    return PMX_OK;
}

extern "C" int pmx_binplan_overflows(pmx_binplan *pl, uint32_t *count)
{
    PMX_REQUIRE(pl != nullptr && count != nullptr, PMX_EINVAL, "NULL argument");
    *count = pl->host_flag ? *(volatile uint32_t *)pl->host_flag : 0u;
    return PMX_OK;
}

extern "C" int pmx_binplan_builds(pmx_binplan *pl, uint32_t *single_pass, uint32_t *two_pass)
{
    PMX_REQUIRE(pl != nullptr && single_pass != nullptr && two_pass != nullptr, PMX_EINVAL, "NULL argument");
    *single_pass = pl->builds[0];
    *two_pass = pl->builds[1];
    return PMX_OK;
}

// ---- [r6] the plan's order of the rows, for the caller ------------------------------------------------------------------
// order[k] = the row that stands k-th when the rows are taken tile by tile (tiles in index order, inside a tile in the
// order the bin pass met them — the order of the rows themselves, block by block), the rows that touch no local cell
// last: what ParticleMesh.tile_order hands a time-stepping caller to re-sort its particle arrays with.  The plan has
// this order already (its index list, with slack between the tiles): an exclusive scan of the counts and one copy,
// ~0.3 ms for 134 M rows where torch.argsort of 64-bit keys took 20.
static __global__ void __launch_bounds__(1024) order_scan_kernel(const uint32_t *counts, int64_t nbuckets, int64_t *first)
{
    __shared__ int64_t sh[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nbuckets; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < nbuckets ? (int64_t)counts[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int64_t t = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nbuckets) first[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
}
static __global__ void __launch_bounds__(TBLOCK) order_copy_kernel(const uint32_t *list, const int64_t *offsets, const uint32_t *counts,
                                                            const int64_t *first, int64_t nbuckets, int64_t *order)
{
    for (int64_t t = blockIdx.x; t < nbuckets; t += gridDim.x) {
        const uint32_t *src = list + offsets[t];
        int64_t *dst = order + first[t];
        const int64_t n = counts[t];
        for (int64_t j = threadIdx.x; j < n; j += TBLOCK) dst[j] = (int64_t)src[j];
    }
}
extern "C" int pmx_binplan_order(pmx_binplan *pl, int64_t *order, void *stream)
{
    PMX_REQUIRE(pl && pl->built, PMX_EINVAL, "bin plan is not built");
    PMX_REQUIRE(order != nullptr || pl->npart == 0, PMX_EINVAL, "order is NULL");
    if (pl->npart == 0) return PMX_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nbuckets = pl->g.ntiles + 1;
    int64_t *first = nullptr;
    PMX_HIP_CHECK(hipMallocAsync((void **)&first, (size_t)nbuckets * sizeof(int64_t), st));
    order_scan_kernel<<<1, 1024, 0, st>>>(pl->counts, nbuckets, first);
    const unsigned grid = (unsigned)(nbuckets < 65535 * 8 ? nbuckets : 65535 * 8);
    order_copy_kernel<<<grid, TBLOCK, 0, st>>>(pl->list, pl->offsets, pl->counts, first, nbuckets, order);
    hipError_t e = hipGetLastError();
    (void)hipFreeAsync(first, st);
    PMX_HIP_CHECK(e);
    return PMX_OK;
}

extern "C" int pmx_binplan_stale(pmx_binplan *pl, uint32_t *count)
{
    PMX_REQUIRE(pl != nullptr && count != nullptr, PMX_EINVAL, "NULL argument");
    *count = pl->host_flag ? *(volatile uint32_t *)(pl->host_flag + 1) : 0u;
    return PMX_OK;
}

extern "C" int pmx_binplan_destroy(pmx_binplan *pl)
{
    if (!pl) return PMX_OK;
    if (pl->tid) (void)hipFree(pl->tid);
    if (pl->list) (void)hipFree(pl->list);
    if (pl->ctl) (void)hipFree(pl->ctl);          // (flags, nheavy and counts live in it)
    if (pl->offsets) (void)hipFree(pl->offsets);
    if (pl->cursor) (void)hipFree(pl->cursor);
    if (pl->host_flag) (void)hipHostFree(pl->host_flag);
    if (pl->halo) (void)hipFree(pl->halo);
    if (pl->pos_copy) (void)hipFree(pl->pos_copy);
    if (pl->inv) (void)hipFree(pl->inv);
    if (pl->out_sorted) (void)hipFree(pl->out_sorted);
    if (pl->host_groups) (void)hipHostFree(pl->host_groups);
    if (pl->mstats) (void)hipFree(pl->mstats);
    if (pl->dscratch) (void)hipFree(pl->dscratch);
    if (pl->dhalo) (void)hipFree(pl->dhalo);
    if (pl->heavy_items) (void)hipFree(pl->heavy_items);
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
    const int S = native_support(p->kind);
    const int T[3] = {T0, T1, T2};
    for (int d = 0; d < 3; d++) {
        PMX_REQUIRE(p->size[d] >= 1, PMX_EUNSUPPORTED, "empty block");
        PMX_REQUIRE(p->period[d] == 0 || p->size[d] <= p->period[d], PMX_EUNSUPPORTED, "block larger than period");
        // every region cell must map to a distinct canvas cell
        int64_t span = p->period[d] > 0 ? p->period[d] : p->size[d];
        PMX_REQUIRE(span >= T[d] + S - 1, PMX_EUNSUPPORTED, "mesh smaller than a tile region");
        if (p->period[d] > 0 && p->size[d] == p->period[d]) {
            // full periodic axis: the last tile's halo wraps onto tile 0; the owned boxes must
            // tile the axis exactly
            PMX_REQUIRE(p->size[d] % T[d] == 0, PMX_EUNSUPPORTED, "periodic axis is not a multiple of the tile");
        } else if (p->period[d] > 0) {
            // a block within S-1 cells of the full period would let a stencil wrap from below
            // onto cells owned by another tile
            PMX_REQUIRE(p->size[d] <= p->period[d] - (S - 1), PMX_EUNSUPPORTED,
                        "block within S-1 cells of the full period");
        }
    }
    return PMX_OK;
}

extern "C" int pmx_binplan_build(pmx_binplan *pl, const pmx_painter *p_, const pmx_vec *pos, int64_t npart,
                                 void *stream)
{
    PMX_REQUIRE(pl != nullptr, PMX_EINVAL, "plan is NULL");
    PMX_REQUIRE(!pl->halo_pending, PMX_EINVAL,
                "the halos of the last paint of this plan are still staged (pmx_halo_merge / pmx_rowfft_halo first)");
    int rc = pmx_binplan_supported(p_, npart);
    if (rc) return rc;
    PMX_REQUIRE(npart == 0 || (vec_ok(pos) && pos->ncol >= 3), PMX_EINVAL, "pos must be (n, >=3) f4/f8");
    hipStream_t st = (hipStream_t)stream;
    pmx_painter p = *p_;
    BinGeom g;
    g.stale = nullptr;
    g.kind = p.kind;
    g.S = native_support(p.kind);
    const int T[3] = {T0, T1, T2};
    g.ntiles = 1;
    for (int d = 0; d < 3; d++) {
        bool full = p.period[d] > 0 && p.size[d] == p.period[d];
        g.o[d] = full ? 0 : g.S - 1;
        g.nt[d] = (int32_t)((p.size[d] + g.o[d] + T[d] - 1) / T[d]);
        g.ntiles *= g.nt[d];
    }
    {
        // [r6] which tiles of a PCS paint on a fixed-point region are deposited four lanes per particle
        // (tile_deposit_quadz): those of plans with the tile-ordered copy (rows in no order: always) and the tiles that
        // hold more than 9/8 particles per cell — the crowded tiles of a clustered set, where the particles of a wave
        // share cells whatever order they come in.  Measured with both forms forced (scripts/r06/quad_forced.sh,
        // profiles/r06_quad/; paint in ms, one lane / four lanes per particle): config 5's per-GPU load 44.3 / 38.9, a
        // Zel'dovich set at 512^3 2.90 / 2.70, at 256^3 0.545 / 0.424, white-noise steps of 1 cell 2.69 / 2.58, a
        // lattice of 2 particles per cell 4.42 / 4.47 — but the benchmark's jittered lattice at one particle per cell
        // 2.30 / 2.58 (its lanes rarely meet; the quad form's extra vector work shows) and steps of 4 cells 2.74 / 2.78:
        // tiles at the mean density of a uniform set keep the lane-per-particle loop.  (A criterion from the row order
        // the bin pass measures was tried first: a clustered set displaces neighbours together — 4 changes of tile per
        // 64 rows, like a lattice — and does not stand out there.)  PMX_QUAD_MIN: experiments.
        static const int quad_min = [] { const char *e = getenv("PMX_QUAD_MIN"); return e ? atoi(e) : (int)(PMX_QUAD_MIN_DEFAULT); }();
        g.quad_min = quad_min;
        // [r6] ... and which tiles of the one-lane loops deal their entries to the lanes (tile_deposit): from two particles
        // per cell on (PMX_DEAL_MIN: experiments)
        static const int deal_min = [] { const char *e = getenv("PMX_DEAL_MIN"); return e ? atoi(e) : (int)(PMX_DEAL_MIN_DEFAULT); }();
        g.deal_min = deal_min;
    }
    g.chunk = 1 << 30;
    {
        // list entries of a tile that its own workgroup takes: four times the mean population, at
        // least 16384; what a crowded tile holds beyond that is cut into pieces of the same size
        int64_t mean = g.ntiles > 0 ? npart / g.ntiles : 0;
#ifndef PMX_MIN_CHUNK
#define PMX_MIN_CHUNK 16384
#endif
#ifndef PMX_CHUNK_FACTOR
#define PMX_CHUNK_FACTOR 4
#endif
        int64_t ch = PMX_CHUNK_FACTOR * mean > PMX_MIN_CHUNK ? PMX_CHUNK_FACTOR * mean : PMX_MIN_CHUNK;
        g.chunk = (int32_t)(ch < (1 << 30) ? ch : (1 << 30));
    }
    PMX_REQUIRE(g.ntiles < 2147483647ll, PMX_EUNSUPPORTED, "more than 2^31 buckets");
    // The slot ranges of the previous build can be reused when it was for the same geometry
    // and about as many particles (a time-stepping caller: particles move a fraction of a tile per
    // step) and reuse has not just failed (back-off after an overflow).  [r5] "About": within an eighth — on several
    // ranks particles migrate, a rank's count changes a little with every step, and the same count was asked for until
    // now: every step of such a run paid the two-pass build (1.6 instead of 0.75 ms at 512^3; 38 instead of 1 ms for
    // a rank of config 5, whose clustered rows queue on the counters of the crowded tiles).  The ranges carry 25 % +
    // 64 slots of slack per tile; what does not fit raises the overflow flag and is repaired as ever.
    const int64_t dn = npart > pl->npart ? npart - pl->npart : pl->npart - npart;
    bool reuse = pl->built && pl->have_history && dn * 8 <= pl->npart && npart > 0 &&
                 same_geometry(p, pl->painter) && pl->g.ntiles == g.ntiles &&
                 (!pl->sorted || pl->cap_copy >= pl->cap_list * 3 * (size_t)pos->elsize);
    if (pl->host_flag) {
        uint32_t seen = *(volatile uint32_t *)pl->host_flag;   // stale at worst: a hint only
        if (seen != pl->seen_overflows) {
            pl->seen_overflows = seen;
            pl->distrust = pl->distrust ? (pl->distrust < 64 ? 2 * pl->distrust : 64) : 1;
            pl->skip = pl->distrust;
            // [r6] ... and the ranges of the builds that follow carry more slack (slot_capacity).  (The back-off above is a
            // host-side hint: a caller that runs ten steps ahead of the device sees the flag ten steps late and the
            // distrust never builds up; the slack, raised once, stays.)  PMX_SLACK_ADAPT=0: the fixed quarter.
            static const bool adapt = [] { const char *e = getenv("PMX_SLACK_ADAPT"); return !(e && atoi(e) == 0); }();
            if (adapt && pl->slack < 2) pl->slack++;
        } else if (pl->last_reuse && pl->distrust > 0) {
            // the previous single-pass build raised no flag (as far as the host has seen): trust returns
            // step by step, so that one overflow late in a long run does not cost 64 two-pass builds
            pl->distrust /= 2;
        }
    }
    if (reuse && pl->skip > 0) {
        pl->skip--;
        reuse = false;
    }
    pl->g = g;
    pl->painter = p;
    pl->npart = npart;
    pl->built = false;
    const int64_t nbuckets = g.ntiles + 1;            // + the bucket of particles in no tile
    size_t np1 = (size_t)(npart > 0 ? npart : 1);
    // every bucket reserves slot_capacity(count, slack) slots: <= 1.25 count + 64 (1.5 count + 256, 2 count + 1024)
    auto list_entries = [&](size_t np) { return (size_t)slot_capacity((int64_t)np, pl->slack) - (size_t)slot_capacity(0, pl->slack)
                                                + (size_t)slot_capacity(0, pl->slack) * (size_t)nbuckets + 64; };
    size_t nlist = list_entries(np1);
    if (np1 * 4 > pl->cap_part || nlist > pl->cap_list) {
        // (an eighth of room to grow: a count that creeps up does not reallocate — and start over — every step)
        const size_t npa = np1 + np1 / 8;
        const size_t nla = list_entries(npa);
        size_t c1 = 0, c3 = 0;
        if (pl->tid) (void)hipFree(pl->tid);
        if (pl->list) (void)hipFree(pl->list);
        pl->tid = nullptr; pl->list = nullptr; pl->cap_part = 0; pl->cap_list = 0;
        rc = plan_ensure((void **)&pl->tid, &c1, npa * 4); if (rc) return rc;
        rc = plan_ensure((void **)&pl->list, &c3, nla * 4); if (rc) return rc;
        pl->cap_part = npa * 4;
        pl->cap_list = nla;
        reuse = false;
    }
    if (reuse && pl->sorted && (size_t)npart > pl->cap_inv) reuse = false;      // (the inverse list of the tile-ordered copy has a slot per row)
    if ((size_t)(nbuckets + 1) > pl->cap_tiles) {
        size_t c1 = 0, c2 = 0, c3 = 0;
        if (pl->ctl) (void)hipFree(pl->ctl);
        if (pl->offsets) (void)hipFree(pl->offsets);
        if (pl->cursor) (void)hipFree(pl->cursor);
        pl->ctl = nullptr; pl->counts = nullptr; pl->flags = nullptr; pl->nheavy = nullptr;
        pl->offsets = nullptr; pl->cursor = nullptr; pl->cap_tiles = 0;
        rc = plan_ensure((void **)&pl->ctl, &c1, 32 + (size_t)(nbuckets + 1) * 4); if (rc) return rc;
        pl->flags = pl->ctl; pl->nheavy = pl->ctl + 4; pl->counts = pl->ctl + 8;
        rc = plan_ensure((void **)&pl->offsets, &c2, (size_t)(nbuckets + 1) * 8); if (rc) return rc;
        rc = plan_ensure((void **)&pl->cursor, &c3, (size_t)(nbuckets + 1) * 8); if (rc) return rc;
        pl->cap_tiles = (size_t)(nbuckets + 1);
        reuse = false;
    }
    {
        size_t cb = pl->cap_heavy * 8;
        rc = plan_ensure((void **)&pl->heavy_items, &cb, (np1 / (size_t)g.chunk + 16) * 8);
        if (rc) return rc;
        pl->cap_heavy = cb / 8;
    }
    if (!pl->host_flag) {
        PMX_HIP_CHECK(hipHostMalloc((void **)&pl->host_flag, 64, hipHostMallocMapped));
        pl->host_flag[0] = 0;
        pl->host_flag[1] = 0;      // particles a tile kernel found outside the region their list entry names (pmx_binplan_stale)
        pl->host_flag[2] = 0;      // [2..3]: the last finished build's measurement of the row order (heavy_items_kernel)
        pl->host_flag[3] = 0;
    }
    uint32_t *measure_out = nullptr;
    pl->g.stale = pl->host_flag + 1;
    PMX_HIP_CHECK(hipMemsetAsync(pl->ctl, 0, 32 + (size_t)nbuckets * 4, st));      // flags, nheavy, counts: one fill
    DVec dpos = dvec(pos);
    if (npart > 0) {
        const unsigned full_grid = grid_for((npart + 3) / 4, TBLOCK);
        const unsigned small_grid = full_grid < 128 ? full_grid : 128;     // gated launches: cheap to skip (a launch of 512 workgroups that return at once still took 22-25 us per rank and cycle, profiles/r03_h_multirank8_*; the repair they stand for is rare and may be slow)
        // contiguous (n, 3) rows on a 16-byte boundary take the dense staging path
        const bool dense = pos->stride1 == pos->elsize && pos->stride0 == 3 * (int64_t)pos->elsize &&
                           (((uintptr_t)pos->data) & 15) == 0;
#define BC2(K, MODE, GRID, GATE, SP)                                                                              \
    do {                                                                                                        \
        if (dense) bin_count_kernel<K, true, MODE, SP><<<GRID, TBLOCK, 0, st>>>(p, g, dpos, npart, pl->tid, pl->counts, \
                pl->flags, pl->offsets, pl->list, pl->host_flag, GATE, inv, copyp, pl->cursor);                 \
        else bin_count_kernel<K, false, MODE, SP><<<GRID, TBLOCK, 0, st>>>(p, g, dpos, npart, pl->tid, pl->counts,  \
                pl->flags, pl->offsets, pl->list, pl->host_flag, GATE, inv, copyp, pl->cursor);                 \
    } while (0)
#define BC(K, MODE, GRID, GATE)                                                                                 \
    do {                                                                                                        \
        if (inv != nullptr) BC2(K, MODE, GRID, GATE, true);                                                     \
        else BC2(K, MODE, GRID, GATE, false);                                                                   \
    } while (0)
#define BCK(MODE, GRID, GATE)                                                                                   \
    do {                                                                                                        \
        switch (p.kind) {                                                                                       \
        case PMX_TUNED_NNB: BC(PMX_TUNED_NNB, MODE, GRID, GATE); break;                                         \
        case PMX_TUNED_CIC: BC(PMX_TUNED_CIC, MODE, GRID, GATE); break;                                         \
        case PMX_TUNED_TSC: BC(PMX_TUNED_TSC, MODE, GRID, GATE); break;                                         \
        default: BC(PMX_TUNED_PCS, MODE, GRID, GATE); break;                                                    \
        }                                                                                                       \
    } while (0)
        const uint32_t *nogate = nullptr;
        uint32_t *inv = nullptr;
        void *copyp = nullptr;               // != NULL: the single-pass rebuild writes the tile-ordered copy itself
        const uint32_t *copy_gate = nullptr; // the gather of the copy then only runs after an overflow repair
        if (!pl->host_groups) {
            PMX_HIP_CHECK(hipHostMalloc((void **)&pl->host_groups, 64, hipHostMallocDefault));
            pl->host_groups[0] = 0;                 // breaks of the tile sequence among ...
            pl->host_groups[1] = 0;                 // ... this many sampled rows of the last build
            pl->host_groups[2] = 0;                 // particles of that build
        }
        if (pl->have_measure) {                     // what the last finished build left in the mapped slot
            pl->host_groups[0] = *(volatile uint32_t *)(pl->host_flag + 2);
            pl->host_groups[1] = *(volatile uint32_t *)(pl->host_flag + 3);
        }
        // breaks of the tile sequence per 64 consecutive rows above which the row order counts as
        // incoherent (lattice order: a handful; random order: 63).  [r5] ~60, not 24: measured at 512^3 with the plan's
        // form forced both ways (scripts/r05/order_threshold.sh; cycle in ms without / with the tile-ordered copy): a
        // lattice with N(0, 1 / 2 / 4 / 8) cells of jitter per row (~15 / 28 / 47 / 59 breaks) 5.44 / 5.83 / 7.01 /
        // 9.50 against 7.06 / 7.67 / 10.58 / 11.04; only rows in no order at all (63) gain: 22.8 against 14.4.
        // Two thresholds: a plan takes the copy above 61.5 and gives it up below 58 (position sets on either side of ONE
        // threshold made the plan start over every step: 11.1 ms at 8 cells of jitter against 9.5 in either form).
        auto incoherent = [&](double breaks, double rows, bool has_copy) { return breaks * 64.0 > (has_copy ? PMX_SORTED_DROP_BREAKS : PMX_SORTED_TAKE_BREAKS) * rows; };
        if (reuse && pl->sort_pref < 0 && pl->host_groups[2] == (uint32_t)npart && pl->host_groups[1] > 4096 &&
            incoherent(pl->host_groups[0], (double)pl->host_groups[1], pl->sorted) != pl->sorted)
            reuse = false;       // the order of the rows changed its character since the plan was built: start over
        if (!reuse) pl->sorted = false;
#ifndef PMX_LEAN_BIN
#define PMX_LEAN_BIN 1
#endif
#ifndef PMX_REPAIR_GRID
#define PMX_REPAIR_GRID 512      // workgroups of the gated repair, resident at once (measured 128 / 256 / 512: the launch that returns at once 4.7 us each; the repair of 512^3 rows 5.3 / 3.4 / 2.6 ms)
#endif
        // the block form of the pass (bin_lean_kernel for dense rows, else bin_block_kernel): into the ranges `offsets`
        // names — or, list_arg == NULL (lean form only), the counts alone
        auto block_pass = [&](uint32_t *list_arg, const uint32_t *repair_gate = nullptr) {
            const int64_t nblocks = (npart + BLOCK_ROWS - 1) / BLOCK_ROWS;
            const unsigned bgrid = (unsigned)(nblocks < 65535 * 8 ? nblocks : 65535 * 8);
            bool whole_b = true;
            for (int d = 0; d < 3; d++) whole_b = whole_b && g.o[d] == 0 && (int)p.period[d] == (int)p.size[d];      // (whole_mesh())
#ifdef PMX_GENERAL_FORMS_ONLY
            whole_b = false;       // (a build switch for measurements, right results: what the forms for blocks of any shape cost on a whole mesh)
#endif
#define BL(K, PE_, WH) do { if (repair_gate) bin_repair_lean_kernel<K, PE_, WH><<<(bgrid < PMX_REPAIR_GRID ? bgrid : PMX_REPAIR_GRID), TBLOCK, 0, st>>>(p, g, dpos, npart, pl->counts, pl->flags, pl->offsets, pl->cursor, list_arg, pl->host_flag, repair_gate); \
                             else bin_lean_kernel<K, PE_, WH><<<bgrid, TBLOCK, 0, st>>>(p, g, dpos, npart, pl->counts, pl->flags, pl->offsets, list_arg, pl->host_flag); } while (0)
#define BB(K)                                                                                                   \
    do {                                                                                                        \
        if (dense && PMX_LEAN_BIN) {                                                                            \
            if (whole_b) { if (dpos.elsize == 8) BL(K, 8, true); else BL(K, 4, true); }                         \
            else { if (dpos.elsize == 8) BL(K, 8, false); else BL(K, 4, false); }                               \
        } else if (dense) bin_block_kernel<K, true><<<bgrid, TBLOCK, 0, st>>>(p, g, dpos, npart, pl->counts, pl->flags, pl->offsets, list_arg, pl->host_flag); \
        else bin_block_kernel<K, false><<<bgrid, TBLOCK, 0, st>>>(p, g, dpos, npart, pl->counts, pl->flags, pl->offsets, list_arg, pl->host_flag); \
    } while (0)
            switch (p.kind) {
            case PMX_TUNED_NNB: BB(PMX_TUNED_NNB); break;
            case PMX_TUNED_CIC: BB(PMX_TUNED_CIC); break;
            case PMX_TUNED_TSC: BB(PMX_TUNED_TSC); break;
            default: BB(PMX_TUNED_PCS); break;
            }
#undef BB
#undef BL
        };
        pl->last_reuse = reuse;
        pl->builds[reuse ? 0 : 1]++;
        if (reuse) {
            // single pass into the previous slot ranges; if a tile overflowed (flags[0]) the
            // exact two-pass build below runs, otherwise its kernels return at once.  Whether the
            // plan carries the tile-ordered copy was decided by its first build.
            inv = pl->sorted ? pl->inv : nullptr;
            copyp = pl->sorted ? pl->pos_copy : nullptr;
            if (pl->sorted) pl->copy_elsize = pos->elsize;      // the single pass rewrites the copy
            copy_gate = pl->sorted ? pl->flags : nullptr;
            // (measured, block against chunk form of the single pass, same box: 512^3 f8 1.01 vs 1.17 ms, 768^3 3.22 vs
            // 3.76, clustered 0.96 vs 1.27, 12-byte rows 0.90 vs 0.96, config 3 0.86 vs 1.08, 256^3 0.17 vs 0.20)
            if (inv == nullptr && pl->form != 2) {
                // rows in a coherent order, no tile-ordered copy: one request per tile and block of rows
                block_pass(pl->list);
            } else
                BCK(1, grid_for((npart + PMX_ONEPASS_U - 1) / PMX_ONEPASS_U, TBLOCK), nogate);
            // the repair, one launch that returns at once unless a tile overflowed (measured: the four gated launches
            // it replaces, zero / count / scan / scatter, cost a slab rank 20 us per build)
            const uint32_t *gate = pl->flags;
            // ([r6] two launches, both returning at once unless a tile overflowed: the scan of the exact counts into
            // new ranges as a workgroup of its own — the stream orders it before the fill; no workgroup waits for another)
            const bool lean_repair = inv == nullptr && pl->form != 2 && dense && PMX_LEAN_BIN && PMX_REPAIR_GRID > 0;
            bin_scan_kernel<<<1, 1024, 0, st>>>(pl->counts, g.ntiles + 1, pl->offsets, pl->cursor, gate, lean_repair ? pl->counts : nullptr, pl->slack);
            if (lean_repair) block_pass(pl->list, gate);
            else BCK(3, small_grid, gate);
        } else {
            // [r5] Dense rows count through the block form too (list == NULL: counts only) and fill their ranges with
            // the same pass a rebuild uses, instead of one device atomic per wave and tile in both passes: nothing
            // for uniform rows (1.6 ms either way at 512^3), but on clustered rows the waves queue on the counters of
            // the crowded tiles — 18.6 + 19.6 ms for the 2.7e8 rows of a config-5 rank against 1.1 ms per block pass.
            // (Rows in no order, which get the tile-ordered copy, need a tile id per row and the inverse list: they
            // are counted again by the per-wave kernel below.)
#ifndef PMX_LEAN_TWOPASS
#define PMX_LEAN_TWOPASS 1
#endif
            const bool lean2 = PMX_LEAN_TWOPASS && dense && PMX_LEAN_BIN && pl->form != 2 && pl->sort_pref != 1;
            if (lean2) block_pass(nullptr);
            else BCK(0, full_grid, nogate);
            // How coherent is the row order?  Every build leaves its measurement in host_groups
            // (asynchronous copy, below); a two-pass build of about as many rows as that one (within an
            // eighth: ghost batches change their size from step to step, overflow repairs, reallocations)
            // decides from it without waiting.  Only a plan object that has never measured rows like these
            // synchronises: once in its life for a time-stepping caller.
            bool want = pl->sort_pref == 1;
            if (pl->sort_pref < 0 && npart >= (1 << 16)) {
                const double was = (double)pl->host_groups[2];
                const bool known = pl->have_measure && pl->host_groups[1] > 4096 &&
                                   fabs((double)npart - was) * 8.0 <= (double)npart;
                if (!known) {
                    PMX_HIP_CHECK(hipMemcpyAsync(pl->host_groups, pl->flags + 1, 8, hipMemcpyDeviceToHost, st));
                    PMX_HIP_CHECK(hipStreamSynchronize(st));
                    pl->host_groups[2] = (uint32_t)npart;
                    pl->have_measure = true;
                }
                want = pl->host_groups[1] > 0 && incoherent(pl->host_groups[0], (double)pl->host_groups[1], false);
            }
            if (want) {
                const size_t es = (size_t)pos->elsize;
                rc = plan_ensure(&pl->pos_copy, &pl->cap_copy, pl->cap_list * 3 * es); if (rc) return rc;
                size_t ci = pl->cap_inv * 4;
                rc = plan_ensure((void **)&pl->inv, &ci, (np1 + np1 / 8) * 4); if (rc) return rc;      // (room to grow, as the list has)
                pl->cap_inv = ci / 4;
                pl->sorted = true;
                pl->copy_elsize = (int)es;
                inv = pl->inv;
            }
            if (lean2 && !want) {
                bin_scan_kernel<<<1, 1024, 0, st>>>(pl->counts, nbuckets, pl->offsets, pl->cursor, nogate, pl->counts, pl->slack);
                block_pass(pl->list);
            } else {
                if (lean2) {
                    PMX_HIP_CHECK(hipMemsetAsync(pl->counts, 0, (size_t)nbuckets * 4, st));
                    BCK(0, full_grid, nogate);
                }
                bin_scan_kernel<<<1, 1024, 0, st>>>(pl->counts, nbuckets, pl->offsets, pl->cursor, nogate, nullptr, pl->slack);
                bin_scatter_kernel<<<full_grid, TBLOCK, 0, st>>>(pl->tid, pl->cursor, npart, pl->list, nogate, inv);
            }
        }
#undef BCK
#undef BC
#undef BC2
        if (pl->sort_pref < 0 && npart >= (1 << 16)) {
            // what this build saw of the row order, read by the NEXT build (stale at worst: a hint); the build's last
            // kernel stores it into the mapped slot
            measure_out = pl->host_flag + 2;
            pl->host_groups[2] = (uint32_t)npart;
            pl->have_measure = true;
        }
        if (pl->sorted) {
            const unsigned cgrid = (unsigned)(nbuckets < 65535 * 8 ? nbuckets : 65535 * 8);
            if (pos->elsize == 8)
                sort_copy_kernel<6><<<cgrid, TBLOCK, 0, st>>>(pl->list, pl->offsets, pl->counts, nbuckets, dpos, (uint32_t *)pl->pos_copy, copy_gate);
            else
                sort_copy_kernel<3><<<cgrid, TBLOCK, 0, st>>>(pl->list, pl->offsets, pl->counts, nbuckets, dpos, (uint32_t *)pl->pos_copy, copy_gate);
        }
    } else {
        bin_scan_kernel<<<1, 1024, 0, st>>>(pl->counts, nbuckets, pl->offsets, pl->cursor, nullptr, nullptr, pl->slack);
    }
    if (npart > 0)
        heavy_items_kernel<<<grid_for(g.ntiles, TBLOCK, 1024), TBLOCK, 0, st>>>(pl->counts, g.ntiles, g.chunk, pl->heavy_items,
                                                                               pl->nheavy, (uint32_t)pl->cap_heavy, pl->flags, measure_out);
    PMX_HIP_CHECK(hipGetLastError());
    pl->built = true;
    pl->have_history = npart > 0;
    return PMX_OK;
}

#endif   // PMX_PART_PLAN

#if PMX_PART_PAINT
template <typename T>
#if PMX_BINNED_PART == 0
static
#endif
int paint_binned_t(pmx_binplan *pl, const pmx_painter &p, void *canvas, DVec pos, DVec mass, double ms,
                   int overwrite, int defer, hipStream_t st)
{
    const BinGeom &g = pl->g;
    PMX_REQUIRE(!pl->halo_pending, PMX_EINVAL,
                "the halos of the previous paint of this plan are still staged (pmx_halo_merge / pmx_rowfft_halo first)");
    size_t need = (size_t)g.ntiles * (size_t)halo_cells(g.S) * sizeof(T);
    int rc = plan_ensure(&pl->halo, &pl->cap_halo, need > 0 ? need : 16);
    if (rc) return rc;
    unsigned grid = (unsigned)(g.ntiles < 65535 * 8 ? g.ntiles : 65535 * 8);
    const int ntw = walk_x(g.S) ? g.nt[0] : g.nt[2];
    const int64_t nwork = (g.ntiles / ntw) * ((ntw + ZSEG - 1) / ZSEG);   // segments of tiles along the walk axis
    unsigned pgrid = (unsigned)(nwork < 65535 * 8 ? nwork : 65535 * 8);
    T *halo = (T *)pl->halo;
    const int sorted = pl->sorted ? 1 : 0;
    if (sorted) {
        // stream the plan's copy of the positions (dense rows of 3 elements, list order)
        PMX_REQUIRE(pos.elsize == pl->copy_elsize, PMX_EINVAL, "positions changed their element type since pmx_binplan_build");
        const int es = pos.elsize;
        pos.data = (const char *)pl->pos_copy;
        pos.stride0 = 3 * es;
        pos.stride1 = es;
    }
    // Fixed-point regions for the windows whose deposit is bound by the LDS atomics (S >= 3); NNB / CIC are
    // bound by memory and keep their doubles.  Per-particle masses: their largest finite magnitude, and whether
    // all of them are finite, are found on the device; a batch with a NaN / Inf mass is served by the
    // floating-point kernels, launched behind the fixed-point ones (each returns at once when the batch is not
    // its own).  A scalar mass decides on the host.
    const bool det = pl->deterministic != 0;
    const bool fixed_kind = det || (PMX_FIXED_POINT && g.S >= PMX_FIXED_MIN_S);
    const double *mstats = nullptr;
    bool run_fixed = fixed_kind, run_float = !fixed_kind;
    if (fixed_kind) {
        if (mass.data && pl->mass_stats_ext) {
            // the caller has run pmx_mass_stats on these very masses (once per mass array and version: a
            // time-stepping caller's masses do not change): no pass over them in front of this paint
            mstats = pl->mass_stats_ext;
            pl->mass_stats_ext = nullptr;
            run_float = true;
        } else if (mass.data) {
            if (!pl->mstats) PMX_HIP_CHECK(hipMalloc((void **)&pl->mstats, 32));
            PMX_HIP_CHECK(hipMemsetAsync(pl->mstats, 0, 32, st));
            mass_stats_kernel<<<grid_for(pl->npart, TBLOCK, 2048), TBLOCK, 0, st>>>(mass, pl->npart, (unsigned long long *)pl->mstats);
            mass_stats_finish_kernel<<<1, 1, 0, st>>>((unsigned long long *)pl->mstats);
            mstats = pl->mstats;
            run_float = true;
        } else if (!(fabs(ms) <= 1.7e308)) {
            run_fixed = false;
            run_float = true;
        }
    }
    // deterministic: the fixed-point kernels write a dense int64 copy of the block (pd: its layout) with the
    // batch's one scale; det_finish_kernel converts into the caller's canvas
    pmx_painter pd = p;
    int32_t *dexp = nullptr;
    double *dhalo = nullptr;
    if (det && run_fixed) {
        const size_t cells = (size_t)p.size[0] * (size_t)p.size[1] * (size_t)p.size[2];
        rc = plan_ensure(&pl->dscratch, &pl->cap_dscratch, cells * 8 + 64); if (rc) return rc;
        rc = plan_ensure(&pl->dhalo, &pl->cap_dhalo, (size_t)g.ntiles * (size_t)halo_cells(g.S) * 8 + 16); if (rc) return rc;
        dexp = (int32_t *)((char *)pl->dscratch + cells * 8);
        dhalo = (double *)pl->dhalo;
        pd.canvas_elsize = 8;
        pd.strides[2] = 8; pd.strides[1] = 8 * p.size[2]; pd.strides[0] = 8 * p.size[2] * p.size[1];
        det_scale_kernel<<<1, 1024, 0, st>>>(p, pl->counts, g.ntiles, mstats, ms, dexp);
    }
#define PT3L(K, TT, MD, ODD, PP, CV, HL, OW, SD, WH, PE_) paint_tile_kernel<K, TT, TileThreads<K, TT>::paint, SD, MD, WH, PE_><<<pgrid, TileThreads<K, TT>::paint, 0, st>>>(PP, g, (char *)(CV), pos, mass, ms, pl->list, pl->offsets, pl->counts, HL, OW, mstats, ODD, dexp, p)
#define PT3P(K, TT, MD, ODD, PP, CV, HL, OW, SD, WH) do { if (pos.elsize == 8) PT3L(K, TT, MD, ODD, PP, CV, HL, OW, SD, WH, 8); else PT3L(K, TT, MD, ODD, PP, CV, HL, OW, SD, WH, 4); } while (0)
#define PT3W(K, TT, MD, ODD, PP, CV, HL, OW, SD) do { if (whole32) PT3P(K, TT, MD, ODD, PP, CV, HL, OW, SD, true); else PT3P(K, TT, MD, ODD, PP, CV, HL, OW, SD, false); } while (0)
#define PT3(K, TT, MD, ODD, PP, CV, HL, OW) do { if (sorted) PT3W(K, TT, MD, ODD, PP, CV, HL, OW, true); else PT3W(K, TT, MD, ODD, PP, CV, HL, OW, false); } while (0)
    // [r5] float canvases, S >= 3: the 32-bit region (paint_tile32_kernel); contributions of either sign need its SIGNED guard
    const bool signed32 = mass.data != nullptr || ms < 0 || p.order[0] != 0 || p.order[1] != 0 || p.order[2] != 0;
    const bool dense32 = pos.stride1 == pos.elsize && pos.stride0 == 3 * pos.elsize;      // (always so for the plan's sorted copy)
    bool whole32 = true;
    for (int d = 0; d < 3; d++) whole32 = whole32 && g.o[d] == 0 && (int)p.period[d] == (int)p.size[d];      // (whole_mesh())
#ifdef PMX_GENERAL_FORMS_ONLY
    whole32 = false;       // (a build switch for measurements, right results: what the forms for blocks of any shape cost on a whole mesh)
#endif
#define PT32L(K, SD, SG, WH, PE_) paint_tile32_kernel<K, Tile32<K>::threads, SD, SG, WH, PE_><<<pgrid, Tile32<K>::threads, 0, st>>>(p, g, (char *)canvas, pos, mass, ms, pl->list, pl->offsets, pl->counts, (float *)halo, overwrite, mstats, 0)
#define PT32P(K, SD, SG, WH) do { if (!dense32) { if constexpr (!SD) PT32L(K, false, SG, WH, 0); } else if (pos.elsize == 8) PT32L(K, SD, SG, WH, 8); else PT32L(K, SD, SG, WH, 4); } while (0)
#define PT32W(K, SD, SG) do { if (whole32) PT32P(K, SD, SG, true); else PT32P(K, SD, SG, false); } while (0)
#define PT32(K, SG) do { if (sorted) PT32W(K, true, SG); else PT32W(K, false, SG); } while (0)
#define PT(K) do { if (run_fixed && det) PT3(K, double, 2, 0, pd, pl->dscratch, dhalo, 1); \
                   else if (run_fixed) { if constexpr (PMX_REGION32 && std::is_same<T, float>::value && Tuned<K>::S >= 3) { if (signed32) PT32(K, true); else PT32(K, false); } \
                                         else PT3(K, T, 1, 0, p, canvas, halo, overwrite); } \
                   if (run_float) PT3(K, T, 0, (run_fixed ? 1 : 0), p, canvas, halo, overwrite); } while (0)
    // (a batch is served either by the fixed-point or by the floating-point kernels: the merge of the other
    // finds only zeros in its staging buffer... the deterministic one has a staging buffer of its own)
#define HM(S_) do { if (defer) break; if (run_fixed && det) { halo_merge_kernel<S_, double, true><<<grid, TBLOCK, 0, st>>>(pd, g, (char *)pl->dscratch, dhalo, pl->counts, 1, mstats, 0); \
                                            if (run_float) halo_merge_kernel<S_, T><<<grid, TBLOCK, 0, st>>>(p, g, (char *)canvas, halo, pl->counts, overwrite, mstats, 1); } \
                    else halo_merge_kernel<S_, T><<<grid, TBLOCK, 0, st>>>(p, g, (char *)canvas, halo, pl->counts, overwrite); } while (0)
    switch (p.kind) {
    case PMX_TUNED_NNB: PT(PMX_TUNED_NNB); break;
    case PMX_TUNED_CIC: PT(PMX_TUNED_CIC); HM(2); break;
    case PMX_TUNED_TSC: PT(PMX_TUNED_TSC); HM(3); break;
    default: PT(PMX_TUNED_PCS); HM(4); break;
    }
#undef PT
#undef PT3
#undef PT3W
#undef PT3P
#undef PT3L
#undef PT32
#undef PT32W
#undef PT32P
#undef PT32L
#undef HM
    // the pieces of crowded tiles (none for a uniform batch: the kernel then returns at once)
    const unsigned hgrid = (unsigned)(pl->cap_heavy < 1024 ? pl->cap_heavy : 1024);
#define PH3(K, TT, MD, ODD, PP, CV) do { if (sorted) paint_heavy_kernel<K, TT, TileThreads<K, TT>::paint, true, MD><<<hgrid, TileThreads<K, TT>::paint, 0, st>>>(PP, g, (char *)(CV), pos, mass, ms, pl->list, pl->offsets, pl->counts, pl->heavy_items, pl->nheavy, (uint32_t)pl->cap_heavy, mstats, ODD, dexp, p); \
                   else paint_heavy_kernel<K, TT, TileThreads<K, TT>::paint, false, MD><<<hgrid, TileThreads<K, TT>::paint, 0, st>>>(PP, g, (char *)(CV), pos, mass, ms, pl->list, pl->offsets, pl->counts, pl->heavy_items, pl->nheavy, (uint32_t)pl->cap_heavy, mstats, ODD, dexp, p); } while (0)
#define PH(K) do { if (run_fixed && det) PH3(K, double, 2, 0, pd, pl->dscratch); \
                   else if (run_fixed) PH3(K, T, 1, 0, p, canvas); \
                   if (run_float) PH3(K, T, 0, (run_fixed ? 1 : 0), p, canvas); } while (0)
    switch (p.kind) {
    case PMX_TUNED_NNB: PH(PMX_TUNED_NNB); break;
    case PMX_TUNED_CIC: PH(PMX_TUNED_CIC); break;
    case PMX_TUNED_TSC: PH(PMX_TUNED_TSC); break;
    default: PH(PMX_TUNED_PCS); break;
    }
#undef PH
#undef PH3
    if (run_fixed && det) {
        const int64_t cells = p.size[0] * p.size[1] * p.size[2];
        det_finish_kernel<T><<<grid_for(cells, TBLOCK, 8192), TBLOCK, 0, st>>>(p, (char *)canvas, (const long long *)pl->dscratch, dexp, overwrite, mstats);
    }
    PMX_HIP_CHECK(hipGetLastError());
    if (defer) {
        pl->halo_pending = 1;
        pl->halo_elsize = (int)sizeof(T);
        pl->halo_canvas = canvas;
    }
    return PMX_OK;
}

#if PMX_BINNED_PART == 4
template int paint_binned_t<float>(pmx_binplan *, const pmx_painter &, void *, DVec, DVec, double, int, int, hipStream_t);
#else
#if PMX_BINNED_PART == 2
extern template int paint_binned_t<float>(pmx_binplan *, const pmx_painter &, void *, DVec, DVec, double, int, int, hipStream_t);
#endif
extern "C" int pmx_mass_stats(const pmx_vec *mass, int64_t n, double *stats, void *stream)
{
    PMX_REQUIRE(stats != nullptr, PMX_EINVAL, "stats is NULL");
    PMX_REQUIRE(n == 0 || vec_ok(mass), PMX_EINVAL, "mass must be (n,) f4/f8");
    hipStream_t st = (hipStream_t)stream;
    PMX_HIP_CHECK(hipMemsetAsync(stats, 0, 32, st));
    if (n > 0) mass_stats_kernel<<<grid_for(n, TBLOCK, 2048), TBLOCK, 0, st>>>(dvec(mass), n, (unsigned long long *)stats);
    mass_stats_finish_kernel<<<1, 1, 0, st>>>((unsigned long long *)stats);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

// can the halo merge of a paint be left to the forward row pass of r2c (pmx_rowfft_halo)?  Whole periodic axes 1 and 2
// (every tile a full one, halos wrap onto tiles); along axis 0 either the same (one rank's mesh) or [r5] a block of
// planes of a larger period (a slab rank: tile space starts S - 1 planes below the block, nothing wraps, what a region
// holds outside the block is dropped by whoever merges); every cell written by the paint (overwrite), dense rows of
// a power-of-two length the row kernel gathers for, floating-point staging.
static bool halo_deferrable(const pmx_binplan *pl, const pmx_painter &p, int overwrite)
{
    const BinGeom &g = pl->g;
    if (!overwrite || pl->deterministic || g.S < 2) return false;
    for (int d = 1; d < 3; d++)
        if (g.o[d] != 0 || p.period[d] != p.size[d]) return false;
    if (!((g.o[0] == 0 && p.period[0] == p.size[0]) || g.o[0] == g.S - 1)) return false;
    if (p.strides[2] != p.canvas_elsize) return false;
    return pmx_rowfft_halo_supported(p.size[2], p.canvas_elsize) == PMX_OK;
}

static int paint_binned_any(pmx_binplan *pl, const pmx_painter *p_, void *canvas, const pmx_vec *pos,
                            const pmx_vec *mass, double mass_scalar, int32_t overwrite, int32_t defer, int32_t *deferred,
                            void *stream)
{
    PMX_REQUIRE(pl && pl->built, PMX_EINVAL, "bin plan is not built");
    PMX_REQUIRE(p_ && same_geometry(*p_, pl->painter), PMX_EINVAL, "painter differs from the one the plan was built for");
    PMX_REQUIRE(canvas != nullptr, PMX_EINVAL, "canvas is NULL");
    PMX_REQUIRE(p_->canvas_elsize == 4 || p_->canvas_elsize == 8, PMX_EINVAL, "canvas must be float or double");
    PMX_REQUIRE(pl->npart == 0 || vec_ok(pos), PMX_EINVAL, "pos");
    pmx_painter p = *p_;
    hipStream_t st = (hipStream_t)stream;
    const int df = (defer && halo_deferrable(pl, p, overwrite)) ? 1 : 0;
    if (deferred) *deferred = df;
    if (p.canvas_elsize == 8) return paint_binned_t<double>(pl, p, canvas, dvec(pos), dvec(mass), mass_scalar, overwrite, df, st);
    return paint_binned_t<float>(pl, p, canvas, dvec(pos), dvec(mass), mass_scalar, overwrite, df, st);
}

extern "C" int pmx_paint_binned(pmx_binplan *pl, const pmx_painter *p_, void *canvas, const pmx_vec *pos,
                                const pmx_vec *mass, double mass_scalar, int32_t overwrite, void *stream)
{
    return paint_binned_any(pl, p_, canvas, pos, mass, mass_scalar, overwrite, 0, nullptr, stream);
}

extern "C" int pmx_paint_binned_defer(pmx_binplan *pl, const pmx_painter *p_, void *canvas, const pmx_vec *pos,
                                      const pmx_vec *mass, double mass_scalar, int32_t overwrite, int32_t *deferred,
                                      void *stream)
{
    PMX_REQUIRE(deferred != nullptr, PMX_EINVAL, "deferred is NULL");
    return paint_binned_any(pl, p_, canvas, pos, mass, mass_scalar, overwrite, 1, deferred, stream);
}

extern "C" int pmx_halo_merge(pmx_binplan *pl, const pmx_painter *p_, void *canvas, void *stream)
{
    PMX_REQUIRE(pl && pl->built, PMX_EINVAL, "bin plan is not built");
    if (!pl->halo_pending) return PMX_OK;
    PMX_REQUIRE(p_ && same_geometry(*p_, pl->painter), PMX_EINVAL, "painter differs from the one the plan was built for");
    PMX_REQUIRE(canvas == pl->halo_canvas, PMX_EINVAL, "the staged halos belong to another canvas");
    PMX_REQUIRE(p_->canvas_elsize == pl->halo_elsize, PMX_EINVAL, "canvas element size differs from the staged halos'");
    const pmx_painter p = *p_;
    const BinGeom &g = pl->g;
    hipStream_t st = (hipStream_t)stream;
    unsigned grid = (unsigned)(g.ntiles < 65535 * 8 ? g.ntiles : 65535 * 8);
#define HMD(S_) do { if (p.canvas_elsize == 8) halo_merge_kernel<S_, double><<<grid, TBLOCK, 0, st>>>(p, g, (char *)canvas, (const double *)pl->halo, pl->counts, 1); \
                     else halo_merge_kernel<S_, float><<<grid, TBLOCK, 0, st>>>(p, g, (char *)canvas, (const float *)pl->halo, pl->counts, 1); } while (0)
    switch (g.S) {
    case 2: HMD(2); break;
    case 3: HMD(3); break;
    case 4: HMD(4); break;
    default: break;
    }
#undef HMD
    pl->halo_pending = 0;
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

// (for pmx_rowfft_halo, csrc/pmx_colfft.hip) the staged halos of the last paint and what the gather has to know
extern "C" int pmx_binplan_halo_source(pmx_binplan *pl, const void *canvas, int32_t elsize, const void **halo,
                                       int32_t *S, int32_t *nt, int32_t consume)
{
    PMX_REQUIRE(pl && pl->built, PMX_EINVAL, "bin plan is not built");
    PMX_REQUIRE(pl->halo_pending, PMX_EINVAL, "no staged halos on this plan");
    PMX_REQUIRE(canvas == pl->halo_canvas, PMX_EINVAL, "the staged halos belong to another canvas");
    PMX_REQUIRE(elsize == pl->halo_elsize, PMX_EINVAL, "element size differs from the staged halos'");
    *halo = pl->halo;
    *S = pl->g.S;
    for (int d = 0; d < 3; d++) nt[d] = pl->g.nt[d];
    nt[3] = pl->g.o[0];          // plane 0 of the block in tile space (0: the whole mesh of one rank)
    if (consume) pl->halo_pending = 0;
    return PMX_OK;
}
#endif   // PMX_BINNED_PART != 4

#endif   // PMX_PART_PAINT

#if PMX_PART_READOUT
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
    const int sorted = pl->sorted ? 1 : 0;
    const DVec caller_out = dout;
    if (sorted) {
        size_t cb = pl->cap_out * 8;
        int rc = plan_ensure((void **)&pl->out_sorted, &cb, pl->cap_list * 8);
        if (rc) return rc;
        pl->cap_out = cb / 8;
        PMX_REQUIRE(dpos.elsize == pl->copy_elsize, PMX_EINVAL, "positions changed their element type since pmx_binplan_build");
        const int es = dpos.elsize;
        dpos.data = (const char *)pl->pos_copy;
        dpos.stride0 = 3 * es;
        dpos.stride1 = es;
        dout.data = (const char *)pl->out_sorted;
        dout.stride0 = 8; dout.stride1 = 0; dout.elsize = 8;
    }
    unsigned grid = (unsigned)(g.ntiles < 65535 * 8 ? g.ntiles : 65535 * 8);
    const bool relax = pl->exact == 0;
    // [r5] the common case — relaxed arithmetic, the index list, dense position rows, a dense result vector — has a loop of its
    // own, which also zeroes the dropped particles and takes the pieces of crowded tiles (one launch instead of three)
#ifndef PMX_LEAN_READOUT
#define PMX_LEAN_READOUT 1
#endif
    const bool lean = PMX_LEAN_READOUT && relax && !sorted && dpos.stride1 == dpos.elsize && dpos.stride0 == 3 * dpos.elsize
                      && dout.stride0 >= dout.elsize && dout.stride0 % dout.elsize == 0 && dout.stride0 < (1 << 20);      // (a dense vector, or a column of the caller's array: F[:, d])
    if (!lean) zero_dropped_kernel<<<256, TBLOCK, 0, st>>>(pl->list, pl->offsets, pl->counts, g.ntiles, dout, sorted);
    // (NNB: one cell, weight 1 — the same bits either way; the lean per-particle setup of the relaxed form is what it takes)
#define RT2(K, T, RX) do { if (sorted) readout_tile_kernel<K, T, TileThreads<K, T>::readout, true, RX><<<grid, TileThreads<K, T>::readout, 0, st>>>(p, g, (const char *)canvas, dpos, dout, pl->list, pl->offsets, pl->counts); \
                      else readout_tile_kernel<K, T, TileThreads<K, T>::readout, false, RX><<<grid, TileThreads<K, T>::readout, 0, st>>>(p, g, (const char *)canvas, dpos, dout, pl->list, pl->offsets, pl->counts); } while (0)
#define RT(K, T) do { if (relax) RT2(K, T, true); else RT2(K, T, false); } while (0)
    bool whole_r = true;
    for (int d = 0; d < 3; d++) whole_r = whole_r && g.o[d] == 0 && (int)p.period[d] == (int)p.size[d];      // (whole_mesh())
#ifdef PMX_GENERAL_FORMS_ONLY
    whole_r = false;       // (a build switch for measurements, right results: what the forms for blocks of any shape cost on a whole mesh)
#endif
#define RLL(K, T, PE_, OE_, WH) readout_tile_lean_kernel<K, T, TileThreads<K, T>::readout, PE_, OE_, WH><<<grid, TileThreads<K, T>::readout, 0, st>>>(p, g, (const char *)canvas, dpos, (char *)const_cast<char *>(dout.data), pl->list, pl->offsets, pl->counts, pl->heavy_items, pl->nheavy, (uint32_t)pl->cap_heavy, (int)dout.stride0)
#define RLW(K, T, PE_, OE_) do { if (whole_r) RLL(K, T, PE_, OE_, true); else RLL(K, T, PE_, OE_, false); } while (0)
#define RLO(K, T, PE_) do { if (dout.elsize == 8) RLW(K, T, PE_, 8); else RLW(K, T, PE_, 4); } while (0)
#define RL(K, T) do { if (dpos.elsize == 8) RLO(K, T, 8); else RLO(K, T, 4); } while (0)
    if (lean) {
        if (p.canvas_elsize == 8) {
            switch (p.kind) {
            case PMX_TUNED_NNB: RL(PMX_TUNED_NNB, double); break;
            case PMX_TUNED_CIC: RL(PMX_TUNED_CIC, double); break;
            case PMX_TUNED_TSC: RL(PMX_TUNED_TSC, double); break;
            default: RL(PMX_TUNED_PCS, double); break;
            }
        } else {
            switch (p.kind) {
            case PMX_TUNED_NNB: RL(PMX_TUNED_NNB, float); break;
            case PMX_TUNED_CIC: RL(PMX_TUNED_CIC, float); break;
            case PMX_TUNED_TSC: RL(PMX_TUNED_TSC, float); break;
            default: RL(PMX_TUNED_PCS, float); break;
            }
        }
    } else
#undef RL
#undef RLO
#undef RLW
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
#undef RT2
    if (!lean) {
        const unsigned hgrid = (unsigned)(pl->cap_heavy < 1024 ? pl->cap_heavy : 1024);
#define RH2(K, T, RX) do { if (sorted) readout_heavy_kernel<K, T, TileThreads<K, T>::readout, true, RX><<<hgrid, TileThreads<K, T>::readout, 0, st>>>(p, g, (const char *)canvas, dpos, dout, pl->list, pl->offsets, pl->counts, pl->heavy_items, pl->nheavy, (uint32_t)pl->cap_heavy); \
                      else readout_heavy_kernel<K, T, TileThreads<K, T>::readout, false, RX><<<hgrid, TileThreads<K, T>::readout, 0, st>>>(p, g, (const char *)canvas, dpos, dout, pl->list, pl->offsets, pl->counts, pl->heavy_items, pl->nheavy, (uint32_t)pl->cap_heavy); } while (0)
#define RH(K, T) do { if (relax) RH2(K, T, true); else RH2(K, T, false); } while (0)
        if (p.canvas_elsize == 8) {
            switch (p.kind) {
            case PMX_TUNED_NNB: RH(PMX_TUNED_NNB, double); break;
            case PMX_TUNED_CIC: RH(PMX_TUNED_CIC, double); break;
            case PMX_TUNED_TSC: RH(PMX_TUNED_TSC, double); break;
            default: RH(PMX_TUNED_PCS, double); break;
            }
        } else {
            switch (p.kind) {
            case PMX_TUNED_NNB: RH(PMX_TUNED_NNB, float); break;
            case PMX_TUNED_CIC: RH(PMX_TUNED_CIC, float); break;
            case PMX_TUNED_TSC: RH(PMX_TUNED_TSC, float); break;
            default: RH(PMX_TUNED_PCS, float); break;
            }
        }
#undef RH
#undef RH2
    }
    if (sorted)
        unsort_kernel<<<grid_for(pl->npart, TBLOCK), TBLOCK, 0, st>>>(pl->out_sorted, pl->inv, pl->npart, caller_out);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_readout_binned_multi(pmx_binplan *pl, const pmx_painter *p_, const void *const *canvases, int32_t ncanvas,
                                        const pmx_vec *pos, const pmx_vec *out, void *stream)
{
    PMX_REQUIRE(pl && pl->built, PMX_EINVAL, "bin plan is not built");
    PMX_REQUIRE(p_ && same_geometry(*p_, pl->painter), PMX_EINVAL, "painter differs from the one the plan was built for");
    PMX_REQUIRE(canvases != nullptr && ncanvas >= 1 && ncanvas <= PMX_MAXFIELDS, PMX_EINVAL, "1 to PMX_MAXFIELDS canvases");
    for (int f = 0; f < ncanvas; f++) PMX_REQUIRE(canvases[f] != nullptr, PMX_EINVAL, "canvas is NULL");
    PMX_REQUIRE(vec_ok(out) && out->ncol >= ncanvas, PMX_EINVAL, "out must be (n, >= ncanvas) f4/f8");
    if (pl->npart == 0) return PMX_OK;
    PMX_REQUIRE(vec_ok(pos), PMX_EINVAL, "pos");
    pmx_painter p = *p_;
    const BinGeom &g = pl->g;
    hipStream_t st = (hipStream_t)stream;
    const DVec dout = dvec(out), dpos = dvec(pos);
    // what the lean loop takes (pmx_readout_binned serves everything else, one canvas at a time)
    const bool lean = pl->exact == 0 && !pl->sorted && dpos.stride1 == dpos.elsize && dpos.stride0 == 3 * dpos.elsize
                      && dout.stride0 >= dout.elsize && dout.stride0 % dout.elsize == 0 && dout.stride0 < (1 << 20)
                      && dout.stride1 % dout.elsize == 0 && dout.stride1 < (1 << 20);
    PMX_REQUIRE(lean, PMX_EUNSUPPORTED, "pmx_readout_binned_multi: relaxed arithmetic, index list, dense rows of three positions");
    CanvasSet cs;
    for (int f = 0; f < PMX_MAXFIELDS; f++) cs.ptr[f] = f < ncanvas ? (const char *)canvases[f] : nullptr;
    cs.n = ncanvas;
    cs.ostride1 = (int32_t)dout.stride1;
    const unsigned grid = (unsigned)(g.ntiles < 65535 * 8 ? g.ntiles : 65535 * 8);
    bool whole_r = true;
    for (int d = 0; d < 3; d++) whole_r = whole_r && g.o[d] == 0 && (int)p.period[d] == (int)p.size[d];
#define RML(K, T, PE_, OE_, WH) readout_tile_multi_kernel<K, T, TileThreads<K, T>::readout, PE_, OE_, WH><<<grid, TileThreads<K, T>::readout, 0, st>>>(p, g, cs, dpos, (char *)const_cast<char *>(dout.data), pl->list, pl->offsets, pl->counts, pl->heavy_items, pl->nheavy, (uint32_t)pl->cap_heavy, (int)dout.stride0)
#define RMW(K, T, PE_, OE_) do { if (whole_r) RML(K, T, PE_, OE_, true); else RML(K, T, PE_, OE_, false); } while (0)
#define RMO(K, T, PE_) do { if (dout.elsize == 8) RMW(K, T, PE_, 8); else RMW(K, T, PE_, 4); } while (0)
#define RM(K, T) do { if (dpos.elsize == 8) RMO(K, T, 8); else RMO(K, T, 4); } while (0)
    if (p.canvas_elsize == 8) {
        switch (p.kind) {
        case PMX_TUNED_NNB: RM(PMX_TUNED_NNB, double); break;
        case PMX_TUNED_CIC: RM(PMX_TUNED_CIC, double); break;
        case PMX_TUNED_TSC: RM(PMX_TUNED_TSC, double); break;
        default: RM(PMX_TUNED_PCS, double); break;
        }
    } else {
        switch (p.kind) {
        case PMX_TUNED_NNB: RM(PMX_TUNED_NNB, float); break;
        case PMX_TUNED_CIC: RM(PMX_TUNED_CIC, float); break;
        case PMX_TUNED_TSC: RM(PMX_TUNED_TSC, float); break;
        default: RM(PMX_TUNED_PCS, float); break;
        }
    }
#undef RM
#undef RMO
#undef RMW
#undef RML
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}
#endif   // PMX_PART_READOUT
