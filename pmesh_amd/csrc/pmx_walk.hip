// pmx_walk.hip — the "walk" form of tile-binned paint / readout for wide windows (TSC, PCS).
//
// Why: the tile kernels (pmx_binned.hip) spend S^3 LDS operations per particle (27 / 64
// ds_add_f64 for TSC / PCS paint, as many ds_read for readout), on addresses that are random
// within the tile: at one particle per cell they are bound by the LDS atomic rate
// (~7 lanes/clk/CU on gfx950, half of it with bank conflicts), not by HBM.  Here a workgroup
// owns a PATCH of P1 x P2 = 16 x 32 columns of the mesh and WALKS along axis 0, one mesh plane
// per step.  Thread (b, c) owns column (b, c): it keeps the S x S x S stencil values of "its"
// cell in REGISTERS — a window that slides along axis 0 — and processes the particles whose
// stencil starts in cell (a, b, c) at step a.  Per step and thread only the plane that leaves
// (paint) or enters (readout) the window touches LDS: S^2 operations per CELL (9 / 16 instead
// of 27 / 64 per particle), with lane <-> c contiguous, i.e. free of bank conflicts and of
// same-address collisions.  The particles of a plane bucket arrive in list order; they are
// matched to their owner threads through LDS (a counter per cell hands out ranks, the first
// KOWN particles of a cell go to its owner, the rest — clustered inputs — take the scatter
// form on a compact list: S^3 LDS operations spread over all threads, exactly the tile
// kernels' arithmetic).
//
// Latency: with the window in registers only 1-2 workgroups fit a CU, so nothing hides a
// dependent index -> position gather.  The positions (and masses / list entries) of plane a+1
// are therefore gathered straight into LDS by LDS-DMA (global_load_lds, per-lane source
// address, no VGPR destination) while plane a is processed; the list entries that address them
// are loaded two planes ahead.  One s_waitcnt vmcnt(0) per step, placed after the arithmetic
// of the step and before the barrier that ends it, retires them.
//
// LDS: two raw particle buffers, a ring of S+1 mesh planes of the patch plus its halo
// ((P1+S-1) x (P2+S-1) cells), the per-cell counters.  Paint: plane a is complete once every
// owner has flushed the oldest plane of its window at step a; its P1 x P2 box goes to the
// canvas with plain row stores, the ring around it (and the S-1 planes that trail a segment)
// to the halo staging buffer, which halo_merge_walk_kernel adds with atomics after the kernel
// boundary, as in the tile form.  Readout: plane a+S-1 is staged at step a; the window shifts
// by one plane (S^2 ds_read), the per-particle sum runs over the registers in the
// reference's lexicographic order (bit-identical to pmx_readout).
//
// Bin geometry (pmx_binplan_build with g.walk): bucket = (patch, plane); planes of a patch
// column are consecutive buckets.  Tile-space coordinate of a particle along axis d:
// c_d = I0w_d + o_d (see the header of pmx_binned.hip); plane a = c_0, patch (c_1 / P1, c_2 / P2).
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>

#include "pmx_binplan.h"

namespace pmx {

#ifndef PMX_WALK_K
#define PMX_WALK_K 1
#endif
constexpr int KOWN = PMX_WALK_K;     // particles per cell and step that the owner thread takes
constexpr int LSEGMAX = 64;          // planes per segment (pmx_binplan_build keeps lseg <= this)
// waves per SIMD the register allocation aims at: TSC two workgroups per CU (<= 128 VGPRs),
// PCS one (its window alone is 128 registers)
#ifndef PMX_WALK_WAVES_TSC
#define PMX_WALK_WAVES_TSC 4
#endif
template <int KIND> constexpr int walk_waves() { return Tuned<KIND>::S >= 4 ? 2 : PMX_WALK_WAVES_TSC; }

template <int KIND> struct Walk {
    static constexpr int S = Tuned<KIND>::S;
    static constexpr int R1 = P1 + S - 1, R2 = P2 + S - 1;
    static constexpr int PLANE = R1 * R2;
    static constexpr int NR = S + 1;                       // ring slots
    static constexpr int RING = (S - 1) * (R2 + P1);       // cells of a plane outside the box
    // particle records per chunk: 1.25 x the mean population of a plane at one particle per
    // cell (three raw buffers of TSC and the rest fit 80 KB: two workgroups per CU); PCS runs
    // one workgroup per CU (registers) and has the LDS for more
#ifdef PMX_WALK_NREC
    static constexpr int NREC = PMX_WALK_NREC;
#else
    static constexpr int NREC = S >= 4 ? 2 * WTHREADS : 5 * WTHREADS / 4;
#endif
    static constexpr int REPS = (NREC + WTHREADS - 1) / WTHREADS;
    // compact numbering of the cells of a plane outside the P1 x P2 box
    __device__ static __forceinline__ int ring_index(int b, int c)
    {
        return b >= P1 ? (b - P1) * R2 + c : (S - 1) * R2 + b * (S - 1) + (c - P2);
    }
    __device__ static __forceinline__ void ring_decode(int r, int *b, int *c)
    {
        if (r < (S - 1) * R2) { *b = P1 + r / R2; *c = r % R2; }
        else { r -= (S - 1) * R2; *b = r / (S - 1); *c = P2 + r % (S - 1); }
    }
};

// staged cells of one unit (a patch segment): the ring of every plane + S-1 trailing planes
template <int KIND> __host__ __device__ inline int64_t unit_halo_cells(int lseg)
{
    return (int64_t)lseg * Walk<KIND>::RING + (int64_t)(Walk<KIND>::S - 1) * Walk<KIND>::PLANE;
}

// ---- LDS-DMA ------------------------------------------------------------------------------
// global_load_lds_*: every active lane copies BYTES from its own global address to
// LDS[m0 + lane * BYTES] (dwordx3: lane * 16; scripts/glds_probe.hip); inactive lanes are skipped;
// no VGPR destination, counted by vmcnt.  M0 is reserved by the
// compiler: saved and restored inside the statement (cdna_hip_programming.md, inline asm).
template <int BYTES> __device__ __forceinline__ void glds(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    static_assert(BYTES == 4 || BYTES == 12 || BYTES == 16, "LDS-DMA widths on gfx950");
    if constexpr (BYTES == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else if constexpr (BYTES == 12)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// every vector memory operation of this wave has completed (LDS-DMA included: the compiler does
// not count the asm statements above)
__device__ __forceinline__ void vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Layout of the dynamic LDS of the walk kernels (byte offsets).
//   PE: position element (float / double); XB: bytes of the extra per-particle word that
//   travels with the position (paint: mass element size, 0 = scalar mass; readout: 4 = the list
//   entry); RT: element of the mesh ring (double for paint, the canvas type for readout).
// Three raw particle buffers (plane k is consumed while k+1 is sorted and k+2 lands), two
// sets of per-cell counters / owner slots / overflow lists, three overflow counters.
template <int KIND, typename PE, int XB, typename RT> struct Smem {
    using W = Walk<KIND>;
    static constexpr int NREC = W::NREC;
    static constexpr bool F8 = sizeof(PE) == 8;
    // one raw buffer
    static constexpr int XY = 0;                                   // f8: (x, y) 16 B; f4: (x, y, z, -) 16 B
    static constexpr int ZLO = 16 * NREC, ZHI = 20 * NREC;         // f8 only
    static constexpr int POSB = F8 ? 24 * NREC : 16 * NREC;
    static constexpr int XLO = POSB, XHI = POSB + 4 * NREC;        // the extra word (low / high dword)
    static constexpr int RAWB = (POSB + XB * NREC + 15) & ~15;
    static constexpr int RING = 3 * RAWB;
    static constexpr int CNT = RING + ((W::NR * W::PLANE * (int)sizeof(RT) + 15) & ~15);   // [2][WTHREADS] u32
    static constexpr int SLOT = CNT + 2 * 4 * WTHREADS;                                    // [2][WTHREADS * KOWN] u16
    static constexpr int XLIST = SLOT + 2 * 2 * WTHREADS * KOWN;                           // [2][NREC] u16
    static constexpr int XKEY = XLIST + 2 * 2 * NREC;                                      // [2][NREC] u16
    static constexpr int NX = XKEY + 2 * 2 * NREC;                                         // [4] u32
    static constexpr int SCNT = NX + 16;
    static constexpr int SOFF = SCNT + 4 * (LSEGMAX + 4);
    static constexpr int TOTAL = SOFF + 8 * (LSEGMAX + 4);
    __device__ static __forceinline__ uint32_t *cnt(unsigned char *smem, int set) { return (uint32_t *)(smem + CNT) + set * WTHREADS; }
    __device__ static __forceinline__ uint16_t *slot(unsigned char *smem, int set) { return (uint16_t *)(smem + SLOT) + set * WTHREADS * KOWN; }
    __device__ static __forceinline__ uint16_t *xlist(unsigned char *smem, int set) { return (uint16_t *)(smem + XLIST) + set * NREC; }
    __device__ static __forceinline__ uint16_t *xkey(unsigned char *smem, int set) { return (uint16_t *)(smem + XKEY) + set * NREC; }
};

struct UnitCoords {
    int B, C, a0, a1;
    int64_t bucket0;
};

__device__ __forceinline__ UnitCoords unit_coords(const BinGeom &g, int64_t unit)
{
    UnitCoords u;
    const int seg = (int)(unit % g.nseg);
    const int64_t patch = unit / g.nseg;
    u.C = (int)(patch % g.nt[2]);
    u.B = (int)(patch / g.nt[2]);
    u.a0 = seg * g.lseg;
    u.a1 = u.a0 + g.lseg < g.nt[0] ? u.a0 + g.lseg : g.nt[0];
    u.bucket0 = patch * g.nt[0];
    return u;
}

// What paint and readout share: the particle pipeline of a unit.
template <int KIND, typename PE, int XB, typename RT> struct Pipe {
    using SM = Smem<KIND, PE, XB, RT>;
    using W = Walk<KIND>;
    static constexpr int NREC = W::NREC, REPS = W::REPS;

    // gather the rows idx[r] of chunk [j0, j0 + nsub) of the list into raw buffer `rawoff`
    // (asynchronous).  xvec: the array the extra word is gathered from by row index (mass), or
    // no data with XB == 4: the list entries themselves (contiguous).
    __device__ static __forceinline__ void issue(unsigned char *smem, int rawoff, const DVec &pos, const DVec &xvec,
                                                 const uint32_t *list, int64_t j0, int nsub, const uint32_t *idx)
    {
        const uint32_t base = (uint32_t)(uintptr_t)smem + rawoff;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
        for (int r = 0; r < REPS; r++) {
            const int jj = r * WTHREADS + threadIdx.x;
            const uint32_t wb = (uint32_t)(r * WTHREADS + wave * 64);   // first record of this wave-instruction
            if (jj < nsub) {
                const char *row = pos.data + (uint64_t)idx[r] * (uint32_t)pos.stride0;   // v_mad_u64_u32
                if constexpr (SM::F8) {
                    glds<16>(row, base + SM::XY + wb * 16);
                    glds<4>(row + 16, base + SM::ZLO + wb * 4);
                    glds<4>(row + 20, base + SM::ZHI + wb * 4);
                } else {
                    glds<12>(row, base + SM::XY + wb * 16);      // dwordx3 lands at a 16-byte lane stride
                }
                if constexpr (XB > 0) {
                    const char *x = xvec.data ? xvec.data + (uint64_t)idx[r] * (uint32_t)xvec.stride0
                                              : (const char *)(list + j0 + jj);
                    glds<4>(x, base + SM::XLO + wb * 4);
                    if constexpr (XB == 8) glds<4>(x + 4, base + SM::XHI + wb * 4);
                }
            }
        }
    }

    // list entries of chunk [j0, j0 + nsub) -> registers (ordinary loads: the compiler waits
    // for them where they are used, one step later)
    __device__ static __forceinline__ void load_idx(const uint32_t *list, int64_t j0, int nsub, uint32_t *idx)
    {
#pragma unroll
        for (int r = 0; r < REPS; r++) {
            const int jj = r * WTHREADS + threadIdx.x;
            idx[r] = jj < nsub ? list[j0 + jj] : 0u;
        }
    }

    __device__ static __forceinline__ void read_pos(const unsigned char *smem, int rawoff, int j, double *x)
    {
        const unsigned char *b = smem + rawoff;
        if constexpr (SM::F8) {
            const double2 xy = *(const double2 *)(b + SM::XY + j * 16);
            const int lo = *(const int *)(b + SM::ZLO + j * 4), hi = *(const int *)(b + SM::ZHI + j * 4);
            x[0] = xy.x; x[1] = xy.y; x[2] = __hiloint2double(hi, lo);
        } else {
            const float *f = (const float *)(b + SM::XY + j * 16);
            x[0] = (double)f[0]; x[1] = (double)f[1]; x[2] = (double)f[2];
        }
    }

    __device__ static __forceinline__ double read_mass(const unsigned char *smem, int rawoff, int j)
    {
        const unsigned char *b = smem + rawoff;
        if constexpr (XB == 8)
            return __hiloint2double(*(const int *)(b + SM::XHI + j * 4), *(const int *)(b + SM::XLO + j * 4));
        else
            return (double)*(const float *)(b + SM::XLO + j * 4);
    }

    __device__ static __forceinline__ uint32_t read_index(const unsigned char *smem, int rawoff, int j)
    {
        return *(const uint32_t *)(smem + rawoff + SM::XLO + j * 4);
    }

    // the records of a chunk are matched to the owner threads of their cells (counter / slot /
    // overflow set `set`).  Only the column (b, c) inside the patch is computed here; a record
    // that falls outside the patch — a plan that no longer matches the positions — is ignored,
    // so nothing indexes outside LDS (the plane is not checked: the window and the ring are
    // addressed relative to the step).
    __device__ static __forceinline__ void sort(const pmx_painter &p, const BinGeom &g, const UnitCoords &u,
                                                unsigned char *smem, int rawoff, int nsub, int set, uint32_t *nxc)
    {
        uint32_t *cnt = SM::cnt(smem, set);
        uint16_t *slot = SM::slot(smem, set);
        uint16_t *xlist = SM::xlist(smem, set), *xkey = SM::xkey(smem, set);
        const int org1 = u.B * P1 - g.o[1], org2 = u.C * P2 - g.o[2];
        const int per1 = (int)p.period[1], per2 = (int)p.period[2], size1 = (int)p.size[1], size2 = (int)p.size[2];
#pragma unroll
        for (int r = 0; r < REPS; r++) {
            const int jj = r * WTHREADS + threadIdx.x;
            if (jj < nsub) {
                double x[3];
                read_pos(smem, rawoff, jj, x);
                int w1 = wrap_fast(Tuned<KIND>::first(x[1] * p.scale[1] + p.translate[1]), per1);
                int w2 = wrap_fast(Tuned<KIND>::first(x[2] * p.scale[2] + p.translate[2]), per2);
                if (per1 > 0 && w1 >= size1) w1 -= per1;
                if (per2 > 0 && w2 >= size2) w2 -= per2;
                const int b = w1 - org1, c = w2 - org2;
                if ((unsigned)b < (unsigned)P1 && (unsigned)c < (unsigned)P2) {
                    const int key = b * P2 + c;
                    const uint32_t rank = atomicAdd(&cnt[key], 1u);
                    if (rank < (uint32_t)KOWN) slot[key * KOWN + rank] = (uint16_t)jj;
                    else {
                        const uint32_t e = atomicAdd(nxc, 1u);
                        xlist[e] = (uint16_t)jj;
                        xkey[e] = (uint16_t)key;
                    }
                }
            }
        }
    }
};

template <int KIND>
__device__ __forceinline__ void weights(const pmx_painter &p, const double *x, double (*V)[Tuned<KIND>::S])
{
    constexpr int S = Tuned<KIND>::S;
    int I[S];
#pragma unroll
    for (int d = 0; d < 3; d++) Tuned<KIND>::axis(x[d] * p.scale[d] + p.translate[d], p.order[d], p.scale[d], I, V[d]);
}

__device__ __forceinline__ int min_u(uint32_t a, int b) { return a < (uint32_t)b ? (int)a : b; }

// One step of the walk handles three planes at once, so that their latencies overlap inside
// every wave and a single barrier ends the step:
//   plane k+2: the gather of its records is issued (LDS-DMA), the list entries of k+3 requested;
//   plane k+1: its records (landed during the previous step) are matched to their owners;
//   plane k  : the owners accumulate, crowded cells scatter, the window gives up its oldest plane;
//   plane k-1: complete since the previous barrier, it leaves for the canvas / the staging.
// XB: 0 scalar mass, 4 / 8: per-particle float / double mass
template <int KIND, typename T, typename PE, int XB>
__global__ void __launch_bounds__(WTHREADS, walk_waves<KIND>()) paint_walk_kernel(pmx_painter p, BinGeom g, char *canvas, DVec pos,
                                                             DVec mass, double mass_scalar, const uint32_t *list,
                                                             const int64_t *offsets, const uint32_t *counts,
                                                             T *halo, uint32_t *unit_flags, int overwrite)
{
    using W = Walk<KIND>;
    using SM = Smem<KIND, PE, XB, double>;
    using PP = Pipe<KIND, PE, XB, double>;
    constexpr int S = W::S, R2 = W::R2, PLANE = W::PLANE, NR = W::NR, RING = W::RING, NREC = W::NREC, REPS = W::REPS;
    extern __shared__ __align__(16) unsigned char smem[];
    double *ring = (double *)(smem + SM::RING);
    uint32_t *nx = (uint32_t *)(smem + SM::NX);
    uint32_t *scnt = (uint32_t *)(smem + SM::SCNT);
    int64_t *soff = (int64_t *)(smem + SM::SOFF);
    const int tid = threadIdx.x, tb = tid / P2, tc = tid % P2;
    const int64_t unit_halo = unit_halo_cells<KIND>(g.lseg);
    for (int64_t unit = blockIdx.x; unit < g.nunits; unit += gridDim.x) {
        const UnitCoords u = unit_coords(g, unit);
        const int Lu = u.a1 - u.a0;
        // populations and list ranges of the planes of this unit (+ zeros behind the end)
        if (tid < LSEGMAX + 4) {
            scnt[tid] = tid < Lu ? counts[u.bucket0 + u.a0 + tid] : 0u;
            soff[tid] = tid < Lu ? offsets[u.bucket0 + u.a0 + tid] : 0;
        }
        // nothing to add in this unit (uniform per workgroup): skip it, its staging stays unused
        const int any = __syncthreads_or(tid < Lu && scnt[tid] != 0);
        if (!any && !overwrite) {
            if (tid == 0) unit_flags[unit] = 0;
            __syncthreads();
            continue;
        }
        if (tid == 0) { unit_flags[unit] = 1; nx[0] = nx[1] = nx[2] = nx[3] = 0; }
        for (int q = tid; q < NR * PLANE; q += WTHREADS) ring[q] = 0;
        SM::cnt(smem, 0)[tid] = 0;
        SM::cnt(smem, 1)[tid] = 0;
        T *hbase = halo + unit * unit_halo;
        double acc[S][S][S];
#pragma unroll
        for (int i = 0; i < S; i++)
#pragma unroll
            for (int j = 0; j < S; j++)
#pragma unroll
                for (int k = 0; k < S; k++) acc[i][j][k] = 0;
        // per-thread constants of the write-out: my cell of the box, my cell of the ring
        const int l1c = u.B * P1 - g.o[1] + tb, l2c = u.C * P2 - g.o[2] + tc;
        const bool cell_ok = l1c >= 0 && l1c < p.size[1] && l2c >= 0 && l2c < p.size[2];
        const int64_t cell_off = l1c * p.strides[1] + l2c * p.strides[2];
        int ringq = 0;
        if (tid < RING) {
            int b, c;
            W::ring_decode(tid, &b, &c);
            ringq = b * R2 + c;
        }
        // prologue: the first chunks of planes 0 and 1 land in raw buffers 0 and 1, the list
        // entries of plane 2 are in flight, plane 0 is sorted
        uint32_t idx[REPS];
        PP::load_idx(list, soff[0], min_u(scnt[0], NREC), idx);
        PP::issue(smem, 0, pos, mass, list, soff[0], min_u(scnt[0], NREC), idx);
        PP::load_idx(list, soff[1], min_u(scnt[1], NREC), idx);
        PP::issue(smem, SM::RAWB, pos, mass, list, soff[1], min_u(scnt[1], NREC), idx);
        PP::load_idx(list, soff[2], min_u(scnt[2], NREC), idx);
        vm_drain();
        __syncthreads();
        PP::sort(p, g, u, smem, 0, min_u(scnt[0], NREC), 0, &nx[0]);
        __syncthreads();

        // the owner part and the scatter part of one chunk whose records are in `rawoff`
        auto accumulate = [&](auto rot, const int a, const int sl0, const int rawoff, const int set, const uint32_t nxv) __attribute__((always_inline)) {
            constexpr int ROT = decltype(rot)::value;
            uint32_t *cnt = SM::cnt(smem, set);
            const uint16_t *slot = SM::slot(smem, set);
            const uint16_t *xlist = SM::xlist(smem, set), *xkey = SM::xkey(smem, set);
            {
                const uint32_t cn = cnt[tid];
                cnt[tid] = 0;
                const int nown = cn < (uint32_t)KOWN ? (int)cn : KOWN;
                for (int r = 0; r < nown; r++) {
                    const int j = slot[tid * KOWN + r];
                    double x[3], V[3][S];
                    PP::read_pos(smem, rawoff, j, x);
                    const double m = XB ? PP::read_mass(smem, rawoff, j) : mass_scalar;
                    weights<KIND>(p, x, V);
#pragma unroll
                    for (int i = 0; i < S; i++) V[0][i] *= m;
#pragma unroll
                    for (int i = 0; i < S; i++)
#pragma unroll
                        for (int jb = 0; jb < S; jb++) {
                            const double fb = V[0][i] * V[1][jb];
                            // one rounding instead of two: the order of the additions into a cell
                            // already differs from the reference's (see the header)
#pragma unroll
                            for (int k = 0; k < S; k++)
                                acc[(i + ROT) % S][jb][k] = __builtin_fma(fb, V[2][k], acc[(i + ROT) % S][jb][k]);
                        }
                }
            }
            // the rest of crowded cells: scatter form on a compact list, taken by a different
            // wave first every step (the SIMDs that host the first waves would do all of it)
            for (uint32_t e = (uint32_t)((tid - 64 * a) & (WTHREADS - 1)); e < nxv; e += WTHREADS) {
                const int j = xlist[e];
                const int key = xkey[e];
                const int b = key / P2, c = key % P2;
                double x[3], V[3][S];
                PP::read_pos(smem, rawoff, j, x);
                const double m = XB ? PP::read_mass(smem, rawoff, j) : mass_scalar;
                weights<KIND>(p, x, V);
#pragma unroll
                for (int i = 0; i < S; i++) V[0][i] *= m;
#pragma unroll
                for (int i = 0; i < S; i++) {
                    int sl = sl0 + i;
                    if (sl >= NR) sl -= NR;
#pragma unroll
                    for (int jb = 0; jb < S; jb++) {
                        const double fb = V[0][i] * V[1][jb];
                        const int rowoff = sl * PLANE + (b + jb) * R2 + c;
#pragma unroll
                        for (int k = 0; k < S; k++) unsafeAtomicAdd(&ring[rowoff + k], fb * V[2][k]);
                    }
                }
            }
        };

        // plane `pk` (index in the unit) leaves the ring: box -> canvas (plain row stores),
        // ring around it (and whole planes behind the segment) -> staging
        auto write_out = [&](const int pk) __attribute__((always_inline)) {
            const int a = u.a0 + pk;
            const int sl = a % NR;
            if (pk < Lu) {
                const int l0 = a - g.o[0];
                const int q = tb * R2 + tc;
                const double v = ring[sl * PLANE + q];
                ring[sl * PLANE + q] = 0;
                if (cell_ok && l0 >= 0 && l0 < p.size[0]) {
                    T *dst = (T *)(canvas + l0 * p.strides[0] + cell_off);
                    if (overwrite) *dst = (T)v;
                    else *dst += (T)v;
                }
                if (tid < RING) {
                    hbase[(int64_t)pk * RING + tid] = (T)ring[sl * PLANE + ringq];
                    ring[sl * PLANE + ringq] = 0;
                }
            } else {
                T *hplane = hbase + (int64_t)Lu * RING + (int64_t)(pk - Lu) * PLANE;
                for (int q = tid; q < PLANE; q += WTHREADS) {
                    hplane[q] = (T)ring[sl * PLANE + q];
                    ring[sl * PLANE + q] = 0;
                }
            }
        };

        auto step = [&](auto rot, const int k) __attribute__((always_inline)) {
            constexpr int ROT = decltype(rot)::value;
            const int a = u.a0 + k;
            const int sl0 = a % NR;
            const uint32_t n = scnt[k];                    // 0 behind the segment
            int b0 = k % 3, b1 = b0 + 1, b2 = b0 + 2;      // raw buffers of planes k, k+1, k+2
            if (b1 >= 3) b1 -= 3;
            if (b2 >= 3) b2 -= 3;
            if (tid == 0) nx[(k + 2) & 3] = 0;             // counter of plane k+2: last read two steps ago
            // plane k+2: gather its records; list entries of plane k+3
            if (k + 2 < Lu) {
                PP::issue(smem, b2 * SM::RAWB, pos, mass, list, soff[k + 2], min_u(scnt[k + 2], NREC), idx);
                if (k + 3 < Lu) PP::load_idx(list, soff[k + 3], min_u(scnt[k + 3], NREC), idx);
            }
            // plane k+1: match its records to their owners
            if (k + 1 < Lu) PP::sort(p, g, u, smem, b1 * SM::RAWB, min_u(scnt[k + 1], NREC), (k + 1) & 1, &nx[(k + 1) & 3]);
            // plane k: accumulate
            if (n > 0) {
                accumulate(rot, a, sl0, b0 * SM::RAWB, k & 1, nx[k & 3]);
                // crowded planes: the chunks behind the first one are fetched on the spot
                for (uint32_t sub0 = NREC; sub0 < n; sub0 += NREC) {
                    const int ns = min_u(n - sub0, NREC);
                    uint32_t idx2[REPS];
                    PP::load_idx(list, soff[k] + sub0, ns, idx2);
                    __syncthreads();                       // the records of the previous chunk are done with
                    if (tid == 0) nx[k & 3] = 0;
                    PP::issue(smem, b0 * SM::RAWB, pos, mass, list, soff[k] + sub0, ns, idx2);
                    vm_drain();
                    __syncthreads();
                    PP::sort(p, g, u, smem, b0 * SM::RAWB, ns, k & 1, &nx[k & 3]);
                    __syncthreads();
                    accumulate(rot, a, sl0, b0 * SM::RAWB, k & 1, nx[k & 3]);
                }
            }
            // the oldest plane of the window is complete for this thread: add it to ring plane k
#pragma unroll
            for (int j = 0; j < S; j++)
#pragma unroll
                for (int kk = 0; kk < S; kk++) {
                    unsafeAtomicAdd(&ring[sl0 * PLANE + (tb + j) * R2 + tc + kk], acc[ROT][j][kk]);
                    acc[ROT][j][kk] = 0;
                }
            // plane k-1 is complete since the last barrier
            if (k > 0) write_out(k - 1);
            vm_drain();                 // the records of plane k+2 have landed (this wave's share)
            __syncthreads();
        };

        const int nsteps = Lu + S - 1;
        for (int k = 0; k < nsteps; k += S) {
            step(std::integral_constant<int, 0>(), k);
            if (S > 1 && k + 1 < nsteps) step(std::integral_constant<int, 1 % S>(), k + 1);
            if (S > 2 && k + 2 < nsteps) step(std::integral_constant<int, 2 % S>(), k + 2);
            if (S > 3 && k + 3 < nsteps) step(std::integral_constant<int, 3 % S>(), k + 3);
        }
        write_out(nsteps - 1);
        __syncthreads();
    }
}

// second pass: add the staged cells of every unit to their owners (after ALL boxes are stored)
template <int KIND, typename T>
__global__ void __launch_bounds__(TBLOCK) halo_merge_walk_kernel(pmx_painter p, BinGeom g, char *canvas, const T *halo,
                                                                 const uint32_t *unit_flags)
{
    using W = Walk<KIND>;
    constexpr int R2 = W::R2, PLANE = W::PLANE, RING = W::RING;
    const int64_t unit_halo = unit_halo_cells<KIND>(g.lseg);
    for (int64_t unit = blockIdx.x; unit < g.nunits; unit += gridDim.x) {
        if (!unit_flags[unit]) continue;
        const UnitCoords u = unit_coords(g, unit);
        const int Lu = u.a1 - u.a0;
        const int nring = Lu * RING, total = nring + (W::S - 1) * PLANE;
        const T *hbase = halo + unit * unit_halo;
        for (int h = threadIdx.x; h < total; h += TBLOCK) {
            const T v = hbase[h];
            if (v == (T)0) continue;
            int pa, b, c;
            if (h < nring) {
                pa = h / RING;
                W::ring_decode(h - pa * RING, &b, &c);
            } else {
                const int hh = h - nring;
                pa = Lu + hh / PLANE;
                const int q = hh % PLANE;
                b = q / R2; c = q % R2;
            }
            const int l0 = wrap_near(u.a0 + pa - g.o[0], p.period[0]);
            const int l1 = wrap_near(u.B * P1 - g.o[1] + b, p.period[1]);
            const int l2 = wrap_near(u.C * P2 - g.o[2] + c, p.period[2]);
            if (l0 < 0 || l0 >= p.size[0] || l1 < 0 || l1 >= p.size[1] || l2 < 0 || l2 >= p.size[2]) continue;
            unsafeAtomicAdd((T *)(canvas + l0 * p.strides[0] + l1 * p.strides[1] + l2 * p.strides[2]), v);
        }
    }
}

// The same pipeline for readout.  One step: the records of plane k+2 are gathered, plane k+1 is
// sorted, the register window of plane k takes mesh plane a+S-1 (staged in the ring during the
// previous step) and the particles of plane k are read out; mesh plane a+S goes from registers
// into the ring and plane a+S+1 is requested from the canvas.
template <int KIND, typename T, typename PE>
__global__ void __launch_bounds__(WTHREADS, walk_waves<KIND>()) readout_walk_kernel(pmx_painter p, BinGeom g, const char *canvas, DVec pos,
                                                               DVec out, const uint32_t *list, const int64_t *offsets,
                                                               const uint32_t *counts)
{
    using W = Walk<KIND>;
    using SM = Smem<KIND, PE, 4, T>;
    using PP = Pipe<KIND, PE, 4, T>;
    constexpr int S = W::S, R2 = W::R2, PLANE = W::PLANE, NR = W::NR, NREC = W::NREC, REPS = W::REPS;
    constexpr int NPL = (PLANE + WTHREADS - 1) / WTHREADS;
    extern __shared__ __align__(16) unsigned char smem[];
    T *ring = (T *)(smem + SM::RING);
    uint32_t *nx = (uint32_t *)(smem + SM::NX);
    uint32_t *scnt = (uint32_t *)(smem + SM::SCNT);
    int64_t *soff = (int64_t *)(smem + SM::SOFF);
    const int tid = threadIdx.x, tb = tid / P2, tc = tid % P2;
    const DVec nox = {nullptr, 0, 0, 8};
    for (int64_t unit = blockIdx.x; unit < g.nunits; unit += gridDim.x) {
        const UnitCoords u = unit_coords(g, unit);
        const int Lu = u.a1 - u.a0;
        if (tid < LSEGMAX + 4) {
            scnt[tid] = tid < Lu ? counts[u.bucket0 + u.a0 + tid] : 0u;
            soff[tid] = tid < Lu ? offsets[u.bucket0 + u.a0 + tid] : 0;
        }
        const int any = __syncthreads_or(tid < Lu && scnt[tid] != 0);
        if (!any) {
            __syncthreads();
            continue;
        }
        if (tid == 0) nx[0] = nx[1] = nx[2] = nx[3] = 0;
        SM::cnt(smem, 0)[tid] = 0;
        SM::cnt(smem, 1)[tid] = 0;
        // the cells of a plane of the patch + halo that this thread stages: their place in the
        // canvas is the same for every plane (cells outside the block read 0)
        int64_t poff[NPL];
        bool pok[NPL];
#pragma unroll
        for (int r = 0; r < NPL; r++) {
            const int q = tid + r * WTHREADS;
            const int b = q / R2, c = q - b * R2;
            const int l1 = wrap_near(u.B * P1 - g.o[1] + b, p.period[1]);
            const int l2 = wrap_near(u.C * P2 - g.o[2] + c, p.period[2]);
            pok[r] = q < PLANE && l1 >= 0 && l1 < p.size[1] && l2 >= 0 && l2 < p.size[2];
            poff[r] = l1 * p.strides[1] + l2 * p.strides[2];
        }
        auto fetch_plane = [&](const int pl, T *v) __attribute__((always_inline)) {
            const int l0 = wrap_near(pl - g.o[0], p.period[0]);
            const bool in0 = l0 >= 0 && l0 < p.size[0];
            const char *base = canvas + l0 * p.strides[0];
#pragma unroll
            for (int r = 0; r < NPL; r++) v[r] = (in0 && pok[r]) ? *(const T *)(base + poff[r]) : (T)0;
        };
        auto store_plane = [&](const int pl, const T *v) __attribute__((always_inline)) {
            const int sl = pl % NR;
#pragma unroll
            for (int r = 0; r < NPL; r++) {
                const int q = tid + r * WTHREADS;
                if (q < PLANE) ring[sl * PLANE + q] = v[r];
            }
        };
        // prologue: planes a0 .. a0+S-1 of the mesh in the ring, plane a0+S requested; records
        // of planes 0 and 1 landed, list entries of plane 2 in flight, plane 0 sorted
        T pv[NPL];
#pragma unroll
        for (int k = 0; k < S; k++) {
            fetch_plane(u.a0 + k, pv);
            store_plane(u.a0 + k, pv);
        }
        fetch_plane(u.a0 + S, pv);
        uint32_t idx[REPS];
        PP::load_idx(list, soff[0], min_u(scnt[0], NREC), idx);
        PP::issue(smem, 0, pos, nox, list, soff[0], min_u(scnt[0], NREC), idx);
        PP::load_idx(list, soff[1], min_u(scnt[1], NREC), idx);
        PP::issue(smem, SM::RAWB, pos, nox, list, soff[1], min_u(scnt[1], NREC), idx);
        PP::load_idx(list, soff[2], min_u(scnt[2], NREC), idx);
        vm_drain();
        __syncthreads();
        PP::sort(p, g, u, smem, 0, min_u(scnt[0], NREC), 0, &nx[0]);
        T win[S][S][S];
#pragma unroll
        for (int i = 0; i < S; i++)
#pragma unroll
            for (int j = 0; j < S; j++)
#pragma unroll
                for (int k = 0; k < S; k++)
                    win[i][j][k] = i < S - 1 ? ring[((u.a0 + i) % NR) * PLANE + (tb + j) * R2 + tc + k] : (T)0;
        __syncthreads();

        auto gather = [&](auto rot, const int a, const int sl0, const int rawoff, const int set, const uint32_t nxv) __attribute__((always_inline)) {
            constexpr int ROT = decltype(rot)::value;
            uint32_t *cnt = SM::cnt(smem, set);
            const uint16_t *slot = SM::slot(smem, set);
            const uint16_t *xlist = SM::xlist(smem, set), *xkey = SM::xkey(smem, set);
            {
                const uint32_t cn = cnt[tid];
                cnt[tid] = 0;
                const int nown = cn < (uint32_t)KOWN ? (int)cn : KOWN;
                for (int r = 0; r < nown; r++) {
                    const int j = slot[tid * KOWN + r];
                    double x[3], V[3][S];
                    PP::read_pos(smem, rawoff, j, x);
                    weights<KIND>(p, x, V);
                    double value = 0;
#pragma unroll
                    for (int i = 0; i < S; i++)
#pragma unroll
                        for (int jb = 0; jb < S; jb++) {
                            const double fb = V[0][i] * V[1][jb];
#pragma unroll
                            for (int k = 0; k < S; k++) value += (double)win[(i + ROT) % S][jb][k] * (fb * V[2][k]);
                        }
                    out.set((int64_t)PP::read_index(smem, rawoff, j), 0, value);
                }
            }
            for (uint32_t e = (uint32_t)((tid - 64 * a) & (WTHREADS - 1)); e < nxv; e += WTHREADS) {
                const int j = xlist[e];
                const int key = xkey[e];
                const int b = key / P2, c = key % P2;
                double x[3], V[3][S];
                PP::read_pos(smem, rawoff, j, x);
                weights<KIND>(p, x, V);
                double value = 0;
#pragma unroll
                for (int i = 0; i < S; i++) {
                    int sl = sl0 + i;
                    if (sl >= NR) sl -= NR;
#pragma unroll
                    for (int jb = 0; jb < S; jb++) {
                        const double fb = V[0][i] * V[1][jb];
                        const int rowoff = sl * PLANE + (b + jb) * R2 + c;
#pragma unroll
                        for (int k = 0; k < S; k++) value += (double)ring[rowoff + k] * (fb * V[2][k]);
                    }
                }
                out.set((int64_t)PP::read_index(smem, rawoff, j), 0, value);
            }
        };

        auto step = [&](auto rot, const int k) __attribute__((always_inline)) {
            constexpr int ROT = decltype(rot)::value;
            const int a = u.a0 + k;
            const int sl0 = a % NR;
            const uint32_t n = scnt[k];
            int b0 = k % 3, b1 = b0 + 1, b2 = b0 + 2;
            if (b1 >= 3) b1 -= 3;
            if (b2 >= 3) b2 -= 3;
            if (tid == 0) nx[(k + 2) & 3] = 0;
            // mesh plane a+S (requested during the previous step) enters the ring for the next
            // step, plane a+S+1 is requested
            if (k + 1 < Lu) {
                store_plane(a + S, pv);
                if (k + 2 < Lu) fetch_plane(a + S + 1, pv);
            }
            if (k + 2 < Lu) {
                PP::issue(smem, b2 * SM::RAWB, pos, nox, list, soff[k + 2], min_u(scnt[k + 2], NREC), idx);
                if (k + 3 < Lu) PP::load_idx(list, soff[k + 3], min_u(scnt[k + 3], NREC), idx);
            }
            if (k + 1 < Lu) PP::sort(p, g, u, smem, b1 * SM::RAWB, min_u(scnt[k + 1], NREC), (k + 1) & 1, &nx[(k + 1) & 3]);
            // the window takes mesh plane a+S-1 (in the ring since the previous barrier)
            {
                int sl = sl0 + S - 1;
                if (sl >= NR) sl -= NR;
#pragma unroll
                for (int j = 0; j < S; j++)
#pragma unroll
                    for (int kk = 0; kk < S; kk++) win[(S - 1 + ROT) % S][j][kk] = ring[sl * PLANE + (tb + j) * R2 + tc + kk];
            }
            if (n > 0) {
                gather(rot, a, sl0, b0 * SM::RAWB, k & 1, nx[k & 3]);
                for (uint32_t sub0 = NREC; sub0 < n; sub0 += NREC) {
                    const int ns = min_u(n - sub0, NREC);
                    uint32_t idx2[REPS];
                    PP::load_idx(list, soff[k] + sub0, ns, idx2);
                    __syncthreads();
                    if (tid == 0) nx[k & 3] = 0;
                    PP::issue(smem, b0 * SM::RAWB, pos, nox, list, soff[k] + sub0, ns, idx2);
                    vm_drain();
                    __syncthreads();
                    PP::sort(p, g, u, smem, b0 * SM::RAWB, ns, k & 1, &nx[k & 3]);
                    __syncthreads();
                    gather(rot, a, sl0, b0 * SM::RAWB, k & 1, nx[k & 3]);
                }
            }
            vm_drain();
            __syncthreads();
        };

        for (int k = 0; k < Lu; k += S) {
            step(std::integral_constant<int, 0>(), k);
            if (S > 1 && k + 1 < Lu) step(std::integral_constant<int, 1 % S>(), k + 1);
            if (S > 2 && k + 2 < Lu) step(std::integral_constant<int, 2 % S>(), k + 2);
            if (S > 3 && k + 3 < Lu) step(std::integral_constant<int, 3 % S>(), k + 3);
        }
        __syncthreads();
    }
}

// the walk kernels gather whole rows by LDS-DMA: columns of a row contiguous, rows (and the
// mass entries) on 4-byte boundaries
bool walk_layout_ok(const pmx_vec *pos)
{
    return pos && pos->data && pos->ncol >= 3 && pos->stride1 == pos->elsize && pos->stride0 % 4 == 0 &&
           ((uintptr_t)pos->data) % 4 == 0;
}

template <typename K> static int set_lds(K kernel, int bytes)
{
    PMX_HIP_CHECK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return PMX_OK;
}

template <int KIND, typename T, typename PE, int XB>
static int paint_walk_launch(pmx_binplan *pl, const pmx_painter &p, void *canvas, DVec pos, DVec mass, double ms,
                             int overwrite, unsigned grid, hipStream_t st)
{
    using SM = Smem<KIND, PE, XB, double>;
    auto k = paint_walk_kernel<KIND, T, PE, XB>;
    static bool configured = false;
    if (!configured) {
        int rc = set_lds(k, SM::TOTAL);
        if (rc) return rc;
        configured = true;
    }
    k<<<grid, WTHREADS, SM::TOTAL, st>>>(p, pl->g, (char *)canvas, pos, mass, ms, pl->list, pl->offsets, pl->counts,
                                         (T *)pl->halo, pl->unit_flags, overwrite);
    return PMX_OK;
}

template <int KIND, typename T>
static int paint_walk_t(pmx_binplan *pl, const pmx_painter &p, void *canvas, DVec pos, DVec mass, double ms,
                        int overwrite, hipStream_t st)
{
    const BinGeom &g = pl->g;
    const size_t need = (size_t)g.nunits * (size_t)unit_halo_cells<KIND>(g.lseg) * sizeof(T);
    int rc = plan_ensure(&pl->halo, &pl->cap_halo, need > 0 ? need : 16);
    if (rc) return rc;
    size_t capb = pl->cap_units * 4;
    rc = plan_ensure((void **)&pl->unit_flags, &capb, (size_t)(g.nunits + 1) * 4);
    if (rc) return rc;
    pl->cap_units = capb / 4;
    const unsigned grid = (unsigned)(g.nunits < 65535 * 8 ? g.nunits : 65535 * 8);
    const int xb = mass.data ? mass.elsize : 0;
    if (pos.elsize == 8) {
        if (xb == 0) rc = paint_walk_launch<KIND, T, double, 0>(pl, p, canvas, pos, mass, ms, overwrite, grid, st);
        else if (xb == 8) rc = paint_walk_launch<KIND, T, double, 8>(pl, p, canvas, pos, mass, ms, overwrite, grid, st);
        else rc = paint_walk_launch<KIND, T, double, 4>(pl, p, canvas, pos, mass, ms, overwrite, grid, st);
    } else {
        if (xb == 0) rc = paint_walk_launch<KIND, T, float, 0>(pl, p, canvas, pos, mass, ms, overwrite, grid, st);
        else if (xb == 8) rc = paint_walk_launch<KIND, T, float, 8>(pl, p, canvas, pos, mass, ms, overwrite, grid, st);
        else rc = paint_walk_launch<KIND, T, float, 4>(pl, p, canvas, pos, mass, ms, overwrite, grid, st);
    }
    if (rc) return rc;
    halo_merge_walk_kernel<KIND, T><<<grid, TBLOCK, 0, st>>>(p, g, (char *)canvas, (const T *)pl->halo, pl->unit_flags);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

int paint_walk(pmx_binplan *pl, const pmx_painter &p, void *canvas, DVec pos, DVec mass, double ms, int overwrite,
               hipStream_t st)
{
    const bool f8 = p.canvas_elsize == 8;
    switch (p.kind) {
    case PMX_TUNED_TSC:
        return f8 ? paint_walk_t<PMX_TUNED_TSC, double>(pl, p, canvas, pos, mass, ms, overwrite, st)
                  : paint_walk_t<PMX_TUNED_TSC, float>(pl, p, canvas, pos, mass, ms, overwrite, st);
    case PMX_TUNED_PCS:
        return f8 ? paint_walk_t<PMX_TUNED_PCS, double>(pl, p, canvas, pos, mass, ms, overwrite, st)
                  : paint_walk_t<PMX_TUNED_PCS, float>(pl, p, canvas, pos, mass, ms, overwrite, st);
    }
    set_error("walk kernels: window kind %d is not built", (int)p.kind);
    return PMX_EUNSUPPORTED;
}

template <int KIND, typename T, typename PE>
static int readout_walk_launch(pmx_binplan *pl, const pmx_painter &p, const void *canvas, DVec pos, DVec out,
                               hipStream_t st)
{
    using SM = Smem<KIND, PE, 4, T>;
    auto k = readout_walk_kernel<KIND, T, PE>;
    static bool configured = false;
    if (!configured) {
        int rc = set_lds(k, SM::TOTAL);
        if (rc) return rc;
        configured = true;
    }
    const BinGeom &g = pl->g;
    const unsigned grid = (unsigned)(g.nunits < 65535 * 8 ? g.nunits : 65535 * 8);
    k<<<grid, WTHREADS, SM::TOTAL, st>>>(p, g, (const char *)canvas, pos, out, pl->list, pl->offsets, pl->counts);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

template <int KIND, typename T>
static int readout_walk_t(pmx_binplan *pl, const pmx_painter &p, const void *canvas, DVec pos, DVec out, hipStream_t st)
{
    if (pos.elsize == 8) return readout_walk_launch<KIND, T, double>(pl, p, canvas, pos, out, st);
    return readout_walk_launch<KIND, T, float>(pl, p, canvas, pos, out, st);
}

int readout_walk(pmx_binplan *pl, const pmx_painter &p, const void *canvas, DVec pos, DVec out, hipStream_t st)
{
    const bool f8 = p.canvas_elsize == 8;
    switch (p.kind) {
    case PMX_TUNED_TSC:
        return f8 ? readout_walk_t<PMX_TUNED_TSC, double>(pl, p, canvas, pos, out, st)
                  : readout_walk_t<PMX_TUNED_TSC, float>(pl, p, canvas, pos, out, st);
    case PMX_TUNED_PCS:
        return f8 ? readout_walk_t<PMX_TUNED_PCS, double>(pl, p, canvas, pos, out, st)
                  : readout_walk_t<PMX_TUNED_PCS, float>(pl, p, canvas, pos, out, st);
    }
    set_error("walk kernels: window kind %d is not built", (int)p.kind);
    return PMX_EUNSUPPORTED;
}

}  // namespace pmx
