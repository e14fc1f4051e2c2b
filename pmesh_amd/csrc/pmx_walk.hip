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
// the window touches LDS: S^2 operations per CELL (9 / 16 instead of 27 / 64 per particle),
// with lane <-> c contiguous, i.e. free of bank conflicts and of same-address collisions.
// The particles of a plane bucket arrive in list order; they are matched to their owner
// threads through LDS (a counter per cell hands out ranks, the first KOWN particles of a
// cell go to its owner, the rest — clustered inputs — take the scatter form on a compact
// list: S^3 LDS operations spread over all threads, exactly the tile kernels' arithmetic).
//
// LDS: a ring of S+1 mesh planes of the patch plus its halo ((P1+S-1) x (P2+S-1) cells), the
// particle records of one step, the per-cell counters.  Paint: plane a is complete once every
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
#define PMX_WALK_K 2
#endif
constexpr int KOWN = PMX_WALK_K;     // particles per cell and step that the owner thread takes
constexpr int NREC = 2 * WTHREADS;   // particle records per sub-step

template <int KIND> struct Walk {
    static constexpr int S = Tuned<KIND>::S;
    static constexpr int R1 = P1 + S - 1, R2 = P2 + S - 1;
    static constexpr int PLANE = R1 * R2;
    static constexpr int NR = S + 1;                       // ring slots
    static constexpr int RING = (S - 1) * (R2 + P1);       // cells of a plane outside the box
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

struct __align__(16) WRec {
    double x[3];     // grid coordinates pos * scale + translate
    double m;        // paint: mass; readout: the particle's row index
};

template <typename E> __device__ __forceinline__ double ld_elem(const DVec &v, int64_t i, int c)
{
    return (double)*(const E *)(v.data + i * v.stride0 + c * v.stride1);
}

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

// Phase A of a step, shared by paint and readout: the particles [j0, j0 + nsub) of the
// plane bucket are loaded, checked against the bucket (a stale plan must not index outside
// LDS), recorded and matched to the owner threads of their cells.
template <int KIND, typename PE, bool PAINT>
__device__ __forceinline__ void walk_sort(const pmx_painter &p, const BinGeom &g, const UnitCoords &u, int a,
                                          const DVec &pos, const DVec &mass, double mass_scalar,
                                          const uint32_t *list, int64_t j0, int nsub, WRec *rec, uint32_t *cnt,
                                          uint16_t *slot, uint16_t *xlist, uint16_t *xkey, uint32_t *nxc)
{
    for (int jj = threadIdx.x; jj < nsub; jj += WTHREADS) {
        const int64_t idx = (int64_t)list[j0 + jj];
        double X[3];
        int cc[3];
        bool ok = true;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            X[d] = ld_elem<PE>(pos, idx, d) * p.scale[d] + p.translate[d];
            ok = ok && (fabs(X[d]) < 1073741824.0);
            int i0w = 0;
            ok = ok && local_base<KIND>(p, d, Tuned<KIND>::first(ok ? X[d] : 0.0), &i0w);
            cc[d] = i0w + g.o[d];
        }
        const int b = cc[1] - u.B * P1, c = cc[2] - u.C * P2;
        ok = ok && cc[0] == a && (unsigned)b < (unsigned)P1 && (unsigned)c < (unsigned)P2;
        if (ok) {
            WRec r;
            r.x[0] = X[0]; r.x[1] = X[1]; r.x[2] = X[2];
            if (PAINT) r.m = mass.data ? mass.get(idx, 0) : mass_scalar;
            else r.m = (double)idx;
            rec[jj] = r;
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

template <int KIND, typename T, typename PE>
__global__ void __launch_bounds__(WTHREADS) paint_walk_kernel(pmx_painter p, BinGeom g, char *canvas, DVec pos,
                                                             DVec mass, double mass_scalar, const uint32_t *list,
                                                             const int64_t *offsets, const uint32_t *counts,
                                                             T *halo, uint32_t *unit_flags, int overwrite)
{
    using W = Walk<KIND>;
    constexpr int S = W::S, R2 = W::R2, PLANE = W::PLANE, NR = W::NR, RING = W::RING;
    __shared__ double ring[NR * PLANE];
    __shared__ WRec rec[NREC];
    __shared__ uint32_t cnt[WTHREADS];
    __shared__ uint16_t slot[WTHREADS * KOWN];
    __shared__ uint16_t xlist[NREC], xkey[NREC];
    __shared__ uint32_t nx[2];
    const int tid = threadIdx.x, tb = tid / P2, tc = tid % P2;
    const int64_t unit_halo = unit_halo_cells<KIND>(g.lseg);
    for (int64_t unit = blockIdx.x; unit < g.nunits; unit += gridDim.x) {
        const UnitCoords u = unit_coords(g, unit);
        const int Lu = u.a1 - u.a0;
        // nothing to add in this unit (uniform per workgroup): skip it, its staging stays unused
        const int any = __syncthreads_or(tid < Lu && counts[u.bucket0 + u.a0 + tid] != 0);
        if (!any && !overwrite) {
            if (tid == 0) unit_flags[unit] = 0;
            continue;
        }
        if (tid == 0) { unit_flags[unit] = 1; nx[0] = 0; nx[1] = 0; }
        for (int q = tid; q < NR * PLANE; q += WTHREADS) ring[q] = 0;
        cnt[tid] = 0;
        T *hbase = halo + unit * unit_halo;
        double acc[S][S][S];
#pragma unroll
        for (int i = 0; i < S; i++)
#pragma unroll
            for (int j = 0; j < S; j++)
#pragma unroll
                for (int k = 0; k < S; k++) acc[i][j][k] = 0;
        int phase = 0;
        __syncthreads();

        auto step = [&](auto rot, const int a) {
            constexpr int ROT = decltype(rot)::value;
            const uint32_t n = a < u.a1 ? counts[u.bucket0 + a] : 0u;
            const int64_t start = a < u.a1 ? offsets[u.bucket0 + a] : 0;
            const int sl0 = a % NR;
            for (uint32_t sub0 = 0; sub0 < n; sub0 += NREC) {
                const int nsub = (int)((n - sub0) < (uint32_t)NREC ? (n - sub0) : (uint32_t)NREC);
                uint32_t *nxc = &nx[phase & 1];
                walk_sort<KIND, PE, true>(p, g, u, a, pos, mass, mass_scalar, list, start + sub0, nsub, rec, cnt,
                                          slot, xlist, xkey, nxc);
                if (tid == 0) nx[(phase + 1) & 1] = 0;
                __syncthreads();
                // owner: the first KOWN particles of my cell go into the register window
                {
                    const uint32_t cn = cnt[tid];
                    cnt[tid] = 0;
                    const int nown = cn < (uint32_t)KOWN ? (int)cn : KOWN;
                    for (int r = 0; r < nown; r++) {
                        const WRec R = rec[slot[tid * KOWN + r]];
                        double V[3][S];
                        int I[S];
#pragma unroll
                        for (int d = 0; d < 3; d++) Tuned<KIND>::axis(R.x[d], p.order[d], p.scale[d], I, V[d]);
#pragma unroll
                        for (int i = 0; i < S; i++) V[0][i] *= R.m;
#pragma unroll
                        for (int i = 0; i < S; i++)
#pragma unroll
                            for (int j = 0; j < S; j++) {
                                const double fb = V[0][i] * V[1][j];
#pragma unroll
                                for (int k = 0; k < S; k++) acc[(i + ROT) % S][j][k] += fb * V[2][k];
                            }
                    }
                }
                // the rest of crowded cells: scatter form, spread over all threads
                {
                    const uint32_t nxv = *nxc;
                    for (uint32_t e = tid; e < nxv; e += WTHREADS) {
                        const WRec R = rec[xlist[e]];
                        const int key = xkey[e];
                        const int b = key / P2, c = key % P2;
                        double V[3][S];
                        int I[S];
#pragma unroll
                        for (int d = 0; d < 3; d++) Tuned<KIND>::axis(R.x[d], p.order[d], p.scale[d], I, V[d]);
#pragma unroll
                        for (int i = 0; i < S; i++) V[0][i] *= R.m;
#pragma unroll
                        for (int i = 0; i < S; i++) {
                            int sl = sl0 + i;
                            if (sl >= NR) sl -= NR;
#pragma unroll
                            for (int j = 0; j < S; j++) {
                                const double fb = V[0][i] * V[1][j];
                                const int rowoff = sl * PLANE + (b + j) * R2 + c;
#pragma unroll
                                for (int k = 0; k < S; k++) unsafeAtomicAdd(&ring[rowoff + k], fb * V[2][k]);
                            }
                        }
                    }
                }
                phase++;
                if (sub0 + NREC < n) __syncthreads();      // the next sub-step reuses the records
            }
            // the oldest plane of the window is complete for this thread: add it to ring plane a
#pragma unroll
            for (int j = 0; j < S; j++)
#pragma unroll
                for (int k = 0; k < S; k++) {
                    unsafeAtomicAdd(&ring[sl0 * PLANE + (tb + j) * R2 + tc + k], acc[ROT][j][k]);
                    acc[ROT][j][k] = 0;
                }
            __syncthreads();
            // plane a is complete: box -> canvas (plain row stores), ring -> staging
            {
                const int pa = a - u.a0;
                const bool trailing = a >= u.a1;
                const int l0 = a - g.o[0];
                const bool in0 = !trailing && l0 >= 0 && l0 < p.size[0];
                T *hplane = trailing ? hbase + (int64_t)Lu * RING + (int64_t)(pa - Lu) * PLANE : hbase + (int64_t)pa * RING;
                for (int q = tid; q < PLANE; q += WTHREADS) {
                    const int b = q / R2, c = q - b * R2;
                    const double v = ring[sl0 * PLANE + q];
                    ring[sl0 * PLANE + q] = 0;
                    if (trailing) hplane[q] = (T)v;
                    else if (b < P1 && c < P2) {
                        const int l1 = u.B * P1 - g.o[1] + b, l2 = u.C * P2 - g.o[2] + c;
                        if (in0 && l1 >= 0 && l1 < p.size[1] && l2 >= 0 && l2 < p.size[2]) {
                            T *dst = (T *)(canvas + l0 * p.strides[0] + l1 * p.strides[1] + l2 * p.strides[2]);
                            if (overwrite) *dst = (T)v;
                            else *dst += (T)v;
                        }
                    } else hplane[W::ring_index(b, c)] = (T)v;
                }
            }
        };

        const int aend = u.a1 + S - 1;
        for (int a = u.a0; a < aend; a += S) {
            step(std::integral_constant<int, 0>(), a);
            if (S > 1 && a + 1 < aend) step(std::integral_constant<int, 1 % S>(), a + 1);
            if (S > 2 && a + 2 < aend) step(std::integral_constant<int, 2 % S>(), a + 2);
            if (S > 3 && a + 3 < aend) step(std::integral_constant<int, 3 % S>(), a + 3);
        }
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

template <int KIND, typename T, typename PE>
__global__ void __launch_bounds__(WTHREADS) readout_walk_kernel(pmx_painter p, BinGeom g, const char *canvas, DVec pos,
                                                               DVec out, const uint32_t *list, const int64_t *offsets,
                                                               const uint32_t *counts)
{
    using W = Walk<KIND>;
    constexpr int S = W::S, R2 = W::R2, PLANE = W::PLANE, NR = W::NR;
    __shared__ T ring[NR * PLANE];
    __shared__ WRec rec[NREC];
    __shared__ uint32_t cnt[WTHREADS];
    __shared__ uint16_t slot[WTHREADS * KOWN];
    __shared__ uint16_t xlist[NREC], xkey[NREC];
    __shared__ uint32_t nx[2];
    const int tid = threadIdx.x, tb = tid / P2, tc = tid % P2;
    const DVec nomass = {nullptr, 0, 0, 8};
    for (int64_t unit = blockIdx.x; unit < g.nunits; unit += gridDim.x) {
        const UnitCoords u = unit_coords(g, unit);
        const int Lu = u.a1 - u.a0;
        const int any = __syncthreads_or(tid < Lu && counts[u.bucket0 + u.a0 + tid] != 0);
        if (!any) continue;
        if (tid == 0) { nx[0] = 0; nx[1] = 0; }
        cnt[tid] = 0;
        // stage plane `pl` (tile-space) of the patch + halo; cells outside the block read 0
        auto load_plane = [&](const int pl) {
            int sl = pl % NR;
            const int l0 = wrap_near(pl - g.o[0], p.period[0]);
            const bool in0 = l0 >= 0 && l0 < p.size[0];
            for (int q = tid; q < PLANE; q += WTHREADS) {
                const int b = q / R2, c = q - b * R2;
                const int l1 = wrap_near(u.B * P1 - g.o[1] + b, p.period[1]);
                const int l2 = wrap_near(u.C * P2 - g.o[2] + c, p.period[2]);
                const bool in = in0 && l1 >= 0 && l1 < p.size[1] && l2 >= 0 && l2 < p.size[2];
                ring[sl * PLANE + q] = in ? *(const T *)(canvas + l0 * p.strides[0] + l1 * p.strides[1] + l2 * p.strides[2]) : (T)0;
            }
        };
#pragma unroll
        for (int k = 0; k < S - 1; k++) load_plane(u.a0 + k);
        __syncthreads();
        T win[S][S][S];
#pragma unroll
        for (int i = 0; i < S; i++)
#pragma unroll
            for (int j = 0; j < S; j++)
#pragma unroll
                for (int k = 0; k < S; k++)
                    win[i][j][k] = i < S - 1 ? ring[((u.a0 + i) % NR) * PLANE + (tb + j) * R2 + tc + k] : (T)0;
        int phase = 0;

        auto step = [&](auto rot, const int a) {
            constexpr int ROT = decltype(rot)::value;
            const uint32_t n = counts[u.bucket0 + a];
            const int64_t start = offsets[u.bucket0 + a];
            const int sl0 = a % NR;
            load_plane(a + S - 1);
            bool fresh = true;      // the window has not taken plane a+S-1 yet
            uint32_t sub0 = 0;
            do {
                const int nsub = (int)((n - sub0) < (uint32_t)NREC ? (n - sub0) : (uint32_t)NREC);
                uint32_t *nxc = &nx[phase & 1];
                walk_sort<KIND, PE, false>(p, g, u, a, pos, nomass, 0.0, list, start + sub0, nsub, rec, cnt, slot,
                                           xlist, xkey, nxc);
                if (tid == 0) nx[(phase + 1) & 1] = 0;
                __syncthreads();
                if (fresh) {
                    int sl = sl0 + S - 1;
                    if (sl >= NR) sl -= NR;
#pragma unroll
                    for (int j = 0; j < S; j++)
#pragma unroll
                        for (int k = 0; k < S; k++) win[(S - 1 + ROT) % S][j][k] = ring[sl * PLANE + (tb + j) * R2 + tc + k];
                    fresh = false;
                }
                if (nsub > 0) {
                    const uint32_t cn = cnt[tid];
                    cnt[tid] = 0;
                    const int nown = cn < (uint32_t)KOWN ? (int)cn : KOWN;
                    for (int r = 0; r < nown; r++) {
                        const WRec R = rec[slot[tid * KOWN + r]];
                        double V[3][S];
                        int I[S];
#pragma unroll
                        for (int d = 0; d < 3; d++) Tuned<KIND>::axis(R.x[d], p.order[d], p.scale[d], I, V[d]);
                        double value = 0;
#pragma unroll
                        for (int i = 0; i < S; i++)
#pragma unroll
                            for (int j = 0; j < S; j++) {
                                const double fb = V[0][i] * V[1][j];
#pragma unroll
                                for (int k = 0; k < S; k++) value += (double)win[(i + ROT) % S][j][k] * (fb * V[2][k]);
                            }
                        out.set((int64_t)R.m, 0, value);
                    }
                    const uint32_t nxv = *nxc;
                    for (uint32_t e = tid; e < nxv; e += WTHREADS) {
                        const WRec R = rec[xlist[e]];
                        const int key = xkey[e];
                        const int b = key / P2, c = key % P2;
                        double V[3][S];
                        int I[S];
#pragma unroll
                        for (int d = 0; d < 3; d++) Tuned<KIND>::axis(R.x[d], p.order[d], p.scale[d], I, V[d]);
                        double value = 0;
#pragma unroll
                        for (int i = 0; i < S; i++) {
                            int sl = sl0 + i;
                            if (sl >= NR) sl -= NR;
#pragma unroll
                            for (int j = 0; j < S; j++) {
                                const double fb = V[0][i] * V[1][j];
                                const int rowoff = sl * PLANE + (b + j) * R2 + c;
#pragma unroll
                                for (int k = 0; k < S; k++) value += (double)ring[rowoff + k] * (fb * V[2][k]);
                            }
                        }
                        out.set((int64_t)R.m, 0, value);
                    }
                    phase++;
                    __syncthreads();       // the records are free for the next step
                }
                sub0 += NREC;
            } while (sub0 < n);
        };

        for (int a = u.a0; a < u.a1; a += S) {
            step(std::integral_constant<int, 0>(), a);
            if (S > 1 && a + 1 < u.a1) step(std::integral_constant<int, 1 % S>(), a + 1);
            if (S > 2 && a + 2 < u.a1) step(std::integral_constant<int, 2 % S>(), a + 2);
            if (S > 3 && a + 3 < u.a1) step(std::integral_constant<int, 3 % S>(), a + 3);
        }
        __syncthreads();
    }
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
    T *halo = (T *)pl->halo;
    if (pos.elsize == 8)
        paint_walk_kernel<KIND, T, double><<<grid, WTHREADS, 0, st>>>(p, g, (char *)canvas, pos, mass, ms, pl->list,
                                                                     pl->offsets, pl->counts, halo, pl->unit_flags, overwrite);
    else
        paint_walk_kernel<KIND, T, float><<<grid, WTHREADS, 0, st>>>(p, g, (char *)canvas, pos, mass, ms, pl->list,
                                                                    pl->offsets, pl->counts, halo, pl->unit_flags, overwrite);
    halo_merge_walk_kernel<KIND, T><<<grid, TBLOCK, 0, st>>>(p, g, (char *)canvas, halo, pl->unit_flags);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

int paint_walk(pmx_binplan *pl, const pmx_painter &p, void *canvas, DVec pos, DVec mass, double ms, int overwrite,
               hipStream_t st)
{
    const bool f8 = p.canvas_elsize == 8;
    switch (p.kind) {
    case PMX_TUNED_CIC:
        return f8 ? paint_walk_t<PMX_TUNED_CIC, double>(pl, p, canvas, pos, mass, ms, overwrite, st)
                  : paint_walk_t<PMX_TUNED_CIC, float>(pl, p, canvas, pos, mass, ms, overwrite, st);
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

template <int KIND, typename T>
static int readout_walk_t(pmx_binplan *pl, const pmx_painter &p, const void *canvas, DVec pos, DVec out, hipStream_t st)
{
    const BinGeom &g = pl->g;
    const unsigned grid = (unsigned)(g.nunits < 65535 * 8 ? g.nunits : 65535 * 8);
    if (pos.elsize == 8)
        readout_walk_kernel<KIND, T, double><<<grid, WTHREADS, 0, st>>>(p, g, (const char *)canvas, pos, out, pl->list,
                                                                       pl->offsets, pl->counts);
    else
        readout_walk_kernel<KIND, T, float><<<grid, WTHREADS, 0, st>>>(p, g, (const char *)canvas, pos, out, pl->list,
                                                                      pl->offsets, pl->counts);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

int readout_walk(pmx_binplan *pl, const pmx_painter &p, const void *canvas, DVec pos, DVec out, hipStream_t st)
{
    const bool f8 = p.canvas_elsize == 8;
    switch (p.kind) {
    case PMX_TUNED_CIC:
        return f8 ? readout_walk_t<PMX_TUNED_CIC, double>(pl, p, canvas, pos, out, st)
                  : readout_walk_t<PMX_TUNED_CIC, float>(pl, p, canvas, pos, out, st);
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
