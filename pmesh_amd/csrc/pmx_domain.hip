// pmx_domain.hip — domain decomposition and particle exchange packing.
//
// Replaces pmesh/domain.py:605-646 (the numpy classification in chunks of
// 49152 rows and the two serial passes of the Cython gridnd_fill,
// pmesh/_domain.pyx:9-122), Layout.exchange's `take` (domain.py:188) and
// Layout.gather's bincountv (domain.py:26-48, 294-295).
//
// Structure on the device: one thread per particle classifies it (target-rank
// bit mask, <= 64 ranks) and a block-level ballot/popcount gives per-block,
// per-rank counts; a tiny scan over blocks gives every block its base offset
// per rank; the fill pass recomputes the intra-block rank of each particle
// with the same ballots, which yields the reference's order: rank-major,
// ascending particle index (stable).
#include <hip/hip_runtime.h>
#include <math.h>

#include "pmx_common.h"

namespace pmx {

constexpr int DBLOCK = 256;
constexpr int DSUB = 8;                       // sub-batches of DBLOCK particles per chunk
constexpr int DCHUNK = DBLOCK * DSUB;         // particles per chunk of the stable multi-split

struct GridD {
    int32_t ndim, periodic, nranks;
    int32_t shape[PMX_MAXDIM];
    const double *edges[PMX_MAXDIM];
    const int32_t *assign;
    const int16_t *degenerate;
};

// numpy float remainder (npy_divmod): sign of the divisor
__device__ inline double np_remainder(double a, double b)
{
    double mod = fmod(a, b);
    if (!b) return mod;
    if (mod) {
        if ((b < 0) != (mod < 0)) mod += b;
    } else {
        mod = copysign(0.0, b);
    }
    return mod;
}

// The same value without the (slow, software) fmod for arguments within one period of
// [0, b): fmod(a, b) is a itself for |a| < b and a - b, exactly (Sterbenz), for b <= a < 2b.
__device__ inline double np_remainder_near(double a, double b)
{
    if (b > 0) {
        if (a >= 0) {
            if (a < b) return a + 0.0;            // -0.0 -> +0.0 as copysign(0, b) does
            if (a < b + b) return a - b;
        } else if (a >= -b) {
            return a == -b ? 0.0 : a + b;         // fmod = a (negative): one rounding in a + b
        }
    }
    return np_remainder(a, b);
}

// numpy.digitize(x, bins, right=False) == searchsorted(bins, x, 'right')
__device__ inline int np_digitize(double x, const double *bins, int n)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (x < bins[mid]) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

__device__ inline int py_mod(int a, int n)
{
    // arguments are within a period or two of [0, n) almost always: no integer division then
    if (a >= 0) {
        if (a < n) return a;
        if (a < n + n) return a - n;
    } else if (a >= -n) {
        return a + n;
    }
    int r = a % n;
    return r < 0 ? r + n : r;
}

// domain.py:608-630, one axis; int16 truncation as the reference's 'i2' arrays
__device__ inline void classify_axis(const GridD &g, int j, double x, double s, int *sil, int *sir, double invw)
{
    const double *edges = g.edges[j];
    int ne = g.shape[j] + 1;
    int l, r;
    if (g.periodic && g.shape[j] == 1) {
        // an axis that is not split (two of three on a slab decomposition): every position,
        // finite or not, resolves to the one domain whatever the smoothing — the general branch
        // gives (0, 1) for values inside the period and (1, 2) otherwise, both of which wrap to
        // domain 0 with a patch of one
        *sil = 0;
        *sir = 1;
        return;
    }
    if (g.periodic) {
        double box = edges[ne - 1];
        if (x >= 0 && x < box) {
            // Fast path, exact: a coordinate inside the box whose smoothing interval [x - s, x + s] stays inside ONE
            // domain [e_k, e_k+1).  Then remainder() is the identity on x, x - s and x + s, all three digitize to
            // k + 1, and the formulas below give (l, r) = (k, k + 1) — the patch of one domain that all but the
            // particles within s of a domain face have.  k comes from a guess (uniform edges: exact almost always)
            // corrected against the edges themselves, i.e. it IS digitize(x) - 1; one or two LDS reads instead of
            // three binary searches and three remainders (classify_kernel: 320 -> ... us per 1.7e7 particles).
            int k = (int)(x * invw);             // invw = shape / box, formed once per thread
            k = max(0, min(k, g.shape[j] - 1));
            while (k > 0 && x < edges[k]) k--;
            while (k < g.shape[j] - 1 && x >= edges[k + 1]) k++;
            if (x - s >= edges[k] && x + s < edges[k + 1]) {
                *sil = (int)(int16_t)k;
                *sir = (int)(int16_t)(k + 1);
                return;
            }
        }
        double c = np_remainder_near(x, box);
        l = np_digitize(np_remainder_near(c - s, box), edges, ne);
        r = np_digitize(np_remainder_near(c + s, box), edges, ne);
        int p = np_digitize(c, edges, ne);
        l = p - py_mod(p - l, g.shape[j]) - 1;
        r = p + py_mod(r - p, g.shape[j]);
    } else {
        l = np_digitize(x - s, edges, ne) - 1;
        r = np_digitize(x + s, edges, ne);
        l = max(0, min(l, g.shape[j]));
        r = max(0, min(r, g.shape[j]));
    }
    *sil = (int)(int16_t)l;
    *sir = (int)(int16_t)r;
}

// gridnd_fill's patch enumeration (_domain.pyx:62-118) -> unique targets as a mask
// ND > 0: the number of dimensions at compile time (every loop over the axes unrolled, the per-axis arrays in
// registers: with a run-time count they are indexed dynamically — scratch or a chain of selects per access)
template <int ND = 0>
__device__ inline uint64_t particle_targets(const GridD &g, const int *sil, const int *sir)
{
    const int nd = ND > 0 ? ND : g.ndim;
    constexpr int NB = ND > 0 ? ND : PMX_MAXDIM;      // loop bounds the compiler can unroll; the axes beyond nd are skipped
    int strides[PMX_MAXDIM];
    strides[nd - 1] = 1;
#pragma unroll
    for (int j = NB - 2; j >= 0; j--) if (j <= nd - 2) strides[j] = strides[j + 1] * g.shape[j + 1];
    int64_t patch = 1;
    int p[PMX_MAXDIM];
#pragma unroll
    for (int j = 0; j < NB; j++) if (j < nd) {
        patch *= sir[j] - sil[j];
        p[j] = sil[j];
    }
    uint64_t mask = 0;
    if (patch == 1) {
        // the common case: the particle and its smoothing region lie in one domain
        int64_t target = 0;
#pragma unroll
        for (int j = 0; j < NB; j++) if (j < nd) {
            int t = p[j];
            if (g.periodic) t = py_mod(t, g.shape[j]);
            target += (int64_t)t * strides[j];
        }
        target = g.assign[target];
        return g.degenerate[target] ? 0 : (uint64_t)1 << target;
    }
    for (int64_t q = 0; q < patch; q++) {
        int64_t target = 0;
#pragma unroll
        for (int j = 0; j < NB; j++) if (j < nd) {
            int t = p[j];
            if (g.periodic) t = py_mod(t, g.shape[j]);
            target += (int64_t)t * strides[j];
        }
        target = g.assign[target];
        // quirk Q3: DomainDegenerate is indexed by the rank after the lookup
        if (!g.degenerate[target]) mask |= (uint64_t)1 << target;
        p[nd - 1]++;
        bool carry = true;
#pragma unroll
        for (int jj = NB - 1; jj > 0; jj--)
            if (jj <= nd - 1 && carry) {
                if (p[jj] == sir[jj]) { p[jj] = sil[jj]; p[jj - 1]++; }
                else carry = false;
            }
    }
    return mask;
}

struct F3 { double v[PMX_MAXDIM]; };

template <int ND>
__global__ void __launch_bounds__(DBLOCK) classify_kernel(GridD g, DVec pos, F3 scale, F3 smoothing,
                                                          int64_t n, uint64_t *masks,
                                                          unsigned long long *counts, int64_t nchunks,
                                                          int64_t *chunk_counts)
{
    const int nd = ND > 0 ? ND : g.ndim;
    // a block walks whole chunks of DCHUNK consecutive particles: the per-chunk, per-rank
    // counts that the fill pass needs fall out of the classification (no second pass over the masks)
    __shared__ unsigned int lcount[PMX_MAXRANKS];      // this chunk
    __shared__ unsigned int tcount[PMX_MAXRANKS];      // all chunks of this block
    // the grid description is read through dependent loads (binary searches, the domain -> rank
    // table): keep it in LDS when it is small (it always is for the ParticleMesh domains)
    constexpr int MAXE = 80, MAXCELLS = 256;
    __shared__ double s_edges[PMX_MAXDIM][MAXE];
    __shared__ int32_t s_assign[MAXCELLS];
    __shared__ int16_t s_degenerate[PMX_MAXRANKS];
    // domains per unit length along every axis (the guess of classify_axis' fast path), from the caller's tables
    double invw[PMX_MAXDIM];
    for (int j = 0; j < PMX_MAXDIM; j++) invw[j] = j < g.ndim ? (double)g.shape[j] / g.edges[j][g.shape[j]] : 0.0;
    {
        int cells = 1;
        bool fits = true;
        for (int j = 0; j < g.ndim; j++) {
            cells *= g.shape[j];
            fits = fits && g.shape[j] + 1 <= MAXE;
        }
        fits = fits && cells <= MAXCELLS;
        if (fits) {
            for (int j = 0; j < g.ndim; j++)
                for (int q = threadIdx.x; q <= g.shape[j]; q += DBLOCK) s_edges[j][q] = g.edges[j][q];
            for (int q = threadIdx.x; q < cells; q += DBLOCK) s_assign[q] = g.assign[q];
            for (int q = threadIdx.x; q < g.nranks; q += DBLOCK) s_degenerate[q] = g.degenerate[q];
            for (int j = 0; j < g.ndim; j++) g.edges[j] = s_edges[j];
            g.assign = s_assign;
            g.degenerate = s_degenerate;
        }
    }
    if (threadIdx.x < PMX_MAXRANKS) tcount[threadIdx.x] = 0;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (threadIdx.x < PMX_MAXRANKS) lcount[threadIdx.x] = 0;
        __syncthreads();
        // ([r4] measured: all DSUB x 3 loads of a chunk requested before the first row is classified — 48 more live
        // registers — took the kernel from 2.39 to 2.93 ms per 1.3e8 rows: it is bound by its instructions, not by
        // the latency of its loads)
        for (int k = 0; k < DSUB; k++) {
            int64_t i = chunk * DCHUNK + k * DBLOCK + threadIdx.x;
            uint64_t m = 0;
            if (i < n) {
                int sil[PMX_MAXDIM], sir[PMX_MAXDIM];
#pragma unroll
                for (int j = 0; j < (ND > 0 ? ND : PMX_MAXDIM); j++) if (j < nd) {
                    // transform0 (pm.py:1788-1790): scale * x in double
                    double x = scale.v[j] * pos.get(i, j);
                    classify_axis(g, j, x, smoothing.v[j], &sil[j], &sir[j], invw[j]);
                }
                m = particle_targets<ND>(g, sil, sir);
                masks[i] = m;
            }
            // (rows in a coherent order: the 64 particles of a wave are bound for ONE rank and nothing else — one
            // comparison and one add instead of a ballot and an add per rank)
            const uint64_t m0 = __shfl(m, 0);
            if (__ballot(m != m0) == 0 && (m0 & (m0 - 1)) == 0) {
                if ((threadIdx.x & 63) == 0 && m0) atomicAdd(&lcount[__ffsll((long long)m0) - 1], 64u);
            } else {
                for (int r = 0; r < g.nranks; r++) {
                    unsigned long long b = __ballot((m >> r) & 1);
                    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&lcount[r], (unsigned)__popcll(b));
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < g.nranks) {
            chunk_counts[(int64_t)threadIdx.x * nchunks + chunk] = lcount[threadIdx.x];
            tcount[threadIdx.x] += lcount[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x < g.nranks && tcount[threadIdx.x])
        atomicAdd(&counts[threadIdx.x], (unsigned long long)tcount[threadIdx.x]);
}

// [r5] The lean form for what ParticleMesh.decompose sends: dense rows of 3 floats or doubles, a periodic grid whose
// tables fit the LDS.  On P ranks every step of a time-stepping caller classifies its particles again
// (examples/nbody.py:199-204) and classify_kernel was the largest kernel of such a cycle (238 us per 1.7e7 rows, 14
// ps per row, against 108 us for the bin pass over the same rows): its loop carries the general patch enumeration, the
// three remainders and binary searches of the slow path and the strided element loads in every trip.  Here a lane loads
// its row in one piece, only the axes that are split are looked at, and the fast path — the coordinate inside the box,
// its smoothing interval inside ONE domain: all but the particles within s of a domain face — is a guess, at most a
// correction against the edges, and two table reads; any other row takes the general functions as before (whole waves
// do, for rows in a coherent order).  Same masks, same counts (tests/test_domain.py).
template <int PE>
__global__ void __launch_bounds__(DBLOCK) classify_lean_kernel(GridD g, const char *data, F3 scale, F3 smoothing,
                                                               int64_t n, uint64_t *masks,
                                                               unsigned long long *counts, int64_t nchunks,
                                                               int64_t *chunk_counts)
{
    __shared__ unsigned int lcount[PMX_MAXRANKS];
    __shared__ unsigned int tcount[PMX_MAXRANKS];
    constexpr int MAXE = 80, MAXCELLS = 256;
    __shared__ double s_edges[3][MAXE];
    __shared__ int32_t s_assign[MAXCELLS];
    __shared__ int16_t s_degenerate[PMX_MAXRANKS];
    double invw[3], box[3];
    int stride[3];
    {
        int cells = 1;
        for (int j = 2; j >= 0; j--) { stride[j] = cells; cells *= g.shape[j]; }
        for (int j = 0; j < 3; j++) {
            box[j] = g.edges[j][g.shape[j]];
            invw[j] = (double)g.shape[j] / box[j];
            for (int q = threadIdx.x; q <= g.shape[j]; q += DBLOCK) s_edges[j][q] = g.edges[j][q];
        }
        for (int q = threadIdx.x; q < cells; q += DBLOCK) s_assign[q] = g.assign[q];
        for (int q = threadIdx.x; q < g.nranks; q += DBLOCK) s_degenerate[q] = g.degenerate[q];
        for (int j = 0; j < 3; j++) g.edges[j] = s_edges[j];
        g.assign = s_assign;
        g.degenerate = s_degenerate;
    }
    if (threadIdx.x < PMX_MAXRANKS) tcount[threadIdx.x] = 0;
    constexpr int ROW = 3 * PE;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (threadIdx.x < PMX_MAXRANKS) lcount[threadIdx.x] = 0;
        __syncthreads();
        // (the next sub-batch's row is in flight while this one is classified)
        auto load = [&](int k, double *x) __attribute__((always_inline)) {
            const int64_t i = chunk * DCHUNK + k * DBLOCK + threadIdx.x;
            const int64_t il = i < n ? i : n - 1;                   // (a row beyond the end reads the last one: no load behind a branch)
            if (PE == 4) { const float *q = (const float *)(data + il * ROW); x[0] = q[0]; x[1] = q[1]; x[2] = q[2]; }
            else { const double *q = (const double *)(data + il * ROW); x[0] = q[0]; x[1] = q[1]; x[2] = q[2]; }
        };
        double xn[3];
        load(0, xn);
#pragma unroll 2
        for (int k = 0; k < DSUB; k++) {
            const int64_t i = chunk * DCHUNK + k * DBLOCK + threadIdx.x;
            double x[3] = {xn[0], xn[1], xn[2]};
            if (k + 1 < DSUB) load(k + 1, xn);
            bool fast = true;
            int target = 0;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (g.shape[j] == 1) continue;                      // (uniform: an axis that is not split resolves to its one domain)
                const double X = scale.v[j] * x[j];                 // transform0 (pm.py:1788-1790): scale * x in double
                const double s = smoothing.v[j];
                int kk = (int)(X * invw[j]);
                kk = max(0, min(kk, g.shape[j] - 1));
                while (kk > 0 && X < s_edges[j][kk]) kk--;
                while (kk < g.shape[j] - 1 && X >= s_edges[j][kk + 1]) kk++;
                fast = fast && X >= 0 && X < box[j] && X - s >= s_edges[j][kk] && X + s < s_edges[j][kk + 1];
                target += kk * stride[j];
                x[j] = X;
            }
            uint64_t m = 0;
            if (i < n) {
                if (fast) {
                    const int rank = s_assign[target];
                    m = s_degenerate[rank] ? 0 : (uint64_t)1 << rank;
                } else {
                    int sil[PMX_MAXDIM], sir[PMX_MAXDIM];
#pragma unroll
                    for (int j = 0; j < 3; j++)
                        classify_axis(g, j, g.shape[j] == 1 ? x[j] : x[j], smoothing.v[j], &sil[j], &sir[j], invw[j]);
                    m = particle_targets<3>(g, sil, sir);
                }
                masks[i] = m;
            }
            const uint64_t m0 = __shfl(m, 0);
            if (__ballot(m != m0) == 0 && (m0 & (m0 - 1)) == 0) {
                if ((threadIdx.x & 63) == 0 && m0) atomicAdd(&lcount[__ffsll((long long)m0) - 1], 64u);
            } else {
                for (int r = 0; r < g.nranks; r++) {
                    unsigned long long b = __ballot((m >> r) & 1);
                    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&lcount[r], (unsigned)__popcll(b));
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < g.nranks) {
            chunk_counts[(int64_t)threadIdx.x * nchunks + chunk] = lcount[threadIdx.x];
            tcount[threadIdx.x] += lcount[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x < g.nranks && tcount[threadIdx.x])
        atomicAdd(&counts[threadIdx.x], (unsigned long long)tcount[threadIdx.x]);
}

// per-chunk (DCHUNK particles), per-rank counts from the masks alone (when the classification
// was not run by this library instance just before, see pmx_decompose_fill)
__global__ void __launch_bounds__(DBLOCK) chunk_count_kernel(const uint64_t *masks, int64_t n,
                                                             int nranks, int64_t nchunks,
                                                             int64_t *chunk_counts)
{
    __shared__ unsigned int lcount[PMX_MAXRANKS];
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (threadIdx.x < PMX_MAXRANKS) lcount[threadIdx.x] = 0;
        __syncthreads();
        for (int k = 0; k < DSUB; k++) {
            int64_t i = chunk * DCHUNK + k * DBLOCK + threadIdx.x;
            uint64_t m = i < n ? masks[i] : 0;
            for (int r = 0; r < nranks; r++) {
                unsigned long long b = __ballot((m >> r) & 1);
                if ((threadIdx.x & 63) == 0 && b) atomicAdd(&lcount[r], (unsigned)__popcll(b));
            }
        }
        __syncthreads();
        if (threadIdx.x < nranks) chunk_counts[(int64_t)threadIdx.x * nchunks + chunk] = lcount[threadIdx.x];
        __syncthreads();
    }
}

// exclusive scan along chunks for each rank, seeded with the rank's offset.
// One block of 1024 threads per rank; sequential over tiles of 1024 chunks.
constexpr int DSCAN = 1024;
__global__ void __launch_bounds__(DSCAN) chunk_scan_kernel(int64_t *chunk_counts, int64_t nchunks,
                                                           const int64_t *offsets)
{
    __shared__ int64_t sh[DSCAN];
    __shared__ int64_t carry;
    int r = blockIdx.x;
    int64_t *row = chunk_counts + (int64_t)r * nchunks;
    if (threadIdx.x == 0) carry = offsets[r];
    __syncthreads();
    for (int64_t base = 0; base < nchunks; base += DSCAN) {
        int64_t i = base + threadIdx.x;
        int64_t v = i < nchunks ? row[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < DSCAN; off <<= 1) {
            int64_t t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        int64_t incl = sh[threadIdx.x];
        if (i < nchunks) row[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == DSCAN - 1) carry += incl;
        __syncthreads();
    }
}

template <typename IDX>
__global__ void __launch_bounds__(DBLOCK) fill_kernel(const uint64_t *masks, int64_t n, int nranks,
                                                      int64_t nchunks, const int64_t *chunk_base,
                                                      IDX *indices)
{
    // per sub-batch of DBLOCK particles: every wave leaves its per-rank populations in LDS (one ballot per rank),
    // one barrier, then every lane finds its slot for each rank it is bound for — its rank inside the wave from
    // the same ballot again, the waves in front of it from the table — and the per-rank cursors advance:
    // three barriers per sub-batch, where the first version held three per rank and sub-batch
    __shared__ unsigned int wcount[DBLOCK / 64][PMX_MAXRANKS];
    __shared__ int64_t rbase[PMX_MAXRANKS];            // next free slot per rank within this chunk
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long below = ((unsigned long long)1 << lane) - 1;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        __syncthreads();
        if (threadIdx.x < nranks) rbase[threadIdx.x] = chunk_base[(int64_t)threadIdx.x * nchunks + chunk];
        for (int k = 0; k < DSUB; k++) {
            int64_t i = chunk * DCHUNK + k * DBLOCK + threadIdx.x;
            uint64_t m = i < n ? masks[i] : 0;
            // (a wave whose particles all stay on one rank — the rule in lattice order — asks one ballot that matters)
            const unsigned long long any = __ballot(m != 0);
            // (the 64 particles of the wave bound for ONE rank and nothing else — the rule in a coherent order: no ballots)
            const uint64_t m0 = __shfl(m, 0);
            const bool uni = any && __ballot(m != m0) == 0 && (m0 & (m0 - 1)) == 0;
            const int r0 = uni ? __ffsll((long long)m0) - 1 : 0;
            if (uni) {
                if (lane < nranks) wcount[wave][lane] = lane == r0 ? 64u : 0u;
            } else {
                for (int r = 0; r < nranks; r++) {
                    unsigned long long b = any ? __ballot((m >> r) & 1) : 0ull;
                    if (lane == 0) wcount[wave][r] = (unsigned)__popcll(b);
                }
            }
            __syncthreads();
            if (uni) {
                unsigned before = (unsigned)lane;
                for (int w = 0; w < wave; w++) before += wcount[w][r0];
                indices[rbase[r0] + before] = (IDX)i;
            } else if (any) {
                for (int r = 0; r < nranks; r++) {
                    const bool hit = (m >> r) & 1;
                    unsigned long long b = __ballot(hit);
                    if (hit) {
                        unsigned before = (unsigned)__popcll(b & below);
                        for (int w = 0; w < wave; w++) before += wcount[w][r];
                        indices[rbase[r] + before] = (IDX)i;
                    }
                }
            }
            __syncthreads();
            if (threadIdx.x < nranks) {
                unsigned tot = 0;
                for (int w = 0; w < DBLOCK / 64; w++) tot += wcount[w][threadIdx.x];
                rbase[threadIdx.x] += tot;
            }
            __syncthreads();      // wcount / rbase are rewritten by the next round
        }
    }
}

template <typename IDX>
__global__ void __launch_bounds__(256) take_rows_kernel(const char *src, int64_t src_stride0,
                                                        int64_t row_bytes, const IDX *indices,
                                                        int64_t nrows, char *dst)
{
    // one thread per 4-byte word of the packed output (rows are f4/f8 columns)
    int64_t words = row_bytes >> 2;
    int64_t total = nrows * words;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t j = t / words, w = t - j * words;
        int64_t i = (int64_t)indices[j];
        *(uint32_t *)(dst + j * row_bytes + 4 * w) = *(const uint32_t *)(src + i * src_stride0 + 4 * w);
    }
}

// the same with the output rows `dst_stride` bytes apart (one column of a row that packs several arrays side
// by side: Layout.exchange(pack=True) without the concatenation of separately gathered columns);
// indices == NULL: row j of the source (the inverse: a column out of packed rows)
template <typename IDX>
__global__ void __launch_bounds__(256) pack_rows_kernel(const char *src, int64_t src_stride0, int64_t row_bytes,
                                                        const IDX *indices, int64_t nrows, char *dst,
                                                        int64_t dst_stride)
{
    int64_t words = row_bytes >> 2;
    int64_t total = nrows * words;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t j = t / words, w = t - j * words;
        int64_t i = indices ? (int64_t)indices[j] : j;
        *(uint32_t *)(dst + j * dst_stride + 4 * w) = *(const uint32_t *)(src + i * src_stride0 + 4 * w);
    }
}

template <typename T, typename IDX>
__global__ void __launch_bounds__(256) scatter_add_kernel(const T *values, int ncol,
                                                          const IDX *indices, int64_t nrows,
                                                          T *out)
{
    int64_t total = nrows * ncol;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t j = t / ncol, c = t - j * ncol;
        unsafeAtomicAdd(&out[(int64_t)indices[j] * ncol + c], values[t]);
    }
}

// per-library scratch for the chunk tables (grown on demand; never shrinks)
struct Scratch {
    void *ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return PMX_OK;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
        PMX_HIP_CHECK(hipMalloc(&ptr, need));
        bytes = need;
        return PMX_OK;
    }
};
static thread_local Scratch g_scratch;
// what the chunk table in g_scratch describes: pmx_decompose_count leaves the per-chunk
// counts of its classification there and pmx_decompose_fill picks them up if it is called
// next, on the same thread, for the same masks (what domain.py does); anything else recounts
struct ChunkTag { const void *masks = nullptr; int64_t npart = -1; int nranks = 0; };
static thread_local ChunkTag g_chunk_tag;

}  // namespace pmx

using namespace pmx;

extern "C" int pmx_decompose_count(const pmx_grid *g, const pmx_vec *pos, const double *scale,
                                   const double *smoothing, int64_t npart, uint64_t *masks,
                                   int64_t *counts, void *stream)
{
    PMX_REQUIRE(g && g->ndim >= 1 && g->ndim <= PMX_MAXDIM, PMX_EINVAL, "bad grid");
    PMX_REQUIRE(g->nranks >= 1 && g->nranks <= PMX_MAXRANKS, PMX_EUNSUPPORTED,
                "more than 64 ranks are not supported");
    PMX_REQUIRE(counts != nullptr, PMX_EINVAL, "counts is NULL");
    hipStream_t st = (hipStream_t)stream;
    PMX_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int64_t) * g->nranks, st));
    if (npart == 0) return PMX_OK;
    PMX_REQUIRE(vec_ok(pos) && pos->ncol >= g->ndim, PMX_EINVAL, "pos must be (n, >=ndim) f4/f8");
    PMX_REQUIRE(masks != nullptr, PMX_EINVAL, "masks is NULL");
    GridD gd;
    gd.ndim = g->ndim;
    gd.periodic = g->periodic;
    gd.nranks = g->nranks;
    F3 sc, sm;
    for (int d = 0; d < PMX_MAXDIM; d++) {
        gd.shape[d] = d < g->ndim ? g->shape[d] : 1;
        gd.edges[d] = d < g->ndim ? g->edges[d] : nullptr;
        sc.v[d] = d < g->ndim ? scale[d] : 1.0;
        sm.v[d] = d < g->ndim ? smoothing[d] : 0.0;
    }
    gd.assign = g->assign;
    gd.degenerate = g->degenerate;
    int64_t nchunks = (npart + DCHUNK - 1) / DCHUNK;
    g_chunk_tag = ChunkTag();
    int rc = g_scratch.ensure(sizeof(int64_t) * nchunks * g->nranks);
    if (rc) return rc;
    unsigned grid = (unsigned)(nchunks < 256 * 16 ? nchunks : 256 * 16);
#ifndef PMX_LEAN_CLASSIFY
#define PMX_LEAN_CLASSIFY 1
#endif
    // the lean form: 3-d, periodic, tables that fit its LDS copies, dense rows of 3 elements
    bool lean = PMX_LEAN_CLASSIFY && gd.ndim == 3 && gd.periodic && pos->ncol == 3 && pos->stride1 == pos->elsize &&
                pos->stride0 == 3 * (int64_t)pos->elsize && (int64_t)gd.shape[0] * gd.shape[1] * gd.shape[2] <= 256;
    for (int d = 0; d < 3 && lean; d++) lean = gd.shape[d] + 1 <= 80;
    if (lean) {
        if (pos->elsize == 8)
            classify_lean_kernel<8><<<grid, DBLOCK, 0, st>>>(gd, (const char *)pos->data, sc, sm, npart, masks,
                                                             (unsigned long long *)counts, nchunks, (int64_t *)g_scratch.ptr);
        else
            classify_lean_kernel<4><<<grid, DBLOCK, 0, st>>>(gd, (const char *)pos->data, sc, sm, npart, masks,
                                                             (unsigned long long *)counts, nchunks, (int64_t *)g_scratch.ptr);
    } else {
        auto classify = gd.ndim == 3 ? classify_kernel<3> : (gd.ndim == 2 ? classify_kernel<2> : classify_kernel<1>);      // (PMX_MAXDIM = 3)
        classify<<<grid, DBLOCK, 0, st>>>(gd, dvec(pos), sc, sm, npart, masks, (unsigned long long *)counts,
                                          nchunks, (int64_t *)g_scratch.ptr);
    }
    PMX_HIP_CHECK(hipGetLastError());
    g_chunk_tag.masks = masks;
    g_chunk_tag.npart = npart;
    g_chunk_tag.nranks = g->nranks;
    return PMX_OK;
}

extern "C" int pmx_decompose_fill(int32_t nranks, const uint64_t *masks, int64_t npart,
                                  const int64_t *offsets, void *indices, int32_t index_elsize,
                                  void *stream)
{
    PMX_REQUIRE(nranks >= 1 && nranks <= PMX_MAXRANKS, PMX_EUNSUPPORTED, "more than 64 ranks");
    PMX_REQUIRE(index_elsize == 4 || index_elsize == 8, PMX_EINVAL, "index_elsize must be 4 or 8");
    if (npart == 0) return PMX_OK;
    hipStream_t st = (hipStream_t)stream;
    int64_t nchunks = (npart + DCHUNK - 1) / DCHUNK;
    const bool have = g_chunk_tag.masks == masks && g_chunk_tag.npart == npart && g_chunk_tag.nranks == nranks;
    int rc = have ? PMX_OK : g_scratch.ensure(sizeof(int64_t) * nchunks * nranks);
    if (rc) return rc;
    int64_t *cc = (int64_t *)g_scratch.ptr;
    unsigned grid = (unsigned)(nchunks < 256 * 16 ? nchunks : 256 * 16);
    if (!have) chunk_count_kernel<<<grid, DBLOCK, 0, st>>>(masks, npart, nranks, nchunks, cc);
    g_chunk_tag = ChunkTag();                 // the scan turns the counts into offsets, in place
    chunk_scan_kernel<<<nranks, DSCAN, 0, st>>>(cc, nchunks, offsets);
    if (index_elsize == 8)
        fill_kernel<int64_t><<<grid, DBLOCK, 0, st>>>(masks, npart, nranks, nchunks, cc, (int64_t *)indices);
    else
        fill_kernel<int32_t><<<grid, DBLOCK, 0, st>>>(masks, npart, nranks, nchunks, cc, (int32_t *)indices);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_take_rows(const void *src, int64_t src_stride0, int64_t row_bytes,
                             const void *indices, int32_t index_elsize, int64_t nrows, void *dst,
                             void *stream)
{
    PMX_REQUIRE(row_bytes > 0 && (row_bytes & 3) == 0, PMX_EINVAL, "row_bytes must be a multiple of 4");
    PMX_REQUIRE(index_elsize == 4 || index_elsize == 8, PMX_EINVAL, "index_elsize must be 4 or 8");
    if (nrows == 0) return PMX_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned grid = grid_for(nrows * (row_bytes >> 2), 256);
    if (index_elsize == 8)
        take_rows_kernel<int64_t><<<grid, 256, 0, st>>>((const char *)src, src_stride0, row_bytes, (const int64_t *)indices, nrows, (char *)dst);
    else
        take_rows_kernel<int32_t><<<grid, 256, 0, st>>>((const char *)src, src_stride0, row_bytes, (const int32_t *)indices, nrows, (char *)dst);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_pack_rows(const void *src, int64_t src_stride0, int64_t row_bytes, const void *indices,
                             int32_t index_elsize, int64_t nrows, void *dst, int64_t dst_stride, void *stream)
{
    PMX_REQUIRE(row_bytes > 0 && (row_bytes & 3) == 0, PMX_EINVAL, "row_bytes must be a multiple of 4");
    PMX_REQUIRE((dst_stride & 3) == 0 && dst_stride >= row_bytes, PMX_EINVAL, "dst_stride must be a multiple of 4, at least row_bytes");
    PMX_REQUIRE(indices == nullptr || index_elsize == 4 || index_elsize == 8, PMX_EINVAL, "index_elsize must be 4 or 8");
    if (nrows == 0) return PMX_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned grid = grid_for(nrows * (row_bytes >> 2), 256);
    if (indices != nullptr && index_elsize == 8)
        pack_rows_kernel<int64_t><<<grid, 256, 0, st>>>((const char *)src, src_stride0, row_bytes, (const int64_t *)indices, nrows, (char *)dst, dst_stride);
    else
        pack_rows_kernel<int32_t><<<grid, 256, 0, st>>>((const char *)src, src_stride0, row_bytes, (const int32_t *)indices, nrows, (char *)dst, dst_stride);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_scatter_add(const void *values, int32_t elsize, int32_t ncol,
                               const void *indices, int32_t index_elsize, int64_t nrows, void *out,
                               int64_t nout, void *stream)
{
    PMX_REQUIRE(elsize == 4 || elsize == 8, PMX_EINVAL, "values must be f4/f8");
    PMX_REQUIRE(index_elsize == 4 || index_elsize == 8, PMX_EINVAL, "index_elsize must be 4 or 8");
    PMX_REQUIRE(ncol >= 1, PMX_EINVAL, "ncol");
    hipStream_t st = (hipStream_t)stream;
    if (nout > 0) PMX_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)nout * ncol * elsize, st));
    if (nrows == 0) return PMX_OK;
    unsigned grid = grid_for(nrows * ncol, 256);
#define LAUNCH(T, I) scatter_add_kernel<T, I><<<grid, 256, 0, st>>>((const T *)values, ncol, (const I *)indices, nrows, (T *)out)
    if (elsize == 8) { if (index_elsize == 8) LAUNCH(double, int64_t); else LAUNCH(double, int32_t); }
    else { if (index_elsize == 8) LAUNCH(float, int64_t); else LAUNCH(float, int32_t); }
#undef LAUNCH
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}
