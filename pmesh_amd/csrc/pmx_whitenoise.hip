// pmx_whitenoise.hip — Gadget / N-GenIC compatible white noise in Fourier space, on the device.
//
// Reference: pmesh.whitenoise.generate for 3-d meshes (pmesh/_whitenoise_generics.h:29-238,
// _whitenoise_imp.c:21-105; called by ParticleMesh.generate_whitenoise, pm.py:1656-1696).
// Scheme: a master random stream, walked in rings of growing |index| over the (i, j) plane,
// hands one 31-bit seed to every column and its mirror images; each column (i, j) then draws
// (phase, amplitude) pairs for k2 = 0..N2/2 from its own stream.  That makes the field
// independent of the domain decomposition and keeps the large scales when the mesh is refined.
// Columns in the "upper" half of the plane take their k2 = 0 and k2 = N2/2 modes from the
// stream of the mirror column, conjugated, which makes the field Hermitian.
//
// Random numbers: Luescher's double-precision RANLUX (ranlxd, luxury 1 = 202 subtract-with-
// borrow steps per 12 delivered numbers; Comput. Phys. Commun. 79 (1994) 100), seeded the way
// GSL's gsl_rng_ranlxd1 is (the reference vendors it: pmesh/gsl/ranlxd.c).  All state is
// integer multiples of 2^-48, so the streams are bit-identical to the reference's; the
// amplitudes go through log/sqrt/sin/cos, which on the device differ from glibc in the last
// bit (tests: <= 4 ulp).
//
// Mapping: the master stream is a strictly sequential chain of N0*N1 draws -> one host core by default (a few
// ms at 512^2; the table is 2 x 4 bytes per local column), or one device thread (wn_master_kernel).  The columns are
// independent: one thread per column and stream ("own" stream: every mode except the mirrored
// planes; "mirror" stream: only the k2 = 0 and N2/2 modes of upper-half columns).  The twelve
// state words live in registers with static indices: a refill is 16 rounds of 12 steps plus 10
// steps, after which the ring is rotated back so that slot 0 is the oldest again; delivered
// numbers are parked in LDS ([slot][thread]) where they can be indexed dynamically.
#include <hip/hip_runtime.h>
#include <math.h>


#include <vector>

#include "pmx_common.h"

namespace pmx {

constexpr double ULP48 = 1.0 / 281474976710656.0;   // 2^-48
constexpr int WN_BLOCK = 128;

// ---- the master stream and the seed tables (one thread: the stream is one sequential chain) ----

struct HostRlx {
    double x[12];
    double borrow;
    int ir, ir_refill;
    __host__ __device__ void seed(unsigned long s)
    {
        int bits[31];
        if (s == 0) s = 1;
        int v = (int)(s & 0xFFFFFFFFUL);    // the reference keeps the seed in an int (ranlxd.c:202-208)
        for (int k = 0; k < 31; k++) { bits[k] = v % 2; v /= 2; }
        int a = 0, b = 18;
        for (int k = 0; k < 12; k++) {
            double acc = 0;
            for (int l = 0; l < 48; l++) {
                double y = (double)((bits[a] + 1) % 2);
                acc += acc + y;
                bits[a] = (bits[a] + bits[b]) % 2;
                a = (a + 1) % 31;
                b = (b + 1) % 31;
            }
            x[k] = ULP48 * acc;
        }
        borrow = 0;
        ir = 11;
        ir_refill = 0;
    }
    __host__ __device__ double uniform()
    {
        ir = (ir + 1) % 12;
        if (ir == ir_refill) {
            int p = ir;
            for (int k = 0; k < 202; k++) {
                double y = x[(p + 7) % 12] - x[p];
                y = y - borrow;
                if (y < 0) { borrow = ULP48; y += 1; } else borrow = 0;
                x[p] = y;
                p = (p + 1) % 12;
            }
            ir = p;
            ir_refill = p;
        }
        return x[ir];
    }
};

// ---- device: per-column streams ----------------------------------------------------------

struct DevRlx {
    double y[12];      // y[0] = oldest ring entry (the next to be replaced)
    double borrow;

    __device__ __forceinline__ void seed(uint32_t s)
    {
        // column seeds are < 2^31 (0x7fffffff * u): plain bits, no sign games
        if (s == 0) s = 1;
        uint32_t bits = s & 0x7fffffffu;
        int a = 0, b = 18;
#pragma unroll 1
        for (int k = 0; k < 12; k++) {
            unsigned long long acc = 0;
#pragma unroll 1
            for (int l = 0; l < 48; l++) {
                uint32_t ba = (bits >> a) & 1u, bb = (bits >> b) & 1u;
                acc = 2 * acc + (1u - ba);
                bits = (bits & ~(1u << a)) | ((ba ^ bb) << a);
                a = a == 30 ? 0 : a + 1;
                b = b == 30 ? 0 : b + 1;
            }
            set(k, (double)acc * ULP48);     // acc < 2^48: exact
        }
        borrow = 0;
    }
    __device__ __forceinline__ void set(int k, double v)
    {
        // dynamic k only here (12 predicated moves), everything else is static
#pragma unroll
        for (int m = 0; m < 12; m++)
            if (m == k) y[m] = v;
    }
    template <int T> __device__ __forceinline__ void step()
    {
        double d = y[(T + 7) % 12] - y[T];
        d = d - borrow;
        if (d < 0) { borrow = ULP48; d += 1; } else borrow = 0;
        y[T] = d;
    }
    template <int N> __device__ __forceinline__ void steps()
    {
        if constexpr (N > 0) {
            steps<N - 1>();
            step<N - 1>();
        }
    }
    // 202 steps, then rotate the ring so that slot 0 is the oldest entry again
    __device__ __forceinline__ void refill()
    {
#pragma unroll 1
        for (int r = 0; r < 16; r++) steps<12>();
        steps<10>();
        double t[12];
#pragma unroll
        for (int m = 0; m < 12; m++) t[m] = y[(m + 10) % 12];
#pragma unroll
        for (int m = 0; m < 12; m++) y[m] = t[m];
    }
};

// The master stream: N-GenIC's ring order (_whitenoise_generics.h:73-93; the mixed use of N0 / N1 in the ring corners
// is the reference's), one 31-bit seed per column and its mirror image, written for the columns of the local block.
struct SeedTables {
    int N0, N1;
    int64_t s0, s1, n0, n1;
    uint32_t *own, *mir;
    __host__ __device__ void assign(HostRlx &master, int i, int j) const
    {
        const unsigned int s = (unsigned int)(0x7fffffff * master.uniform());
        const int ii[2] = {i, (N0 - i) % N0};
        const int jj[2] = {j, (N1 - j) % N1};
        for (int a = 0; a < 2; a++) {
            // only the pairings (direct, direct) and (mirrored, mirrored) are ever read back
            const int64_t li = ii[a] - s0, lj = jj[a] - s1;
            if (li >= 0 && li < n0 && lj >= 0 && lj < n1) (a == 0 ? own : mir)[li * n1 + lj] = s;
        }
    }
    __host__ __device__ void fill(HostRlx &master, uint32_t seed) const
    {
        master.seed(seed);
        for (int i = 0; i < N0 / 2; i++) {
            for (int j = 0; j < i; j++) assign(master, i, j);
            for (int j = 0; j < i + 1; j++) assign(master, j, i);
            for (int j = 0; j < i; j++) assign(master, N0 - 1 - i, j);
            for (int j = 0; j < i + 1; j++) assign(master, N1 - 1 - j, i);
            for (int j = 0; j < i; j++) assign(master, i, N1 - 1 - j);
            for (int j = 0; j < i + 1; j++) assign(master, j, N0 - 1 - i);
            for (int j = 0; j < i; j++) assign(master, N0 - 1 - i, N1 - 1 - j);
            for (int j = 0; j < i + 1; j++) assign(master, N1 - 1 - j, N0 - 1 - i);
        }
    }
};

// [r5] The same on the device (pmx_whitenoise_master(1)): a strictly sequential chain of N0 * N1 draws (17 subtract-with-
// borrow steps each, every step waiting for the borrow of the one before) on ONE thread, its twelve state words in LDS.
// Measured on MI355X: 0.12 / 0.48 / 1.9 s at 256^2 / 512^2 / 1024^2 columns against 3 / 13 / 50 ms of one host core plus a
// copy of 8 bytes per column — a chain with no parallelism is the host's work, which is why the host form stays the
// default; the device form never copies or waits (the tables do not exist on the host) and is tested against the same
// golden spectrum.  (What would parallelise it is the skip-ahead of the recurrence — Luescher's equivalent linear
// congruential form modulo 2^576 - 2^240 + 1 — a 576-bit modular power per thread: sized, not built.)
static __global__ void __launch_bounds__(64) wn_master_kernel(uint32_t seed, SeedTables tb)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    __shared__ HostRlx master;
    tb.fill(master, seed);
}

// per host thread (ranks that run as threads of one process — the tests' thread ranks, a multi-threaded caller — set and
// read their own switch; pmx_whitenoise reads it ONCE at entry)
static thread_local int wn_master_on_device = 0;

struct WnGeom {
    int64_t nmesh[3], start[3], size[3], strides[3];
    int32_t unitary, elsize;
};

// blockIdx.y = 0: the column's own stream; 1: the mirror stream (upper-half columns only)
// sign = +1: the modes k2 = 0..N2/2; sign = -1 (full, uncompressed spectra only): the same draws
// from the conjugate quadrant's stream, written at N2 - k2 with the opposite imaginary part
// (_whitenoise_generics.h:133-147, 188-195).  The -1 pass is launched first so that the +1 pass
// owns the Nyquist plane, as the reference's loop order does.
__global__ void __launch_bounds__(WN_BLOCK) whitenoise_kernel(WnGeom g, const uint32_t *seed_own,
                                                              const uint32_t *seed_mirror, char *canvas, int sign)
{
    __shared__ double parked[12][WN_BLOCK];
    const int tid = threadIdx.x;
    const int64_t col = blockIdx.x * (int64_t)WN_BLOCK + tid;
    const int which = blockIdx.y;
    if (col >= g.size[0] * g.size[1]) return;
    const int64_t li = col / g.size[1], lj = col - li * g.size[1];
    const int64_t i = g.start[0] + li, j = g.start[1] + lj;
    const int64_t N0 = g.nmesh[0], N1 = g.nmesh[1], N2 = g.nmesh[2];
    const int64_t ci = (N0 - i) % N0, cj = (N1 - j) % N1;
    const bool mirror = (ci == i && cj < j) || (ci < i && cj != j) || (ci < i && cj == j);
    if (which == 1 && !mirror) return;

    DevRlx rng;
    rng.seed(which == 0 ? (sign > 0 ? seed_own[col] : seed_mirror[col]) : seed_mirror[col]);
    int have = 12;                                   // delivered numbers already consumed
    auto draw = [&]() -> double {
        if (have == 12) {
            rng.refill();
#pragma unroll
            for (int m = 0; m < 12; m++) parked[m][tid] = rng.y[m];
            have = 0;
        }
        return parked[have++][tid];
    };

    const bool selfconj_ij = ci == i && cj == j;
    char *colbase = canvas + li * g.strides[0] + lj * g.strides[1];
#pragma unroll 1
    for (int64_t k = 0; k <= N2 / 2; k++) {
        // _whitenoise_imp.c:21-27: phase first, then a non-zero amplitude deviate
        double phase = draw() * 2 * M_PI;
        double ampl;
        do ampl = draw(); while (ampl == 0);
        const bool plane = k == 0 || k == N2 / 2;
        const bool use_conj = mirror && plane;
        if (which == 0 ? use_conj : !plane) continue;      // this mode belongs to the other stream
        // the reference tests the UNREFLECTED k for membership, then writes at the reflected
        // index if that is inside the block (_whitenoise_generics.h:158-166, 11-27)
        if (k - g.start[2] < 0 || k - g.start[2] >= g.size[2]) continue;
        ampl = g.unitary ? 1.0 : sqrt(-log(ampl));
        double re = ampl * cos(phase), im = ampl * sin(phase);
        if (g.elsize == 8) { re = (double)(float)re; im = (double)(float)im; }
        const int64_t k2 = sign < 0 ? N2 - k : k;
        if (sign < 0) im = -im;
        if (use_conj) im = -im;
        if (selfconj_ij && (N2 - k2) % N2 == k2) {
            im = 0;                                       // self-conjugate mode: real
            if (g.unitary) re = 1;
        }
        if (i == 0 && j == 0 && k2 == 0) re = im = 0;     // the mean is set by the caller
        const int64_t r2 = k2 - g.start[2];
        if (r2 < 0 || r2 >= g.size[2]) continue;
        char *p = colbase + r2 * g.strides[2];
        if (g.elsize == 16) { ((double *)p)[0] = re; ((double *)p)[1] = im; }
        else { ((float *)p)[0] = (float)re; ((float *)p)[1] = (float)im; }
    }
}

}  // namespace pmx

using namespace pmx;

extern "C" int pmx_whitenoise_master(int32_t on_device)
{
    pmx::wn_master_on_device = on_device ? 1 : 0;
    return PMX_OK;
}

extern "C" int pmx_whitenoise(uint32_t seed, int32_t unitary, const int64_t *nmesh, const int64_t *start,
                              const int64_t *size, const int64_t *strides, int32_t elsize, void *canvas,
                              void *stream)
{
    PMX_REQUIRE(nmesh && start && size && strides, PMX_EINVAL, "NULL geometry");
    PMX_REQUIRE(elsize == 8 || elsize == 16, PMX_EINVAL, "canvas must be complex64 or complex128");
    for (int d = 0; d < 3; d++) {
        PMX_REQUIRE(nmesh[d] >= 1 && nmesh[d] < (1ll << 30), PMX_EINVAL, "bad mesh size");
        PMX_REQUIRE(size[d] >= 0 && start[d] >= 0, PMX_EINVAL, "bad block");
    }
    PMX_REQUIRE(start[0] + size[0] <= nmesh[0] && start[1] + size[1] <= nmesh[1], PMX_EINVAL,
                "block outside the mesh");
    PMX_REQUIRE(start[2] + size[2] <= nmesh[2], PMX_EINVAL, "block outside the mesh");
    // a block that holds modes beyond the Nyquist plane asks for the full (uncompressed) spectrum
    // of a complex-to-complex mesh: two passes, the negative k2 first (_whitenoise_generics.h:47-66)
    const bool full = start[2] + size[2] > nmesh[2] / 2 + 1;
    const int64_t ncol = size[0] * size[1];
    if (ncol == 0 || size[2] == 0) return PMX_OK;
    PMX_REQUIRE(canvas != nullptr, PMX_EINVAL, "canvas is NULL");
    hipStream_t st = (hipStream_t)stream;

    // ---- master stream: seeds of the local columns, on the host (default) or on the device (pmx_whitenoise_master)
    const int N0 = (int)nmesh[0], N1 = (int)nmesh[1];
    uint32_t *dseed = nullptr;
    PMX_HIP_CHECK(hipMallocAsync((void **)&dseed, (size_t)ncol * 8, st));
    std::vector<uint32_t> tables;
    const bool on_device = wn_master_on_device != 0;
    if (on_device) {
        hipError_t e0 = hipMemsetAsync(dseed, 0, (size_t)ncol * 8, st);
        if (e0 != hipSuccess) { (void)hipFreeAsync(dseed, st); PMX_HIP_CHECK(e0); }
        wn_master_kernel<<<1, 64, 0, st>>>(seed, SeedTables{N0, N1, start[0], start[1], size[0], size[1], dseed, dseed + ncol});
    } else {
        tables.assign((size_t)ncol * 2, 0u);
        HostRlx master;
        SeedTables{N0, N1, start[0], start[1], size[0], size[1], tables.data(), tables.data() + ncol}.fill(master, seed);
        hipError_t e1 = hipMemcpyAsync(dseed, tables.data(), (size_t)ncol * 8, hipMemcpyHostToDevice, st);
        if (e1 != hipSuccess) { (void)hipFreeAsync(dseed, st); PMX_HIP_CHECK(e1); }
    }
    WnGeom g;
    for (int d = 0; d < 3; d++) { g.nmesh[d] = nmesh[d]; g.start[d] = start[d]; g.size[d] = size[d]; g.strides[d] = strides[d]; }
    g.unitary = unitary ? 1 : 0;
    g.elsize = elsize;
    dim3 grid((unsigned)((ncol + WN_BLOCK - 1) / WN_BLOCK), 2);
    if (full) whitenoise_kernel<<<grid, WN_BLOCK, 0, st>>>(g, dseed, dseed + ncol, (char *)canvas, -1);
    whitenoise_kernel<<<grid, WN_BLOCK, 0, st>>>(g, dseed, dseed + ncol, (char *)canvas, +1);
    hipError_t e3 = hipGetLastError();
    (void)hipFreeAsync(dseed, st);          // (stream ordered: behind the kernels that read the tables)
    // the host tables go out of scope on return: wait for their copy (the device form neither copies nor waits)
    hipError_t e4 = on_device ? hipSuccess : hipStreamSynchronize(st);
    PMX_HIP_CHECK(e3);
    PMX_HIP_CHECK(e4);
    return PMX_OK;
}
