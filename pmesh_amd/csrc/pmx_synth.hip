// pmx_synth.hip — synthetic particle sets for bench.py, generated
// in HBM so the benchmark never stages 3 GB of positions through PCIe.
// "uniform": lattice + hashed jitter (splitmix64 finaliser), bit-reproducible
// against oracle/pmesh_oracle.c:pmo_synth_uniform.  "clustered": lattice +
// plane-wave Zel'dovich displacements (sin() differs from the host libm in the
// last bits, so only statistically equal to the oracle's).
#include <hip/hip_runtime.h>
#include <math.h>

#include "pmx_common.h"

namespace pmx {

__device__ inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) synth_uniform_kernel(DVec pos, int64_t nlat, double boxsize,
                                                            uint64_t seed, int64_t g0, int64_t n)
{
    const double h = boxsize / nlat;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t g = g0 + t;
        int64_t ijk[3] = {g / (nlat * nlat), (g / nlat) % nlat, g % nlat};
        for (int a = 0; a < 3; a++) {
            uint64_t r = mix64(seed ^ (uint64_t)(3 * g + a));
            double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
            double x = (ijk[a] + 0.5) * h + (u - 0.5) * 0.8 * h;
            pos.set(t, a, x);
        }
    }
}

struct Modes { double m[32][8]; int n; };

__global__ void __launch_bounds__(256) synth_clustered_kernel(DVec pos, int64_t nlat, double boxsize,
                                                              Modes md, double shift, int64_t g0, int64_t n)
{
    const double h = boxsize / nlat;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t g = g0 + t;
        int64_t ijk[3] = {g / (nlat * nlat), (g / nlat) % nlat, g % nlat};
        double q[3], x[3];
        for (int a = 0; a < 3; a++) { q[a] = (ijk[a] + 0.5 + shift) * h; x[a] = q[a]; }
        for (int m = 0; m < md.n; m++) {
            const double *mm = md.m[m];
            double ph = 2 * M_PI * (mm[0] * q[0] + mm[1] * q[1] + mm[2] * q[2]) / boxsize + mm[7];
            double s = mm[6] * sin(ph);
            for (int a = 0; a < 3; a++) x[a] += s * mm[3 + a];
        }
        for (int a = 0; a < 3; a++) {
            double y = fmod(x[a], boxsize);
            if (y < 0) y += boxsize;
            pos.set(t, a, y);
        }
    }
}

}  // namespace pmx

using namespace pmx;

extern "C" int pmx_synth_uniform(const pmx_vec *pos, int64_t nlat, double boxsize, uint64_t seed,
                                 int64_t g0, int64_t npart, void *stream)
{
    PMX_REQUIRE(vec_ok(pos) && pos->ncol >= 3, PMX_EINVAL, "pos must be (n,3) f4/f8");
    if (npart == 0) return PMX_OK;
    synth_uniform_kernel<<<grid_for(npart, 256), 256, 0, (hipStream_t)stream>>>(dvec(pos), nlat, boxsize, seed, g0, npart);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_synth_clustered(const pmx_vec *pos, int64_t nlat, double boxsize,
                                   const double *modes, int32_t nmodes, double shift, int64_t g0,
                                   int64_t npart, void *stream)
{
    PMX_REQUIRE(vec_ok(pos) && pos->ncol >= 3, PMX_EINVAL, "pos must be (n,3) f4/f8");
    PMX_REQUIRE(nmodes >= 0 && nmodes <= 32, PMX_EINVAL, "at most 32 modes");
    if (npart == 0) return PMX_OK;
    Modes md;
    md.n = nmodes;
    for (int m = 0; m < nmodes; m++)
        for (int c = 0; c < 8; c++) md.m[m][c] = modes[8 * m + c];
    synth_clustered_kernel<<<grid_for(npart, 256), 256, 0, (hipStream_t)stream>>>(dvec(pos), nlat, boxsize, md, shift, g0, npart);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}
