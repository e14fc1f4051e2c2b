// pmx_transfer.hip — apply-transfer on the complex field, one thread per mode.
//
// Replaces the Python slab loop of Field.apply (pmesh/pm.py:617-648) for the
// transfer functions used in the PM cycle (examples/nbody.py:154-181,
// pmesh/transfer.py:69-112,232-240, window compensation window.py:65-80).
// Coordinates follow _init_o_coords (pm.py:1200-1226): w = 2 pi/N (i - N[i>=N/2]),
// k = w N / L, Nyquist negative.  HBM-bound: one complex read + one complex
// write per mode; the k-vectors are recomputed from the index (no coordinate
// arrays are read).
#include <hip/hip_runtime.h>
#include <math.h>

#include "pmx_common.h"

namespace pmx {

struct TGeom {
    int64_t shape[3], in_strides[3], out_strides[3], start[3], nmesh[3];
    double boxsize[3];
    double dw[3], nl[3];   // 2 pi / N and N / L per axis
    // memory-order permutation: ax[2] is the fastest-varying axis in memory
    int32_t ax[3];
    int32_t ndim;
};

// grid.y walks the slowest memory axis, grid.x / threads the flattened two fast axes:
// 32-bit index arithmetic only, consecutive threads touch consecutive modes.
// SIMPLE: no Gaussian, no deconvolution, spectral or no gradient — the transfers of the PM
// cycle proper (dx1, potential): no transcendental code, few registers, high occupancy.
template <typename T, bool SIMPLE>
__global__ void __launch_bounds__(256) transfer_kernel(pmx_transfer t, TGeom g, const char *in, char *out)
{
    const uint32_t n1 = (uint32_t)g.shape[g.ax[1]], n2 = (uint32_t)g.shape[g.ax[2]];
    const uint32_t inner = n1 * n2;
    for (int64_t i0 = blockIdx.y; i0 < g.shape[g.ax[0]]; i0 += gridDim.y)
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < inner; q += gridDim.x * blockDim.x) {
        const uint32_t i1 = q / n2;
        const int64_t v0 = i0, v1 = i1, v2 = q - i1 * n2;   // indices in memory order
        int64_t idx[3];
#pragma unroll
        for (int d = 0; d < 3; d++) idx[d] = (g.ax[0] == d) ? v0 : ((g.ax[1] == d) ? v1 : v2);
        double kk[3] = {0, 0, 0}, ww[3] = {0, 0, 0}, k2 = 0;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if (d >= g.ndim) break;
            int64_t gi = idx[d] + g.start[d];
            double wi = (double)gi;
            if (gi >= g.nmesh[d] / 2) wi -= g.nmesh[d];
            wi *= g.dw[d];               // 2 pi / N   (pm.py:1217)
            ww[d] = wi;
            kk[d] = wi * g.nl[d];        // w N / L    (pm.py:1218)
            k2 += kk[d] * kk[d];
        }
        double re = t.amplitude, im = 0;
        if (t.laplace_pow) {
            double qq = (k2 == 0) ? 1.0 : k2;
            if (t.laplace_pow == -1) re *= 1.0 / qq;
            else if (t.laplace_pow == 1) re *= qq;
            else if (!SIMPLE) re *= pow(qq, (double)t.laplace_pow);
        }
        if (!SIMPLE && t.gauss_r != 0) re *= exp(-0.5 * k2 * t.gauss_r * t.gauss_r);
        if (!SIMPLE && t.deconv_pow) {
            for (int d = 0; d < g.ndim; d++) {
                double x = 0.5 * ww[d];
                double s;
                if (x < 1e-5 && x > -1e-5) { double x2 = x * x; s = 1.0 - x2 / 6. + x2 * x2 / 120.; }
                else s = sin(x) / x;
                double sp = s;
                for (int e = 1; e < t.deconv_pow; e++) sp *= s;
                re /= sp;
            }
        }
        if (t.grad_dir >= 0) {
            int d = t.grad_dir;
            double D;
            if (SIMPLE || t.grad_kind == 0) D = kk[d];
            else {
                double C = g.boxsize[d] / g.nmesh[d];
                double w = kk[d] * C;
                D = 1.0 / C * 1 / 6.0 * (8 * sin(w) - sin(2 * w));
            }
            im = re * D;
            re = 0;
        }
        int64_t io = idx[0] * g.in_strides[0] + idx[1] * g.in_strides[1] + idx[2] * g.in_strides[2];
        int64_t oo = idx[0] * g.out_strides[0] + idx[1] * g.out_strides[1] + idx[2] * g.out_strides[2];
        const T *a = (const T *)(in + io);
        T *b = (T *)(out + oo);
        double ar = a[0], ai = a[1];
        b[0] = (T)(re * ar - im * ai);
        b[1] = (T)(re * ai + im * ar);
    }
}

}  // namespace pmx

using namespace pmx;

extern "C" int pmx_apply_transfer(const pmx_transfer *t, int32_t ndim, int32_t elsize,
                                  const void *in, const int64_t *in_strides, void *out,
                                  const int64_t *out_strides, const int64_t *shape,
                                  const int64_t *start, const int64_t *nmesh,
                                  const double *boxsize, void *stream)
{
    PMX_REQUIRE(t && ndim >= 1 && ndim <= 3, PMX_EINVAL, "bad arguments");
    PMX_REQUIRE(elsize == 4 || elsize == 8, PMX_EINVAL, "elsize must be 4 or 8");
    PMX_REQUIRE(t->grad_dir < ndim, PMX_EINVAL, "grad_dir out of range");
    TGeom g;
    g.ndim = ndim;
    for (int d = 0; d < 3; d++) {
        bool on = d < ndim;
        g.shape[d] = on ? shape[d] : 1;
        g.in_strides[d] = on ? in_strides[d] : 0;
        g.out_strides[d] = on ? out_strides[d] : 0;
        g.start[d] = on ? start[d] : 0;
        g.nmesh[d] = on ? nmesh[d] : 1;
        g.boxsize[d] = on ? boxsize[d] : 1.0;
        g.dw[d] = 2 * M_PI / g.nmesh[d];
        g.nl[d] = g.nmesh[d] / g.boxsize[d];
    }
    // order axes by decreasing output stride so consecutive threads touch
    // consecutive memory whatever the (transposed) layout is
    int ax[3] = {0, 1, 2};
    for (int a = 0; a < 3; a++)
        for (int b = a + 1; b < 3; b++) {
            int64_t sa = llabs(g.out_strides[ax[a]]), sb = llabs(g.out_strides[ax[b]]);
            bool swap = sa < sb || (sa == sb && g.shape[ax[a]] == 1 && g.shape[ax[b]] != 1);
            if (swap) { int tmp = ax[a]; ax[a] = ax[b]; ax[b] = tmp; }
        }
    for (int a = 0; a < 3; a++) g.ax[a] = ax[a];
    int64_t total = g.shape[0] * g.shape[1] * g.shape[2];
    if (total == 0) return PMX_OK;
    int64_t inner = g.shape[g.ax[1]] * g.shape[g.ax[2]];
    PMX_REQUIRE(inner < (1ll << 31), PMX_EUNSUPPORTED, "plane of more than 2^31 modes");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((inner + 255) / 256), (unsigned)(g.shape[g.ax[0]] < 65535 ? g.shape[g.ax[0]] : 65535));
    bool simple = t->gauss_r == 0 && t->deconv_pow == 0 && (t->grad_dir < 0 || t->grad_kind == 0) &&
                  t->laplace_pow >= -1 && t->laplace_pow <= 1;
    if (elsize == 8) {
        if (simple) transfer_kernel<double, true><<<grid, 256, 0, st>>>(*t, g, (const char *)in, (char *)out);
        else transfer_kernel<double, false><<<grid, 256, 0, st>>>(*t, g, (const char *)in, (char *)out);
    } else {
        if (simple) transfer_kernel<float, true><<<grid, 256, 0, st>>>(*t, g, (const char *)in, (char *)out);
        else transfer_kernel<float, false><<<grid, 256, 0, st>>>(*t, g, (const char *)in, (char *)out);
    }
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}
