// pmx_window_dev.h — device-side window arithmetic shared by the direct and the
// tile-binned kernels (see pmx_window.hip for the provenance of every formula).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "pmx_common.h"

namespace pmx {

__host__ __device__ inline int native_support(int kind)
{
    switch (kind) {
    case PMX_NEAREST: case PMX_TUNED_NNB: return 1;
    case PMX_LINEAR: case PMX_TUNED_CIC: return 2;
    case PMX_QUADRATIC: case PMX_TUNED_TSC: return 3;
    case PMX_CUBIC: case PMX_TUNED_PCS: return 4;
    }
    // table driven: lanczos n has support 2n, acg n has support n (_window_lanczos.h:2057, _window_acg.h:2057)
    if (kind >= PMX_LANCZOS2 && kind <= PMX_LANCZOS6) return 2 * (kind - PMX_LANCZOS2 + 2);
    if (kind >= PMX_ACG2 && kind <= PMX_ACG6) return kind - PMX_ACG2 + 2;
    // wavelet scaling functions (_window_wavelets.h: _<name>_nativesupport)
    switch (kind) {
    case PMX_DB6: case PMX_SYM6: return 7;
    case PMX_DB12: case PMX_SYM12: return 10;
    case PMX_DB20: return 13;
    case PMX_SYM20: return 12;
    }
    return -1;
}

struct WInfo {
    int support;
    int left;
    double vfactor, shift;
};

// pmesh_window_info_init (_window_imp.c:24-47)
__host__ __device__ inline WInfo winfo_init(int nativesupport, double support)
{
    WInfo w;
    if (support <= 0) {
        w.support = nativesupport;
        support = nativesupport;
    } else {
        w.support = (int)support;
        w.support += (support != (double)w.support);
    }
    w.left = (w.support - 1) / 2;
    w.shift = support / 2.0 - w.support / 2;
    w.vfactor = nativesupport / (1. * support);
    return w;
}

__device__ __forceinline__ int wrap1(int i, int64_t n)
{
    if (n <= 0) return i;
    int m = (int)n;
    int r = i % m;
    return r < 0 ? r + m : r;
}

// ---- one axis of SETUP_KERNEL_* (tuned_nnb.h:1-27, tuned_cic.h:1-32,
//      tuned_tsc.h:1-37, tuned_pcs.h:1-52) ---------------------------------
template <int KIND> struct Tuned;

template <> struct Tuned<PMX_TUNED_NNB> {
    static constexpr int S = 1;
    // first stencil index alone (the same expression as in axis())
    __device__ static __forceinline__ int first(double X) { return (int)floor(X + 0.5); }
    __device__ static __forceinline__ void axis(double X, int order, double scale, int *I, double *V)
    {
        I[0] = (int)floor(X + 0.5);
        V[0] = (order == 0) ? 1 : 0;
    }
};

template <> struct Tuned<PMX_TUNED_CIC> {
    static constexpr int S = 2;
    __device__ static __forceinline__ int first(double X) { return (int)floor(X); }
    __device__ static __forceinline__ void axis(double X, int order, double scale, int *I, double *V)
    {
        I[0] = (int)floor(X);
        I[1] = I[0] + 1;
        if (order == 0) {
            V[1] = X - I[0];
            V[0] = 1. - V[1];
        } else {
            V[1] = scale;
            V[0] = -scale;
        }
    }
};

template <> struct Tuned<PMX_TUNED_TSC> {
    static constexpr int S = 3;
    __device__ static __forceinline__ int first(double X) { return (int)floor(X + 0.5) - 1; }
    __device__ static __forceinline__ void axis(double X, int order, double scale, int *I, double *V)
    {
        I[1] = (int)floor(X + 0.5);
        I[0] = I[1] - 1;
        I[2] = I[1] + 1;
        if (order == 0) {
            V[1] = 0.75 - (X - I[1]) * (X - I[1]);
            V[0] = (1.5 - (X - I[0])) * (1.5 - (X - I[0])) * 0.5;
            V[2] = (1.5 + (X - I[2])) * (1.5 + (X - I[2])) * 0.5;
        } else {
            V[1] = -2 * (X - I[1]) * scale;
            V[0] = -(1.5 - (X - I[0])) * scale;
            V[2] = (1.5 + (X - I[2])) * scale;
        }
    }
};

template <> struct Tuned<PMX_TUNED_PCS> {
    static constexpr int S = 4;
    __device__ static __forceinline__ int first(double X) { return (int)floor(X) - 1; }
    __device__ static __forceinline__ void axis(double X, int order, double scale, int *I, double *V)
    {
        I[1] = (int)floor(X);
        I[0] = I[1] - 1;
        I[2] = I[1] + 1;
        I[3] = I[2] + 1;
        if (order == 0) {
            V[1] = 1.0 / 6.0 * (4 - 6 * (X - I[1]) * (X - I[1]) + 3 * (X - I[1]) * (X - I[1]) * (X - I[1]));
            V[2] = 1.0 / 6.0 * (4 - 6 * (X - I[2]) * (X - I[2]) - 3 * (X - I[2]) * (X - I[2]) * (X - I[2]));
            V[0] = 1.0 / 6.0 * (2 - (X - I[0])) * (2 - (X - I[0])) * (2 - (X - I[0]));
            V[3] = 1.0 / 6.0 * (2 + (X - I[3])) * (2 + (X - I[3])) * (2 + (X - I[3]));
        } else {
            // quirk Q1: no scale factor in the tuned PCS derivative
            V[1] = +1.0 / 6.0 * (-12 * (X - I[1]) + 9 * (X - I[1]) * (X - I[1]));
            V[2] = -1.0 / 6.0 * (+12 * (X - I[2]) + 9 * (X - I[2]) * (X - I[2]));
            V[0] = -1.0 / 2.0 * (2 - (X - I[0])) * (2 - (X - I[0]));
            V[3] = +1.0 / 2.0 * (2 + (X - I[3])) * (2 + (X - I[3]));
        }
    }
};

// ---- the same weights from fewer instructions (the RELAXED forms of the tile kernels) ----------------------
// The cell index of a particle is floor(pos * scale + translate [+ 0.5]) in double precision without FMA,
// bit for bit the reference's (Tuned<KIND>::first); everything AFTER the index may differ from the reference's
// arithmetic within the tolerance SURVEY.md 8(d) / BASELINE.json's north_star state for field values.  Here the
// weights are polynomials in ONE offset d = X - I_ref (CIC: the first cell, d in [0, 1); TSC: the centre cell,
// d in [-1/2, 1/2); PCS: the second cell, d in [0, 1)) evaluated with fused multiply-adds in the precision F
// of the canvas (float canvases: d is formed in double, converted once, and the weights cost packed-rate fp32
// instructions instead of half-rate fp64 ones).  Same polynomials as _window_tuned_cic/tsc/pcs.h, expanded:
//   TSC  W(0) = 3/4 - d^2,  W(-1) = (1/2 - d)^2 / 2,  W(+1) = (1/2 + d)^2 / 2
//   PCS  W(0) = 2/3 - d^2 + d^3/2,  W(1) = 2/3 - e^2 - e^3/2 (e = d - 1),  W(-1) = -e^3/6,  W(2) = d^3/6
// Errors: <= 3 ulp of F per weight (relative 3e-16 / 2e-7), against 1-2 ulp of double for the reference's form.
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <int KIND, typename F> struct Fast;

template <typename F> struct Fast<PMX_TUNED_NNB, F> {
    static constexpr int REF = 0;      // d = X - (I[0] + REF)
    __device__ static __forceinline__ void axis(F d, int order, F scale, F *V) { V[0] = order == 0 ? (F)1 : (F)0; }
};
template <typename F> struct Fast<PMX_TUNED_CIC, F> {
    static constexpr int REF = 0;
    __device__ static __forceinline__ void axis(F d, int order, F scale, F *V)
    {
        if (order == 0) { V[1] = d; V[0] = (F)1 - d; }
        else { V[1] = scale; V[0] = -scale; }
    }
};
template <typename F> struct Fast<PMX_TUNED_TSC, F> {
    static constexpr int REF = 1;
    __device__ static __forceinline__ void axis(F d, int order, F scale, F *V)
    {
        const F lo = (F)0.5 - d, hi = (F)0.5 + d;
        if (order == 0) {
            V[1] = fma_(-d, d, (F)0.75);
            V[0] = ((F)0.5 * lo) * lo;
            V[2] = ((F)0.5 * hi) * hi;
        } else {
            V[1] = ((F)-2 * d) * scale;
            V[0] = -lo * scale;
            V[2] = hi * scale;
        }
    }
};
template <typename F> struct Fast<PMX_TUNED_PCS, F> {
    static constexpr int REF = 1;
    __device__ static __forceinline__ void axis(F d, int order, F scale, F *V)
    {
        const F e = d - (F)1, d2 = d * d, e2 = e * e;
        if (order == 0) {
            V[1] = fma_(d2, fma_((F)0.5, d, (F)-1), (F)(2.0 / 3.0));
            V[2] = fma_(e2, fma_((F)-0.5, e, (F)-1), (F)(2.0 / 3.0));
            V[0] = e2 * (e * (F)(-1.0 / 6.0));
            V[3] = d2 * (d * (F)(1.0 / 6.0));
        } else {
            // quirk Q1: no scale factor in the tuned PCS derivative
            V[1] = d * fma_((F)1.5, d, (F)-2);
            V[2] = e * fma_((F)-1.5, e, (F)-2);
            V[0] = (F)-0.5 * e2;
            V[3] = (F)0.5 * d2;
        }
    }
};

}  // namespace pmx
