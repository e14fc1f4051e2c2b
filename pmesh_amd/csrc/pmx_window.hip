// pmx_window.hip — paint / readout, direct form (one thread per particle).
//
// Replaces the reference's per-particle C kernels and the Python-level loop
// around them: pmesh/_window.pyx:128-205 -> pmesh/_window_generics.h:4-142 ->
// pmesh/_window_tuned_{nnb,cic,tsc,pcs}.h.  Arithmetic follows SURVEY.md
// Appendix A: positions are widened to double, X = pos*scale + translate is a
// separate multiply and add (this library is built with -ffp-contract=off),
// the base index is (int)floor(X) [CIC/PCS] or (int)floor(X+0.5) [NNB/TSC],
// weights use the UNWRAPPED indices and each sub-expression is written as in
// the reference header so that weights are bit-identical; the window product
// is ((V0*mass)*V1)*V2, left to right.
//
// This file is the order-independent baseline: it works for particles in any
// order and canvases with any strides, scattering with hardware
// global_atomic_add_f64/f32.  The LDS-tiled kernels for tile-sorted particles
// are in pmx_binned.hip.
//
// Deliberate, documented deviations (the reference's behaviour there is
// undefined): a particle whose grid coordinate is NaN or |X| >= 2^30 is
// dropped (the reference spins in its wrap loop or overflows the int cast).
#include <hip/hip_runtime.h>
#include <math.h>

#include <mutex>

#include "pmx_common.h"
#include "pmx_window_dev.h"

namespace pmx {

template <typename T> __device__ __forceinline__ void canvas_add(char *canvas, int64_t off, double f)
{
    unsafeAtomicAdd((T *)(canvas + off), (T)f);
}

// _{nnb,cic,tsc,pcs}_tuned_{paint,readout}{1,2,3}.  `nd` is a compile-time
// constant at the fast-path call sites (the loops unroll) and a run-time value
// in the general kernel.
template <int KIND, typename T, bool PAINT>
__device__ __forceinline__ double tuned_particle(const pmx_painter &p, char *canvas,
                                                 const double *x, double mass, int nd)
{
    constexpr int S = Tuned<KIND>::S;
    int I[PMX_MAXDIM][S];
    double V[PMX_MAXDIM][S];
#pragma unroll
    for (int d = 0; d < PMX_MAXDIM; d++) {
        if (d >= nd) break;
        double X = x[d] * p.scale[d] + p.translate[d];
        if (!(fabs(X) < 1073741824.0)) return 0.0;  // NaN / out of int range: dropped
        Tuned<KIND>::axis(X, p.order[d], p.scale[d], I[d], V[d]);
        // consecutive indices: wrap the first, step the rest
        int r = wrap1(I[d][0], p.period[d]);
        I[d][0] = r;
#pragma unroll
        for (int a = 1; a < S; a++) {
            r = r + 1;
            if (p.period[d] > 0) {
                while (r >= p.period[d]) r -= (int)p.period[d];
            }
            I[d][a] = r;
        }
    }
    if (PAINT) {
#pragma unroll
        for (int a = 0; a < S; a++) V[0][a] *= mass;
    }
    const int Sb = nd > 1 ? S : 1, Sc = nd > 2 ? S : 1;
    double value = 0;
#pragma unroll
    for (int a = 0; a < S; a++) {
        bool in0 = !(I[0][a] < 0 || I[0][a] >= p.size[0]);
        int64_t off0 = I[0][a] * p.strides[0];
#pragma unroll
        for (int b = 0; b < S; b++) {
            if (b >= Sb) break;
            double fb = V[0][a];
            bool in1 = in0;
            int64_t off1 = off0;
            if (nd > 1) {
                fb = fb * V[1][b];
                in1 = in1 && !(I[1][b] < 0 || I[1][b] >= p.size[1]);
                off1 += I[1][b] * p.strides[1];
            }
#pragma unroll
            for (int c = 0; c < S; c++) {
                if (c >= Sc) break;
                double f = fb;
                bool in2 = in1;
                int64_t off = off1;
                if (nd > 2) {
                    f = f * V[2][c];
                    in2 = in2 && !(I[2][c] < 0 || I[2][c] >= p.size[2]);
                    off += I[2][c] * p.strides[2];
                }
                if (PAINT) {
                    if (in2) canvas_add<T>(canvas, off, f);
                } else {
                    // _REd3 returns 0 outside (generics.h:162-167)
                    value += in2 ? (double)(*(const T *)(canvas + off)) * f : 0.0;
                }
            }
        }
    }
    return value;
}

template <int KIND, int ND, typename T>
__global__ void __launch_bounds__(256) paint_tuned_kernel(pmx_painter p, char *canvas, DVec pos,
                                                          DVec mass, double mass_scalar, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        double x[PMX_MAXDIM];
#pragma unroll
        for (int d = 0; d < ND; d++) x[d] = pos.get(i, d);
        double m = mass.data ? mass.get(i, 0) : mass_scalar;
        tuned_particle<KIND, T, true>(p, canvas, x, m, ND);
    }
}

template <int KIND, int ND, typename T>
__global__ void __launch_bounds__(256) readout_tuned_kernel(pmx_painter p, const char *canvas,
                                                            DVec pos, DVec out, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        double x[PMX_MAXDIM];
#pragma unroll
        for (int d = 0; d < ND; d++) x[d] = pos.get(i, d);
        double v = tuned_particle<KIND, T, false>(p, const_cast<char *>(canvas), x, 0.0, ND);
        out.set(i, 0, v);
    }
}

// lookup table of a table-driven window (device pointer)
struct TableD {
    const double *v;
    int n;
    double step;
    double hsupport;    // wavelets: the table starts at -support/2 (not symmetric); else < 0
};

// `_<name>_kernel` of the generated headers (makelanczos.py:20-30): linear interpolation
__device__ inline double table_kernel(const TableD &t, double x)
{
    if (t.hsupport >= 0) {
        // makewavelets.py:26-35: one-sided table over [0, support), no symmetry
        x += t.hsupport;
        double f = x / t.step;
        if (f < 0) return 0;
        int i = (int)f;
        f -= i;
        if (i >= t.n - 1) return 0;
        return t.v[i] * (1 - f) + t.v[i + 1] * f;
    }
    x = fabs(x);
    double f = x / t.step;
    int i = (int)f;
    if (i < 0) return 0;
    if (i >= t.n - 1) return 0;
    f -= i;
    return t.v[i] * (1 - f) + t.v[i + 1] * f;
}

// `_<name>_diff` (makelanczos.py:31-46): slope of the table segment
__device__ inline double table_diff(const TableD &t, double x)
{
    if (t.hsupport >= 0) {
        // makewavelets.py:36-46 (the index truncates towards zero, as `int i = x / step` does)
        x += t.hsupport;
        int i = (int)(x / t.step);
        if (i < 0) return 0;
        if (i >= t.n - 1) return 0;
        return (t.v[i + 1] - t.v[i]) / t.step;
    }
    double factor;
    if (x >= 0) factor = 1;
    else { factor = -1; x = -x; }
    int i = (int)(x / t.step);
    if (i < 0) return 0;
    if (i >= t.n - 1) return 0;
    double f = t.v[i + 1] - t.v[i];
    return factor * f / t.step;
}

// ---- analytic kernels of the generic path (_window_imp.c:108-236) ---------
__device__ inline double k_eval(int kind, double x)
{
    switch (kind) {
    case PMX_NEAREST: case PMX_TUNED_NNB:
        return (x < 0.5 && x >= -0.5) ? 1.0 : 0.0;
    case PMX_LINEAR: case PMX_TUNED_CIC:
        x = fabs(x);
        return (x < 1.0) ? 1.0 - x : 0.0;
    case PMX_QUADRATIC: case PMX_TUNED_TSC:
        x = fabs(x);
        if (x <= 0.5) return 0.75 - x * x;
        if (x < 1.5) { x = 1.5 - x; return (x * x) * 0.5; }
        return 0;
    default: {
        x = fabs(x);
        double xx = x * x;
        if (x < 1.0) return 1.0 / 6.0 * (4 - 6 * xx + 3 * xx * x);
        if (x < 2) return 1.0 / 6.0 * (2 - x) * (2 - x) * (2 - x);
        return 0;
    }
    }
}

__device__ inline double d_eval(int kind, double x)
{
    double factor;
    switch (kind) {
    case PMX_NEAREST: case PMX_TUNED_NNB:
        return 0;
    case PMX_LINEAR: case PMX_TUNED_CIC:
        if (x < 0) { factor = 1; x = -x; }
        else if (x > 0) factor = -1;
        else factor = 0;
        return (x < 1.0) ? factor : 0.0;
    case PMX_QUADRATIC: case PMX_TUNED_TSC:
        if (x < 0) { x = -x; factor = -1; } else factor = +1;
        if (x <= 0.5) return factor * (-2 * x);
        if (x < 1.5) return factor * (-(1.5 - x));
        return 0;
    default: {
        if (x < 0) { factor = -1; x = -x; } else factor = +1;
        double xx = x * x;
        if (x < 1.0) return factor * (1.0 / 6.0) * (-12 * x + 9 * xx);
        if (x < 2.0) return factor * (-1.0 / 2.0) * (2 - x) * (2 - x);
        return 0;
    }
    }
}

template <typename T, bool PAINT>
__device__ double tuned_dispatch(const pmx_painter &p, char *canvas, const double *x, double m)
{
    switch (p.kind) {
    case PMX_TUNED_NNB: return tuned_particle<PMX_TUNED_NNB, T, PAINT>(p, canvas, x, m, p.ndim);
    case PMX_TUNED_CIC: return tuned_particle<PMX_TUNED_CIC, T, PAINT>(p, canvas, x, m, p.ndim);
    case PMX_TUNED_TSC: return tuned_particle<PMX_TUNED_TSC, T, PAINT>(p, canvas, x, m, p.ndim);
    default: return tuned_particle<PMX_TUNED_PCS, T, PAINT>(p, canvas, x, m, p.ndim);
    }
}

// _generic_paint / _generic_readout (_window_generics.h:4-142) with _fill_k
// (_window_imp.c:50-83): any kind, any integer support <= 32, per-particle
// hsml; tuned kinds take their fast path per particle when the integer support
// equals the native one (quirk Q6).  p.support holds the effective integer
// support of the window object (pmesh_painter_init, _window_imp.c:456-458).
// P: pmx_painter (1..3 dimensions, MAXD = PMX_MAXDIM) or pmx_painter_nd (up to PMX_MAXDIM_ND: pmx_paint_nd /
// pmx_readout_nd; no tuned fast path there: the reference's own is for ndim <= 3, _window_imp.c:486-520)
template <typename T, bool PAINT, typename P = pmx_painter, int MAXD = PMX_MAXDIM>
__global__ void __launch_bounds__(256) general_kernel(P p, char *canvas, DVec pos,
                                                      DVec mass, double mass_scalar, DVec hsml,
                                                      DVec out, int64_t n, TableD tab)
{
    const int nd = p.ndim;
    const int ns = native_support(p.kind);
    const bool tuned = p.kind >= PMX_TUNED_NNB && p.kind <= PMX_TUNED_PCS;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        double x[MAXD];
        for (int d = 0; d < nd; d++) x[d] = pos.get(i, d);
        double m = PAINT ? (mass.data ? mass.get(i, 0) : mass_scalar) : 0.0;
        double h = hsml.data ? hsml.get(i, 0) : 1.0;
        WInfo w = winfo_init(ns, p.support * h);
        double value = 0;
        bool done = false;
        if constexpr (MAXD == PMX_MAXDIM) {
            if (tuned && w.support == ns) {
                value = tuned_dispatch<T, PAINT>(p, canvas, x, m);
                done = true;
            }
        }
        if (!done) {
            // weights are tabulated per particle up to PMX_MAXSUPPORT points per axis and
            // evaluated on the fly beyond (the reference sizes its table dynamically,
            // _window_generics.h:24; test_lanczos_resize uses support 400)
            const bool cached = w.support <= PMX_MAXSUPPORT;
            int ipos[MAXD];
            double dxs[MAXD];
            double k[MAXD][PMX_MAXSUPPORT];
            bool finite = true;
            auto weight = [&](int d, int j) -> double {
                double xx = (dxs[d] - j) * w.vfactor;
                if (tab.v) {
                    if (p.order[d] == 0) return table_kernel(tab, xx) * w.vfactor;
                    return table_diff(tab, xx) * p.scale[d] * w.vfactor * w.vfactor;
                }
                if (p.order[d] == 0) return k_eval(p.kind, xx) * w.vfactor;
                return d_eval(p.kind, xx) * p.scale[d] * w.vfactor * w.vfactor;
            };
            for (int d = 0; d < nd; d++) {
                double g = x[d] * p.scale[d] + p.translate[d];
                if (!(fabs(g) < 1073741824.0)) { finite = false; break; }
                ipos[d] = (int)(floor(g + w.shift) - w.left);
                dxs[d] = g - ipos[d];
                if (cached)
                    for (int j = 0; j < w.support; j++) k[d][j] = weight(d, j);
            }
            if (finite) {
                int rel[MAXD];
                for (int d = 0; d < MAXD; d++) rel[d] = 0;
                const int s2 = w.support;
                while (rel[0] != s2) {
                    double kernel = 1.0;
                    int64_t ind = 0;
                    bool outside = false;
                    for (int d = 0; d < nd; d++) {
                        int r = rel[d];
                        kernel *= cached ? k[d][r] : weight(d, r);
                        int t = wrap1(ipos[d] + r, p.period[d]);
                        if (t >= p.size[d] || t < 0) { outside = true; break; }
                        ind += p.strides[d] * t;
                    }
                    if (!outside) {
                        if (PAINT) canvas_add<T>(canvas, ind, m * kernel);
                        else value += kernel * (double)(*(const T *)(canvas + ind));
                    }
                    rel[nd - 1]++;
                    for (int d = nd - 1; d > 0; d--)
                        if (rel[d] == s2) { rel[d - 1]++; rel[d] = 0; }
                }
            }
        }
        if (!PAINT) out.set(i, 0, value);
    }
}

static int check_painter(const pmx_painter *p)
{
    PMX_REQUIRE(p != nullptr, PMX_EINVAL, "painter is NULL");
    PMX_REQUIRE(p->ndim >= 1 && p->ndim <= PMX_MAXDIM, PMX_EUNSUPPORTED, "ndim must be 1..3");
    PMX_REQUIRE(p->canvas_elsize == 4 || p->canvas_elsize == 8, PMX_EINVAL,
                "canvas must be float or double (_window.pyx:135)");
    PMX_REQUIRE(native_support(p->kind) > 0, PMX_EUNSUPPORTED, "window kind not built");
    for (int d = 0; d < p->ndim; d++) {
        PMX_REQUIRE(p->period[d] >= 0 && p->period[d] < (1 << 30), PMX_EINVAL, "bad period");
        PMX_REQUIRE(p->size[d] >= 0 && p->size[d] < (1ll << 31), PMX_EINVAL, "bad size");
    }
    return PMX_OK;
}

template <int KIND, typename T>
static void launch_paint_nd(const pmx_painter &p, void *canvas, DVec pos, DVec mass, double ms,
                            int64_t n, hipStream_t st)
{
    dim3 block(256), grid(grid_for(n, 256));
    switch (p.ndim) {
    case 1: paint_tuned_kernel<KIND, 1, T><<<grid, block, 0, st>>>(p, (char *)canvas, pos, mass, ms, n); break;
    case 2: paint_tuned_kernel<KIND, 2, T><<<grid, block, 0, st>>>(p, (char *)canvas, pos, mass, ms, n); break;
    default: paint_tuned_kernel<KIND, 3, T><<<grid, block, 0, st>>>(p, (char *)canvas, pos, mass, ms, n); break;
    }
}

template <int KIND, typename T>
static void launch_readout_nd(const pmx_painter &p, const void *canvas, DVec pos, DVec out,
                              int64_t n, hipStream_t st)
{
    dim3 block(256), grid(grid_for(n, 256));
    switch (p.ndim) {
    case 1: readout_tuned_kernel<KIND, 1, T><<<grid, block, 0, st>>>(p, (const char *)canvas, pos, out, n); break;
    case 2: readout_tuned_kernel<KIND, 2, T><<<grid, block, 0, st>>>(p, (const char *)canvas, pos, out, n); break;
    default: readout_tuned_kernel<KIND, 3, T><<<grid, block, 0, st>>>(p, (const char *)canvas, pos, out, n); break;
    }
}

template <typename T>
static void launch_paint_kind(const pmx_painter &p, void *canvas, DVec pos, DVec mass, double ms,
                              int64_t n, hipStream_t st)
{
    switch (p.kind) {
    case PMX_TUNED_NNB: launch_paint_nd<PMX_TUNED_NNB, T>(p, canvas, pos, mass, ms, n, st); break;
    case PMX_TUNED_CIC: launch_paint_nd<PMX_TUNED_CIC, T>(p, canvas, pos, mass, ms, n, st); break;
    case PMX_TUNED_TSC: launch_paint_nd<PMX_TUNED_TSC, T>(p, canvas, pos, mass, ms, n, st); break;
    default: launch_paint_nd<PMX_TUNED_PCS, T>(p, canvas, pos, mass, ms, n, st); break;
    }
}

template <typename T>
static void launch_readout_kind(const pmx_painter &p, const void *canvas, DVec pos, DVec out,
                                int64_t n, hipStream_t st)
{
    switch (p.kind) {
    case PMX_TUNED_NNB: launch_readout_nd<PMX_TUNED_NNB, T>(p, canvas, pos, out, n, st); break;
    case PMX_TUNED_CIC: launch_readout_nd<PMX_TUNED_CIC, T>(p, canvas, pos, out, n, st); break;
    case PMX_TUNED_TSC: launch_readout_nd<PMX_TUNED_TSC, T>(p, canvas, pos, out, n, st); break;
    default: launch_readout_nd<PMX_TUNED_PCS, T>(p, canvas, pos, out, n, st); break;
    }
}

// true if every particle takes the tuned fast path
static bool is_fast(const pmx_painter &p, const pmx_vec *hsml)
{
    if (p.kind < PMX_TUNED_NNB || p.kind > PMX_TUNED_PCS) return false;
    if (hsml && hsml->data) return false;
    WInfo w = winfo_init(native_support(p.kind), (double)p.support);
    return w.support == native_support(p.kind);
}

// device tables of the table-driven kinds, per device
struct TableSlot { double *v = nullptr; int n = 0; double step = 0; };
static TableSlot g_tables[16][PMX_SYM20 + 1];
static std::mutex g_tables_mutex;

static int lookup_table(int kind, TableD *t)
{
    t->v = nullptr; t->n = 0; t->step = 0; t->hsupport = -1.0;
    if (kind < PMX_LANCZOS2) return PMX_OK;
    if (kind >= PMX_DB6) t->hsupport = 0.5 * native_support(kind);
    int dev = 0;
    PMX_HIP_CHECK(hipGetDevice(&dev));
    PMX_REQUIRE(dev < 16, PMX_EUNSUPPORTED, "device index above 15");
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    const TableSlot &sl = g_tables[dev][kind];
    PMX_REQUIRE(sl.v != nullptr, PMX_EINVAL, "table-driven window used before pmx_window_set_table");
    t->v = sl.v; t->n = sl.n; t->step = sl.step;
    return PMX_OK;
}

}  // namespace pmx

using namespace pmx;

extern "C" int pmx_window_set_table(int32_t kind, const double *values, int32_t n, double step)
{
    PMX_REQUIRE(kind >= PMX_LANCZOS2 && kind <= PMX_SYM20, PMX_EINVAL, "not a table-driven window kind");
    PMX_REQUIRE(values != nullptr && n >= 2 && step > 0, PMX_EINVAL, "bad table");
    int dev = 0;
    PMX_HIP_CHECK(hipGetDevice(&dev));
    PMX_REQUIRE(dev < 16, PMX_EUNSUPPORTED, "device index above 15");
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    TableSlot &sl = g_tables[dev][kind];
    if (sl.v) (void)hipFree(sl.v);
    sl.v = nullptr;
    PMX_HIP_CHECK(hipMalloc((void **)&sl.v, sizeof(double) * n));
    PMX_HIP_CHECK(hipMemcpy(sl.v, values, sizeof(double) * n, hipMemcpyHostToDevice));
    sl.n = n;
    sl.step = step;
    return PMX_OK;
}

extern "C" int pmx_window_info(int32_t kind, int32_t support, int32_t *nativesupport,
                               int32_t *eff_support)
{
    int ns = native_support(kind);
    PMX_REQUIRE(ns > 0, PMX_EUNSUPPORTED, "window kind not built");
    WInfo w = winfo_init(ns, (double)support);
    if (nativesupport) *nativesupport = ns;
    if (eff_support) *eff_support = w.support;
    return PMX_OK;
}

static double sinc_unnormed(double x)
{
    // _window_imp.c:13-22
    if (x < 1e-5 && x > -1e-5) {
        double x2 = x * x;
        return 1.0 - x2 / 6. + x2 * x2 / 120.;
    }
    return sin(x) / x;
}

extern "C" int pmx_fwindow(int32_t kind, int32_t support, const double *w, int64_t n, double *out)
{
    int ns = native_support(kind);
    PMX_REQUIRE(ns > 0, PMX_EUNSUPPORTED, "window kind not built");
    WInfo wi = winfo_init(ns, (double)support);
    if (kind >= PMX_LANCZOS2) {
        for (int64_t i = 0; i < n; i++) out[i] = 1.0;   // fwindow == NULL: "not implemented" (_window_imp.c:482-484)
        return PMX_OK;
    }
    for (int64_t i = 0; i < n; i++) {
        // pmesh_painter_get_fwindow (_window_imp.c:473-485): sinc^p at w / vfactor
        double t = sinc_unnormed(0.5 * (w[i] / wi.vfactor));
        double r = t;
        for (int q = 1; q < ns; q++) r = r * t;
        out[i] = r;
    }
    return PMX_OK;
}

static bool canvas_empty(const pmx_painter &p)
{
    for (int d = 0; d < p.ndim; d++)
        if (p.size[d] == 0) return true;
    return false;
}

extern "C" int pmx_paint(const pmx_painter *p_, void *canvas, const pmx_vec *pos,
                         const pmx_vec *mass, double mass_scalar, const pmx_vec *hsml,
                         int64_t npart, void *stream)
{
    int rc = check_painter(p_);
    if (rc) return rc;
    if (npart == 0) return PMX_OK;  // empty pos: no-op
    if (canvas_empty(*p_)) return PMX_OK;  // a rank that holds no cells: every contribution is outside
    PMX_REQUIRE(canvas != nullptr, PMX_EINVAL, "canvas is NULL");
    PMX_REQUIRE(vec_ok(pos) && pos->ncol >= p_->ndim, PMX_EINVAL, "pos must be (n, >=ndim) f4/f8");
    PMX_REQUIRE(!mass || !mass->data || vec_ok(mass), PMX_EINVAL, "mass must be f4/f8");
    PMX_REQUIRE(!hsml || !hsml->data || vec_ok(hsml), PMX_EINVAL, "hsml must be f4/f8");
    pmx_painter p = *p_;
    // effective integer support of the window object (_window_imp.c:456-458)
    p.support = winfo_init(native_support(p.kind), (double)p.support).support;
    hipStream_t st = (hipStream_t)stream;
    DVec dpos = dvec(pos), dmass = dvec(mass), dh = dvec(hsml), none = dvec(nullptr);
    TableD tab;
    rc = lookup_table(p.kind, &tab);
    if (rc) return rc;
    if (is_fast(*p_, hsml)) {
        if (p.canvas_elsize == 8) launch_paint_kind<double>(p, canvas, dpos, dmass, mass_scalar, npart, st);
        else launch_paint_kind<float>(p, canvas, dpos, dmass, mass_scalar, npart, st);
    } else {
        dim3 block(256), grid(grid_for(npart, 256));
        if (p.canvas_elsize == 8)
            general_kernel<double, true><<<grid, block, 0, st>>>(p, (char *)canvas, dpos, dmass, mass_scalar, dh, none, npart, tab);
        else
            general_kernel<float, true><<<grid, block, 0, st>>>(p, (char *)canvas, dpos, dmass, mass_scalar, dh, none, npart, tab);
    }
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_readout(const pmx_painter *p_, const void *canvas, const pmx_vec *pos,
                           const pmx_vec *hsml, const pmx_vec *out, int64_t npart, void *stream)
{
    int rc = check_painter(p_);
    if (rc) return rc;
    if (npart == 0) return PMX_OK;
    // an empty block has no storage: the kernels then write 0 for every particle (all cells outside)
    PMX_REQUIRE(canvas != nullptr || canvas_empty(*p_), PMX_EINVAL, "canvas is NULL");
    PMX_REQUIRE(vec_ok(pos) && pos->ncol >= p_->ndim, PMX_EINVAL, "pos must be (n, >=ndim) f4/f8");
    PMX_REQUIRE(vec_ok(out), PMX_EINVAL, "out must be f4/f8");
    PMX_REQUIRE(!hsml || !hsml->data || vec_ok(hsml), PMX_EINVAL, "hsml must be f4/f8");
    pmx_painter p = *p_;
    p.support = winfo_init(native_support(p.kind), (double)p.support).support;
    hipStream_t st = (hipStream_t)stream;
    DVec dpos = dvec(pos), dh = dvec(hsml), dout = dvec(out), none = dvec(nullptr);
    TableD tab;
    rc = lookup_table(p.kind, &tab);
    if (rc) return rc;
    if (is_fast(*p_, hsml)) {
        if (p.canvas_elsize == 8) launch_readout_kind<double>(p, canvas, dpos, dout, npart, st);
        else launch_readout_kind<float>(p, canvas, dpos, dout, npart, st);
    } else {
        dim3 block(256), grid(grid_for(npart, 256));
        if (p.canvas_elsize == 8)
            general_kernel<double, false><<<grid, block, 0, st>>>(p, (char *)canvas, dpos, none, 0.0, dh, dout, npart, tab);
        else
            general_kernel<float, false><<<grid, block, 0, st>>>(p, (char *)canvas, dpos, none, 0.0, dh, dout, npart, tab);
    }
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

// ---- meshes of more than three dimensions (include/pmesh_amd.h: pmx_painter_nd) --------------------------------------
static int check_painter_nd(const pmx_painter_nd *p)
{
    PMX_REQUIRE(p != nullptr, PMX_EINVAL, "painter is NULL");
    PMX_REQUIRE(p->ndim >= 1 && p->ndim <= PMX_MAXDIM_ND, PMX_EUNSUPPORTED, "ndim must be 1..PMX_MAXDIM_ND");
    PMX_REQUIRE(p->canvas_elsize == 4 || p->canvas_elsize == 8, PMX_EINVAL,
                "canvas must be float or double (_window.pyx:135)");
    PMX_REQUIRE(native_support(p->kind) > 0, PMX_EUNSUPPORTED, "window kind not built");
    return PMX_OK;
}

static bool canvas_empty_nd(const pmx_painter_nd &p)
{
    for (int d = 0; d < p.ndim; d++)
        if (p.size[d] == 0) return true;
    return false;
}

extern "C" int pmx_paint_nd(const pmx_painter_nd *p_, void *canvas, const pmx_vec *pos,
                            const pmx_vec *mass, double mass_scalar, const pmx_vec *hsml,
                            int64_t npart, void *stream)
{
    int rc = check_painter_nd(p_);
    if (rc) return rc;
    if (npart == 0) return PMX_OK;
    if (canvas_empty_nd(*p_)) return PMX_OK;
    PMX_REQUIRE(canvas != nullptr, PMX_EINVAL, "canvas is NULL");
    PMX_REQUIRE(vec_ok(pos) && pos->ncol >= p_->ndim, PMX_EINVAL, "pos must be (n, >=ndim) f4/f8");
    PMX_REQUIRE(!mass || !mass->data || vec_ok(mass), PMX_EINVAL, "mass must be f4/f8");
    PMX_REQUIRE(!hsml || !hsml->data || vec_ok(hsml), PMX_EINVAL, "hsml must be f4/f8");
    pmx_painter_nd p = *p_;
    p.support = winfo_init(native_support(p.kind), (double)p.support).support;
    hipStream_t st = (hipStream_t)stream;
    DVec dpos = dvec(pos), dmass = dvec(mass), dh = dvec(hsml), none = dvec(nullptr);
    TableD tab;
    rc = lookup_table(p.kind, &tab);
    if (rc) return rc;
    dim3 block(256), grid(grid_for(npart, 256));
    if (p.canvas_elsize == 8)
        general_kernel<double, true, pmx_painter_nd, PMX_MAXDIM_ND><<<grid, block, 0, st>>>(p, (char *)canvas, dpos, dmass, mass_scalar, dh, none, npart, tab);
    else
        general_kernel<float, true, pmx_painter_nd, PMX_MAXDIM_ND><<<grid, block, 0, st>>>(p, (char *)canvas, dpos, dmass, mass_scalar, dh, none, npart, tab);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_readout_nd(const pmx_painter_nd *p_, const void *canvas, const pmx_vec *pos,
                              const pmx_vec *hsml, const pmx_vec *out, int64_t npart, void *stream)
{
    int rc = check_painter_nd(p_);
    if (rc) return rc;
    if (npart == 0) return PMX_OK;
    PMX_REQUIRE(canvas != nullptr || canvas_empty_nd(*p_), PMX_EINVAL, "canvas is NULL");
    PMX_REQUIRE(vec_ok(pos) && pos->ncol >= p_->ndim, PMX_EINVAL, "pos must be (n, >=ndim) f4/f8");
    PMX_REQUIRE(vec_ok(out), PMX_EINVAL, "out must be f4/f8");
    PMX_REQUIRE(!hsml || !hsml->data || vec_ok(hsml), PMX_EINVAL, "hsml must be f4/f8");
    pmx_painter_nd p = *p_;
    p.support = winfo_init(native_support(p.kind), (double)p.support).support;
    hipStream_t st = (hipStream_t)stream;
    DVec dpos = dvec(pos), dh = dvec(hsml), dout = dvec(out), none = dvec(nullptr);
    TableD tab;
    rc = lookup_table(p.kind, &tab);
    if (rc) return rc;
    dim3 block(256), grid(grid_for(npart, 256));
    if (p.canvas_elsize == 8)
        general_kernel<double, false, pmx_painter_nd, PMX_MAXDIM_ND><<<grid, block, 0, st>>>(p, (char *)canvas, dpos, none, 0.0, dh, dout, npart, tab);
    else
        general_kernel<float, false, pmx_painter_nd, PMX_MAXDIM_ND><<<grid, block, 0, st>>>(p, (char *)canvas, dpos, none, 0.0, dh, dout, npart, tab);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}
