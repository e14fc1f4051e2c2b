/*
 * pmesh_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's paint / readout / decompose
 * arithmetic, written from the specification in SURVEY.md Appendix A and the
 * reference sources cited per function (paths relative to /root/reference).
 * It exists to CHECK the HIP library: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  Nothing under pmesh_amd/
 * links, imports or falls back to this file.
 *
 * Parity status: PINNED.  tests/test_oracle.py compares every function here
 * (a) against the golden vectors generated from the compiled reference
 * extensions (tests/golden/ npz files, generator tests/golden/make_golden.py),
 * (b) against the reference's inline known answers (pmesh/tests/test_window.py)
 * and (c), in this container, bit-for-bit against oracle/_ref (the reference's
 * own _window_imp.c compiled from where it lies).
 *
 * Must be compiled WITHOUT fp contraction (-ffp-contract=off) so that
 * pos*scale + translate rounds twice as in the reference build (gcc -O2,
 * x86-64 baseline; pmesh/_window_tuned_cic.h:8).
 *
 * The exported functions have the same signatures as the pmx_* entry points
 * of include/pmesh_amd.h, with HOST pointers.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pmesh_amd.h"

/* ------------------------------------------------------------------ helpers */

static inline double ld_real(const void *base, int elsize, int64_t off)
{
    const char *p = (const char *)base + off;
    if (elsize == 8) return *(const double *)p;
    return (double)*(const float *)p;
}

static inline double vec_get(const pmx_vec *v, int64_t i, int c)
{
    return ld_real(v->data, v->elsize, i * v->stride0 + c * v->stride1);
}

/* integer index wrap: the reference uses while-loops (tuned_cic.h:26-32);
 * the modulo gives the same value for every representable index. */
static inline int64_t wrap_idx(int64_t i, int64_t n)
{
    if (n <= 0) return i;
    i %= n;
    if (i < 0) i += n;
    return i;
}

/* (int) floor(x) as the reference casts it (tuned_cic.h:10) */
static inline int ifloor(double x) { return (int)floor(x); }

/* --------------------------------------------------- window bookkeeping (a3) */

typedef struct winfo {
    int support;
    double vfactor;
    double shift;
    int left;
} winfo;

static int native_support(int kind)
{
    /* pmesh_painter_init, _window_imp.c:258-282, 401-454 */
    switch (kind) {
    case PMX_NEAREST: case PMX_TUNED_NNB: return 1;
    case PMX_LINEAR: case PMX_TUNED_CIC: return 2;
    case PMX_QUADRATIC: case PMX_TUNED_TSC: return 3;
    case PMX_CUBIC: case PMX_TUNED_PCS: return 4;
    }
    return -1;
}

/* pmesh_window_info_init, _window_imp.c:24-47 */
static void winfo_init(winfo *w, int nativesupport, double support)
{
    if (support <= 0) {
        w->support = nativesupport;
        support = nativesupport;
    } else {
        w->support = (int)support;
        w->support += (support != (double)w->support); /* round up */
    }
    w->left = (w->support - 1) / 2;
    w->shift = support / 2.0 - w->support / 2;
    w->vfactor = nativesupport / (1. * support);
}

int pmo_window_info(int32_t kind, int32_t support, int32_t *nativesupport, int32_t *eff_support)
{
    int ns = native_support(kind);
    if (ns < 0) return PMX_EUNSUPPORTED;
    winfo w;
    winfo_init(&w, ns, (double)support);
    if (nativesupport) *nativesupport = ns;
    if (eff_support) *eff_support = w.support;
    return PMX_OK;
}

/* ------------------------------------------- analytic kernels (generic path) */

/* _window_imp.c:108-236 */
static double k_nearest(double x) { return (x < 0.5 && x >= -0.5) ? 1.0 : 0.0; }
static double d_nearest(double x) { (void)x; return 0.0; }

static double k_linear(double x)
{
    x = fabs(x);
    return (x < 1.0) ? 1.0 - x : 0.0;
}
static double d_linear(double x)
{
    double factor;
    if (x < 0) { factor = 1; x = -x; }
    else if (x > 0) factor = -1;
    else factor = 0;
    return (x < 1.0) ? factor : 0.0;
}

static double k_quadratic(double x)
{
    x = fabs(x);
    if (x <= 0.5) return 0.75 - x * x;
    if (x < 1.5) { x = 1.5 - x; return (x * x) * 0.5; }
    return 0;
}
static double d_quadratic(double x)
{
    double factor;
    if (x < 0) { x = -x; factor = -1; } else factor = +1;
    if (x <= 0.5) return factor * (-2 * x);
    if (x < 1.5) return factor * (-(1.5 - x));
    return 0;
}

static double k_cubic(double x)
{
    x = fabs(x);
    double xx = x * x;
    if (x < 1.0) return 1.0 / 6.0 * (4 - 6 * xx + 3 * xx * x);
    if (x < 2) return 1.0 / 6.0 * (2 - x) * (2 - x) * (2 - x);
    return 0;
}
static double d_cubic(double x)
{
    double factor;
    if (x < 0) { factor = -1; x = -x; } else factor = +1;
    double xx = x * x;
    if (x < 1.0) return factor * (1.0 / 6.0) * (-12 * x + 9 * xx);
    if (x < 2.0) return factor * (-1.0 / 2.0) * (2 - x) * (2 - x);
    return 0;
}

typedef double (*kfunc)(double);

static void pick_kernel(int kind, kfunc *k, kfunc *d)
{
    switch (kind) {
    case PMX_NEAREST: case PMX_TUNED_NNB: *k = k_nearest; *d = d_nearest; break;
    case PMX_LINEAR: case PMX_TUNED_CIC: *k = k_linear; *d = d_linear; break;
    case PMX_QUADRATIC: case PMX_TUNED_TSC: *k = k_quadratic; *d = d_quadratic; break;
    default: *k = k_cubic; *d = d_cubic; break;
    }
}

/* sinc and the Fourier windows, _window_imp.c:13-22, 121-244 */
static double sinc_unnormed(double x)
{
    if (x < 1e-5 && x > -1e-5) {
        double x2 = x * x;
        return 1.0 - x2 / 6. + x2 * x2 / 120.;
    }
    return sin(x) / x;
}

int pmo_fwindow(int32_t kind, int32_t support, const double *w, int64_t n, double *out)
{
    int ns = native_support(kind);
    if (ns < 0) return PMX_EUNSUPPORTED;
    winfo wi;
    winfo_init(&wi, ns, (double)support);
    for (int64_t i = 0; i < n; i++) {
        /* pmesh_painter_get_fwindow, _window_imp.c:473-485 */
        double t = sinc_unnormed(0.5 * (w[i] / wi.vfactor));
        double r = t;
        for (int p = 1; p < ns; p++) r = r * t; /* t, t*t, t*t*t, t*t*t*t */
        out[i] = r;
    }
    return PMX_OK;
}

/* --------------------------------------------------- tuned per-axis weights */

/* One axis of SETUP_KERNEL_{NNB,CIC,TSC,PCS}: unwrapped indices I[0..S) and
 * weights V[0..S) (tuned_nnb.h:1-27, tuned_cic.h:1-32, tuned_tsc.h:1-37,
 * tuned_pcs.h:1-52).  Every sub-expression is written as in the header. */
static int tuned_axis(int kind, double X, int order, double scale, int *I, double *V)
{
    switch (kind) {
    case PMX_TUNED_NNB:
        I[0] = ifloor(X + 0.5);
        V[0] = (order == 0) ? 1 : 0;
        return 1;
    case PMX_TUNED_CIC:
        I[0] = ifloor(X);
        I[1] = I[0] + 1;
        if (order == 0) {
            V[1] = X - I[0];
            V[0] = 1. - V[1];
        } else {
            V[1] = scale;
            V[0] = -scale;
        }
        return 2;
    case PMX_TUNED_TSC:
        I[1] = ifloor(X + 0.5);
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
        return 3;
    case PMX_TUNED_PCS:
        I[1] = ifloor(X);
        I[0] = I[1] - 1;
        I[2] = I[1] + 1;
        I[3] = I[2] + 1;
        if (order == 0) {
            V[1] = 1.0 / 6.0 * (4 - 6 * (X - I[1]) * (X - I[1])
                                + 3 * (X - I[1]) * (X - I[1]) * (X - I[1]));
            V[2] = 1.0 / 6.0 * (4 - 6 * (X - I[2]) * (X - I[2])
                                - 3 * (X - I[2]) * (X - I[2]) * (X - I[2]));
            V[0] = 1.0 / 6.0 * (2 - (X - I[0])) * (2 - (X - I[0])) * (2 - (X - I[0]));
            V[3] = 1.0 / 6.0 * (2 + (X - I[3])) * (2 + (X - I[3])) * (2 + (X - I[3]));
        } else {
            /* quirk Q1: no scale factor in the tuned PCS derivative */
            V[1] = +1.0 / 6.0 * (-12 * (X - I[1]) + 9 * (X - I[1]) * (X - I[1]));
            V[2] = -1.0 / 6.0 * (+12 * (X - I[2]) + 9 * (X - I[2]) * (X - I[2]));
            V[0] = -1.0 / 2.0 * (2 - (X - I[0])) * (2 - (X - I[0]));
            V[3] = +1.0 / 2.0 * (2 + (X - I[3])) * (2 + (X - I[3]));
        }
        return 4;
    }
    return 0;
}

static inline void canvas_add(void *canvas, int elsize, int64_t off, double f)
{
    char *p = (char *)canvas + off;
    if (elsize == 8) *(double *)p += f;
    else *(float *)p += f; /* (float)((double)*p + f), quirk Q7 */
}

static inline double canvas_get(const void *canvas, int elsize, int64_t off)
{
    return ld_real(canvas, elsize, off);
}

/* _{nnb,cic,tsc,pcs}_tuned_paint{1,2,3} */
static void tuned_paint(const pmx_painter *p, void *canvas, const double *pos, double weight)
{
    int I[PMX_MAXDIM][4];
    double V[PMX_MAXDIM][4];
    int S = 0;
    int nd = p->ndim;
    for (int d = 0; d < nd; d++) {
        double X = pos[d] * p->scale[d] + p->translate[d];
        S = tuned_axis(p->kind, X, p->order[d], p->scale[d], I[d], V[d]);
        for (int a = 0; a < S; a++) I[d][a] = (int)wrap_idx(I[d][a], p->period[d]);
    }
    for (int a = 0; a < S; a++) V[0][a] *= weight;

    int Sb = nd > 1 ? S : 1, Sc = nd > 2 ? S : 1;
    for (int a = 0; a < S; a++)
        for (int b = 0; b < Sb; b++)
            for (int c = 0; c < Sc; c++) {
                double f = V[0][a];
                int64_t off;
                if (I[0][a] < 0 || I[0][a] >= p->size[0]) continue;
                off = I[0][a] * p->strides[0];
                if (nd > 1) {
                    f = f * V[1][b];
                    if (I[1][b] < 0 || I[1][b] >= p->size[1]) continue;
                    off += I[1][b] * p->strides[1];
                }
                if (nd > 2) {
                    f = f * V[2][c];
                    if (I[2][c] < 0 || I[2][c] >= p->size[2]) continue;
                    off += I[2][c] * p->strides[2];
                }
                canvas_add(canvas, p->canvas_elsize, off, f);
            }
}

static double tuned_readout(const pmx_painter *p, const void *canvas, const double *pos)
{
    int I[PMX_MAXDIM][4];
    double V[PMX_MAXDIM][4];
    int S = 0;
    int nd = p->ndim;
    for (int d = 0; d < nd; d++) {
        double X = pos[d] * p->scale[d] + p->translate[d];
        S = tuned_axis(p->kind, X, p->order[d], p->scale[d], I[d], V[d]);
        for (int a = 0; a < S; a++) I[d][a] = (int)wrap_idx(I[d][a], p->period[d]);
    }
    double value = 0;
    int Sb = nd > 1 ? S : 1, Sc = nd > 2 ? S : 1;
    for (int a = 0; a < S; a++)
        for (int b = 0; b < Sb; b++)
            for (int c = 0; c < Sc; c++) {
                double f = V[0][a];
                int64_t off;
                int inside = !(I[0][a] < 0 || I[0][a] >= p->size[0]);
                off = I[0][a] * p->strides[0];
                if (nd > 1) {
                    f = f * V[1][b];
                    inside &= !(I[1][b] < 0 || I[1][b] >= p->size[1]);
                    off += I[1][b] * p->strides[1];
                }
                if (nd > 2) {
                    f = f * V[2][c];
                    inside &= !(I[2][c] < 0 || I[2][c] >= p->size[2]);
                    off += I[2][c] * p->strides[2];
                }
                /* _REd3 returns 0 outside; value += 0 keeps the reference's sum */
                value += inside ? canvas_get(canvas, p->canvas_elsize, off) * f : 0;
            }
    return value;
}

/* -------------------------------------------------------- generic path (a8) */

/* _fill_k, _window_imp.c:50-83 */
static void fill_k(const pmx_painter *p, const winfo *w, kfunc kern, kfunc diff,
                   const double *pos, int *ipos, double *k)
{
    for (int d = 0; d < p->ndim; d++) {
        double *kd = &k[w->support * d];
        double g = pos[d] * p->scale[d] + p->translate[d];
        ipos[d] = floor(g + w->shift) - w->left;
        double dx = g - ipos[d];
        for (int i = 0; i < w->support; i++) {
            double x = (dx - i) * w->vfactor;
            if (p->order[d] == 0) kd[i] = kern(x) * w->vfactor;
            else kd[i] = diff(x) * p->scale[d] * w->vfactor * w->vfactor;
        }
    }
}

static int is_tuned(int kind) { return kind >= PMX_TUNED_NNB && kind <= PMX_TUNED_PCS; }

/* _generic_paint / _generic_readout, _window_generics.h:4-142.  mode 0 paint. */
static double generic_one(const pmx_painter *p, void *canvas, const double *pos, double weight,
                          double hsml, int paint)
{
    winfo w;
    int ns = native_support(p->kind);
    /* painter->support was replaced by its effective integer value in
     * pmesh_painter_init (_window_imp.c:456-458) before this product */
    winfo w0;
    winfo_init(&w0, ns, (double)p->support);
    winfo_init(&w, ns, w0.support * hsml);

    /* fast path: tuned kind, ndim<=3, integer support equals native (quirk Q6) */
    if (is_tuned(p->kind) && p->ndim <= 3 && w.support == ns) {
        if (paint) { tuned_paint(p, canvas, pos, weight); return 0; }
        return tuned_readout(p, canvas, pos);
    }

    kfunc kern, diff;
    pick_kernel(p->kind, &kern, &diff);
    int ipos[PMX_MAXDIM];
    double k[PMX_MAXDIM * PMX_MAXSUPPORT];
    if (w.support > PMX_MAXSUPPORT) return 0;
    fill_k(p, &w, kern, diff, pos, ipos, k);

    int rel[PMX_MAXDIM] = {0, 0, 0};
    int s2 = w.support;
    int nd = p->ndim;
    double value = 0;
    while (rel[0] != s2) {
        double kernel = 1.0;
        int64_t ind = 0;
        int outside = 0;
        for (int d = 0; d < nd; d++) {
            int r = rel[d];
            int64_t t = ipos[d] + r;
            kernel *= k[w.support * d + r];
            t = wrap_idx(t, p->period[d]);
            if (t >= p->size[d] || t < 0) { outside = 1; break; }
            ind += p->strides[d] * t;
        }
        if (!outside) {
            if (paint) canvas_add(canvas, p->canvas_elsize, ind, weight * kernel);
            else value += kernel * canvas_get(canvas, p->canvas_elsize, ind);
        }
        rel[nd - 1]++;
        for (int d = nd - 1; d > 0; d--)
            if (rel[d] == s2) { rel[d - 1]++; rel[d] = 0; }
    }
    return value;
}

/* -------------------------------------------------------------- entry points */

static int check_painter(const pmx_painter *p)
{
    if (!p) return PMX_EINVAL;
    if (p->ndim < 1 || p->ndim > PMX_MAXDIM) return PMX_EUNSUPPORTED;
    if (p->canvas_elsize != 4 && p->canvas_elsize != 8) return PMX_EINVAL;
    if (native_support(p->kind) < 0) return PMX_EUNSUPPORTED;
    return PMX_OK;
}

/* the particle loop of _window.pyx:157-165 */
int pmo_paint(const pmx_painter *p, void *canvas, const pmx_vec *pos, const pmx_vec *mass,
              double mass_scalar, const pmx_vec *hsml, int64_t npart, void *stream)
{
    (void)stream;
    int rc = check_painter(p);
    if (rc) return rc;
    for (int64_t i = 0; i < npart; i++) {
        double x[PMX_MAXDIM];
        for (int d = 0; d < p->ndim; d++) x[d] = vec_get(pos, i, d);
        double m = (mass && mass->data) ? vec_get(mass, i, 0) : mass_scalar;
        double h = (hsml && hsml->data) ? vec_get(hsml, i, 0) : 1.0;
        generic_one(p, canvas, x, m, h, 1);
    }
    return PMX_OK;
}

/* the particle loop of _window.pyx:198-205 */
int pmo_readout(const pmx_painter *p, const void *canvas, const pmx_vec *pos, const pmx_vec *hsml,
                const pmx_vec *out, int64_t npart, void *stream)
{
    (void)stream;
    int rc = check_painter(p);
    if (rc) return rc;
    for (int64_t i = 0; i < npart; i++) {
        double x[PMX_MAXDIM];
        for (int d = 0; d < p->ndim; d++) x[d] = vec_get(pos, i, d);
        double h = (hsml && hsml->data) ? vec_get(hsml, i, 0) : 1.0;
        double v = generic_one(p, (void *)canvas, x, 0.0, h, 0);
        char *o = (char *)out->data + i * out->stride0;
        if (out->elsize == 8) *(double *)o = v;
        else *(float *)o = (float)v;
    }
    return PMX_OK;
}

/* ---- meshes of more than three dimensions (include/pmesh_amd.h: pmx_painter_nd) ----
 * _generic_paint / _generic_readout (_window_generics.h:4-142) with _fill_k (_window_imp.c:50-83) for any ndim up
 * to PMX_MAXDIM_ND; the reference's tuned fast path exists for ndim <= 3 only (_window_imp.c:486-520), so every
 * kind takes the generic product here.  mode 0 paint. */
static double generic_one_nd(const pmx_painter_nd *p, void *canvas, const double *pos, double weight,
                             double hsml, int paint)
{
    winfo w, w0;
    int ns = native_support(p->kind);
    winfo_init(&w0, ns, (double)p->support);
    winfo_init(&w, ns, w0.support * hsml);
    kfunc kern, diff;
    pick_kernel(p->kind, &kern, &diff);
    int ipos[PMX_MAXDIM_ND];
    double k[PMX_MAXDIM_ND * PMX_MAXSUPPORT];
    if (w.support > PMX_MAXSUPPORT) return 0;
    int nd = p->ndim;
    for (int d = 0; d < nd; d++) {
        double *kd = &k[w.support * d];
        double g = pos[d] * p->scale[d] + p->translate[d];
        ipos[d] = floor(g + w.shift) - w.left;
        double dx = g - ipos[d];
        for (int i = 0; i < w.support; i++) {
            double x = (dx - i) * w.vfactor;
            if (p->order[d] == 0) kd[i] = kern(x) * w.vfactor;
            else kd[i] = diff(x) * p->scale[d] * w.vfactor * w.vfactor;
        }
    }
    int rel[PMX_MAXDIM_ND] = {0};
    int s2 = w.support;
    double value = 0;
    while (rel[0] != s2) {
        double kernel = 1.0;
        int64_t ind = 0;
        int outside = 0;
        for (int d = 0; d < nd; d++) {
            int r = rel[d];
            int64_t t = ipos[d] + r;
            kernel *= k[w.support * d + r];
            t = wrap_idx(t, p->period[d]);
            if (t >= p->size[d] || t < 0) { outside = 1; break; }
            ind += p->strides[d] * t;
        }
        if (!outside) {
            if (paint) canvas_add(canvas, p->canvas_elsize, ind, weight * kernel);
            else value += kernel * canvas_get(canvas, p->canvas_elsize, ind);
        }
        rel[nd - 1]++;
        for (int d = nd - 1; d > 0; d--)
            if (rel[d] == s2) { rel[d - 1]++; rel[d] = 0; }
    }
    return value;
}

static int check_painter_nd(const pmx_painter_nd *p)
{
    if (!p) return PMX_EINVAL;
    if (p->ndim < 1 || p->ndim > PMX_MAXDIM_ND) return PMX_EUNSUPPORTED;
    if (p->canvas_elsize != 4 && p->canvas_elsize != 8) return PMX_EINVAL;
    if (native_support(p->kind) < 0) return PMX_EUNSUPPORTED;
    return PMX_OK;
}

int pmo_paint_nd(const pmx_painter_nd *p, void *canvas, const pmx_vec *pos, const pmx_vec *mass,
                 double mass_scalar, const pmx_vec *hsml, int64_t npart, void *stream)
{
    (void)stream;
    int rc = check_painter_nd(p);
    if (rc) return rc;
    for (int64_t i = 0; i < npart; i++) {
        double x[PMX_MAXDIM_ND];
        for (int d = 0; d < p->ndim; d++) x[d] = vec_get(pos, i, d);
        double m = (mass && mass->data) ? vec_get(mass, i, 0) : mass_scalar;
        double h = (hsml && hsml->data) ? vec_get(hsml, i, 0) : 1.0;
        generic_one_nd(p, canvas, x, m, h, 1);
    }
    return PMX_OK;
}

int pmo_readout_nd(const pmx_painter_nd *p, const void *canvas, const pmx_vec *pos, const pmx_vec *hsml,
                   const pmx_vec *out, int64_t npart, void *stream)
{
    (void)stream;
    int rc = check_painter_nd(p);
    if (rc) return rc;
    for (int64_t i = 0; i < npart; i++) {
        double x[PMX_MAXDIM_ND];
        for (int d = 0; d < p->ndim; d++) x[d] = vec_get(pos, i, d);
        double h = (hsml && hsml->data) ? vec_get(hsml, i, 0) : 1.0;
        double v = generic_one_nd(p, (void *)canvas, x, 0.0, h, 0);
        char *o = (char *)out->data + i * out->stride0;
        if (out->elsize == 8) *(double *)o = v;
        else *(float *)o = (float)v;
    }
    return PMX_OK;
}

/* ------------------------------------------------------------- decomposition */

/* numpy's float remainder (npy_divmod): result carries the sign of b */
static double np_remainder(double a, double b)
{
    double mod = fmod(a, b);
    if (!b) return mod;
    if (mod) {
        if ((b < 0) != (mod < 0)) mod += b;
    } else {
        mod = copysign(0, b);
    }
    return mod;
}

/* numpy.digitize(x, bins, right=False) for increasing bins
 * == searchsorted(bins, x, side='right') */
static int np_digitize(double x, const double *bins, int n)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (x < bins[mid]) hi = mid;   /* NaN compares false: goes right, as numpy sorts NaN last */
        else lo = mid + 1;
    }
    return lo;
}

static int py_mod(int a, int n)
{
    int r = a % n;
    if (r < 0) r += n;
    return r;
}

/* domain.py:608-630: [sil, sir) of one particle along one axis, as int16 */
static void classify_axis(const pmx_grid *g, int j, double x, double s, int16_t *sil, int16_t *sir)
{
    const double *edges = g->edges[j];
    int ne = g->shape[j] + 1;
    if (g->periodic) {
        double box = edges[ne - 1];
        double c = np_remainder(x, box);
        int l = np_digitize(np_remainder(c - s, box), edges, ne);
        int r = np_digitize(np_remainder(c + s, box), edges, ne);
        int p = np_digitize(c, edges, ne);
        l = p - py_mod(p - l, g->shape[j]) - 1;
        r = p + py_mod(r - p, g->shape[j]);
        *sil = (int16_t)l;
        *sir = (int16_t)r;
    } else {
        int l = np_digitize(x - s, edges, ne);
        int r = np_digitize(x + s, edges, ne);
        l = l - 1;
        if (l < 0) l = 0;
        if (l > g->shape[j]) l = g->shape[j];
        if (r < 0) r = 0;
        if (r > g->shape[j]) r = g->shape[j];
        *sil = (int16_t)l;
        *sir = (int16_t)r;
    }
}

/* gridnd_fill (_domain.pyx:62-118) for one particle: sorted unique target
 * ranks as a bit mask (nranks <= 64). */
static uint64_t particle_targets(const pmx_grid *g, const int16_t *sil, const int16_t *sir)
{
    int nd = g->ndim;
    int strides[PMX_MAXDIM];
    strides[nd - 1] = 1;
    for (int j = nd - 2; j >= 0; j--) strides[j] = strides[j + 1] * g->shape[j + 1];
    int64_t patch = 1;
    int p[PMX_MAXDIM];
    for (int j = 0; j < nd; j++) {
        patch *= sir[j] - sil[j];
        p[j] = sil[j];
    }
    uint64_t mask = 0;
    for (int64_t kkk = 0; kkk < patch; kkk++) {
        int64_t target = 0;
        for (int j = 0; j < nd; j++) {
            int t = p[j];
            if (g->periodic) t = (int)wrap_idx(t, g->shape[j]);
            target += (int64_t)t * strides[j];
        }
        target = g->assign[target];
        /* quirk Q3: indexed by rank after the DomainAssign lookup */
        if (!g->degenerate[target]) mask |= (uint64_t)1 << target;
        p[nd - 1]++;
        for (int jj = nd - 1; jj > 0; jj--) {
            if (p[jj] == sir[jj]) { p[jj] = sil[jj]; p[jj - 1]++; }
            else break;
        }
    }
    return mask;
}

int pmo_decompose_count(const pmx_grid *g, const pmx_vec *pos, const double *scale,
                        const double *smoothing, int64_t npart, uint64_t *masks,
                        int64_t *counts, void *stream)
{
    (void)stream;
    if (!g || g->ndim < 1 || g->ndim > PMX_MAXDIM) return PMX_EINVAL;
    if (g->nranks > PMX_MAXRANKS) return PMX_EUNSUPPORTED;
    for (int r = 0; r < g->nranks; r++) counts[r] = 0;
    for (int64_t i = 0; i < npart; i++) {
        int16_t sil[PMX_MAXDIM], sir[PMX_MAXDIM];
        for (int j = 0; j < g->ndim; j++) {
            /* transform0 (pm.py:1788-1790): scale * x in double */
            double x = scale[j] * vec_get(pos, i, j);
            classify_axis(g, j, x, smoothing[j], &sil[j], &sir[j]);
        }
        uint64_t m = particle_targets(g, sil, sir);
        masks[i] = m;
        for (int r = 0; r < g->nranks; r++)
            if (m >> r & 1) counts[r]++;
    }
    return PMX_OK;
}

int pmo_decompose_fill(int32_t nranks, const uint64_t *masks, int64_t npart,
                       const int64_t *offsets, void *indices, int32_t index_elsize, void *stream)
{
    (void)stream;
    int64_t cur[PMX_MAXRANKS];
    if (nranks > PMX_MAXRANKS) return PMX_EUNSUPPORTED;
    for (int r = 0; r < nranks; r++) cur[r] = offsets[r];
    for (int64_t i = 0; i < npart; i++) {
        uint64_t m = masks[i];
        for (int r = 0; r < nranks; r++) {
            if (!(m >> r & 1)) continue;
            if (index_elsize == 8) ((int64_t *)indices)[cur[r]] = i;
            else ((int32_t *)indices)[cur[r]] = (int32_t)i;
            cur[r]++;
        }
    }
    return PMX_OK;
}

/* debugging aid for the tests: the raw [sil, sir) intervals of domain.py:618-621 */
int pmo_decompose_intervals(const pmx_grid *g, const pmx_vec *pos, const double *scale,
                            const double *smoothing, int64_t npart, int16_t *sil, int16_t *sir)
{
    for (int64_t i = 0; i < npart; i++)
        for (int j = 0; j < g->ndim; j++)
            classify_axis(g, j, scale[j] * vec_get(pos, i, j), smoothing[j],
                          &sil[j * npart + i], &sir[j * npart + i]);
    return PMX_OK;
}

int pmo_take_rows(const void *src, int64_t src_stride0, int64_t row_bytes, const void *indices,
                  int32_t index_elsize, int64_t nrows, void *dst, void *stream)
{
    (void)stream;
    for (int64_t j = 0; j < nrows; j++) {
        int64_t i = index_elsize == 8 ? ((const int64_t *)indices)[j] : ((const int32_t *)indices)[j];
        memcpy((char *)dst + j * row_bytes, (const char *)src + i * src_stride0, row_bytes);
    }
    return PMX_OK;
}

/* the same into rows dst_stride bytes apart (a column of packed rows; domain.py:59-80 pack_arrays +
 * 188 take); indices == NULL: row j of the source */
int pmo_pack_rows(const void *src, int64_t src_stride0, int64_t row_bytes, const void *indices,
                  int32_t index_elsize, int64_t nrows, void *dst, int64_t dst_stride, void *stream)
{
    (void)stream;
    for (int64_t j = 0; j < nrows; j++) {
        int64_t i = !indices ? j : (index_elsize == 8 ? ((const int64_t *)indices)[j] : ((const int32_t *)indices)[j]);
        memcpy((char *)dst + j * dst_stride, (const char *)src + i * src_stride0, row_bytes);
    }
    return PMX_OK;
}

/* bincountv (domain.py:26-48): numpy.bincount accumulates the weights in
 * double, in ascending j, and the per-bin totals are then cast to the output
 * dtype (out[ind] = bincount(...) overwrites all nout rows). */
int pmo_scatter_add(const void *values, int32_t elsize, int32_t ncol, const void *indices,
                    int32_t index_elsize, int64_t nrows, void *out, int64_t nout, void *stream)
{
    (void)stream;
    if (nout == 0) {        /* accumulate into what `out` already holds (include/pmesh_amd.h) */
        for (int64_t j = 0; j < nrows; j++) {
            int64_t i = index_elsize == 8 ? ((const int64_t *)indices)[j] : ((const int32_t *)indices)[j];
            if (i < 0) return PMX_EINVAL;
            for (int c = 0; c < ncol; c++) {
                if (elsize == 8) ((double *)out)[i * ncol + c] += ((const double *)values)[j * ncol + c];
                else ((float *)out)[i * ncol + c] += ((const float *)values)[j * ncol + c];
            }
        }
        return PMX_OK;
    }
    double *acc = (double *)calloc((size_t)(nout * ncol) + 1, sizeof(double));
    if (!acc) return PMX_ENOMEM;
    for (int64_t j = 0; j < nrows; j++) {
        int64_t i = index_elsize == 8 ? ((const int64_t *)indices)[j] : ((const int32_t *)indices)[j];
        if (i < 0 || i >= nout) { free(acc); return PMX_EINVAL; }
        for (int c = 0; c < ncol; c++) {
            double v = elsize == 8 ? ((const double *)values)[j * ncol + c]
                                   : (double)((const float *)values)[j * ncol + c];
            acc[i * ncol + c] += v;
        }
    }
    for (int64_t q = 0; q < nout * ncol; q++) {
        if (elsize == 8) ((double *)out)[q] = acc[q];
        else ((float *)out)[q] = (float)acc[q];
    }
    free(acc);
    return PMX_OK;
}

/* ------------------------------------------------------------ apply-transfer */

/* Field.apply with the transfer functions of examples/nbody.py:154-181 written
 * out per mode; arithmetic order follows the numpy expressions there. */
int pmo_apply_transfer(const pmx_transfer *t, int32_t ndim, int32_t elsize, const void *in,
                       const int64_t *in_strides, void *out, const int64_t *out_strides,
                       const int64_t *shape, const int64_t *start, const int64_t *nmesh,
                       const double *boxsize, void *stream)
{
    (void)stream;
    int64_t n[3] = {1, 1, 1}, is[3] = {0, 0, 0}, os[3] = {0, 0, 0}, st[3] = {0, 0, 0};
    int64_t nm[3] = {1, 1, 1};
    double L[3] = {1, 1, 1};
    for (int d = 0; d < ndim; d++) {
        n[d] = shape[d]; is[d] = in_strides[d]; os[d] = out_strides[d];
        st[d] = start[d]; nm[d] = nmesh[d]; L[d] = boxsize[d];
    }
    for (int64_t i = 0; i < n[0]; i++)
        for (int64_t j = 0; j < n[1]; j++)
            for (int64_t k = 0; k < n[2]; k++) {
                int64_t idx[3] = {i + st[0], j + st[1], k + st[2]};
                double kk[3] = {0, 0, 0}, ww[3] = {0, 0, 0};
                double k2 = 0;
                for (int d = 0; d < ndim; d++) {
                    /* _init_o_coords, pm.py:1200-1226 */
                    double wi = (double)idx[d];
                    if (idx[d] >= nm[d] / 2) wi -= nm[d];
                    wi *= (2 * M_PI / nm[d]);
                    ww[d] = wi;
                    kk[d] = wi * nm[d] / L[d];
                    k2 += kk[d] * kk[d];
                }
                double re = t->amplitude, im = 0;
                if (t->laplace_pow) {
                    double q = (k2 == 0) ? 1.0 : k2;
                    re *= pow(q, (double)t->laplace_pow);
                }
                if (t->gauss_r != 0) re *= exp(-0.5 * k2 * t->gauss_r * t->gauss_r);
                if (t->deconv_pow) {
                    for (int d = 0; d < ndim; d++) {
                        double s = sinc_unnormed(0.5 * ww[d]);
                        re /= pow(s, (double)t->deconv_pow);
                    }
                }
                if (t->grad_dir >= 0) {
                    int d = t->grad_dir;
                    double D;
                    if (t->grad_kind == 0) D = kk[d];
                    else {
                        double C = L[d] / nm[d];
                        double w = kk[d] * C;
                        D = 1.0 / C * 1 / 6.0 * (8 * sin(w) - sin(2 * w));
                    }
                    im = re * D;
                    re = 0;
                }
                int64_t io = i * is[0] + j * is[1] + k * is[2];
                int64_t oo = i * os[0] + j * os[1] + k * os[2];
                if (elsize == 8) {
                    const double *a = (const double *)((const char *)in + io);
                    double *b = (double *)((char *)out + oo);
                    double ar = a[0], ai = a[1];
                    b[0] = re * ar - im * ai;
                    b[1] = re * ai + im * ar;
                } else {
                    const float *a = (const float *)((const char *)in + io);
                    float *b = (float *)((char *)out + oo);
                    double ar = a[0], ai = a[1];
                    b[0] = (float)(re * ar - im * ai);
                    b[1] = (float)(re * ai + im * ar);
                }
            }
    return PMX_OK;
}

/* ------------------------------------------------------- synthetic particles */

static inline uint64_t mix64(uint64_t z)
{
    /* splitmix64 finaliser */
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline void vec_set(const pmx_vec *v, int64_t i, int c, double x)
{
    char *p = (char *)v->data + i * v->stride0 + c * v->stride1;
    if (v->elsize == 8) *(double *)p = x;
    else *(float *)p = (float)x;
}

int pmo_synth_uniform(const pmx_vec *pos, int64_t nlat, double boxsize, uint64_t seed, int64_t g0,
                      int64_t npart, void *stream)
{
    (void)stream;
    double h = boxsize / nlat;
    for (int64_t n = 0; n < npart; n++) {
        int64_t g = g0 + n;
        int64_t ijk[3] = {g / (nlat * nlat), (g / nlat) % nlat, g % nlat};
        for (int a = 0; a < 3; a++) {
            uint64_t r = mix64(seed ^ (uint64_t)(3 * g + a));
            double u = (double)(r >> 11) * (1.0 / 9007199254740992.0); /* 2^-53 */
            double x = (ijk[a] + 0.5) * h + (u - 0.5) * 0.8 * h;
            vec_set(pos, n, a, x);
        }
    }
    return PMX_OK;
}

int pmo_synth_clustered(const pmx_vec *pos, int64_t nlat, double boxsize, const double *modes,
                        int32_t nmodes, double shift, int64_t g0, int64_t npart, void *stream)
{
    (void)stream;
    double h = boxsize / nlat;
    for (int64_t n = 0; n < npart; n++) {
        int64_t g = g0 + n;
        int64_t ijk[3] = {g / (nlat * nlat), (g / nlat) % nlat, g % nlat};
        double q[3], x[3];
        for (int a = 0; a < 3; a++) { q[a] = (ijk[a] + 0.5 + shift) * h; x[a] = q[a]; }
        for (int m = 0; m < nmodes; m++) {
            const double *md = modes + 8 * m;
            double ph = 2 * M_PI * (md[0] * q[0] + md[1] * q[1] + md[2] * q[2]) / boxsize + md[7];
            double s = md[6] * sin(ph);
            for (int a = 0; a < 3; a++) x[a] += s * md[3 + a];
        }
        for (int a = 0; a < 3; a++) {
            double y = fmod(x[a], boxsize);
            if (y < 0) y += boxsize;
            vec_set(pos, n, a, y);
        }
    }
    return PMX_OK;
}

/* ------------------------------------------------------------ white noise */

/* Gadget / N-GenIC compatible white noise in Fourier space
 * (pmesh/_whitenoise_generics.h:29-238, _whitenoise_imp.c:21-105, pm.py:1656-1696).
 *
 * Random numbers: Luescher's double-precision RANLUX ("ranlxd", luxury level 1: 202
 * subtract-with-borrow steps per 12 delivered numbers; M. Luescher, Comput. Phys. Commun. 79
 * (1994) 100, and the v2 double-precision variant), seeded as GSL's gsl_rng_ranlxd1 does
 * (pmesh/gsl/ranlxd.c:191-235): 31 seed bits feed a lagged-Fibonacci bit generator
 * (lags 31, 13) that fills twelve 48-bit fractions.  State: x[0..11] in (2^-48)Z, a borrow,
 * the ring position `ir` (the oldest entry, next to be replaced; the short lag is 7 slots
 * ahead).  One step: y = x[ir+7] - x[ir] - borrow; y < 0 -> y += 1, borrow = 2^-48.  A draw
 * advances ir and returns x[ir]; when ir comes back to where the last refill ended, 202 steps
 * are made first (ranlxd.c:63-171: its three loops are exactly 202 such steps). */
typedef struct {
    double x[12];
    double borrow;
    int ir, ir_refill;
} rlx_t;

static void rlx_seed(rlx_t *s, unsigned long seed)
{
    const double ulp48 = 1.0 / 281474976710656.0;
    int bits[31];
    if (seed == 0) seed = 1;
    int v = (int)(seed & 0xFFFFFFFFUL);          /* an int in the reference: bit 31 makes it negative */
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
        s->x[k] = ulp48 * acc;
    }
    s->borrow = 0;
    s->ir = 11;
    s->ir_refill = 0;
}

static double rlx_uniform(rlx_t *s)
{
    const double ulp48 = 1.0 / 281474976710656.0;
    s->ir = (s->ir + 1) % 12;
    if (s->ir == s->ir_refill) {
        int ir = s->ir;
        for (int k = 0; k < 202; k++) {
            double y = s->x[(ir + 7) % 12] - s->x[ir];
            y = y - s->borrow;
            if (y < 0) { s->borrow = ulp48; y += 1; } else s->borrow = 0;
            s->x[ir] = y;
            ir = (ir + 1) % 12;
        }
        s->ir = ir;
        s->ir_refill = ir;
    }
    return s->x[s->ir];
}

/* one (phase, amplitude) draw: _whitenoise_imp.c:21-27 */
static void wn_sample(rlx_t *s, double *ampl, double *phase)
{
    *phase = rlx_uniform(s) * 2 * M_PI;
    do *ampl = rlx_uniform(s); while (*ampl == 0);
}

typedef struct {
    int64_t nmesh[3], start[3], size[3];
    uint32_t *table[2][2];      /* seeds of column (i, j) and of its mirror images */
} wn_seeds;

/* _whitenoise_imp.c:30-53: the next number of the master stream seeds column (i, j) and its
 * conjugate partners (N0-i, j), (i, N1-j), (N0-i, N1-j); each goes to the table of the
 * quadrant pairing it was reached through, if that column is held locally */
static void wn_assign(wn_seeds *w, int i, int j, rlx_t *master)
{
    unsigned int seed = 0x7fffffff * rlx_uniform(master);
    int ii[2] = {i, (int)((w->nmesh[0] - i) % w->nmesh[0])};
    int jj[2] = {j, (int)((w->nmesh[1] - j) % w->nmesh[1])};
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) {
            int64_t li = ii[a] - w->start[0], lj = jj[b] - w->start[1];
            if (li >= 0 && li < w->size[0] && lj >= 0 && lj < w->size[1])
                w->table[a][b][li * w->size[1] + lj] = seed;
        }
}

int pmo_whitenoise(uint32_t seed, int32_t unitary, const int64_t *nmesh, const int64_t *start,
                   const int64_t *size, const int64_t *strides, int32_t elsize, void *canvas, void *stream)
{
    (void)stream;
    if (elsize != 8 && elsize != 16) return PMX_EINVAL;
    wn_seeds w;
    for (int d = 0; d < 3; d++) { w.nmesh[d] = nmesh[d]; w.start[d] = start[d]; w.size[d] = size[d]; }
    size_t ncol = (size_t)(size[0] * size[1]);
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) {
            w.table[a][b] = (uint32_t *)calloc(ncol + 1, sizeof(uint32_t));
            if (!w.table[a][b]) return PMX_ENOMEM;
        }
    const int N0 = (int)nmesh[0], N1 = (int)nmesh[1], N2 = (int)nmesh[2];
    /* the master stream walks rings of growing index i over the four corners of the (i, j)
     * plane (_whitenoise_generics.h:73-90; N-GenIC's order, which makes the large scales of a
     * finer mesh equal those of a coarser one).  The mixed use of N0 / N1 is the reference's. */
    rlx_t master;
    rlx_seed(&master, seed);
    for (int i = 0; i < N0 / 2; i++) {
        for (int j = 0; j < i; j++) wn_assign(&w, i, j, &master);
        for (int j = 0; j < i + 1; j++) wn_assign(&w, j, i, &master);
        for (int j = 0; j < i; j++) wn_assign(&w, N0 - 1 - i, j, &master);
        for (int j = 0; j < i + 1; j++) wn_assign(&w, N1 - 1 - j, i, &master);
        for (int j = 0; j < i; j++) wn_assign(&w, i, N1 - 1 - j, &master);
        for (int j = 0; j < i + 1; j++) wn_assign(&w, j, N0 - 1 - i, &master);
        for (int j = 0; j < i; j++) wn_assign(&w, N0 - 1 - i, N1 - 1 - j, &master);
        for (int j = 0; j < i + 1; j++) wn_assign(&w, N1 - 1 - j, N0 - 1 - i, &master);
    }
    /* half spectrum requested (no local mode with k2 > N2/2): one pass, sign +1; otherwise
     * the negative half first, then the positive one (which then owns the Nyquist plane) */
    int full = 0;
    for (int64_t k = N2 / 2 + 1; k < N2; k++)
        if (k - start[2] >= 0 && k - start[2] < size[2]) { full = 1; break; }
    const int signs[2] = {full ? -1 : 1, full ? 1 : 0};

    for (int64_t i = start[0]; i < start[0] + size[0]; i++) {
        int64_t ci = (N0 - i) % N0;
        for (int64_t j = start[1]; j < start[1] + size[1]; j++) {
            int64_t cj = (N1 - j) % N1;
            /* columns in the "upper" half take the k2 = 0 and k2 = N2/2 planes from the
             * generator of their mirror column and conjugate (generics.h:121-131) */
            int mirror = (ci == i && cj < j) || (ci < i && cj != j) || (ci < i && cj == j);
            int64_t col = (i - start[0]) * size[1] + (j - start[1]);
            for (int is = 0; is < 2 && signs[is] != 0; is++) {
                int sign = signs[is];
                rlx_t lower, self;
                rlx_seed(&lower, w.table[mirror][mirror][col]);
                rlx_seed(&self, sign == 1 ? w.table[0][0][col] : w.table[1][1][col]);
                for (int64_t k = 0; k <= N2 / 2; k++) {
                    int use_conj = mirror && (k == 0 || k == N2 / 2);
                    double ampl, phase;
                    if (use_conj) { wn_sample(&self, &ampl, &phase); wn_sample(&lower, &ampl, &phase); }
                    else { wn_sample(&lower, &ampl, &phase); wn_sample(&self, &ampl, &phase); }
                    int64_t k2 = sign == -1 ? N2 - k : k;
                    /* generics.h:158-166 tests the UNREFLECTED k for membership, then writes at the
                     * reflected index if that is inside the block (generics.h:11-27) */
                    if (!(k - start[2] >= 0 && k - start[2] < size[2])) continue;
                    ampl = unitary ? 1.0 : sqrt(-log(ampl));
                    /* two adjacent calls, as in the reference: gcc pairs them into one sincos(),
                     * whose last bit can differ from separate cos() / sin() calls */
                    double re = ampl * cos(phase);
                    double im = ampl * sin(phase);
                    if (elsize == 8) { re = (float)re; im = (float)im; }
                    if (sign == -1) im = -im;
                    if (use_conj) im *= -1;
                    if ((N0 - i) % N0 == i && (N1 - j) % N1 == j && (N2 - k2) % N2 == k2) {
                        im = 0;
                        if (unitary) re = 1;
                    }
                    if (i == 0 && j == 0 && k2 == 0) re = im = 0;
                    int64_t r2 = k2 - start[2];
                    if (r2 < 0 || r2 >= size[2]) continue;
                    char *p = (char *)canvas + (i - start[0]) * strides[0] + (j - start[1]) * strides[1] + r2 * strides[2];
                    if (elsize == 16) { ((double *)p)[0] = re; ((double *)p)[1] = im; }
                    else { ((float *)p)[0] = (float)re; ((float *)p)[1] = (float)im; }
                }
            }
        }
    }
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) free(w.table[a][b]);
    return PMX_OK;
}
