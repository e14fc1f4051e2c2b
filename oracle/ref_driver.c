/*
 * ref_driver.c — TEST INFRASTRUCTURE ONLY (see oracle/pmesh_oracle.c header).
 *
 * A particle loop around the REFERENCE's own per-particle C entry points
 * (pmesh/_window_imp.h:76-86), so that the reference window kernels can be
 * driven without Cython.  It is compiled together with
 * /root/reference/pmesh/_window_imp.c, from where that file lies, into
 * oracle/_ref/libpmesh_ref.so by oracle/Makefile; no reference source is
 * copied.  The loop restates pmesh/_window.pyx:128-205 (copy the painter,
 * set geometry, pmesh_painter_init, then one call per particle with pos
 * widened to double, mass, and hsml or 1.0).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "_window_imp.h" /* from -I/root/reference/pmesh */
#include "../include/pmesh_amd.h"

static double ld(const pmx_vec *v, int64_t i, int c)
{
    const char *p = (const char *)v->data + i * v->stride0 + c * v->stride1;
    return v->elsize == 8 ? *(const double *)p : (double)*(const float *)p;
}

/* pmx kind -> reference enum (_window_imp.h:4-28) */
static int ref_type(int kind)
{
    switch (kind) {
    case PMX_NEAREST: return PMESH_PAINTER_NEAREST;
    case PMX_LINEAR: return PMESH_PAINTER_LINEAR;
    case PMX_QUADRATIC: return PMESH_PAINTER_QUADRATIC;
    case PMX_CUBIC: return PMESH_PAINTER_CUBIC;
    case PMX_TUNED_NNB: return PMESH_PAINTER_TUNED_NNB;
    case PMX_TUNED_CIC: return PMESH_PAINTER_TUNED_CIC;
    case PMX_TUNED_TSC: return PMESH_PAINTER_TUNED_TSC;
    case PMX_TUNED_PCS: return PMESH_PAINTER_TUNED_PCS;
    case PMX_LANCZOS2: return PMESH_PAINTER_LANCZOS2;
    case PMX_LANCZOS3: return PMESH_PAINTER_LANCZOS3;
    case PMX_LANCZOS4: return PMESH_PAINTER_LANCZOS4;
    case PMX_LANCZOS5: return PMESH_PAINTER_LANCZOS5;
    case PMX_LANCZOS6: return PMESH_PAINTER_LANCZOS6;
    case PMX_ACG2: return PMESH_PAINTER_ACG2;
    case PMX_ACG3: return PMESH_PAINTER_ACG3;
    case PMX_ACG4: return PMESH_PAINTER_ACG4;
    case PMX_ACG5: return PMESH_PAINTER_ACG5;
    case PMX_ACG6: return PMESH_PAINTER_ACG6;
    case PMX_DB6: return PMESH_PAINTER_DB6;
    case PMX_DB12: return PMESH_PAINTER_DB12;
    case PMX_DB20: return PMESH_PAINTER_DB20;
    case PMX_SYM6: return PMESH_PAINTER_SYM6;
    case PMX_SYM12: return PMESH_PAINTER_SYM12;
    case PMX_SYM20: return PMESH_PAINTER_SYM20;
    }
    return kind - 100; /* kind >= 100: raw reference enum value (lanczos, acg, ...) */
}

static void setup(PMeshPainter *painter, const pmx_painter *p, void *canvas)
{
    /* ResampleWindow.__init__ (_window.pyx:104-114) */
    memset(painter, 0, sizeof(*painter));
    painter->support = p->support;
    painter->type = (PMeshPainterType)ref_type(p->kind);
    painter->ndim = 0;
    painter->canvas_dtype_elsize = 0;
    pmesh_painter_init(painter);
    /* paint()/readout() prologue (_window.pyx:137-155) */
    painter->ndim = p->ndim;
    painter->canvas = canvas;
    painter->canvas_dtype_elsize = p->canvas_elsize;
    for (int d = 0; d < p->ndim; d++) {
        painter->order[d] = p->order[d];
        painter->Nmesh[d] = p->period[d];
        painter->scale[d] = p->scale[d];
        painter->translate[d] = p->translate[d];
        painter->size[d] = p->size[d];
        painter->strides[d] = p->strides[d];
    }
    pmesh_painter_init(painter);
}

int ref_paint(const pmx_painter *p, void *canvas, const pmx_vec *pos, const pmx_vec *mass,
              double mass_scalar, const pmx_vec *hsml, int64_t npart, void *stream)
{
    (void)stream;
    PMeshPainter painter[1];
    setup(painter, p, canvas);
    for (int64_t i = 0; i < npart; i++) {
        double x[32];
        for (int d = 0; d < p->ndim; d++) x[d] = ld(pos, i, d);
        double m = (mass && mass->data) ? ld(mass, i, 0) : mass_scalar;
        double h = (hsml && hsml->data) ? ld(hsml, i, 0) : 1.0;
        pmesh_painter_paint(painter, x, m, h);
    }
    return 0;
}

int ref_readout(const pmx_painter *p, const void *canvas, const pmx_vec *pos, const pmx_vec *hsml,
                const pmx_vec *out, int64_t npart, void *stream)
{
    (void)stream;
    PMeshPainter painter[1];
    setup(painter, p, (void *)canvas);
    for (int64_t i = 0; i < npart; i++) {
        double x[32];
        for (int d = 0; d < p->ndim; d++) x[d] = ld(pos, i, d);
        double h = (hsml && hsml->data) ? ld(hsml, i, 0) : 1.0;
        double v = pmesh_painter_readout(painter, x, h);
        char *o = (char *)out->data + i * out->stride0;
        if (out->elsize == 8) *(double *)o = v;
        else *(float *)o = (float)v;
    }
    return 0;
}

int ref_window_info(int32_t kind, int32_t support, int32_t *nativesupport, int32_t *eff_support)
{
    PMeshPainter painter[1];
    memset(painter, 0, sizeof(*painter));
    painter->support = support;
    painter->type = (PMeshPainterType)ref_type(kind);
    pmesh_painter_init(painter);
    if (nativesupport) *nativesupport = (int)painter->nativesupport;
    if (eff_support) *eff_support = painter->support;
    return 0;
}

int ref_fwindow(int32_t kind, int32_t support, const double *w, int64_t n, double *out)
{
    PMeshPainter painter[1];
    memset(painter, 0, sizeof(*painter));
    painter->support = support;
    painter->type = (PMeshPainterType)ref_type(kind);
    pmesh_painter_init(painter);
    for (int64_t i = 0; i < n; i++) out[i] = pmesh_painter_get_fwindow(painter, w[i]);
    return 0;
}

/* ---- meshes of more than three dimensions: the same loop around the reference's entry points with the geometry of a
 * pmx_painter_nd (the reference's painter holds up to 32 dimensions, _window_imp.h:50-60) */
static void setup_nd(PMeshPainter *painter, const pmx_painter_nd *p, void *canvas)
{
    memset(painter, 0, sizeof(*painter));
    painter->support = p->support;
    painter->type = (PMeshPainterType)ref_type(p->kind);
    painter->ndim = 0;
    painter->canvas_dtype_elsize = 0;
    pmesh_painter_init(painter);
    painter->ndim = p->ndim;
    painter->canvas = canvas;
    painter->canvas_dtype_elsize = p->canvas_elsize;
    for (int d = 0; d < p->ndim; d++) {
        painter->order[d] = p->order[d];
        painter->Nmesh[d] = p->period[d];
        painter->scale[d] = p->scale[d];
        painter->translate[d] = p->translate[d];
        painter->size[d] = p->size[d];
        painter->strides[d] = p->strides[d];
    }
    pmesh_painter_init(painter);
}

int ref_paint_nd(const pmx_painter_nd *p, void *canvas, const pmx_vec *pos, const pmx_vec *mass,
                 double mass_scalar, const pmx_vec *hsml, int64_t npart, void *stream)
{
    (void)stream;
    PMeshPainter painter[1];
    setup_nd(painter, p, canvas);
    for (int64_t i = 0; i < npart; i++) {
        double x[32];
        for (int d = 0; d < p->ndim; d++) x[d] = ld(pos, i, d);
        double m = (mass && mass->data) ? ld(mass, i, 0) : mass_scalar;
        double h = (hsml && hsml->data) ? ld(hsml, i, 0) : 1.0;
        pmesh_painter_paint(painter, x, m, h);
    }
    return 0;
}

int ref_readout_nd(const pmx_painter_nd *p, const void *canvas, const pmx_vec *pos, const pmx_vec *hsml,
                   const pmx_vec *out, int64_t npart, void *stream)
{
    (void)stream;
    PMeshPainter painter[1];
    setup_nd(painter, p, (void *)canvas);
    for (int64_t i = 0; i < npart; i++) {
        double x[32];
        for (int d = 0; d < p->ndim; d++) x[d] = ld(pos, i, d);
        double h = (hsml && hsml->data) ? ld(hsml, i, 0) : 1.0;
        double v = pmesh_painter_readout(painter, x, h);
        char *o = (char *)out->data + i * out->stride0;
        if (out->elsize == 8) *(double *)o = v;
        else *(float *)o = (float)v;
    }
    return 0;
}
