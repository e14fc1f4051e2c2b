// pmx_binned.hip — tile-binned paint / readout (LDS-tiled).  Placeholder entry
// points until the tiled kernels land; they fail loudly rather than fall back.
#include "pmx_common.h"

extern "C" int pmx_binplan_create(pmx_binplan **plan, const pmx_painter *p, int64_t max_particles)
{
    pmx::set_error("pmx_binplan_create: not built yet");
    return PMX_EUNSUPPORTED;
}
extern "C" int pmx_binplan_destroy(pmx_binplan *plan) { return PMX_OK; }
extern "C" int pmx_binplan_build(pmx_binplan *plan, const pmx_painter *p, const pmx_vec *pos,
                                 const pmx_vec *mass, double mass_scalar, int64_t npart, void *stream)
{
    pmx::set_error("pmx_binplan_build: not built yet");
    return PMX_EUNSUPPORTED;
}
extern "C" int pmx_paint_binned(pmx_binplan *plan, const pmx_painter *p, void *canvas, void *stream)
{
    pmx::set_error("pmx_paint_binned: not built yet");
    return PMX_EUNSUPPORTED;
}
extern "C" int pmx_readout_binned(pmx_binplan *plan, const pmx_painter *p, const void *canvas,
                                  const pmx_vec *out, void *stream)
{
    pmx::set_error("pmx_readout_binned: not built yet");
    return PMX_EUNSUPPORTED;
}
