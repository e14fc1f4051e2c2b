// pmx_binned_paint_f4.hip — the paint launcher of pmx_binned.hip for float canvases as a compilation unit of its own
// (see PMX_BINNED_PART there).
#define PMX_BINNED_PART 4
#include "pmx_binned.hip"
