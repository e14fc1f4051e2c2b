// pmx_binned_paint.hip — the paint launchers of pmx_binned.hip as a compilation unit of their own (see
// PMX_BINNED_PART there): the tile-binned paint kernels in all their template forms.
#define PMX_BINNED_PART 2
#include "pmx_binned.hip"
