// pmx_binned_readout.hip — the readout launchers of pmx_binned.hip as a compilation unit of their own (see
// PMX_BINNED_PART there): the tile-binned readout kernels in all their template forms.
#define PMX_BINNED_PART 3
#include "pmx_binned.hip"
