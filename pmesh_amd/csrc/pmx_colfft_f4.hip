// pmx_colfft_f4.hip — the float instantiations of pmx_colfft.hip as a compilation unit of their own (see
// PMX_COLFFT_PART there).
#define PMX_COLFFT_PART 2
#include "pmx_colfft.hip"
