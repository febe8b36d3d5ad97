// k_scanw2.hip — second translation unit of k_scanw (scanw.hpp): the instantiations for padded dimensions 128, 256, 384, 512,
// 1024 and 1536, compiled in parallel with k_scanw.hip (768 / 960 / runtime dimension).  gfx950 only.
#define RBQ_SCANW_TU2 1
#include "k_scanw.hip"
