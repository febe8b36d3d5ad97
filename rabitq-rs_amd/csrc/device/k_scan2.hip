// k_scan2.hip — second translation unit of k_scan (scan.hpp): the instantiations for padded dimensions 256, 384, 512, 1024
// and 1536, compiled in parallel with k_scan.hip (128 / 768 / 960 / runtime dimension).  gfx950 only.
#define RBQ_SCAN_TU2 1
#include "k_scan.hip"
