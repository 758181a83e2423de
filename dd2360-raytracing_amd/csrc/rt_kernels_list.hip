// rt_kernels_list.hip — the hitable_list (USE_OCTREE off) instantiations of rt_kernels.hip as their own translation unit,
// compiled WITH SLP vectorisation: packed fp32 makes the list scan 10 % faster on gfx950 and the tree walk slower (Makefile).
#define RT_TU_LIST 1
#include "rt_kernels.hip"
