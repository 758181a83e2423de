// rt_kernels_contract.hip — rt_kernels.hip compiled a third time, with FMA contraction allowed (-ffp-contract=fast, namespace
// rt::fmac): the kernels behind rt_world_set_arith(RT_ARITH_CONTRACT).  The reference's own build contracts (nvcc -fmad=true by
// default, Makefile:9); this mode models that one behaviour of a real CUDA build, is reported separately and is never the parity mode.
#define RT_TU_CONTRACT 1
#include "rt_kernels.hip"
