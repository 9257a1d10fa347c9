// VALU NeuTra kernels with 16 chains per wave (neutra_kernels.hpp), every conditioner width bucket
#include "neutra_kernels.hpp"

NFMC_NEUTRA_RPW_UNIT(16)
