// fit_rows_kernel<HP = 8, ...>: the row-per-wave gradient kernel of the flow (re)fit (fit_rows.hpp) for conditioners of
// width 5 .. 8 (the default RealNVP from d = 44 up: C3's and C5's flows) -- a unit of its own for the build time; launched
// from fit_kernels.hip.
#include "fit_rows.hpp"

NFMC_FIT_ROWS_UNIT(8, fit_rows_launch_h8)
