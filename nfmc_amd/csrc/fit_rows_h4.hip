// fit_rows_kernel<HP = 4, ...>: the row-per-wave gradient kernel of the flow (re)fit (fit_rows.hpp) for conditioners of
// width <= 4 -- a unit of its own for the build time; launched from fit_kernels.hip.
#include "fit_rows.hpp"

NFMC_FIT_ROWS_UNIT(4, fit_rows_launch_h4)
