// FixedIMH as a data-parallel problem with SPLINE couplings (c-rqnsf, util.py:288-289): the kernels of imh_parallel.hpp
// with the register-layout spline flow (FlowB<..., NB = 8>).  A unit of its own for the build time.
#include "imh_parallel.hpp"

namespace nfmc {

int launch_imh_rqs(int cpl, int lpc, int hp, const NfmcFlowMhArgs& a, const ImhWork& w, hipStream_t st, int* grid_c, int* dp_out, bool dry) {
    int rc = NFMC_EUNSUPPORTED;
#define M(CPL, LPC)               \
    if (cpl == CPL && lpc == LPC) \
        rc = hp == 4 ? launch_imh<CPL, LPC, 4, kRqsBins>(a, w, st, grid_c, dp_out, dry) : launch_imh<CPL, LPC, 8, kRqsBins>(a, w, st, grid_c, dp_out, dry);
    NFMC_FOR_PCFG(M)
#undef M
    return rc;
}

}  // namespace nfmc
