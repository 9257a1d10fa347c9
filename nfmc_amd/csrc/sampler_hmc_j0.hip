// hmc sampler kernels, jump-tail conditioner width 0 (0 = no jump tail): see sampler_impl.hpp
#include "sampler_impl.hpp"

namespace nfmc {
int launch_hmc_j0(const NfmcHmcArgs& a, const JumpDev& jd, Cfg c, bool fast, int64_t tiles, int grid, hipStream_t st) {
    int rc = NFMC_EUNSUPPORTED;
#define M(CPL, LPC) \
    if (c.cpl == CPL && c.lpc == LPC) rc = launch_hmc_cfg<CPL, LPC, 0>(a, jd, fast, tiles, grid, st);
    NFMC_FOR_CFG(M)
#undef M
    return rc;
}
}  // namespace nfmc
