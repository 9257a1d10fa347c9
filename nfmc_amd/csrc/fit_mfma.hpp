// Interface between fit_kernels.hip (the C ABI of the flow fit) and fit_mfma.hip (the matrix-core gradient kernel for
// conditioners of width 33..128 at d = 64 / 128).
#pragma once

#include "common.hpp"

namespace nfmc {

struct FitMfmaArgs {
    NfmcRealNVP f;
    NfmcPotential pot;
    const float* x;
    int64_t n;
    const float* xv;
    int64_t nv;
    float* partial;
    int64_t pstride;
    int64_t ea_off;
    int d4;
    int64_t n_params;
    int64_t tiles, vtiles;    // 128-row tiles of the batch, of the validation rows
    float* ck;                // checkpoint areas: gridDim.x * 8 waves * n_coupling * CkLayout::kLayerFloats floats
    const float* run_state;
};

int nfmc_mfma_supported(int32_t d, int32_t n_hidden, int32_t n_hidden_layers);   // neutra_mfma.hip
// checkpoint floats the kernel needs for `grid` workgroups; workgroups of a launch; the launch itself (0 or an error code)
int64_t fit_mfma_ck_floats(int d, int hp, int n_hl, int n_coupling, int grid);
int fit_mfma_grid(int64_t n, int64_t nv);
int fit_mfma_launch(bool rkl, const FitMfmaArgs& a, int grid, hipStream_t st);

}  // namespace nfmc
