// C entry points of the VALU NeuTra path (kernels: neutra_kernels.hpp, instantiated in neutra_kernels_r64 / r32 / r16.hip)
#include "neutra_kernels.hpp"

using namespace nfmc;

// rows-per-wave variant -> its unit's launcher
#define NFMC_NEUTRA_RPW_CALL(FN, ...)                          \
    (rpw == 64 ? FN##64(__VA_ARGS__) : (rpw == 32 ? FN##32(__VA_ARGS__) : FN##16(__VA_ARGS__)))

extern "C" int64_t nfmc_neutra_scratch_bytes(int64_t n, int32_t d, int32_t n_hidden, int32_t n_hidden_layers,
                                             int32_t n_coupling) {
    if (n <= 0 || d <= 0 || n_hidden <= 32 || n_hidden_layers <= 0 || n_coupling < 0) return 0;
    if (nfmc_mfma_wide_supported(d, n_hidden, n_hidden_layers))   // the composed trajectory of mfma_wide.hip: four (n, d) arrays, three (n,)
        return nfmc_neutra_wide_scratch_floats(n, d) * (int64_t)sizeof(float);
    // momentum, gradient, U~ and H0 of every chain, then the activation checkpoints of every resident wave
    // (mfma_flow.hpp: CkLayout -- hidden activations, alpha, beta of every coupling layer, 1 KB tiles)
    const int64_t th = nfmc_realnvp_padded_hidden(n_hidden) / 16, td = d / 16;
    const int64_t layer_floats = ((n_hidden_layers > 1 ? 2 : 1) * th + td) * 256;
    const int64_t tiles = (n + kMfmaChains - 1) / kMfmaChains;
    const int64_t slots = tiles < kCkMaxGrid ? tiles : kCkMaxGrid;
    int64_t floats = 2 * n * (int64_t)d + 2 * n + slots * kMfmaWaves * n_coupling * layer_floats;
#ifdef NFMC_TRACE
    floats += 2 * kMfmaWaves * (4096 + 8192);   // diagnostic build: the phase and step mark streams of workgroup 0
#endif
    return floats * (int64_t)sizeof(float);
}

extern "C" int nfmc_neutra_potential_grad_f32(const NfmcRealNVP* flow, const NfmcPotential* pot, const float* z,
                                              int64_t n, float* u_out, float* grad_out, nfmc_stream_t stream) {
    if (flow && flow->n_hidden > 32 && flow->n_bins == 0) return nfmc_neutra_potential_grad_mfma_f32(flow, pot, z, n, u_out, grad_out, stream);
    int rc = check_flow_neutra(flow);
    if (rc) return rc;
    if (!pot || !z || n <= 0) return NFMC_EINVAL;
    if (pot->kind != NFMC_POT_QUADRATIC && pot->kind != NFMC_POT_FUNNEL) return NFMC_EUNSUPPORTED;
    const int rpw = neutra_rows_per_wave(flow->d, 3);
    if (!rpw) return NFMC_ESHAPE;
    const int64_t tiles = (n + rpw - 1) / rpw;
    const int grid = (int)(tiles < 4 * kMaxGrid ? tiles : 4 * kMaxGrid);
    const size_t lds = (size_t)3 * rpw * tile_stride(flow->d) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if ((rc = NFMC_NEUTRA_RPW_CALL(neutra_grad_launch_r, hp_bucket_n(flow->n_hidden), lds, grid, st, *flow, *pot, z, n, u_out, grad_out,
                                   tiles)))
        return rc;
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

extern "C" int nfmc_neutra_hmc_steps_f32(const NfmcNeutraHmcArgs* args, nfmc_stream_t stream) {
    if (!args) return NFMC_EINVAL;
    NfmcNeutraHmcArgs a = *args;
    if (a.stats.sum_x && a.stats.defer) return NFMC_EUNSUPPORTED;   // NeuTra folds its statistics per call
    if (int rr = rng_default_only(a.rng)) return rr;
    if (a.flow.n_hidden > 32 && a.flow.n_bins == 0) {
        if (!a.z || a.n <= 0 || a.n_steps <= 0 || a.n_leapfrog <= 0 || !(a.step_size > 0.f)) return NFMC_EINVAL;
        if (a.n_steps > NFMC_MAX_STEPS_PER_CALL) return NFMC_ESHAPE;
        if (a.pot.kind != NFMC_POT_QUADRATIC && a.pot.kind != NFMC_POT_FUNNEL) return NFMC_EUNSUPPORTED;
        if (a.stats.sum_x && (!a.stats.sum_x2 || !a.stats.counters || !a.stats.scratch)) return NFMC_EINVAL;
        if (a.adjust && (a.rng.replay_normals != nullptr) != (a.rng.replay_uniforms != nullptr)) return NFMC_EINVAL;
        if (!store_ok(a.samples)) return NFMC_EINVAL;
        return nfmc_neutra_hmc_steps_mfma_f32(&a, a.scratch, a.scratch_bytes, stream);
    }
    int rc = check_flow_neutra(&a.flow);
    if (rc) return rc;
    if (!a.z || a.n <= 0 || a.n_steps <= 0 || a.n_leapfrog <= 0 || !(a.step_size > 0.f)) return NFMC_EINVAL;
    if (a.n_steps > NFMC_MAX_STEPS_PER_CALL) return NFMC_ESHAPE;
    if (a.pot.kind != NFMC_POT_QUADRATIC && a.pot.kind != NFMC_POT_FUNNEL) return NFMC_EUNSUPPORTED;
    if (a.stats.sum_x && (!a.stats.sum_x2 || !a.stats.counters || !a.stats.scratch)) return NFMC_EINVAL;
    if (a.adjust && (a.rng.replay_normals != nullptr) != (a.rng.replay_uniforms != nullptr)) return NFMC_EINVAL;
    if (!store_ok(a.samples)) return NFMC_EINVAL;
    const int d = a.flow.d;
    const int dp = padded_d(d);
    const int rpw = neutra_rows_per_wave(d, 4);
    if (!rpw) return NFMC_ESHAPE;
    const int64_t tiles = (a.n + rpw - 1) / rpw;
    const int grid = (int)(tiles < kMaxGrid ? tiles : kMaxGrid);
    if (a.stats.sum_x && a.stats.scratch_bytes < (int64_t)grid * (2 * dp + kStatTail) * (int64_t)sizeof(double))
        return NFMC_ESCRATCH;
    const size_t lds = (size_t)4 * rpw * tile_stride(d) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if ((rc = NFMC_NEUTRA_RPW_CALL(neutra_hmc_launch_r, hp_bucket_n(a.flow.n_hidden), lds, grid, st, a, tiles, dp))) return rc;
    NFMC_HIP_CHECK_LAUNCH();
    if (a.stats.sum_x) {
        hipLaunchKernelGGL(stats_finish_kernel<true>, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, st, a.stats.scratch,
                           grid, dp, d, a.stats, (unsigned long long)a.n * (unsigned long long)a.n_steps);
        NFMC_HIP_CHECK_LAUNCH();
    }
    return NFMC_OK;
}
