// K3 / K4 (+K6, K7) on the matrix cores: RealNVP forward / inverse / log-density and the flow-proposal Metropolis
// step for conditioners of width 33..128 at d = 64 / 128 (the shapes of mfma_device.hpp).  Same contracts as the
// one-chain-per-lane kernels of flow_kernels.hip, which they replace for these shapes:
//   Flow.log_prob / bijection.forward                               jump.py:218, imh.py:214
//   Flow.sample(n, return_log_prob=True) / bijection.inverse        jump.py:205, imh.py:221, neutra.py:60
//   the jump of JumpNFMC.sample (jump.py:205-243) / the loop body of FixedIMH.sample (imh.py:220-249)
// 16 chains per wave in the MFMA C layout, 8 waves per workgroup, weights through the LDS pipeline (mfma_flow.hpp).
#include "mfma_flow.hpp"

namespace nfmc {

// latent z ~ N(0, I) for the wave's chains in C layout: tile position p holds logical coordinate (rev ? d-1-p : p).
// Native: Philox stream kTagLatent, one block per 4 consecutive logical coordinates; replay: (n_steps, n, d) array.
template <int TD>
__device__ __forceinline__ float draw_latent_c(f32x4 (&z)[TD], const NfmcRng& rng, int64_t rrow, int64_t n, int s,
                                               int half, bool rev) {
    constexpr int d = 16 * TD;
    const uint32_t gchain = (uint32_t)(rng.chain_offset + (uint64_t)rrow);
    float ss = 0.f;
#pragma unroll
    for (int m = 0; m < TD; ++m) {
        const int p0 = 16 * m + 4 * half;   // tile position of this lane's 4-block
        float zz[4];
        if (rng.replay_normals) {
            const float* src = rng.replay_normals + ((int64_t)s * n + rrow) * d;
#pragma unroll
            for (int j = 0; j < 4; ++j) zz[j] = src[rev ? d - 1 - (p0 + j) : p0 + j];
        } else {
            const int blk = rev ? (d - 4 - p0) >> 2 : p0 >> 2;
            float w[4];
            philox_normal4(gchain, rng.step0 + (uint32_t)s, (uint32_t)blk, kTagLatent, (uint32_t)rng.seed,
                           (uint32_t)(rng.seed >> 32), w);
#pragma unroll
            for (int j = 0; j < 4; ++j) zz[j] = rev ? w[3 - j] : w[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            z[m][j] = zz[j];
            ss = fmaf(zz[j], zz[j], ss);
        }
    }
    return chain_sum(ss);
}

template <int TD>
__device__ __forceinline__ float sum_squares_c(const f32x4 (&z)[TD]) {
    float ss = 0.f;
#pragma unroll
    for (int m = 0; m < TD; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) ss = fmaf(z[m][t], z[m][t], ss);
    return chain_sum(ss);
}

template <int TD, int TH, int NHL>
__global__ void __launch_bounds__(kMfmaBlock, 2) realnvp_forward_mfma_kernel(NfmcRealNVP f, const float* __restrict__ x,
                                                                          int64_t n, float* __restrict__ z,
                                                                          float* __restrict__ logdet,
                                                                          float* __restrict__ log_prob, int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int d = 16 * TD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, half = lane >> 4;
    const bool rev = (f.n_coupling & 1) != 0;
    WeightPipe wp{lds, 0};
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row = tile * kMfmaChains + wave * 16 + col;
        const bool active = row < n;
        const int64_t rrow = active ? row : n - 1;
        f32x4 v[TD];
        load_ctiles<TD>(v, x, rrow, d, half, false);
        const float ld = chain_sum(flow_forward_sweep_c<TD, TH, NHL>(v, f, wp, col, half));
        const float ss = sum_squares_c<TD>(v);
        if (active) {
            if (half == 0) {
                if (logdet) logdet[row] = ld;
                if (log_prob) log_prob[row] = -0.5f * ss - 0.5f * (float)d * kLog2Pi + ld;
            }
            if (z) store_ctiles<TD>(v, z, row, d, half, rev);
        }
    }
}

template <int TD, int TH, int NHL>
__global__ void __launch_bounds__(kMfmaBlock, 2) realnvp_inverse_mfma_kernel(NfmcRealNVP f, const float* __restrict__ z,
                                                                          int64_t n, float* __restrict__ x,
                                                                          float* __restrict__ logdet,
                                                                          float* __restrict__ log_q, NfmcRng rng,
                                                                          int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int d = 16 * TD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, half = lane >> 4;
    const bool rev = (f.n_coupling & 1) != 0;
    WeightPipe wp{lds, 0};
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row = tile * kMfmaChains + wave * 16 + col;
        const bool active = row < n;
        const int64_t rrow = active ? row : n - 1;
        f32x4 v[TD];
        float ss;
        if (z) {
            load_ctiles<TD>(v, z, rrow, d, half, rev);
            ss = sum_squares_c<TD>(v);
        } else {
            ss = draw_latent_c<TD>(v, rng, rrow, n, 0, half, rev);
        }
        const float ld = chain_sum(flow_inverse_sweep_c<TD, TH, NHL>(v, f, wp, col, half));
        if (active) {
            if (half == 0) {
                if (logdet) logdet[row] = ld;
                if (log_q) log_q[row] = -0.5f * ss - 0.5f * (float)d * kLog2Pi - ld;
            }
            if (x) store_ctiles<TD>(v, x, row, d, half, false);
        }
    }
}

template <int TD, int TH, int NHL>
__global__ void __launch_bounds__(kMfmaBlock, 2) flow_mh_mfma_kernel(NfmcFlowMhArgs a, int64_t tiles, int dp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int d = 16 * TD;
    const NfmcRealNVP& f = a.flow;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, half = lane >> 4;
    const bool rev = (f.n_coupling & 1) != 0;
    const int64_t n = a.n;
    const float base_c = -0.5f * (float)d * kLog2Pi;
    double* red = reinterpret_cast<double*>(lds + kMfmaStatOffset);  // [8 waves][2*d + 2]
    WeightPipe wp{lds, 0};
    uint32_t n_acc = 0, n_bad = 0;
    for (int i = threadIdx.x; i < kMfmaWaves * (2 * d + 2); i += kMfmaBlock) red[i] = 0.0;
    __syncthreads();

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row_t = tile * kMfmaChains + wave * 16 + col;
        const bool active = row_t < n;
        const int64_t rrow_t = active ? row_t : n - 1;
        f32x4 x[TD];
        load_ctiles<TD>(x, a.x, rrow_t, d, half, false);
        float u_x = potential_value_c<TD>(x, a.pot, half, lane);                  // jump.py:212 / imh.py:224
        float f_x;
        if (a.logq_cached) {
            f_x = a.logq[rrow_t];
        } else {                                                                    // flow.log_prob(x): jump.py:218 / imh.py:214
            f32x4 w[TD];
#pragma unroll
            for (int m = 0; m < TD; ++m) w[m] = x[m];
            const float ld = chain_sum(flow_forward_sweep_c<TD, TH, NHL>(w, f, wp, col, half));
            f_x = -0.5f * sum_squares_c<TD>(w) + base_c + ld;
        }
        StoreCursor keep(a.samples);
        for (int s = 0; s < a.n_steps; ++s) {
            // per-step opaque copies of the row index (cf. neutra_mfma.hip: addresses must not live through the GEMMs)
            int64_t row = row_t, rrow = rrow_t;
            asm volatile("" : "+v"(row), "+v"(rrow));
            f32x4 xp[TD];
            const float ss = draw_latent_c<TD>(xp, a.rng, rrow, n, s, half, rev);  // flow.sample: jump.py:205 / imh.py:221
            const float ldi = chain_sum(flow_inverse_sweep_c<TD, TH, NHL>(xp, f, wp, col, half));
            asm volatile("" : "+v"(row), "+v"(rrow));
            const float f_xp = -0.5f * ss + base_c - ldi;
            const float u_xp = potential_value_c<TD>(xp, a.pot, half, lane);        // jump.py:213 / imh.py:225
            const float lr = (-u_xp) - (-u_x) + f_x - f_xp;                         // util.py:392
            bool accept = true;
            if (a.adjusted) {
                float u;
                if (a.rng.replay_uniforms) {
                    u = a.rng.replay_uniforms[(int64_t)s * n + rrow];
                } else {
                    const uint4 r = philox4x32_10((uint32_t)(a.rng.chain_offset + (uint64_t)rrow),
                                                  a.rng.step0 + (uint32_t)s, 0u, kTagJump, (uint32_t)a.rng.seed,
                                                  (uint32_t)(a.rng.seed >> 32));
                    u = u32_to_uniform(r.x);
                }
                accept = fast_ln(u) < lr;                                           // jump.py:225 / imh.py:229-230
                if (active && half == 0 && !(fabsf(lr) <= 3.0e38f)) n_bad++;
            }
            accept = accept && active;
            if (accept) {                                                           // jump.py:231 / imh.py:232-233
#pragma unroll
                for (int m = 0; m < TD; ++m) x[m] = xp[m];
                f_x = f_xp;
                u_x = u_xp;
                if (half == 0) n_acc++;
            }
            float* kept = keep.next(n * (int64_t)d);
            if (active) {
                if (half == 0) {
                    if (a.masks_out) a.masks_out[(int64_t)s * n + row] = accept ? 1 : 0;
                    if (a.log_ratio_out) a.log_ratio_out[(int64_t)s * n + row] = lr;
                }
                if (kept) store_ctiles<TD>(x, kept, row, d, half, false);
            }
            if (a.stats.sum_x) {  // K7: sums over the 16 chains of the wave, kept per wave in LDS
#pragma unroll
                for (int m = 0; m < TD; ++m)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float xv = active ? x[m][t] : 0.f;
                        double v1 = (double)xv, v2 = (double)xv * (double)xv;
                        for (int k = 1; k < 16; k <<= 1) {
                            v1 += __shfl_xor(v1, k, kWave);
                            v2 += __shfl_xor(v2, k, kWave);
                        }
                        if (col == 0) {
                            const int c = 16 * m + 4 * half + t;   // x-space: position = coordinate
                            red[wave * (2 * d + 2) + c] += v1;
                            red[wave * (2 * d + 2) + d + c] += v2;
                        }
                    }
            }
        }
        if (active) {
            store_ctiles<TD>(x, a.x, row_t, d, half, false);
            if (half == 0) a.logq[row_t] = f_x;
        }
    }
    if (a.stats.sum_x) {
        for (int m = 1; m < 16; m <<= 1) {   // counted on lane group 0 only
            n_acc += __shfl_xor(n_acc, m, kWave);
            n_bad += __shfl_xor(n_bad, m, kWave);
        }
        if (lane == 0) {
            red[wave * (2 * d + 2) + 2 * d] = (double)n_acc;
            red[wave * (2 * d + 2) + 2 * d + 1] = (double)n_bad;
        }
        __syncthreads();
        const bool defer = a.stats.defer != 0;   // deferred: add to the caller-zeroed slab (nfmc_stats_fold_f32)
        const int slot = defer ? a.stats.tail_slot : 0;
        double* out = a.stats.scratch + (size_t)blockIdx.x * (2 * dp + kStatTail);
        for (int i = threadIdx.x; i < 2 * dp + kStatTail; i += kMfmaBlock) {
            int srci = -1;
            if (i < dp) srci = i < d ? i : -1;
            else if (i < 2 * dp) srci = (i - dp) < d ? d + (i - dp) : -1;
            else if (i == 2 * dp + slot) srci = 2 * d;
            else if (i == 2 * dp + slot + 1) srci = 2 * d + 1;
            double v = 0.0;
            if (srci >= 0)
                for (int w = 0; w < kMfmaWaves; ++w) v += red[w * (2 * d + 2) + srci];
            out[i] = defer ? out[i] + v : v;
        }
    }
}

template <class K>
static int set_lds_mfma(K kern) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfmaLdsBytes);
    return e == hipSuccess ? 0 : (int)e;
}

static int grid_for_tiles(int64_t tiles) { return (int)(tiles < kMaxGrid ? tiles : kMaxGrid); }

template <int TD, int TH, int NHL>
static int launch_forward(const NfmcRealNVP& f, const float* x, int64_t n, float* z, float* logdet, float* log_prob,
                          int64_t tiles, hipStream_t st) {
    auto kern = realnvp_forward_mfma_kernel<TD, TH, NHL>;
    if (int rc = set_lds_mfma(kern)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for_tiles(tiles)), dim3(kMfmaBlock), kMfmaLdsBytes, st, f, x, n, z, logdet, log_prob,
                       tiles);
    return 0;
}

template <int TD, int TH, int NHL>
static int launch_inverse(const NfmcRealNVP& f, const float* z, int64_t n, float* x, float* logdet, float* log_q,
                          const NfmcRng& rng, int64_t tiles, hipStream_t st) {
    auto kern = realnvp_inverse_mfma_kernel<TD, TH, NHL>;
    if (int rc = set_lds_mfma(kern)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for_tiles(tiles)), dim3(kMfmaBlock), kMfmaLdsBytes, st, f, z, n, x, logdet, log_q, rng,
                       tiles);
    return 0;
}

template <int TD, int TH, int NHL>
static int launch_flow_mh(const NfmcFlowMhArgs& a, int64_t tiles, int grid, int dp, hipStream_t st) {
    auto kern = flow_mh_mfma_kernel<TD, TH, NHL>;
    if (int rc = set_lds_mfma(kern)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kMfmaBlock), kMfmaLdsBytes, st, a, tiles, dp);
    return 0;
}

}  // namespace nfmc

using namespace nfmc;

#define NFMC_FLOW_MFMA_DISPATCH(TDV, THV, NHLV, CALL)                                   \
    if (TDV == 4 && THV == 4 && NHLV == 1) { constexpr int TD = 4, TH = 4, NHL = 1; CALL; }      \
    else if (TDV == 4 && THV == 4 && NHLV == 2) { constexpr int TD = 4, TH = 4, NHL = 2; CALL; } \
    else if (TDV == 4 && THV == 8 && NHLV == 1) { constexpr int TD = 4, TH = 8, NHL = 1; CALL; } \
    else if (TDV == 4 && THV == 8 && NHLV == 2) { constexpr int TD = 4, TH = 8, NHL = 2; CALL; } \
    else if (TDV == 8 && THV == 4 && NHLV == 1) { constexpr int TD = 8, TH = 4, NHL = 1; CALL; } \
    else if (TDV == 8 && THV == 4 && NHLV == 2) { constexpr int TD = 8, TH = 4, NHL = 2; CALL; } \
    else if (TDV == 8 && THV == 8 && NHLV == 1) { constexpr int TD = 8, TH = 8, NHL = 1; CALL; } \
    else if (TDV == 8 && THV == 8 && NHLV == 2) { constexpr int TD = 8, TH = 8, NHL = 2; CALL; } \
    else return NFMC_EUNSUPPORTED;

int nfmc::nfmc_realnvp_forward_mfma_f32(const NfmcRealNVP* f, const float* x, int64_t n, float* z, float* logdet,
                                        float* log_prob, nfmc_stream_t stream) {
    const int td = f->d / 16, th = nfmc_realnvp_padded_hidden(f->n_hidden) / 16, nhl = f->n_hidden_layers;
    const int64_t tiles = (n + kMfmaChains - 1) / kMfmaChains;
    int rc = 0;
    NFMC_FLOW_MFMA_DISPATCH(td, th, nhl, rc = (launch_forward<TD, TH, NHL>(*f, x, n, z, logdet, log_prob, tiles, (hipStream_t)stream)))
    if (rc) return rc;
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

int nfmc::nfmc_realnvp_inverse_mfma_f32(const NfmcRealNVP* f, const float* z, int64_t n, float* x, float* logdet,
                                        float* log_q, const NfmcRng* rng, nfmc_stream_t stream) {
    const int td = f->d / 16, th = nfmc_realnvp_padded_hidden(f->n_hidden) / 16, nhl = f->n_hidden_layers;
    const int64_t tiles = (n + kMfmaChains - 1) / kMfmaChains;
    NfmcRng r = {};
    if (rng) r = *rng;
    int rc = 0;
    NFMC_FLOW_MFMA_DISPATCH(td, th, nhl, rc = (launch_inverse<TD, TH, NHL>(*f, z, n, x, logdet, log_q, r, tiles, (hipStream_t)stream)))
    if (rc) return rc;
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

int nfmc::nfmc_flow_mh_steps_mfma_f32(const NfmcFlowMhArgs& a, nfmc_stream_t stream, int* grid_out, int* dp_out) {
    const int d = a.flow.d;
    const int td = d / 16, th = nfmc_realnvp_padded_hidden(a.flow.n_hidden) / 16, nhl = a.flow.n_hidden_layers;
    const int64_t tiles = (a.n + kMfmaChains - 1) / kMfmaChains;
    const int grid = grid_for_tiles(tiles), dp = padded_d(d);
    if (a.stats.sum_x && a.stats.scratch_bytes < (int64_t)grid * (2 * dp + kStatTail) * (int64_t)sizeof(double))
        return NFMC_ESCRATCH;
    if (check_defer(a.stats, dp, d)) return NFMC_EINVAL;
    int rc = 0;
    NFMC_FLOW_MFMA_DISPATCH(td, th, nhl, rc = (launch_flow_mh<TD, TH, NHL>(a, tiles, grid, dp, (hipStream_t)stream)))
    if (rc) return rc;
    *grid_out = grid;
    *dp_out = dp;
    return NFMC_OK;
}
