// Stand-alone pieces of the path: K6 accept/select, K7 moments, the Langevin proposal / log-ratio for
// targets whose U and grad U come from outside (arbitrary Python callables differentiated by torch
// autograd on the GPU), and the native Philox streams on their own (pinned against oracle/philox.py).
#include "common.hpp"

namespace nfmc {

// ---- native streams ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) philox_normals_kernel(NfmcRng rng, uint32_t tag, int64_t n, int d, int nblk,
                                                                float* __restrict__ out) {
    const int64_t total = n * nblk;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t row = t / nblk;
        const int blk = (int)(t - row * nblk);
        float z[4];
        if (rng.rounds == 7)
            philox_normal4<7>((uint32_t)(rng.chain_offset + (uint64_t)row), rng.step0, (uint32_t)blk, tag, (uint32_t)rng.seed,
                              (uint32_t)(rng.seed >> 32), z);
        else
            philox_normal4<10>((uint32_t)(rng.chain_offset + (uint64_t)row), rng.step0, (uint32_t)blk, tag, (uint32_t)rng.seed,
                               (uint32_t)(rng.seed >> 32), z);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * blk + k < d) out[row * d + 4 * blk + k] = z[k];
    }
}

__device__ __forceinline__ float stream_uniform(const NfmcRng& rng, uint32_t tag, int64_t row) {
    const uint32_t gchain = (uint32_t)(rng.chain_offset + (uint64_t)row);
    const uint32_t k0 = (uint32_t)rng.seed, k1 = (uint32_t)(rng.seed >> 32);
    const bool r7 = rng.rounds == 7;
    if (tag == kTagAccept) {
        const uint4 r = r7 ? philox4x32<7>(gchain, rng.step0 >> 2, 0u, kTagAccept, k0, k1)
                           : philox4x32<10>(gchain, rng.step0 >> 2, 0u, kTagAccept, k0, k1);
        return u32_to_uniform(pick_word(r, rng.step0 & 3u));
    }
    const uint4 r = r7 ? philox4x32<7>(gchain, rng.step0, 0u, tag, k0, k1) : philox4x32<10>(gchain, rng.step0, 0u, tag, k0, k1);
    return u32_to_uniform(r.x);
}

__global__ void __launch_bounds__(kBlock) philox_uniforms_kernel(NfmcRng rng, uint32_t tag, int64_t n,
                                                                 float* __restrict__ out) {
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < n; row += (int64_t)gridDim.x * kBlock)
        out[row] = stream_uniform(rng, tag, row);
}

// ---- K7 alone: per-coordinate sums of x and x^2 over `rows` rows -----------------------------------
// One workgroup covers a slab of rows; lane t walks coordinates t, t+256, ... (coalesced across the
// row), partials go to the scratch slab and stats_finish_kernel folds them (deterministic).
__global__ void __launch_bounds__(kBlock) moments_kernel(const float* __restrict__ x, int64_t rows, int d, int dp,
                                                         double* __restrict__ scratch) {
    const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per;
    const int64_t r1 = r0 + per < rows ? r0 + per : rows;
    double* out = scratch + (size_t)blockIdx.x * (2 * dp + kStatTail);
    for (int c = threadIdx.x; c < dp; c += kBlock) {
        double s = 0.0, s2 = 0.0;
        if (c < d)
            for (int64_t r = r0; r < r1; ++r) {
                const float v = x[r * d + c];
                s += (double)v;
                s2 += (double)v * (double)v;
            }
        out[c] = s;
        out[dp + c] = s2;
    }
    if (threadIdx.x < kStatTail) out[2 * dp + threadIdx.x] = 0.0;
}

// ---- K6: Metropolis test + masked row copy (+ carried per-chain scalars) + moments ------------------
// One wave handles 64/LPC chains like the fused samplers so the statistics share block_stats_flush.
template <int CPL, int LPC>
__global__ void __launch_bounds__(kBlock) select_kernel(NfmcSelectArgs a, int64_t tiles) {
    constexpr int CPW = kWave / LPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC, cw = lane / LPC;
    const int d = a.d;
    float sx[CPL], sxx[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) sx[i] = sxx[i] = 0.f;
    uint32_t n_acc = 0, n_bad = 0;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row = (tile * kWavesPerBlock + wave) * CPW + cw;
        const bool active = row < a.n;
        bool accept = active;
        if (a.log_ratio && active) {
            const float lr = a.log_ratio[row];
            const float u = a.uniforms ? a.uniforms[row] : stream_uniform(a.rng, (uint32_t)a.rng_tag, row);
            accept = fast_ln(u) < lr;  // util.py:382-392 ratio, langevin.py:106 / jump.py:225 / imh.py:229 test
            if (g == 0 && !(fabsf(lr) <= 3.0e38f)) n_bad++;
        }
        float x[CPL], xp[CPL];
        load_row<CPL, LPC, false>(a.x, row, d, g, active, x);
        load_row<CPL, LPC, false>(a.x_prime, row, d, g, active, xp);
        const uint64_t am = __ballot(accept);
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            x[i] = select_f32(am, xp[i], x[i]);
            sx[i] += x[i];
            sxx[i] = fmaf(x[i], x[i], sxx[i]);
        }
        if (accept) store_row<CPL, LPC, false>(a.x, row, d, g, true, x);
        if (g == 0 && active) {
            if (accept) {
                n_acc++;
                for (int k = 0; k < a.n_carry; ++k) a.carry[k][row] = a.carry_prime[k][row];
            }
            if (a.mask_out) a.mask_out[row] = accept ? 1 : 0;
        }
    }
    // per-lane counters -> wave totals
    for (int m = 1; m < kWave; m <<= 1) {
        n_acc += __shfl_xor(n_acc, m, kWave);
        n_bad += __shfl_xor(n_bad, m, kWave);
    }
    if (a.stats.sum_x) block_stats_flush<CPL, LPC>(sx, sxx, n_acc, n_bad, a.stats);
}

// ---- Langevin proposal / log-ratio from external U, grad U ------------------------------------------
__global__ void __launch_bounds__(kBlock) langevin_propose_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ grad_u,
                                                                  const float* __restrict__ imd, float h, float sqrt2h,
                                                                  int64_t n, int d, NfmcRng rng,
                                                                  float* __restrict__ x_prime) {
    const int nblk = (d + 3) / 4;
    const int64_t total = n * nblk;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t row = t / nblk;
        const int blk = (int)(t - row * nblk);
        float z[4];
        if (rng.replay_normals) {
#pragma unroll
            for (int k = 0; k < 4; ++k) z[k] = (4 * blk + k < d) ? rng.replay_normals[row * d + 4 * blk + k] : 0.f;
        } else {
            philox_normal4((uint32_t)(rng.chain_offset + (uint64_t)row), rng.step0, (uint32_t)blk, kTagNoise,
                           (uint32_t)rng.seed, (uint32_t)(rng.seed >> 32), z);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * blk + k;
            if (c < d) {
                const float m = imd ? imd[c] : 1.f;
                const int64_t o = row * d + c;
                x_prime[o] = fmaf(sqrt2h / m, z[k], fmaf((-h) / (m * m), grad_u[o], x[o]));  // langevin.py:74-76
            }
        }
    }
}

// one wave per row; lanes stride the coordinates, wave reduction
__global__ void __launch_bounds__(kBlock) langevin_log_ratio_kernel(
    const float* __restrict__ x, const float* __restrict__ xp, const float* __restrict__ u,
    const float* __restrict__ up, const float* __restrict__ gu, const float* __restrict__ gup,
    const float* __restrict__ imd, float h, int64_t n, int d, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * kBlock) >> 6;
    const float inv4h = 1.f / (4.f * h);
    for (int64_t row = wid; row < n; row += nw) {
        float acc = 0.f;
        for (int c = lane; c < d; c += kWave) {
            const float m = imd ? imd[c] : 1.f;
            const float A = 1.f / (m * m);
            const int64_t o = row * d + c;
            const float tf = (xp[o] - x[o]) + h * A * gu[o];
            const float tb = (x[o] - xp[o]) + h * A * gup[o];
            acc += inv4h * (1.f / A) * (tf * tf - tb * tb);  // langevin.py:31-42
        }
        acc = group_allreduce<64>(acc);
        if (lane == 0) out[row] = (u[row] - up[row]) + acc;  // langevin.py:88-105, util.py:392
    }
}

static int grid_for(int64_t work_items) {
    int64_t g = (work_items + kBlock - 1) / kBlock;
    return (int)(g < 1 ? 1 : (g > kMaxGrid ? kMaxGrid : g));
}

}  // namespace nfmc

using namespace nfmc;

extern "C" int nfmc_philox_normals_f32(const NfmcRng* rng, int32_t tag, int64_t n, int32_t d, float* out,
                                       nfmc_stream_t stream) {
    if (!rng || !out || n <= 0 || d <= 0 || !rng_rounds_ok(*rng, true)) return NFMC_EINVAL;
    const int nblk = (d + 3) / 4;
    hipLaunchKernelGGL(philox_normals_kernel, dim3(grid_for(n * nblk)), dim3(kBlock), 0, (hipStream_t)stream, *rng,
                       (uint32_t)tag, n, d, nblk, out);
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

extern "C" int nfmc_philox_uniforms_f32(const NfmcRng* rng, int32_t tag, int64_t n, float* out, nfmc_stream_t stream) {
    if (!rng || !out || n <= 0 || !rng_rounds_ok(*rng, true)) return NFMC_EINVAL;
    if (tag != (int)kTagAccept && tag != (int)kTagJump) return NFMC_EINVAL;
    hipLaunchKernelGGL(philox_uniforms_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, *rng,
                       (uint32_t)tag, n, out);
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

extern "C" int nfmc_moments_update_f32(const float* x, int64_t rows, int32_t d, const NfmcStats* stats,
                                       nfmc_stream_t stream) {
    if (!x || !stats || !stats->sum_x || !stats->sum_x2 || !stats->counters || !stats->scratch) return NFMC_EINVAL;
    if (rows <= 0 || d <= 0) return NFMC_EINVAL;
    if (d > 1024) return NFMC_ESHAPE;
    const int dp = padded_d(d);
    int grid = (int)(rows < 1024 ? (rows + 3) / 4 : 256);
    if (grid < 1) grid = 1;
    if (stats->scratch_bytes < (int64_t)grid * (2 * dp + kStatTail) * (int64_t)sizeof(double)) return NFMC_ESCRATCH;
    if (stats->defer) return NFMC_EUNSUPPORTED;   // K7 alone always folds at once
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(moments_kernel, dim3(grid), dim3(kBlock), 0, st, x, rows, (int)d, dp, stats->scratch);
    NFMC_HIP_CHECK_LAUNCH();
    hipLaunchKernelGGL(stats_finish_kernel<true>, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, st, stats->scratch, grid, dp, (int)d, *stats, 0ull);
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

extern "C" int nfmc_stats_fold_f32(const NfmcStats* stats, int32_t d, uint64_t attempted,
                                   unsigned long long* jump_counters, uint64_t jump_attempted, nfmc_stream_t stream) {
    if (!stats || !stats->sum_x || !stats->sum_x2 || !stats->counters || !stats->scratch) return NFMC_EINVAL;
    if (d <= 0) return NFMC_EINVAL;
    if (d > 1024) return NFMC_ESHAPE;
    const int dp = padded_d(d);
    if (stats->scratch_bytes < stats_scratch_doubles(dp) * (int64_t)sizeof(double)) return NFMC_ESCRATCH;
    hipLaunchKernelGGL(stats_finish_kernel<true>, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, (hipStream_t)stream,
                       stats->scratch, kMaxGrid, dp, (int)d, *stats, (unsigned long long)attempted, jump_counters,
                       (unsigned long long)jump_attempted);
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

template <int CPL, int LPC>
static void launch_select(const NfmcSelectArgs& a, int64_t tiles, int grid, hipStream_t st) {
    hipLaunchKernelGGL((select_kernel<CPL, LPC>), dim3(grid), dim3(kBlock), 0, st, a, tiles);
}

extern "C" int nfmc_mh_accept_select_f32(const NfmcSelectArgs* args, nfmc_stream_t stream) {
    if (!args || !args->x || !args->x_prime) return NFMC_EINVAL;
    NfmcSelectArgs a = *args;
    if (a.n <= 0 || a.d <= 0 || a.n_carry < 0 || a.n_carry > 2) return NFMC_EINVAL;
    if (int rr = rng_default_only(a.rng)) return rr;
    if (a.d > 1024) return NFMC_ESHAPE;
    for (int k = 0; k < a.n_carry; ++k)
        if (!a.carry[k] || !a.carry_prime[k]) return NFMC_EINVAL;
    if (a.stats.sum_x && (!a.stats.sum_x2 || !a.stats.counters || !a.stats.scratch)) return NFMC_EINVAL;
    if (a.rng_tag != (int)kTagAccept && a.rng_tag != (int)kTagJump) return NFMC_EINVAL;
    const int dp = padded_d(a.d);
    const int lpc = dp / 4 > 64 ? 64 : dp / 4;  // CPL = 4 up to d = 256, CPL = 16 above
    const int cpl = dp / lpc;
    const int cpw = kWave / lpc;
    const int64_t tiles = (a.n + (int64_t)kWavesPerBlock * cpw - 1) / ((int64_t)kWavesPerBlock * cpw);
    const int grid = (int)(tiles < kMaxGrid ? tiles : kMaxGrid);
    if (a.stats.sum_x && a.stats.scratch_bytes < (int64_t)grid * (2 * dp + kStatTail) * (int64_t)sizeof(double))
        return NFMC_ESCRATCH;
    if (check_defer(a.stats, dp, a.d)) return NFMC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (cpl == 4) {
        switch (lpc) {
            case 1: launch_select<4, 1>(a, tiles, grid, st); break;
            case 2: launch_select<4, 2>(a, tiles, grid, st); break;
            case 4: launch_select<4, 4>(a, tiles, grid, st); break;
            case 8: launch_select<4, 8>(a, tiles, grid, st); break;
            case 16: launch_select<4, 16>(a, tiles, grid, st); break;
            case 32: launch_select<4, 32>(a, tiles, grid, st); break;
            default: launch_select<4, 64>(a, tiles, grid, st); break;
        }
    } else if (cpl == 8) {
        launch_select<8, 64>(a, tiles, grid, st);
    } else {
        launch_select<16, 64>(a, tiles, grid, st);
    }
    NFMC_HIP_CHECK_LAUNCH();
    if (a.stats.sum_x && !a.stats.defer) {
        hipLaunchKernelGGL(stats_finish_kernel<true>, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, st, a.stats.scratch, grid, dp, a.d, a.stats,
                           (unsigned long long)a.n);
        NFMC_HIP_CHECK_LAUNCH();
    }
    return NFMC_OK;
}

extern "C" int nfmc_langevin_propose_f32(const float* x, const float* grad_u, const float* inv_mass_diag,
                                         float step_size, int64_t n, int32_t d, const NfmcRng* rng, float* x_prime,
                                         nfmc_stream_t stream) {
    if (!x || !grad_u || !rng || !x_prime || n <= 0 || d <= 0 || !(step_size > 0.f)) return NFMC_EINVAL;
    if (int rr = rng_default_only(*rng)) return rr;
    const float sqrt2h = (float)sqrt(2.0 * (double)step_size);
    hipLaunchKernelGGL(langevin_propose_kernel, dim3(grid_for(n * ((d + 3) / 4))), dim3(kBlock), 0, (hipStream_t)stream,
                       x, grad_u, inv_mass_diag, step_size, sqrt2h, n, (int)d, *rng, x_prime);
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

extern "C" int nfmc_langevin_log_ratio_f32(const float* x, const float* x_prime, const float* u, const float* u_prime,
                                           const float* grad_u, const float* grad_u_prime, const float* inv_mass_diag,
                                           float step_size, int64_t n, int32_t d, float* log_ratio,
                                           nfmc_stream_t stream) {
    if (!x || !x_prime || !u || !u_prime || !grad_u || !grad_u_prime || !log_ratio) return NFMC_EINVAL;
    if (n <= 0 || d <= 0 || !(step_size > 0.f)) return NFMC_EINVAL;
    hipLaunchKernelGGL(langevin_log_ratio_kernel, dim3(grid_for(n * 64)), dim3(kBlock), 0, (hipStream_t)stream, x,
                       x_prime, u, u_prime, grad_u, grad_u_prime, inv_mass_diag, step_size, n, (int)d, log_ratio);
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

extern "C" int nfmc_limits(NfmcLimits* out) {
    if (!out) return NFMC_EINVAL;
    out->abi_version = NFMC_ABI_VERSION;
    out->max_d_sampler = 1024;
    out->max_d_flow = 512;
    out->max_hidden_valu = 32;
    out->max_hidden = 128;
    out->max_steps_per_call = NFMC_MAX_STEPS_PER_CALL;
    return NFMC_OK;
}

extern "C" const char* nfmc_error_string(int code) {
    switch (code) {
        case NFMC_OK: return "ok";
        case NFMC_EINVAL: return "invalid argument (NULL pointer or non-positive size)";
        case NFMC_ESHAPE: return "shape outside the supported range (see nfmc_limits)";
        case NFMC_EALIGN: return "pointer alignment";
        case NFMC_EUNSUPPORTED: return "no kernel for this request";
        case NFMC_ESCRATCH: return "statistics scratch buffer too small (nfmc_stats_scratch_bytes)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown nfmc error";
    }
}
