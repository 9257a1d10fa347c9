// K3/K4 (+K6, K7): RealNVP forward / inverse / log-density and the flow-proposal Metropolis step.
//
// Replaces the torchflows calls of the reference:
//   Flow.log_prob / bijection.forward          jump.py:218, imh.py:214
//   Flow.sample(n, return_log_prob=True) / bijection.inverse   jump.py:205, imh.py:221, neutra.py:60
// and, fused around them, the jump of JumpNFMC.sample (jump.py:205-243) / the loop body of
// FixedIMH.sample (imh.py:220-249): target calls, log alpha (util.py:382-392), log u < log alpha,
// masked state + cached-log-q update, counters, streaming moments, sample store.
//
// One wave per workgroup, one chain per lane, the (64, d) tiles of the wave in LDS (flow_device.hpp).
#include "mfma_device.hpp"

namespace nfmc {

constexpr int kFlowBlock = 64;

// flow_b_kernels.hip: register-layout path for narrow conditioners
int flow_mh_b_launch(const NfmcFlowMhArgs& a, hipStream_t st, int* grid_out, int* dp_out, bool dry);
constexpr int kMaxSlots = 8;  // d <= 512 -> at most 8 coordinates per lane in the column-sum pass

template <int HP>
__global__ void __launch_bounds__(kFlowBlock) realnvp_forward_kernel(NfmcRealNVP f, const float* __restrict__ x,
                                                                     int64_t n, float* __restrict__ z,
                                                                     float* __restrict__ logdet,
                                                                     float* __restrict__ log_prob, int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FlowGeom g = make_geom(f);
    const int stride = tile_stride(g.d);
    const int lane = threadIdx.x;
    float* row = lds + lane * stride;
    float* hbuf = lds + 64 * stride;  // wide conditioners only
    const bool rev = (g.n_coupling & 1) != 0;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t r0 = tile * 64;
        __syncthreads();
        tile_load(lds, stride, x, r0, n, g.d);
        __syncthreads();
        const float ld = flow_forward_row<HP>(row, f, g, hbuf);
        float ss = 0.f;
        for (int c = 0; c < g.d; ++c) ss = fmaf(row[c], row[c], ss);
        if (r0 + lane < n) {
            if (logdet) logdet[r0 + lane] = ld;
            if (log_prob) log_prob[r0 + lane] = -0.5f * ss - 0.5f * (float)g.d * kLog2Pi + ld;
        }
        __syncthreads();
        if (z) tile_store(lds, stride, z, r0, n, g.d, rev);
    }
}

// latent for this lane's chain into its row (tile columns in latent order); returns sum z^2
__device__ __forceinline__ float draw_latent_row(float* __restrict__ row, const FlowGeom& g, const NfmcRng& rng,
                                                 int64_t chain_row, int64_t n, int s) {
    float ss = 0.f;
    if (rng.replay_normals) {
        const float* p = rng.replay_normals + ((int64_t)s * n + chain_row) * g.d;
        for (int c = 0; c < g.d; ++c) {
            const float v = chain_row < n ? p[c] : 0.f;
            row[latent_col(c, g)] = v;
            ss = fmaf(v, v, ss);
        }
    } else {
        const uint32_t gchain = (uint32_t)(rng.chain_offset + (uint64_t)chain_row);
        const uint32_t k0 = (uint32_t)rng.seed, k1 = (uint32_t)(rng.seed >> 32);
        const int nblk = (g.d + 3) >> 2;
        for (int b = 0; b < nblk; ++b) {
            float zz[4];
            philox_normal4(gchain, rng.step0 + (uint32_t)s, (uint32_t)b, kTagLatent, k0, k1, zz);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = 4 * b + k;
                if (c < g.d) {
                    row[latent_col(c, g)] = zz[k];
                    ss = fmaf(zz[k], zz[k], ss);
                }
            }
        }
    }
    return ss;
}

template <int HP>
__global__ void __launch_bounds__(kFlowBlock) realnvp_inverse_kernel(NfmcRealNVP f, const float* __restrict__ z,
                                                                     int64_t n, float* __restrict__ x,
                                                                     float* __restrict__ logdet,
                                                                     float* __restrict__ log_q, NfmcRng rng,
                                                                     int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FlowGeom g = make_geom(f);
    const int stride = tile_stride(g.d);
    const int lane = threadIdx.x;
    float* row = lds + lane * stride;
    float* hbuf = lds + 64 * stride;  // wide conditioners only
    const bool rev = (g.n_coupling & 1) != 0;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t r0 = tile * 64;
        float ss = 0.f;
        __syncthreads();
        if (z) {
            tile_load(lds, stride, z, r0, n, g.d, rev);
            __syncthreads();
            for (int c = 0; c < g.d; ++c) ss = fmaf(row[c], row[c], ss);
        } else {
            ss = draw_latent_row(row, g, rng, r0 + lane, n, 0);
        }
        const float ld = flow_inverse_row<HP>(row, f, g, hbuf);
        if (r0 + lane < n) {
            if (logdet) logdet[r0 + lane] = ld;
            if (log_q) log_q[r0 + lane] = -0.5f * ss - 0.5f * (float)g.d * kLog2Pi - ld;
        }
        __syncthreads();
        if (x) tile_store(lds, stride, x, r0, n, g.d);
    }
}

template <int HP>
__global__ void __launch_bounds__(kFlowBlock) flow_mh_kernel(NfmcFlowMhArgs a, int64_t tiles, int dp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const NfmcRealNVP& f = a.flow;
    const FlowGeom g = make_geom(f);
    const int d = g.d;
    const int stride = tile_stride(d);
    const int lane = threadIdx.x;
    float* xt = lds;                  // current states of the wave's 64 chains
    float* pt = lds + 64 * stride;    // proposals
    float* xr = xt + lane * stride;
    float* pr = pt + lane * stride;
    float* hbuf = lds + 128 * stride;  // wide conditioners only
    const int64_t n = a.n;
    const float base_c = -0.5f * (float)d * kLog2Pi;

    double sx[kMaxSlots], sxx[kMaxSlots];
#pragma unroll
    for (int k = 0; k < kMaxSlots; ++k) sx[k] = sxx[k] = 0.0;
    uint32_t n_acc = 0, n_bad = 0;

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t r0 = tile * 64;
        const int64_t row = r0 + lane;
        const bool active = row < n;
        const int rows = (int)(n - r0 < 64 ? n - r0 : 64);
        __syncthreads();
        tile_load(xt, stride, a.x, r0, n, d);
        __syncthreads();
        float u_x = potential_row(xr, a.pot, d);  // jump.py:212 / imh.py:224
        float f_x;
        if (a.logq_cached) {
            f_x = active ? a.logq[row] : 0.f;
        } else {  // flow.log_prob(x): jump.py:218 / imh.py:214
            for (int c = 0; c < d; ++c) pr[c] = xr[c];
            const float ld = flow_forward_row<HP>(pr, f, g, hbuf);
            float ss = 0.f;
            for (int c = 0; c < d; ++c) ss = fmaf(pr[c], pr[c], ss);
            f_x = -0.5f * ss + base_c + ld;
        }
        StoreCursor keep(a.samples);
        for (int s = 0; s < a.n_steps; ++s) {
            const float ss = draw_latent_row(pr, g, a.rng, row, n, s);       // flow.sample: jump.py:205 / imh.py:221
            const float ldi = flow_inverse_row<HP>(pr, f, g, hbuf);
            const float f_xp = -0.5f * ss + base_c - ldi;
            const float u_xp = potential_row(pr, a.pot, d);                   // jump.py:213 / imh.py:225
            const float lr = (-u_xp) - (-u_x) + f_x - f_xp;                   // util.py:392
            bool accept = true;
            if (a.adjusted) {
                float u;
                if (a.rng.replay_uniforms) {
                    u = active ? a.rng.replay_uniforms[(int64_t)s * n + row] : 0.5f;
                } else {
                    const uint4 r = philox4x32_10((uint32_t)(a.rng.chain_offset + (uint64_t)row),
                                                  a.rng.step0 + (uint32_t)s, 0u, kTagJump, (uint32_t)a.rng.seed,
                                                  (uint32_t)(a.rng.seed >> 32));
                    u = u32_to_uniform(r.x);
                }
                accept = fast_ln(u) < lr;                                      // jump.py:225 / imh.py:229-230
                if (active && !(fabsf(lr) <= 3.0e38f)) n_bad++;
            }
            accept = accept && active;
            if (accept) {                                                      // jump.py:231 / imh.py:232-233
                for (int c = 0; c < d; ++c) xr[c] = pr[c];
                f_x = f_xp;
                u_x = u_xp;
                n_acc++;
            }
            if (active) {
                if (a.masks_out) a.masks_out[(int64_t)s * n + row] = accept ? 1 : 0;
                if (a.log_ratio_out) a.log_ratio_out[(int64_t)s * n + row] = lr;
            }
            __syncthreads();
            if (a.stats.sum_x) {  // K7: column sums of the wave tile, lane owns columns lane, lane+64, ...
#pragma unroll
                for (int k = 0; k < kMaxSlots; ++k) {
                    const int c = lane + 64 * k;
                    if (c < d) {
                        float t1 = 0.f, t2 = 0.f;
                        for (int r = 0; r < rows; ++r) {
                            const float v = xt[r * stride + c];
                            t1 += v;
                            t2 = fmaf(v, v, t2);
                        }
                        sx[k] += (double)t1;
                        sxx[k] += (double)t2;
                    }
                }
            }
            if (float* kept = keep.next(n * (int64_t)d)) tile_store(xt, stride, kept, r0, n, d);
            __syncthreads();
        }
        tile_store(xt, stride, a.x, r0, n, d);
        if (active) a.logq[row] = f_x;
    }
    if (a.stats.sum_x) {
        for (int m = 1; m < kWave; m <<= 1) {
            n_acc += __shfl_xor(n_acc, m, kWave);
            n_bad += __shfl_xor(n_bad, m, kWave);
        }
        double* out = a.stats.scratch + (size_t)blockIdx.x * (2 * dp + kStatTail);
        const bool defer = a.stats.defer != 0;   // deferred: add to the caller-zeroed slab (nfmc_stats_fold_f32)
        const int slot = defer ? a.stats.tail_slot : 0;
        if (!defer)
            for (int c = lane; c < 2 * dp + kStatTail; c += kWave) out[c] = 0.0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kMaxSlots; ++k) {
            const int c = lane + 64 * k;
            if (c < d) {
                out[c] = defer ? out[c] + sx[k] : sx[k];
                out[dp + c] = defer ? out[dp + c] + sxx[k] : sxx[k];
            }
        }
        if (lane == 0) {
            out[2 * dp + slot] = defer ? out[2 * dp + slot] + (double)n_acc : (double)n_acc;
            out[2 * dp + slot + 1] = defer ? out[2 * dp + slot + 1] + (double)n_bad : (double)n_bad;
        }
    }
}

// ------------------------------------------------------------------------------------------------
static int check_flow(const NfmcRealNVP* f) {
    if (!f || !f->ea0_log_scale || !f->ea0_shift || !f->ea1_log_scale || !f->ea1_shift) return NFMC_EINVAL;
    if (f->d <= 0 || f->n_coupling < 0 || f->n_hidden <= 0 || f->n_hidden_layers <= 0) return NFMC_EINVAL;
    if (f->n_coupling > 0 && !f->weights) return NFMC_EINVAL;
    if (f->d < 2 && f->n_coupling > 0) return NFMC_ESHAPE;
    if (f->d > 512) return NFMC_ESHAPE;
    if (f->n_hidden > 128) return NFMC_EUNSUPPORTED;
    if (!(f->min_scale >= 0.f && f->min_scale <= 1.f)) return NFMC_EINVAL;   // 1 = additive coupling (exp(-inf) + 1)
    if (f->n_bins != 0) {   // spline couplings: 8 bins, VALU conditioners
        if (f->n_bins != kRqsBins || !(f->spline_bound > 0.f)) return NFMC_EINVAL;
        if (f->n_hidden > 32) return NFMC_EUNSUPPORTED;
    }
    if (f->n_coupling > 0 &&
        f->layer_stride < nfmc_coupling_layer_floats(f->d, f->n_hidden, f->n_hidden_layers, f->n_bins))
        return NFMC_EINVAL;
    return NFMC_OK;
}

// shapes served by the matrix-core flow kernels (flow_mfma.hip); NFMC_FLOW_NO_MFMA=1 keeps the VALU kernels (A/B, tests)
static bool use_mfma_flow(const NfmcRealNVP* f) {
    return f->n_bins == 0 && f->n_coupling > 0 && nfmc_mfma_supported(f->d, f->n_hidden, f->n_hidden_layers) &&
           !getenv("NFMC_FLOW_NO_MFMA");
}

// d = 256 / 512 with a wide conditioner: the streamed matrix-core kernels (mfma_wide.hip)
static bool use_mfma_wide(const NfmcRealNVP* f) {
    return f->n_bins == 0 && f->n_coupling > 0 && nfmc_mfma_wide_supported(f->d, f->n_hidden, f->n_hidden_layers) &&
           !getenv("NFMC_FLOW_NO_MFMA");
}

static bool al16(const void* p) { return ((uintptr_t)p & 15u) == 0; }   // NULL counts as aligned

static int hp_bucket(int h) { return h <= 4 ? 4 : (h <= 8 ? 8 : (h <= 16 ? 16 : 32)); }

constexpr size_t kMaxLdsBytes = 160 * 1024;   // per workgroup on gfx950

template <class K>
static int set_lds(K kernel, size_t bytes) {
    if (bytes > kMaxLdsBytes) return NFMC_EUNSUPPORTED;   // the wave tile(s) of this d do not fit the LDS
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // namespace nfmc

using namespace nfmc;

extern "C" int32_t nfmc_realnvp_padded_hidden(int32_t n_hidden) {
    if (n_hidden <= 0) return 0;
    if (n_hidden <= 32) return hp_bucket(n_hidden);
    return n_hidden <= 64 ? 64 : (n_hidden <= 128 ? 128 : 0);
}

extern "C" int64_t nfmc_flow_scratch_bytes(const NfmcRealNVP* flow, int64_t n, int32_t with_gradient) {
    if (!flow || n <= 0 || flow->n_bins != 0 || flow->n_coupling <= 0) return 0;
    if (!nfmc_mfma_wide_supported(flow->d, flow->n_hidden, flow->n_hidden_layers)) return 0;
    return nfmc_wide_slab_floats(n, flow->d, with_gradient ? 2 : 1) * (int64_t)sizeof(float);
}

extern "C" int64_t nfmc_realnvp_layer_floats(int32_t d, int32_t n_hidden, int32_t n_hidden_layers) {
    if (d <= 0 || n_hidden <= 0 || n_hidden_layers <= 0) return 0;
    const int64_t hp = nfmc_realnvp_padded_hidden(n_hidden);
    if (hp == 0) return 0;
    if (hp > 32) return mfma_layer_floats(d, (int)hp, n_hidden_layers);
    const int64_t d_a = d / 2, d_b = d - d_a;
    return d_a * hp + hp + (int64_t)(n_hidden_layers - 1) * (hp * hp + hp) + 2 * d_b * hp + 2 * d_b;
}

extern "C" int64_t nfmc_coupling_layer_floats(int32_t d, int32_t n_hidden, int32_t n_hidden_layers, int32_t n_bins) {
    if (n_bins == 0) return nfmc_realnvp_layer_floats(d, n_hidden, n_hidden_layers);
    if (d <= 0 || n_hidden <= 0 || n_hidden > 32 || n_hidden_layers <= 0 || n_bins != kRqsBins) return 0;
    const int64_t hp = nfmc_realnvp_padded_hidden(n_hidden);
    const int64_t d_a = d / 2, d_b = d - d_a, rows = (3 * (int64_t)n_bins - 1) * d_b;
    return d_a * hp + hp + (int64_t)(n_hidden_layers - 1) * (hp * hp + hp) + rows * hp + rows;
}

#define NFMC_HP_DISPATCH(HPV, CALL)             \
    switch (HPV) {                              \
        case 4: { constexpr int HP = 4; CALL; } break;   \
        case 8: { constexpr int HP = 8; CALL; } break;   \
        case 16: { constexpr int HP = 16; CALL; } break; \
        case 32: { constexpr int HP = 32; CALL; } break; \
        case 64: { constexpr int HP = 64; CALL; } break; \
        default: { constexpr int HP = 128; CALL; } break; \
    }

// LDS floats behind the wave tiles for the wide conditioners' activation buffer
static size_t hbuf_bytes(int n_hidden) { return n_hidden > 32 ? (size_t)nfmc_realnvp_padded_hidden(n_hidden) * 64 * sizeof(float) : 0; }

extern "C" int nfmc_realnvp_forward_f32(const NfmcRealNVP* flow, const float* x, int64_t n, float* z, float* logdet,
                                        float* log_prob, nfmc_stream_t stream) {
    int rc = check_flow(flow);
    if (rc) return rc;
    if (!x || n <= 0) return NFMC_EINVAL;
    if (use_mfma_flow(flow) && al16(x) && al16(z)) return nfmc_realnvp_forward_mfma_f32(flow, x, n, z, logdet, log_prob, stream);
    if (use_mfma_wide(flow) && al16(x) && al16(z)) {   // a caller that brought no workspace (NfmcRealNVP.scratch): the kernels below
        rc = nfmc_realnvp_forward_wide_f32(flow, x, n, z, logdet, log_prob, stream);
        if (!(rc == NFMC_ESCRATCH && !flow->scratch)) return rc;
    }
    const int64_t tiles = (n + 63) / 64;
    const int grid = (int)(tiles < 4 * kMaxGrid ? tiles : 4 * kMaxGrid);
    const size_t lds = (size_t)64 * tile_stride(flow->d) * sizeof(float) + hbuf_bytes(flow->n_hidden);
    hipStream_t st = (hipStream_t)stream;
    NFMC_HP_DISPATCH(nfmc_realnvp_padded_hidden(flow->n_hidden), {
        if ((rc = set_lds(realnvp_forward_kernel<HP>, lds))) return rc;
        hipLaunchKernelGGL((realnvp_forward_kernel<HP>), dim3(grid), dim3(kFlowBlock), lds, st, *flow, x, n, z, logdet,
                           log_prob, tiles);
    })
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

extern "C" int nfmc_realnvp_inverse_f32(const NfmcRealNVP* flow, const float* z, int64_t n, float* x, float* logdet,
                                        float* log_q, const NfmcRng* rng, nfmc_stream_t stream) {
    int rc = check_flow(flow);
    if (rc) return rc;
    if (n <= 0 || (!z && !rng)) return NFMC_EINVAL;
    NfmcRng r = {};
    if (rng) r = *rng;
    if (int rr = rng_default_only(r)) return rr;
    r.replay_normals = nullptr;  // explicit latents come through `z`
    if (use_mfma_flow(flow) && al16(x) && al16(z)) return nfmc_realnvp_inverse_mfma_f32(flow, z, n, x, logdet, log_q, &r, stream);
    if (use_mfma_wide(flow) && al16(x) && al16(z)) {
        rc = nfmc_realnvp_inverse_wide_f32(flow, z, n, x, logdet, log_q, &r, stream);
        if (!(rc == NFMC_ESCRATCH && !flow->scratch)) return rc;
    }
    const int64_t tiles = (n + 63) / 64;
    const int grid = (int)(tiles < 4 * kMaxGrid ? tiles : 4 * kMaxGrid);
    const size_t lds = (size_t)64 * tile_stride(flow->d) * sizeof(float) + hbuf_bytes(flow->n_hidden);
    hipStream_t st = (hipStream_t)stream;
    NFMC_HP_DISPATCH(nfmc_realnvp_padded_hidden(flow->n_hidden), {
        if ((rc = set_lds(realnvp_inverse_kernel<HP>, lds))) return rc;
        hipLaunchKernelGGL((realnvp_inverse_kernel<HP>), dim3(grid), dim3(kFlowBlock), lds, st, *flow, z, n, x, logdet,
                           log_q, r, tiles);
    })
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

static int check_flow_mh(const NfmcFlowMhArgs& a) {
    int rc = check_flow(&a.flow);
    if (rc) return rc;
    if (!a.x || !a.logq || a.n <= 0 || a.n_steps <= 0) return NFMC_EINVAL;
    if (a.n_steps > NFMC_MAX_STEPS_PER_CALL) return NFMC_ESHAPE;
    if (a.pot.kind != NFMC_POT_QUADRATIC && a.pot.kind != NFMC_POT_FUNNEL) return NFMC_EUNSUPPORTED;
    if (a.stats.sum_x && (!a.stats.sum_x2 || !a.stats.counters || !a.stats.scratch)) return NFMC_EINVAL;
    if (a.adjusted && (a.rng.replay_normals != nullptr) != (a.rng.replay_uniforms != nullptr)) return NFMC_EINVAL;
    if (!store_ok(a.samples) || !rng_rounds_ok(a.rng, true)) return NFMC_EINVAL;
    return NFMC_OK;
}

static size_t flow_mh_tile_lds(const NfmcRealNVP& f) { return (size_t)2 * 64 * tile_stride(f.d) * sizeof(float) + hbuf_bytes(f.n_hidden); }

extern "C" int nfmc_flow_mh_supported_f32(const NfmcFlowMhArgs* args) {
    if (!args) return NFMC_EINVAL;
    const NfmcFlowMhArgs& a = *args;
    int rc = check_flow_mh(a);
    if (rc) return rc;
    int grid = 0, dp = 0;
    rc = getenv("NFMC_FLOW_TILE_PATH") ? NFMC_EUNSUPPORTED : flow_mh_b_launch(a, nullptr, &grid, &dp, true);
    if (rc != NFMC_EUNSUPPORTED) return rc;
    if (rng_rounds(a.rng) != 10) return NFMC_EUNSUPPORTED;   // the opt-in stream exists in the register kernels only
    if (use_mfma_flow(&a.flow) && al16(a.x) && al16(a.samples.base)) return NFMC_OK;
    // wide conditioners at the streamed matrix-core shapes (mfma_wide.hip): the caller composes the step from the flow's
    // own passes, which run there (d = 256, H = 128 x 2, 65536 chains: 0.53 ms per pass against 5.3 for the
    // one-chain-per-lane kernel this entry point would use; nfmc_flow_mh_steps_f32 itself still takes the call).  Round 4: the
    // streamed matrix-core kernel takes them when the caller supplies its workspace and the rows are 16-byte aligned
    if (use_mfma_wide(&a.flow))
        return (a.flow.scratch && a.flow.scratch_bytes >= nfmc_flow_scratch_bytes(&a.flow, a.n, 1) && al16(a.x) && al16(a.samples.base) &&
                rng_rounds(a.rng) == 10)
                   ? NFMC_OK
                   : NFMC_EUNSUPPORTED;
    return flow_mh_tile_lds(a.flow) <= kMaxLdsBytes ? NFMC_OK : NFMC_EUNSUPPORTED;
}

extern "C" int nfmc_flow_mh_steps_f32(const NfmcFlowMhArgs* args, nfmc_stream_t stream) {
    if (!args) return NFMC_EINVAL;
    NfmcFlowMhArgs a = *args;
    int rc = check_flow_mh(a);
    if (rc) return rc;
    const int d = a.flow.d;
    int dp = padded_d(d);
    hipStream_t st = (hipStream_t)stream;
    int grid = 0;
    rc = getenv("NFMC_FLOW_TILE_PATH") ? NFMC_EUNSUPPORTED : flow_mh_b_launch(a, st, &grid, &dp, false);
    if (rc == NFMC_EUNSUPPORTED && rng_rounds(a.rng) != 10) return NFMC_EUNSUPPORTED;
    if (rc == NFMC_EUNSUPPORTED && use_mfma_flow(&a.flow) && al16(a.x) && al16(a.samples.base)) {
        // wide conditioners at d = 64 / 128 (16-byte aligned rows): matrix cores
        rc = nfmc_flow_mh_steps_mfma_f32(a, stream, &grid, &dp);
        if (rc) return rc;
    } else if (rc == NFMC_EUNSUPPORTED && use_mfma_wide(&a.flow) && a.flow.scratch && al16(a.x) && al16(a.samples.base)) {
        // wide conditioners at the other multiples of 32 up to d = 512: matrix cores, state streamed through the caller's slab
        rc = nfmc_flow_mh_steps_wide_f32(a, stream, &grid, &dp);
        if (rc) return rc;
    } else if (rc == NFMC_EUNSUPPORTED) {  // wider conditioners: one chain per lane, wave tiles in LDS
        dp = padded_d(d);
        const int64_t tiles = (a.n + 63) / 64;
        grid = (int)(tiles < kMaxGrid ? tiles : kMaxGrid);
        if (a.stats.sum_x && a.stats.scratch_bytes < (int64_t)grid * (2 * dp + kStatTail) * (int64_t)sizeof(double))
            return NFMC_ESCRATCH;
        if (check_defer(a.stats, dp, d)) return NFMC_EINVAL;
        const size_t lds = flow_mh_tile_lds(a.flow);
        NFMC_HP_DISPATCH(nfmc_realnvp_padded_hidden(a.flow.n_hidden), {
            if ((rc = set_lds(flow_mh_kernel<HP>, lds))) return rc;
            hipLaunchKernelGGL((flow_mh_kernel<HP>), dim3(grid), dim3(kFlowBlock), lds, st, a, tiles, dp);
        })
    } else if (rc) {
        return rc;
    }
    NFMC_HIP_CHECK_LAUNCH();
    if (a.stats.sum_x && !a.stats.defer) {
        hipLaunchKernelGGL(stats_finish_kernel<true>, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, st, a.stats.scratch, grid, dp, d, a.stats,
                           (unsigned long long)a.n * (unsigned long long)a.n_steps);
        NFMC_HIP_CHECK_LAUNCH();
    }
    return NFMC_OK;
}
