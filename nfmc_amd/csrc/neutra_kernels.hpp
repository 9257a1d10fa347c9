// (kernels and device code of neutra_kernels.hip and its per-rows-per-wave units neutra_kernels_r*.hip)
// K5 + K2: NeuTra -- HMC in the flow's latent space on the adjusted potential
//     U~(z) = U(f^-1(z)) - log|det J_{f^-1}(z)|                 (neutra.py:58-68)
// with grad U~ from a hand-written reverse sweep through the coupling stack (the reference gets it from
// torch.autograd through torchflows, hmc.py:40-48), fused with the leapfrog integrator, the Hamiltonian
// test and the statistics (hmc.py:61-77,96-126; mcmc/base.py:74-90).
//
// VALU path (conditioner width <= 32): one chain per lane, four (64, d) wave tiles in LDS
//   zt latent position | pt momentum | wt work (x and the layer inputs re-materialised on the way back) | gt gradient
// Coupling layers are invertible, so the reverse sweep needs no stored activations: the layer input is
// rebuilt from its output (v_b = alpha y_b + beta) and the conditioner's hidden stack is recomputed from
// the pass-through half.  Weights are wave-uniform (scalar loads).
//
// Adjacent leapfrog half-steps evaluate grad U~ at the same z; the value is computed once and reused
// (bitwise what the reference's two evaluations return), counters still report the reference's 2L(+2).
#pragma once

#include "mfma_device.hpp"

namespace nfmc {

constexpr int kMaxHiddenLayers = 4;

// Reverse sweep through one inverse coupling layer.  On entry wrow holds the layer OUTPUT y and grow
// dL/dy; on exit wrow holds the layer INPUT v and grow dL/dv, where L = U(x) + sum_layers sum_t log alpha.
template <int HP>
__device__ __forceinline__ void coupling_inverse_backward(float* __restrict__ wrow, float* __restrict__ grow,
                                                          const float* __restrict__ W, const FlowGeom& g, bool rev) {
    // recompute the hidden stack from the pass-through half, keeping every layer's activations
    float hs[kMaxHiddenLayers][HP];
    const float* b1 = W + (int64_t)g.d_a * HP;
#pragma unroll
    for (int k = 0; k < HP; ++k) hs[0][k] = b1[k];
    for (int j = 0; j < g.d_a; ++j) {
        const float xj = wrow[phys(j, g.d, rev)];
        const float* w = W + (int64_t)j * HP;
#pragma unroll
        for (int k = 0; k < HP; ++k) hs[0][k] = fmaf(w[k], xj, hs[0][k]);
    }
#pragma unroll
    for (int k = 0; k < HP; ++k) hs[0][k] = fast_tanh(hs[0][k]);
    const float* Wh0 = b1 + HP;
#pragma unroll
    for (int l = 1; l < kMaxHiddenLayers; ++l) {
        if (l < g.n_hl) {
            const float* Wh = Wh0 + (int64_t)(l - 1) * (HP * HP + HP);
            const float* bh = Wh + HP * HP;
#pragma unroll
            for (int k = 0; k < HP; ++k) hs[l][k] = bh[k];
#pragma unroll
            for (int i = 0; i < HP; ++i) {
#pragma unroll
                for (int k = 0; k < HP; ++k) hs[l][k] = fmaf(Wh[i * HP + k], hs[l - 1][i], hs[l][k]);
            }
#pragma unroll
            for (int k = 0; k < HP; ++k) hs[l][k] = fast_tanh(hs[l][k]);
        }
    }
    float hl[HP];  // activations of the last hidden layer
#pragma unroll
    for (int k = 0; k < HP; ++k) {
        hl[k] = hs[0][k];
#pragma unroll
        for (int l = 1; l < kMaxHiddenLayers; ++l)
            if (l == g.n_hl - 1) hl[k] = hs[l][k];
    }
    // output layer: transform parameters per target coordinate, their gradients, and dL/dh_last
    const float* W3 = w3_of(W, g, HP);
    const float* b3 = W3 + (int64_t)g.out_rows * HP;
    float dh[HP];
#pragma unroll
    for (int k = 0; k < HP; ++k) dh[k] = 0.f;
    if (g.n_bins > 0) {
        // rational-quadratic spline couplings ('c-rqnsf'): 3K - 1 conditioner outputs per target coordinate
        constexpr int P = 3 * kRqsBins - 1;
        for (int t = 0; t < g.d_b; ++t) {
            float raw[P], draw[P];
            const float* wr = W3 + (int64_t)t * P * HP;
#pragma unroll
            for (int q = 0; q < P; ++q) {
                float u = b3[t * P + q];
#pragma unroll
                for (int k = 0; k < HP; ++k) u = fmaf(wr[q * HP + k], hl[k], u);
                raw[q] = u;
            }
            const int p = phys(g.d_a + t, g.d, rev);
            float v, gv;
            rqs_inverse_backward(wrow[p], grow[p], raw, g.bound, draw, v, gv);
#pragma unroll
            for (int q = 0; q < P; ++q) {
#pragma unroll
                for (int k = 0; k < HP; ++k) dh[k] = fmaf(wr[q * HP + k], draw[q], dh[k]);
            }
            grow[p] = gv;
            wrow[p] = v;                                 // rebuild the layer input
        }
    } else
    for (int t = 0; t < g.d_b; ++t) {
        float ua = b3[t], ub = b3[g.d_b + t];
        const float* wa = W3 + (int64_t)t * HP;
        const float* wb = W3 + (int64_t)(g.d_b + t) * HP;
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            ua = fmaf(wa[k], hl[k], ua);
            ub = fmaf(wb[k], hl[k], ub);
        }
        const float alpha = fast_exp(fmaf(0.5f, ua, g.log1m)) + g.m;
        const float beta = 0.5f * ub;
        const float ra = __builtin_amdgcn_rcpf(alpha);
        const int p = phys(g.d_a + t, g.d, rev);
        const float y = wrow[p];
        const float gy = grow[p];
        const float gv = gy * ra;                        // dL/dv_t
        const float d_alpha = fmaf(-gv, y, ra);          // -gy*y/alpha + 1/alpha (log alpha term of this layer)
        const float d_ua = 0.5f * d_alpha * (alpha - g.m);
        const float d_ub = -0.5f * gv;
#pragma unroll
        for (int k = 0; k < HP; ++k) dh[k] = fmaf(wa[k], d_ua, fmaf(wb[k], d_ub, dh[k]));
        grow[p] = gv;
        wrow[p] = fmaf(alpha, y, beta);                  // rebuild the layer input
    }
    // back through the hidden stack
#pragma unroll
    for (int l = kMaxHiddenLayers - 1; l >= 1; --l) {
        if (l < g.n_hl) {
            const float* Wh = Wh0 + (int64_t)(l - 1) * (HP * HP + HP);
            float dpre[HP], dprev[HP];
#pragma unroll
            for (int k = 0; k < HP; ++k) dpre[k] = dh[k] * (1.f - hs[l][k] * hs[l][k]);
#pragma unroll
            for (int i = 0; i < HP; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < HP; ++k) acc = fmaf(Wh[i * HP + k], dpre[k], acc);
                dprev[i] = acc;
            }
#pragma unroll
            for (int k = 0; k < HP; ++k) dh[k] = dprev[k];
        }
    }
    float dpre[HP];
#pragma unroll
    for (int k = 0; k < HP; ++k) dpre[k] = dh[k] * (1.f - hs[0][k] * hs[0][k]);
    for (int j = 0; j < g.d_a; ++j) {
        const float* w = W + (int64_t)j * HP;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < HP; ++k) acc = fmaf(w[k], dpre[k], acc);
        grow[phys(j, g.d, rev)] += acc;
    }
}

// U~(z) and grad U~(z) for this lane's chain.  zrow: latent (tile columns in latent order), read only;
// wrow: scratch, ends holding z again (rebuilt); grow: gradient in the same column order as zrow.
template <int HP>
__device__ __forceinline__ float adjusted_potential_grad_row(const float* __restrict__ zrow, float* __restrict__ wrow,
                                                             float* __restrict__ grow, const NfmcRealNVP& f,
                                                             const FlowGeom& g, const NfmcPotential& pot) {
    for (int c = 0; c < g.d; ++c) wrow[c] = zrow[c];
    const float ld = flow_inverse_row<HP>(wrow, f, g);          // w = x, ld = logdet_inverse (neutra.py:60)
    const float u = potential_value_grad_row(wrow, grow, pot, g.d);  // U(x), dU/dx (neutra.py:62)
    // reverse sweep, mirror image of flow_inverse_row
    for (int c = 0; c < g.d; ++c) {                               // EA0^-1
        const float s = fast_exp(-f.ea0_log_scale[c]);
        grow[c] *= s;
    }
    for (int c = 0; c < g.d; ++c) wrow[c] = fmaf(fast_exp(f.ea0_log_scale[c]), wrow[c], f.ea0_shift[c]);
    for (int l = 0; l < g.n_coupling; ++l)
        coupling_inverse_backward<HP>(wrow, grow, f.weights + l * g.layer_stride, g, (l & 1) == 0);
    const bool rev_last = (g.n_coupling & 1) != 0;
    for (int c = 0; c < g.d; ++c) {                               // EA1^-1
        const int p = phys(c, g.d, rev_last);
        grow[p] *= fast_exp(-f.ea1_log_scale[c]);
    }
    return u - ld;                                                // neutra.py:63-64
}

constexpr int kNeutraBlock = 64;
constexpr int kNeutraSlots = 8;

// Rows (chains) per wave.  The kernels keep 3 (gradient) or 4 (trajectory) tiles of RPW x d floats in LDS; at 64 rows they fit
// the CU's 160 KB up to d ~ 156 / 208, and until round 3 wider events sent NeuTra to torch autograd on the GPU.  With 32 or 16
// rows per wave (the other lanes idle in the per-row phases, all 64 lanes still share the column phases: tile IO, statistics)
// every d <= 512 has a kernel.
static int neutra_rows_per_wave(int d, int tiles_of_d) {
    for (int rpw = 64; rpw >= 16; rpw >>= 1)
        if ((size_t)tiles_of_d * rpw * tile_stride(d) * sizeof(float) <= 150 * 1024) return rpw;
    return 0;
}
template <int RPW>
__device__ __forceinline__ void tile_load_rows(float* __restrict__ tile, int stride, const float* __restrict__ src, int64_t r0,
                                               int64_t n, int d, bool rev) {
    const int lane = threadIdx.x & 63;
    const int64_t rows = n - r0 < RPW ? n - r0 : RPW;
    const int total = (int)rows * d;
    const float* s = src + r0 * d;
    for (int i = lane; i < RPW * d; i += kWave) {
        const int r = i / d, c = i - r * d;
        tile[r * stride + (rev ? d - 1 - c : c)] = i < total ? s[i] : 0.f;  // rows beyond n: zeros
    }
}
template <int RPW>
__device__ __forceinline__ void tile_store_rows(const float* __restrict__ tile, int stride, float* __restrict__ dst, int64_t r0,
                                                int64_t n, int d, bool rev) {
    const int lane = threadIdx.x & 63;
    const int64_t rows = n - r0 < RPW ? n - r0 : RPW;
    const int total = (int)rows * d;
    float* o = dst + r0 * d;
    for (int i = lane; i < total; i += kWave) {
        const int r = i / d, c = i - r * d;
        o[i] = tile[r * stride + (rev ? d - 1 - c : c)];
    }
}

template <int HP, int RPW>
__global__ void __launch_bounds__(kNeutraBlock) neutra_potential_grad_kernel(NfmcRealNVP f, NfmcPotential pot,
                                                                             const float* __restrict__ z, int64_t n,
                                                                             float* __restrict__ u_out,
                                                                             float* __restrict__ grad_out,
                                                                             int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FlowGeom g = make_geom(f);
    const int stride = tile_stride(g.d);
    const int lane = threadIdx.x;
    float* zt = lds;
    float* wt = lds + RPW * stride;
    float* gt = lds + 2 * RPW * stride;
    const bool rev = (g.n_coupling & 1) != 0;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t r0 = tile * RPW;
        __syncthreads();
        tile_load_rows<RPW>(zt, stride, z, r0, n, g.d, rev);
        __syncthreads();
        if (lane < RPW) {
            const float u = adjusted_potential_grad_row<HP>(zt + lane * stride, wt + lane * stride, gt + lane * stride, f,
                                                            g, pot);
            if (r0 + lane < n && u_out) u_out[r0 + lane] = u;
        }
        __syncthreads();
        if (grad_out) tile_store_rows<RPW>(gt, stride, grad_out, r0, n, g.d, rev);
    }
}

template <int HP, int RPW>
__global__ void __launch_bounds__(kNeutraBlock) neutra_hmc_kernel(NfmcNeutraHmcArgs a, int64_t tiles, int dp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const NfmcRealNVP& f = a.flow;
    const FlowGeom g = make_geom(f);
    const int d = g.d;
    const int stride = tile_stride(d);
    const int lane = threadIdx.x;
    float* zt = lds;
    float* pt = lds + RPW * stride;
    float* wt = lds + 2 * RPW * stride;
    float* gt = lds + 3 * RPW * stride;
    const bool rowlane = lane < RPW;   // lanes that own a chain (all 64 lanes share the column phases)
    float* zr = zt + lane * stride;
    float* pr = pt + lane * stride;
    float* wr = wt + lane * stride;
    float* gr = gt + lane * stride;
    const int64_t n = a.n;
    const bool rev = (g.n_coupling & 1) != 0;
    const float h = a.step_size, hh = a.step_size / 2;

    double sx[kNeutraSlots], sxx[kNeutraSlots];
#pragma unroll
    for (int k = 0; k < kNeutraSlots; ++k) sx[k] = sxx[k] = 0.0;
    uint32_t n_acc = 0, n_bad = 0;

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t r0 = tile * RPW;
        const int64_t row = r0 + lane;
        const bool active = rowlane && row < n;
        const int rows = (int)(n - r0 < RPW ? n - r0 : RPW);
        const uint32_t gchain = (uint32_t)(a.rng.chain_offset + (uint64_t)row);
        __syncthreads();
        tile_load_rows<RPW>(zt, stride, a.z, r0, n, d, rev);
        __syncthreads();
        float u_cur = rowlane ? adjusted_potential_grad_row<HP>(zr, wr, gr, f, g, a.pot) : 0.f;  // U~(z), grad at the current state
        uint4 ur = make_uint4(0, 0, 0, 0);
        StoreCursor keep(a.samples);
        for (int s = 0; s < a.n_steps; ++s) {
            bool accept = false;
            float lr = 0.f;
            if (rowlane) {
            // momentum p = eps / sqrt(m)  (hmc.py:100); tile columns are latent positions: logical c <-> col latent_col(c)
            float kin0 = 0.f;
            if (a.rng.replay_normals) {
                const float* src = a.rng.replay_normals + ((int64_t)s * n + row) * d;
                for (int c = 0; c < d; ++c) {
                    const float m = a.inv_mass_diag ? a.inv_mass_diag[c] : 1.f;
                    const float v = (active ? src[c] : 0.f) * (1.f / sqrtf(m));
                    pr[latent_col(c, g)] = v;
                    kin0 = fmaf(v * v, m, kin0);
                }
            } else {
                const uint32_t k0 = (uint32_t)a.rng.seed, k1 = (uint32_t)(a.rng.seed >> 32);
                for (int b = 0; b < (d + 3) / 4; ++b) {
                    float zz[4];
                    philox_normal4(gchain, a.rng.step0 + (uint32_t)s, (uint32_t)b, kTagNoise, k0, k1, zz);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = 4 * b + k;
                        if (c < d) {
                            const float m = a.inv_mass_diag ? a.inv_mass_diag[c] : 1.f;
                            const float v = zz[k] * (1.f / sqrtf(m));
                            pr[latent_col(c, g)] = v;
                            kin0 = fmaf(v * v, m, kin0);
                        }
                    }
                }
            }
            const float h0 = u_cur + 0.5f * kin0;                      // hmc.py:103-106
            float u_new = u_cur;
            // gr holds grad U~(z) here (from the previous trajectory's last evaluation or the tile prologue);
            // a rejected trajectory restores it below by re-evaluating at the restored z.
            for (int l = 0; l < a.n_leapfrog; ++l) {                    // hmc.py:67-71
                for (int c = 0; c < d; ++c) {
                    const int col = latent_col(c, g);
                    const float m = a.inv_mass_diag ? a.inv_mass_diag[c] : 1.f;
                    const float pv = fmaf(-hh, gr[col], pr[col]);
                    pr[col] = pv;
                    zr[col] = fmaf(h, pv * m, zr[col]);
                }
                u_new = adjusted_potential_grad_row<HP>(zr, wr, gr, f, g, a.pot);
                for (int c = 0; c < d; ++c) pr[c] = fmaf(-hh, gr[c], pr[c]);
            }
            accept = true;
            if (a.adjust) {
                float kin1 = 0.f;
                for (int c = 0; c < d; ++c) {
                    const float m = a.inv_mass_diag ? a.inv_mass_diag[c] : 1.f;
                    const float v = pr[latent_col(c, g)];
                    kin1 = fmaf(v * v, m, kin1);
                }
                lr = h0 - (u_new + 0.5f * kin1);                       // hmc.py:107-111
                float u;
                if (a.rng.replay_uniforms) {
                    u = active ? a.rng.replay_uniforms[(int64_t)s * n + row] : 0.5f;
                } else {
                    const uint32_t step = a.rng.step0 + (uint32_t)s;
                    if (s == 0 || (step & 3u) == 0u)
                        ur = philox4x32_10(gchain, step >> 2, 0u, kTagAccept, (uint32_t)a.rng.seed,
                                           (uint32_t)(a.rng.seed >> 32));
                    u = u32_to_uniform(pick_word(ur, step & 3u));
                }
                accept = fast_ln(u) < lr;                               // hmc.py:112-113
                if (active && !(fabsf(lr) <= 3.0e38f)) n_bad++;
            }
            accept = accept && active;
            if (accept) {
                u_cur = u_new;
                n_acc++;
            } else {
                // restore the trajectory's start point from HBM (the tile is only written back on accept)
                for (int c = 0; c < d; ++c) zr[latent_col(c, g)] = active ? a.z[row * d + c] : 0.f;
            }
            }   // rowlane
            __syncthreads();
            // write accepted rows back so HBM always holds the current state (row-contiguous stores)
            if (accept)
                for (int c = 0; c < d; ++c) a.z[row * d + c] = zr[latent_col(c, g)];
            if (rowlane && !accept && s + 1 < a.n_steps) u_cur = adjusted_potential_grad_row<HP>(zr, wr, gr, f, g, a.pot);
            if (active) {
                if (a.masks_out) a.masks_out[(int64_t)s * n + row] = accept ? 1 : 0;
                if (a.log_ratio_out) a.log_ratio_out[(int64_t)s * n + row] = lr;
            }
            __syncthreads();
            if (a.stats.sum_x) {  // moments of the latent state (reference quirk: SURVEY App. C #1)
#pragma unroll
                for (int k = 0; k < kNeutraSlots; ++k) {
                    const int c = lane + 64 * k;
                    if (c < d) {
                        const int col = latent_col(c, g);
                        float t1 = 0.f, t2 = 0.f;
                        for (int r = 0; r < rows; ++r) {
                            const float v = zt[r * stride + col];
                            t1 += v;
                            t2 = fmaf(v, v, t2);
                        }
                        sx[k] += (double)t1;
                        sxx[k] += (double)t2;
                    }
                }
            }
            if (float* kept = keep.next(n * (int64_t)d)) tile_store_rows<RPW>(zt, stride, kept, r0, n, d, rev);
            __syncthreads();
        }
    }
    if (a.stats.sum_x) {
        for (int m = 1; m < kWave; m <<= 1) {
            n_acc += __shfl_xor(n_acc, m, kWave);
            n_bad += __shfl_xor(n_bad, m, kWave);
        }
        double* out = a.stats.scratch + (size_t)blockIdx.x * (2 * dp + kStatTail);
        for (int c = lane; c < 2 * dp + kStatTail; c += kWave) out[c] = 0.0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kNeutraSlots; ++k) {
            const int c = lane + 64 * k;
            if (c < d) {
                out[c] = sx[k];
                out[dp + c] = sxx[k];
            }
        }
        if (lane == 0) {
            out[2 * dp] = (double)n_acc;
            out[2 * dp + 1] = (double)n_bad;
        }
    }
}

static int check_flow_neutra(const NfmcRealNVP* f) {
    if (!f || !f->ea0_log_scale || !f->ea0_shift || !f->ea1_log_scale || !f->ea1_shift) return NFMC_EINVAL;
    if (f->d <= 0 || f->n_coupling < 0 || f->n_hidden <= 0 || f->n_hidden_layers <= 0) return NFMC_EINVAL;
    if (f->n_coupling > 0 && !f->weights) return NFMC_EINVAL;
    if (f->d < 2 && f->n_coupling > 0) return NFMC_ESHAPE;
    if (f->d > 512) return NFMC_ESHAPE;
    if (f->n_hidden_layers > kMaxHiddenLayers) return NFMC_ESHAPE;
    if (f->n_hidden > 32 || (f->n_bins != 0 && f->n_bins != kRqsBins)) return NFMC_EUNSUPPORTED;
    if (f->n_coupling > 0 && f->layer_stride < nfmc_coupling_layer_floats(f->d, f->n_hidden, f->n_hidden_layers, f->n_bins))
        return NFMC_EINVAL;
    return NFMC_OK;
}

static int hp_bucket_n(int h) { return h <= 4 ? 4 : (h <= 8 ? 8 : (h <= 16 ? 16 : 32)); }

template <class K>
static int set_lds_n(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return NFMC_ESHAPE;
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// The kernels are instantiated per rows-per-wave variant in a unit of its own (neutra_kernels_r64 / r32 / r16.hip: together
// they were 222 s of compile time in one unit, the critical path of the build); the C entry points in neutra_kernels.hip
// call these launchers.  hp: conditioner width bucket (hp_bucket_n).  Return 0 or an error code.
#define NFMC_NEUTRA_RPW_DECL(RPWV)                                                                                              \
    int neutra_grad_launch_r##RPWV(int hp, size_t lds, int grid, hipStream_t st, const NfmcRealNVP& f, const NfmcPotential& pot, \
                                  const float* z, int64_t n, float* u_out, float* grad_out, int64_t tiles);                      \
    int neutra_hmc_launch_r##RPWV(int hp, size_t lds, int grid, hipStream_t st, const NfmcNeutraHmcArgs& a, int64_t tiles, int dp);
NFMC_NEUTRA_RPW_DECL(64)
NFMC_NEUTRA_RPW_DECL(32)
NFMC_NEUTRA_RPW_DECL(16)

#define NFMC_NEUTRA_HP_SWITCH(CALL)                             \
    if (hp == 4) { constexpr int HP = 4; CALL; }                \
    else if (hp == 8) { constexpr int HP = 8; CALL; }           \
    else if (hp == 16) { constexpr int HP = 16; CALL; }         \
    else if (hp == 32) { constexpr int HP = 32; CALL; }         \
    else return NFMC_EUNSUPPORTED;

#define NFMC_NEUTRA_RPW_UNIT(RPWV)                                                                                              \
    namespace nfmc {                                                                                                            \
    int neutra_grad_launch_r##RPWV(int hp, size_t lds, int grid, hipStream_t st, const NfmcRealNVP& f, const NfmcPotential& pot, \
                                  const float* z, int64_t n, float* u_out, float* grad_out, int64_t tiles) {                     \
        int rc = 0;                                                                                                             \
        NFMC_NEUTRA_HP_SWITCH({                                                                                                 \
            if ((rc = set_lds_n(neutra_potential_grad_kernel<HP, RPWV>, lds))) return rc;                                       \
            hipLaunchKernelGGL((neutra_potential_grad_kernel<HP, RPWV>), dim3(grid), dim3(kNeutraBlock), lds, st, f, pot, z, n,  \
                               u_out, grad_out, tiles);                                                                         \
        })                                                                                                                      \
        return 0;                                                                                                               \
    }                                                                                                                           \
    int neutra_hmc_launch_r##RPWV(int hp, size_t lds, int grid, hipStream_t st, const NfmcNeutraHmcArgs& a, int64_t tiles, int dp) { \
        int rc = 0;                                                                                                             \
        NFMC_NEUTRA_HP_SWITCH({                                                                                                 \
            if ((rc = set_lds_n(neutra_hmc_kernel<HP, RPWV>, lds))) return rc;                                                  \
            hipLaunchKernelGGL((neutra_hmc_kernel<HP, RPWV>), dim3(grid), dim3(kNeutraBlock), lds, st, a, tiles, dp);           \
        })                                                                                                                      \
        return 0;                                                                                                               \
    }                                                                                                                           \
    }

}  // namespace nfmc
