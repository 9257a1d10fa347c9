// RealNVP forward / inverse sweeps on the matrix cores for the wave's 16 chains (layout and GEMM scheme:
// mfma_device.hpp).  Shared by the flow kernels (flow_mfma.hip: bijection.forward / inverse, Flow.log_prob / sample,
// the flow-proposal Metropolis step) and by NeuTra (neutra_mfma.hip), whose gradient starts with the inverse sweep.
#pragma once

#include "mfma_device.hpp"

namespace nfmc {

// ---- one coupling layer in C layout.  INVERSE: target half v_b = (y_b - beta) / alpha, else z_b = alpha x_b + beta.
// Returns this lane's share of the layer's logdet in THAT direction.  Three steps of the weight pipeline.
template <int TD, int TH, int NHL, bool REV, bool INVERSE>
__device__ __forceinline__ float coupling_c(f32x4 (&x)[TD], const MLayer& L, float mscale, float log1m,
                                            WeightPipe& wp, int col, int half) {
    constexpr int TS = TD / 2, SRC0 = REV ? TS : 0, TGT0 = REV ? 0 : TS, D2 = 8 * TD, hp = 16 * TH;
    f32x4 hl[TH];   // activations of the last hidden layer
    {
        f32x4 src[TS], h1[TH];
#pragma unroll
        for (int ms = 0; ms < TS; ++ms) src[ms] = x[SRC0 + ms];
        if constexpr (NHL > 1) hidden_stack<TS, TH, NHL>(src, h1, hl, L, REV, wp, col, half);
        else hidden_stack<TS, TH, NHL>(src, hl, h1, L, REV, wp, col, half);
    }
    wp.template stage<hp, D2, 1, D2, 2 * D2>(L.W3, REV, false, L.b3, 2 * D2, REV);
    const float* img = wp.img();
    const float* vec = wp.vec();
    float ld = 0.f;
    f32x4 ua2[2], ub2[2];   // per parity of the target tile: the next tile's bias is read while this one is in flight
    // steps 2 mt / 2 mt + 1: the alpha / beta rows of target tile mt (a pair: separate accumulators, same operand)
    gemm_phase<TH, 2 * TS>(
        [&](int i) { return img + (16 * ((i & 1) * TS + (i >> 1)) + col) * (hp + 4) + 4 * half; },
        [&](int i) { ((i & 1) ? ub2 : ua2)[(i >> 1) & 1] = vec_tile(vec, (i & 1) * TS + (i >> 1), half); },
        [&](int i) -> f32x4& { return ((i & 1) ? ub2 : ua2)[(i >> 1) & 1]; },
        [&](int) -> const f32x4(&)[TH] { return hl; },
        [&](int i) {
            if ((i & 1) == 0) return;
            const int mt = i >> 1;
            const f32x4 ua = ua2[mt & 1], ub = ub2[mt & 1];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float alpha = fast_exp(fmaf(0.5f, ua[t], log1m)) + mscale;
                const float la = fast_ln(alpha);
                if constexpr (INVERSE) {
                    x[TGT0 + mt][t] = (x[TGT0 + mt][t] - 0.5f * ub[t]) * __builtin_amdgcn_rcpf(alpha);
                    ld -= la;
                } else {
                    x[TGT0 + mt][t] = fmaf(alpha, x[TGT0 + mt][t], 0.5f * ub[t]);
                    ld += la;
                }
            }
        });
    return ld;
}

// ---- z -> x in place (x: in z at tile positions in latent order, out x); returns this lane's share of logdet_inverse
template <int TD, int TH, int NHL>
__device__ __forceinline__ float flow_inverse_sweep_c(f32x4 (&x)[TD], const NfmcRealNVP& f, WeightPipe& wp, int col,
                                                      int half) {
    constexpr int d = 16 * TD, hp = 16 * TH;
    const bool rev_last = (f.n_coupling & 1) != 0;
    const float log1m = __logf(1.f - f.min_scale);
    float ldp = 0.f;
#pragma unroll
    for (int m = 0; m < TD; ++m) {  // EA1^-1
        const f32x4 ls = rev_last ? vec_tile_rev(f.ea1_log_scale, m, half, d) : vec_tile(f.ea1_log_scale, m, half);
        const f32x4 sh = rev_last ? vec_tile_rev(f.ea1_shift, m, half, d) : vec_tile(f.ea1_shift, m, half);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            x[m][t] = (x[m][t] - sh[t]) * fast_exp(-ls[t]);
            ldp -= ls[t];
        }
    }
    for (int l = f.n_coupling - 1; l >= 0; --l) {
        const MLayer L = mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL);
        if ((l & 1) == 0) ldp += coupling_c<TD, TH, NHL, true, true>(x, L, f.min_scale, log1m, wp, col, half);
        else ldp += coupling_c<TD, TH, NHL, false, true>(x, L, f.min_scale, log1m, wp, col, half);
    }
#pragma unroll
    for (int m = 0; m < TD; ++m) {  // EA0^-1
        const f32x4 ls = vec_tile(f.ea0_log_scale, m, half), sh = vec_tile(f.ea0_shift, m, half);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            x[m][t] = (x[m][t] - sh[t]) * fast_exp(-ls[t]);
            ldp -= ls[t];
        }
    }
    return ldp;
}

// ---- x -> z in place (z left at tile positions in latent order); returns this lane's share of logdet_forward
template <int TD, int TH, int NHL>
__device__ __forceinline__ float flow_forward_sweep_c(f32x4 (&x)[TD], const NfmcRealNVP& f, WeightPipe& wp, int col,
                                                      int half) {
    constexpr int d = 16 * TD, hp = 16 * TH;
    const bool rev_last = (f.n_coupling & 1) != 0;
    const float log1m = __logf(1.f - f.min_scale);
    float ldp = 0.f;
#pragma unroll
    for (int m = 0; m < TD; ++m) {  // EA0
        const f32x4 ls = vec_tile(f.ea0_log_scale, m, half), sh = vec_tile(f.ea0_shift, m, half);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            x[m][t] = fmaf(fast_exp(ls[t]), x[m][t], sh[t]);
            ldp += ls[t];
        }
    }
    for (int l = 0; l < f.n_coupling; ++l) {
        const MLayer L = mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL);
        if ((l & 1) == 0) ldp += coupling_c<TD, TH, NHL, true, false>(x, L, f.min_scale, log1m, wp, col, half);
        else ldp += coupling_c<TD, TH, NHL, false, false>(x, L, f.min_scale, log1m, wp, col, half);
    }
#pragma unroll
    for (int m = 0; m < TD; ++m) {  // EA1 (acts on logical latent coordinates)
        const f32x4 ls = rev_last ? vec_tile_rev(f.ea1_log_scale, m, half, d) : vec_tile(f.ea1_log_scale, m, half);
        const f32x4 sh = rev_last ? vec_tile_rev(f.ea1_shift, m, half, d) : vec_tile(f.ea1_shift, m, half);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            x[m][t] = fmaf(fast_exp(ls[t]), x[m][t], sh[t]);
            ldp += ls[t];
        }
    }
    return ldp;
}

// ---- closed-form potential value in C layout (tile position = coordinate of x)
template <int TD>
__device__ __forceinline__ float potential_value_c(const f32x4 (&x)[TD], const NfmcPotential& p, int half, int lane) {
    constexpr int d = 16 * TD;
    if (p.kind == NFMC_POT_FUNNEL) {
        const float x0 = __shfl(x[0][0], lane & 15, kWave);  // coordinate 0 = tile 0, reg 0, lane group 0
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < TD; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) s = fmaf(x[m][t], (m == 0 && t == 0 && half == 0) ? 0.f : x[m][t], s);
        s = chain_sum(s);
        const float inv_s2 = 1.f / (p.a_scalar * p.a_scalar);
        return 0.5f * x0 * x0 * inv_s2 + 0.5f * fast_exp(-x0) * s + 0.5f * (float)(d - 1) * x0;
    }
    float u = 0.f;
#pragma unroll
    for (int m = 0; m < TD; ++m) {
        f32x4 a, b;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = p.a_scalar;
            b[t] = p.b_scalar;
        }
        if (p.a) a = vec_tile(p.a, m, half);
        if (p.b) b = vec_tile(p.b, m, half);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float dlt = x[m][t] - b[t];
            u = fmaf(a[t] * dlt, dlt, u);
        }
    }
    return chain_sum(u);
}

}  // namespace nfmc
