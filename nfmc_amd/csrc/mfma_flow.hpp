// RealNVP forward / inverse sweeps on the matrix cores for the wave's 16 chains (layout and GEMM scheme:
// mfma_device.hpp).  Shared by the flow kernels (flow_mfma.hip: bijection.forward / inverse, Flow.log_prob / sample,
// the flow-proposal Metropolis step) and by NeuTra (neutra_mfma.hip), whose gradient starts with the inverse sweep.
#pragma once

#include "mfma_device.hpp"

namespace nfmc {

// ---- NeuTra's activation checkpoints (neutra_mfma.hip): what the reverse sweep needs of every coupling layer -- the
// hidden activations, alpha and beta -- is written by the inverse sweep and read back by the reverse sweep instead of
// being recomputed (3 of the 7 GEMMs of a layer's reverse sweep and the rebuild of tanh'(pre1): 37.5 % of the
// multiply-adds of a gradient).  One area per resident wave (workgroup slot x wave, NOT per chain tile: a workgroup's
// next chain tile reuses it, so the working set stays at grid x 8 waves x n_coupling x kLayerFloats x 4 B and mostly in
// the memory-side cache); a tile is 64 lanes x 16 B, lane-contiguous: every store / load instruction moves one full KB.
template <int TD, int TH, int NHL>
struct CkLayout {
    static constexpr int kHl = 0, kH1 = TH, kAlpha = (NHL > 1 ? 2 : 1) * TH, kBeta = kAlpha + TD / 2;
    static constexpr int kTiles = kBeta + TD / 2;
    static constexpr int kLayerFloats = kTiles * 256;
};
// `ck` already carries this lane's offset (4 * lane floats)
__device__ __forceinline__ f32x4* ck_tile(float* ck, int tile) { return reinterpret_cast<f32x4*>(ck + tile * 256); }
// The layer the inverse sweep finishes with is the one the reverse sweep starts with: its last-hidden-layer activations,
// alpha and beta stay in REGISTERS across the potential (nothing else is live there but x and g) instead of going
// through the checkpoint area -- a third of the checkpoint traffic of a two-layer flow and one exposed load latency.
template <int TD, int TH>
struct CkKeep {
    f32x4 hl[TH], al[TD / 2], be[TD / 2];
};

// ---- the two ElementwiseAffine layers and the mass diagonal as LDS planes in TILE-POSITION order (plane stride 128
// floats): log_scale, exp(log_scale), exp(-log_scale), shift of EA0 and of EA1, then inv_mass_diag (or ones).  NeuTra's
// trajectory kernel fills them once per workgroup; every gradient then reads 16-byte LDS tiles instead of global memory
// (measured on the kernel's timeline, tools/trace_c4.py: the global reads cost ~4 us per gradient, all of it exposed).
struct EaPlanes {
    static constexpr int kLs0 = 0, kE0 = 1, kEi0 = 2, kSh0 = 3, kLs1 = 4, kE1 = 5, kEi1 = 6, kSh1 = 7, kMass = 8, kPlanes = 9;
    static constexpr int kStride = 128;
    static constexpr int kFloats = kPlanes * kStride;
};
__device__ __forceinline__ void ea_planes_fill(float* eac, const NfmcRealNVP& f, const float* inv_mass_diag, int d) {
    const bool rev_last = (f.n_coupling & 1) != 0;
    for (int pos = threadIdx.x; pos < d; pos += blockDim.x) {
        const int c1 = rev_last ? d - 1 - pos : pos;   // EA1 and the mass act on logical latent coordinates
        const float l0 = f.ea0_log_scale[pos], l1 = f.ea1_log_scale[c1];
        eac[EaPlanes::kLs0 * EaPlanes::kStride + pos] = l0;
        eac[EaPlanes::kE0 * EaPlanes::kStride + pos] = fast_exp(l0);
        eac[EaPlanes::kEi0 * EaPlanes::kStride + pos] = fast_exp(-l0);
        eac[EaPlanes::kSh0 * EaPlanes::kStride + pos] = f.ea0_shift[pos];
        eac[EaPlanes::kLs1 * EaPlanes::kStride + pos] = l1;
        eac[EaPlanes::kE1 * EaPlanes::kStride + pos] = fast_exp(l1);
        eac[EaPlanes::kEi1 * EaPlanes::kStride + pos] = fast_exp(-l1);
        eac[EaPlanes::kSh1 * EaPlanes::kStride + pos] = f.ea1_shift[c1];
        eac[EaPlanes::kMass * EaPlanes::kStride + pos] = inv_mass_diag ? inv_mass_diag[c1] : 1.f;
    }
}
// one tile of (log_scale, exp(sign * log_scale), shift): from the LDS planes, or from global memory (rev: the arrays are
// in logical order and tile position p holds logical d-1-p)
template <bool LDS>
__device__ __forceinline__ void ea_tile(f32x4& ls, f32x4& e, f32x4& sh, const float* eac, int p_ls, int p_e, int p_sh,
                                        const float* g_ls, const float* g_sh, int m, int half, int d, bool rev, float sign) {
    if constexpr (LDS) {
        ls = vec_tile(eac + p_ls * EaPlanes::kStride, m, half);
        e = vec_tile(eac + p_e * EaPlanes::kStride, m, half);
        sh = vec_tile(eac + p_sh * EaPlanes::kStride, m, half);
    } else {
        ls = rev ? vec_tile_rev(g_ls, m, half, d) : vec_tile(g_ls, m, half);
        sh = rev ? vec_tile_rev(g_sh, m, half, d) : vec_tile(g_sh, m, half);
#pragma unroll
        for (int t = 0; t < 4; ++t) e[t] = fast_exp(sign * ls[t]);
    }
}

// ---- one coupling layer in C layout.  INVERSE: target half v_b = (y_b - beta) / alpha, else z_b = alpha x_b + beta.
// Returns this lane's share of the layer's logdet in THAT direction.  Three steps of the weight pipeline.
// CK: keep what NeuTra's reverse sweep needs of this layer (hidden activations, alpha, beta) in the wave's checkpoint
// area `ck` (layout: ck_tile) instead of having the reverse sweep recompute it.
template <int TD, int TH, int NHL, bool REV, bool INVERSE, bool CK = false, bool KEEP = false>
__device__ __forceinline__ float coupling_c(f32x4 (&x)[TD], const MLayer& L, float mscale, float log1m,
                                            WeightPipe& wp, int col, int half, float* ck = nullptr,
                                            CkKeep<TD, TH>* keep = nullptr) {
    constexpr int TS = TD / 2, SRC0 = REV ? TS : 0, TGT0 = REV ? 0 : TS, D2 = 8 * TD, hp = 16 * TH;
    f32x4 hl[TH];   // activations of the last hidden layer
    {
        f32x4 src[TS], h1[TH];
#pragma unroll
        for (int ms = 0; ms < TS; ++ms) src[ms] = x[SRC0 + ms];
        if constexpr (NHL > 1) hidden_stack<TS, TH, NHL>(src, h1, hl, L, REV, wp, col, half);
        else hidden_stack<TS, TH, NHL>(src, hl, h1, L, REV, wp, col, half);
        if constexpr (CK) {
#pragma unroll
            for (int m = 0; m < TH; ++m) {
                if constexpr (KEEP) keep->hl[m] = hl[m];
                else *ck_tile(ck, CkLayout<TD, TH, NHL>::kHl + m) = hl[m];
                if constexpr (NHL > 1) *ck_tile(ck, CkLayout<TD, TH, NHL>::kH1 + m) = h1[m];
            }
        }
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
            f32x4 al, be;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float alpha = fast_exp(fmaf(0.5f, ua[t], log1m)) + mscale;
                const float la = fast_ln(alpha);
                al[t] = alpha;
                be[t] = 0.5f * ub[t];
                if constexpr (INVERSE) {
                    x[TGT0 + mt][t] = (x[TGT0 + mt][t] - 0.5f * ub[t]) * __builtin_amdgcn_rcpf(alpha);
                    ld -= la;
                } else {
                    x[TGT0 + mt][t] = fmaf(alpha, x[TGT0 + mt][t], 0.5f * ub[t]);
                    ld += la;
                }
            }
            if constexpr (CK && KEEP) {
                keep->al[mt] = al;
                keep->be[mt] = be;
            } else if constexpr (CK) {
                *ck_tile(ck, CkLayout<TD, TH, NHL>::kAlpha + mt) = al;
                *ck_tile(ck, CkLayout<TD, TH, NHL>::kBeta + mt) = be;
            }
        });
    return ld;
}

// ---- z -> x in place (x: in z at tile positions in latent order, out x); returns this lane's share of logdet_inverse
// `eac`: the elementwise-affine constants in LDS (EaPlanes, filled once per kernel by ea_planes_fill) or null (global reads)
template <int TD, int TH, int NHL, bool CK = false, bool EAC = false>
__device__ __forceinline__ float flow_inverse_sweep_c(f32x4 (&x)[TD], const NfmcRealNVP& f, WeightPipe& wp, int col,
                                                      int half, float* ck = nullptr, const float* eac = nullptr,
                                                      CkKeep<TD, TH>* keep = nullptr) {
    constexpr int d = 16 * TD, hp = 16 * TH;
    const bool rev_last = (f.n_coupling & 1) != 0;
    const float log1m = __logf(1.f - f.min_scale);
    float ldp = 0.f;
#pragma unroll
    for (int m = 0; m < TD; ++m) {  // EA1^-1
        f32x4 ls, sh, ei;
        ea_tile<EAC>(ls, ei, sh, eac, EaPlanes::kLs1, EaPlanes::kEi1, EaPlanes::kSh1, f.ea1_log_scale, f.ea1_shift, m, half, d, rev_last, -1.f);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            x[m][t] = (x[m][t] - sh[t]) * ei[t];
            ldp -= ls[t];
        }
    }
    for (int l = f.n_coupling - 1; l >= (CK ? 1 : 0); --l) {
        const MLayer L = mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL);
        float* ckl = CK ? ck + (size_t)l * CkLayout<TD, TH, NHL>::kLayerFloats : nullptr;
        if ((l & 1) == 0) ldp += coupling_c<TD, TH, NHL, true, true, CK>(x, L, f.min_scale, log1m, wp, col, half, ckl);
        else ldp += coupling_c<TD, TH, NHL, false, true, CK>(x, L, f.min_scale, log1m, wp, col, half, ckl);
    }
    if constexpr (CK) {   // layer 0 (even: reversed), the reverse sweep's first: kept in registers (CkKeep)
        if (f.n_coupling > 0) {
            const MLayer L = mfma_layer(f.weights, d, hp, NHL);
            ldp += coupling_c<TD, TH, NHL, true, true, true, true>(x, L, f.min_scale, log1m, wp, col, half, ck, keep);
        }
    }
#pragma unroll
    for (int m = 0; m < TD; ++m) {  // EA0^-1
        f32x4 ls, sh, ei;
        ea_tile<EAC>(ls, ei, sh, eac, EaPlanes::kLs0, EaPlanes::kEi0, EaPlanes::kSh0, f.ea0_log_scale, f.ea0_shift, m, half, d, false, -1.f);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            x[m][t] = (x[m][t] - sh[t]) * ei[t];
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

// ---- closed-form potential and gradient in C layout (tile position = coordinate of x)
template <int TD>
__device__ __forceinline__ float potential_value_grad_c(const f32x4 (&x)[TD], f32x4 (&g)[TD], const NfmcPotential& p,
                                                        int half, int lane) {
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
        const float e = fast_exp(-x0);
        const float hd = 0.5f * (float)(d - 1);
#pragma unroll
        for (int m = 0; m < TD; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) g[m][t] = x[m][t] * e;
        if (half == 0) g[0][0] = x0 * inv_s2 - 0.5f * e * s + hd;
        return 0.5f * x0 * x0 * inv_s2 + 0.5f * e * s + hd * x0;
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
            g[m][t] = 2.f * a[t] * dlt;
        }
    }
    return chain_sum(u);
}

}  // namespace nfmc
