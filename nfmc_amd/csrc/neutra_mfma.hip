// K5 on the matrix cores: adjusted potential U~(z) = U(f^-1(z)) - logdet_inv(z), its gradient, and the
// NeuTra-HMC leapfrog around it, for RealNVP conditioners of width 33..128 (BASELINE config 4: d = 128,
// n_hidden = 128).  Layout and GEMM scheme: mfma_device.hpp.  Replaces neutra.py:58-68 + hmc.py:40-77,96-126.
//
// 16 chains per wave (v_mfma_f32_16x16x4_f32), 8 waves = 128 chains per workgroup, two waves per SIMD.
// One launch of `neutra_leapfrog_mfma_kernel` = one whole trajectory of every chain: momentum draw and H0, the
// n_leapfrog steps with position and gradient tiles resident in registers from one step to the next, then the
// Hamiltonian test, the masked update and the statistics.  Only the momentum tile goes through caller-supplied
// HBM scratch (stored before each gradient, reloaded after it: it must not be live through the GEMMs):
// 2*4*d bytes per chain per leapfrog against ~0.5 MFLOP of conditioner GEMMs.
#include "mfma_flow.hpp"

namespace nfmc {

// ---- reverse sweep through one inverse coupling layer: (y, dL/dy) -> (v, dL/dv); L = U(x) + sum log alpha.
// Seven steps of the weight pipeline (five with one hidden layer), one image each:
//   W1 -> h1 | Wh -> h2 | W3 -> u_alpha, u_beta of every target tile (then the elementwise backward: du, dv)
//   | W3^T -> dL/dh_last | Wh^T -> dL/dh1 | W1 again -> tanh'(pre1) tile by tile | W1^T -> dL/dy_source.
// Register budget (two waves per SIMD, 256 VGPRs): x, g and at most THREE hidden-width tile sets are live at
// any point (h_last, the du/dv tiles, dL/dh); with two hidden layers the first layer's activations are not kept
// across the transposed products but rebuilt where tanh' is needed (+7 % multiply-adds instead of 32 registers).
// alpha and beta of one layer, which the reverse sweep needs first (requested by adjusted_grad_c, possibly ahead of time)
template <int TD>
struct CkAffine {
    f32x4 al[TD / 2], be[TD / 2];
};
template <int TD, int TH, int NHL>
__device__ __forceinline__ void ck_request(CkAffine<TD>& c, float* ck) {
    using CL = CkLayout<TD, TH, NHL>;
#pragma unroll
    for (int mt = 0; mt < TD / 2; ++mt) {
        c.al[mt] = *ck_tile(ck, CL::kAlpha + mt);
        c.be[mt] = *ck_tile(ck, CL::kBeta + mt);
    }
}

template <int TD, int TH, int NHL, bool REV, bool CK, bool FROMREG = false>
__device__ __forceinline__ void coupling_inverse_backward_c(f32x4 (&x)[TD], f32x4 (&g)[TD], const MLayer& L,
                                                            float mscale, float log1m, WeightPipe& wp, int col,
                                                            int half, float* ck, const CkAffine<TD>& kept,
                                                            const f32x4 (*hl_kept)[TH] = nullptr) {
    constexpr int TS = TD / 2, SRC0 = REV ? TS : 0, TGT0 = REV ? 0 : TS, D2 = 8 * TD, hp = 16 * TH;
    using CL = CkLayout<TD, TH, NHL>;
    f32x4 hl[TH];   // activations of the LAST hidden layer
    f32x4 du[TS], dv[TS];
    // elementwise backward of the affine map of one target tile, given alpha and beta
    auto affine_backward = [&](int mt, const f32x4& al, const f32x4& be) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float alpha = al[t];
            const float ra = __builtin_amdgcn_rcpf(alpha);
            const float y = x[TGT0 + mt][t];
            const float gv = g[TGT0 + mt][t] * ra;
            const float d_alpha = fmaf(-gv, y, ra);
            du[mt][t] = 0.5f * d_alpha * (alpha - mscale);
            dv[mt][t] = -0.5f * gv;
            g[TGT0 + mt][t] = gv;
            x[TGT0 + mt][t] = fmaf(alpha, y, be[t]);
        }
    };
    if constexpr (CK) {
        // everything the inverse sweep kept of this layer: no hidden stack, no W3 product
#pragma unroll
        for (int mt = 0; mt < TS; ++mt) affine_backward(mt, kept.al[mt], kept.be[mt]);
        // the last hidden layer's activations are needed in the epilogues of the next GEMM only: requested here, they
        // arrive under its weight copy
#pragma unroll
        for (int m = 0; m < TH; ++m) {
            if constexpr (FROMREG) hl[m] = (*hl_kept)[m];
            else hl[m] = *ck_tile(ck, CL::kHl + m);
        }
        wp.mark(15);
    } else {
        {
            f32x4 src[TS], h1[TH];
#pragma unroll
            for (int ms = 0; ms < TS; ++ms) src[ms] = x[SRC0 + ms];
            if constexpr (NHL > 1) hidden_stack<TS, TH, NHL>(src, h1, hl, L, REV, wp, col, half);
            else hidden_stack<TS, TH, NHL>(src, hl, h1, L, REV, wp, col, half);
        }
        // conditioner outputs of every target tile, then the elementwise backward of the affine map
        wp.template stage<hp, D2, 1, D2, 2 * D2>(L.W3, REV, false, L.b3, 2 * D2, REV);
        const float* img = wp.img();
        const float* vec = wp.vec();
        f32x4 ua2[2], ub2[2];   // per parity of the target tile: the next tile's bias is read while this one is in flight
        gemm_phase<TH, 2 * TS>(
            [&](int i) { return img + (16 * ((i & 1) * TS + (i >> 1)) + col) * (hp + 4) + 4 * half; },
            [&](int i) { ((i & 1) ? ub2 : ua2)[(i >> 1) & 1] = vec_tile(vec, (i & 1) * TS + (i >> 1), half); },
            [&](int i) -> f32x4& { return ((i & 1) ? ub2 : ua2)[(i >> 1) & 1]; },
            [&](int) -> const f32x4(&)[TH] { return hl; },
            [&](int i) {
                if ((i & 1) == 0) return;
                const int mt = i >> 1;
                f32x4 al, be;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    al[t] = fast_exp(fmaf(0.5f, ua2[mt & 1][t], log1m)) + mscale;
                    be[t] = 0.5f * ub2[mt & 1][t];
                }
                affine_backward(mt, al, be);
            });
    }
    // dL/dh_last = W3^T [du; dv], through tanh of the last hidden layer (its activations die here)
    f32x4 dh[TH];
    {
        wp.template stage<2 * D2, 1, D2, 1, hp>(L.W3T, false, REV, nullptr, 0, false);
        const float* img = wp.img();
        // step i: hidden tile mo = 2 (i / 4) + (i & 1), operand half (i >> 1) & 1 (du, then dv): pairs are the same half of
        // two DIFFERENT hidden tiles (separate accumulators); the dv half continues the sum the du half began
        gemm_phase<TS, 2 * TH>(
            [&](int i) { return img + (16 * (2 * (i >> 2) + (i & 1)) + col) * (2 * D2 + 4) + 4 * half + ((i >> 1) & 1) * D2; },
            [&](int i) {
                if (((i >> 1) & 1) == 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) dh[2 * (i >> 2) + (i & 1)][t] = 0.f;
                }
            },
            [&](int i) -> f32x4& { return dh[2 * (i >> 2) + (i & 1)]; },
            [&](int i) -> const f32x4(&)[TS] { return ((i >> 1) & 1) ? dv : du; },
            [&](int i) {
                if (((i >> 1) & 1) == 0) return;
                const int mo = 2 * (i >> 2) + (i & 1);
#pragma unroll
                for (int t = 0; t < 4; ++t) dh[mo][t] *= (1.f - hl[mo][t] * hl[mo][t]);
            });
    }
    if constexpr (NHL > 1) {
        f32x4 h1k[CK ? TH : 1];   // checkpointed first-layer activations: requested before the GEMM that hides the latency
        if constexpr (CK) {
#pragma unroll
            for (int mo = 0; mo < TH; ++mo) h1k[mo] = *ck_tile(ck, CL::kH1 + mo);
        }
        // dL/dh1 = Wh^T dpre2 ...
        {
            wp.template stage<hp, 1, 1, 1, hp>(L.WhT, false, false, nullptr, 0, false);
            const float* img = wp.img();
            gemm_phase<TH, TH>([&](int mo) { return img + (16 * mo + col) * (hp + 4) + 4 * half; },
                               [&](int mo) {
#pragma unroll
                                   for (int t = 0; t < 4; ++t) hl[mo][t] = 0.f;
                               },
                               [&](int mo) -> f32x4& { return hl[mo]; },
                               [&](int) -> const f32x4(&)[TH] { return dh; },
                               [&](int) {});
        }
        // ... times tanh'(pre1): h1 from the checkpoint, or rebuilt (h1 = tanh(W1 src + b1)) one tile at a time
        if constexpr (CK) {
#pragma unroll
            for (int mo = 0; mo < TH; ++mo) {
#pragma unroll
                for (int t = 0; t < 4; ++t) dh[mo][t] = hl[mo][t] * (1.f - h1k[mo][t] * h1k[mo][t]);
            }
        } else {
            wp.template stage<D2, 1, D2, 1, hp>(L.W1, false, REV, L.b1, hp, false);
            const float* img = wp.img();
            const float* vec = wp.vec();
            f32x4 src[TS];
#pragma unroll
            for (int ms = 0; ms < TS; ++ms) src[ms] = x[SRC0 + ms];
            f32x4 h1t[2];
            gemm_phase<TS, TH>([&](int mo) { return img + (16 * mo + col) * (D2 + 4) + 4 * half; },
                               [&](int mo) { h1t[mo & 1] = vec_tile(vec, mo, half); },
                               [&](int mo) -> f32x4& { return h1t[mo & 1]; },
                               [&](int) -> const f32x4(&)[TS] { return src; },
                               [&](int mo) {
                                   const f32x4 h = tanh4(h1t[mo & 1]);
#pragma unroll
                                   for (int t = 0; t < 4; ++t) dh[mo][t] = hl[mo][t] * (1.f - h[t] * h[t]);
                               });
        }
    }
    wp.template stage<hp, D2, 1, 1, D2>(L.W1T, REV, false, nullptr, 0, false);
    const float* img = wp.img();
    gemm_phase<TH, TS>([&](int ms) { return img + (16 * ms + col) * (hp + 4) + 4 * half; }, [&](int) {},
                       [&](int ms) -> f32x4& { return g[SRC0 + ms]; },
                       [&](int) -> const f32x4(&)[TH] { return dh; }, [&](int) {});
}

// ---- U~(z), grad U~(z) for the wave's 16 chains.  x: in z (tile positions in latent order), out z again
// (rebuilt through the inverse of every step); g: gradient in the same positions.  Workgroup-collective.
// CK: the inverse sweep leaves every layer's activations in the wave's checkpoint area `ck` (mfma_flow.hpp: CkLayout)
// and the reverse sweep reads them back; without, the reverse sweep recomputes them (no scratch needed).
template <int TD, int TH, int NHL, bool CK = false, bool EAC = false>
__device__ __forceinline__ float adjusted_grad_c(f32x4 (&x)[TD], f32x4 (&g)[TD], const NfmcRealNVP& f,
                                                 const NfmcPotential& pot, WeightPipe& wp, int col, int half, int lane,
                                                 float* ck = nullptr, const float* eac = nullptr) {
    constexpr int d = 16 * TD, hp = 16 * TH;
    using CL = CkLayout<TD, TH, NHL>;
    const bool rev_last = (f.n_coupling & 1) != 0;
    const float log1m = __logf(1.f - f.min_scale);
    wp.mark(10);
    CkKeep<TD, TH> keep;   // layer 0's activations, alpha, beta: in registers from the inverse sweep to the reverse sweep
    float ldp = flow_inverse_sweep_c<TD, TH, NHL, CK, EAC>(x, f, wp, col, half, ck, eac, &keep);
    wp.mark(11);
    CkAffine<TD> kept;
    const float u = potential_value_grad_c<TD>(x, g, pot, half, lane);
    // reverse sweep
#pragma unroll
    for (int m = 0; m < TD; ++m) {
        f32x4 ls, sh, e, ei;
        ea_tile<EAC>(ls, e, sh, eac, EaPlanes::kLs0, EaPlanes::kE0, EaPlanes::kSh0, f.ea0_log_scale, f.ea0_shift, m, half, d, false, 1.f);
        if constexpr (EAC) ei = vec_tile(eac + EaPlanes::kEi0 * EaPlanes::kStride, m, half);
        else
#pragma unroll
            for (int t = 0; t < 4; ++t) ei[t] = fast_exp(-ls[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            g[m][t] *= ei[t];
            x[m][t] = fmaf(e[t], x[m][t], sh[t]);
        }
    }
    if constexpr (CK) {   // layer 0 first, from the registers the inverse sweep left (CkKeep); they die here
        if (f.n_coupling > 0) {
            const MLayer L = mfma_layer(f.weights, d, hp, NHL);
#pragma unroll
            for (int mt = 0; mt < TD / 2; ++mt) {
                kept.al[mt] = keep.al[mt];
                kept.be[mt] = keep.be[mt];
            }
            coupling_inverse_backward_c<TD, TH, NHL, true, true, true>(x, g, L, f.min_scale, log1m, wp, col, half, ck, kept, &keep.hl);
        }
    }
    for (int l = CK ? 1 : 0; l < f.n_coupling; ++l) {
        const MLayer L = mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL);
        float* ckl = CK ? ck + (size_t)l * CL::kLayerFloats : nullptr;
        // Requested HERE, not ahead of time: every attempt to hide this latency (the next layer's tiles before this layer's
        // last GEMM, alpha / beta alone, the momentum tiles before the last GEMM) cost more in spills than it hid
        if constexpr (CK) ck_request<TD, TH, NHL>(kept, ckl);
        if ((l & 1) == 0) coupling_inverse_backward_c<TD, TH, NHL, true, CK>(x, g, L, f.min_scale, log1m, wp, col, half, ckl, kept);
        else coupling_inverse_backward_c<TD, TH, NHL, false, CK>(x, g, L, f.min_scale, log1m, wp, col, half, ckl, kept);
    }
#pragma unroll
    for (int m = 0; m < TD; ++m) {
        f32x4 ls, sh, e, ei;
        ea_tile<EAC>(ls, e, sh, eac, EaPlanes::kLs1, EaPlanes::kE1, EaPlanes::kSh1, f.ea1_log_scale, f.ea1_shift, m, half, d, rev_last, 1.f);
        if constexpr (EAC) ei = vec_tile(eac + EaPlanes::kEi1 * EaPlanes::kStride, m, half);
        else
#pragma unroll
            for (int t = 0; t < 4; ++t) ei[t] = fast_exp(-ls[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            g[m][t] *= ei[t];
            x[m][t] = fmaf(e[t], x[m][t], sh[t]);
        }
    }
    wp.mark(12);
    return u - chain_sum(ldp);
}

// ------------------------------------------------------------------------------------------------
template <int TD, int TH, int NHL>
__global__ void __launch_bounds__(kMfmaBlock, 2) neutra_grad_mfma_kernel(NfmcRealNVP f, NfmcPotential pot,
                                                                      const float* __restrict__ z, int64_t n,
                                                                      float* __restrict__ u_out,
                                                                      float* __restrict__ grad_out, int64_t tiles) {
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
        f32x4 x[TD], g[TD];
        load_ctiles<TD>(x, z, rrow, d, half, rev);
        const float u = adjusted_grad_c<TD, TH, NHL>(x, g, f, pot, wp, col, half, lane);
        if (active) {
            if (u_out && half == 0) u_out[row] = u;
            if (grad_out) store_ctiles<TD>(g, grad_out, row, d, half, rev);
        }
    }
}

struct LeapArgs {
    NfmcNeutraHmcArgs a;
    float *p, *gz, *uz, *h0;  // scratch: momentum; gradient and U~ at the current state; H0
    float* ck;                 // scratch: activation checkpoints, one area per (workgroup slot, wave) -- mfma_flow.hpp: CkLayout
    int step;                  // transition index within this call
    float* sample_row;         // store row this transition is kept in (NfmcSampleStore walked on the host), or NULL
};

// LDS of the trajectory kernel: the matrix-core carve-up (mfma_device.hpp), then the EaPlanes constants
constexpr size_t kLeapLdsBytes = kMfmaLdsBytes + EaPlanes::kFloats * sizeof(float);

template <int TD, int TH, int NHL>
__global__ void __launch_bounds__(kMfmaBlock, 2) neutra_leapfrog_mfma_kernel(LeapArgs A, int64_t tiles, int dp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int d = 16 * TD;
    const NfmcNeutraHmcArgs& a = A.a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, half = lane >> 4;
    const bool rev = (a.flow.n_coupling & 1) != 0;
    const int64_t n = a.n;
    const float h = a.step_size, hh = a.step_size / 2;
    const int s = A.step;

    // statistics accumulate in LDS behind the weight images (nothing extra stays live across the GEMMs)
    double* red = reinterpret_cast<double*>(lds + kMfmaStatOffset);  // [8 waves][2*d + 2]
    float* const eac = lds + kMfmaLdsBytes / sizeof(float);          // EaPlanes: elementwise-affine constants, mass diagonal
    const float* const massp = eac + EaPlanes::kMass * EaPlanes::kStride;
    WeightPipe wp{lds, 0};
    uint32_t n_acc = 0, n_bad = 0;
    float* const ck = A.ck + ((size_t)blockIdx.x * kMfmaWaves + __builtin_amdgcn_readfirstlane(wave)) * a.flow.n_coupling *
                                 CkLayout<TD, TH, NHL>::kLayerFloats + 4 * lane;
    for (int i = threadIdx.x; i < kMfmaWaves * (2 * d + 2); i += kMfmaBlock) red[i] = 0.0;
    ea_planes_fill(eac, a.flow, a.inv_mass_diag, d);
    __syncthreads();
    const int L = a.n_leapfrog;
#ifdef NFMC_TRACE
    {
        unsigned long long* tbase = reinterpret_cast<unsigned long long*>(
            A.ck + (size_t)gridDim.x * kMfmaWaves * a.flow.n_coupling * CkLayout<TD, TH, NHL>::kLayerFloats);
        if (blockIdx.x == 0) wp.tr = tbase + wave * 4096;
#ifdef NFMC_TRACE_STEPS
        if (threadIdx.x < kMfmaWaves) {
            g_step_buf[threadIdx.x] = blockIdx.x == 0 ? tbase + kMfmaWaves * 4096 + threadIdx.x * 8192 : nullptr;
            g_step_cnt[threadIdx.x] = 0;
        }
        if (threadIdx.x == 0) g_step_magic = 0x5A17C0DEu;
        __syncthreads();
#endif
    }
#endif

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row_t = tile * kMfmaChains + wave * 16 + col;
        const bool active = row_t < n;
        const int64_t rrow_t = active ? row_t : n - 1;
        f32x4 x[TD], g[TD];
        float u = 0.f;
        {   // ---- start of the trajectory: momentum draw, H0, the first half step (hmc.py:100-106, 68-70)
            int64_t row_in = row_t, rrow_in = rrow_t;
            asm volatile("" : "+v"(row_in), "+v"(rrow_in));
            f32x4 p[TD];
            load_ctiles<TD>(x, a.z, rrow_in, d, half, rev);
            load_ctiles<TD>(g, A.gz, rrow_in, d, half, rev);
            float kin = 0.f;
            const uint32_t gchain = (uint32_t)(a.rng.chain_offset + (uint64_t)rrow_in);
#pragma unroll
            for (int m = 0; m < TD; ++m) {
                const int p0 = 16 * m + 4 * half;  // tile position of this lane's 4-block (= one Philox block)
                float zz[4];
                if (a.rng.replay_normals) {
                    const float* src = a.rng.replay_normals + ((int64_t)s * n + rrow_in) * d;
#pragma unroll
                    for (int j = 0; j < 4; ++j) zz[j] = src[rev ? d - 1 - (p0 + j) : p0 + j];
                } else {
                    const int blk = rev ? (d - 4 - p0) >> 2 : p0 >> 2;
                    float w[4];
                    philox_normal4(gchain, a.rng.step0 + (uint32_t)s, (uint32_t)blk, kTagNoise, (uint32_t)a.rng.seed,
                                   (uint32_t)(a.rng.seed >> 32), w);
#pragma unroll
                    for (int j = 0; j < 4; ++j) zz[j] = rev ? w[3 - j] : w[j];
                }
                const f32x4 mass = vec_tile(massp, m, half);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float mm = mass[j];
                    const float v = zz[j] * (1.f / sqrtf(mm));   // hmc.py:100
                    p[m][j] = v;
                    kin = fmaf(v * v, mm, kin);
                }
            }
            kin = chain_sum(kin);
            if (active && half == 0) A.h0[row_in] = A.uz[row_in] + 0.5f * kin;  // hmc.py:103-106
#pragma unroll
            for (int m = 0; m < TD; ++m) {
                const f32x4 mass = vec_tile(massp, m, half);
#pragma unroll
                for (int t = 0; t < 4; ++t) {  // hmc.py:68-70
                    p[m][t] = fmaf(-hh, g[m][t], p[m][t]);
                    x[m][t] = fmaf(h, p[m][t] * mass[t], x[m][t]);
                }
            }
            if (active) store_ctiles<TD>(p, A.p, row_in, d, half, rev);  // momentum is not live across the GEMMs
        }
        for (int l = 0; l < L; ++l) {
        f32x4 p[TD];
        u = adjusted_grad_c<TD, TH, NHL, true, true>(x, g, a.flow, a.pot, wp, col, half, lane, ck, eac);
        // everything below addresses HBM by the row index: an opaque copy keeps that address arithmetic from
        // being computed before the gradient and held in registers through it (cf. stage_matrix)
        int64_t row = row_t, rrow = rrow_t;
        asm volatile("" : "+v"(row), "+v"(rrow));
        load_ctiles<TD>(p, A.p, rrow, d, half, rev);
#pragma unroll
        for (int m = 0; m < TD; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) p[m][t] = fmaf(-hh, g[m][t], p[m][t]);  // hmc.py:71
        if (l + 1 < L) {
            // the next step's first half (hmc.py:68-70) on the momentum just loaded: ONE load and ONE store of the momentum
            // tile per leapfrog step and no store -> load round trip between the two half steps (same fmas, same order)
#pragma unroll
            for (int m = 0; m < TD; ++m) {
                const f32x4 mass = vec_tile(massp, m, half);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    p[m][t] = fmaf(-hh, g[m][t], p[m][t]);
                    x[m][t] = fmaf(h, p[m][t] * mass[t], x[m][t]);
                }
            }
            if (active) store_ctiles<TD>(p, A.p, row, d, half, rev);
            continue;
        }
        // ---- end of the trajectory: Hamiltonian test, masked update, statistics
        wp.mark(20);
        bool accept = true;
        float lr = 0.f;
        if (a.adjust) {
            float kin = 0.f;
#pragma unroll
            for (int m = 0; m < TD; ++m) {
                const f32x4 mass = vec_tile(massp, m, half);
#pragma unroll
                for (int t = 0; t < 4; ++t) kin = fmaf(p[m][t] * p[m][t], mass[t], kin);
            }
            kin = chain_sum(kin);
            lr = A.h0[rrow] - (u + 0.5f * kin);  // hmc.py:107-111
            float uni;
            if (a.rng.replay_uniforms) {
                uni = a.rng.replay_uniforms[(int64_t)s * n + rrow];
            } else {
                const uint32_t step = a.rng.step0 + (uint32_t)s;
                const uint4 r = philox4x32_10((uint32_t)(a.rng.chain_offset + (uint64_t)rrow), step >> 2, 0u,
                                              kTagAccept, (uint32_t)a.rng.seed, (uint32_t)(a.rng.seed >> 32));
                uni = u32_to_uniform(pick_word(r, step & 3u));
            }
            accept = fast_ln(uni) < lr;  // hmc.py:112-113
            if (active && half == 0 && !(fabsf(lr) <= 3.0e38f)) n_bad++;
        }
        accept = accept && active;
        f32x4 zc[TD];
        if (accept) {
#pragma unroll
            for (int m = 0; m < TD; ++m) zc[m] = x[m];
            store_ctiles<TD>(x, a.z, row, d, half, rev);
            store_ctiles<TD>(g, A.gz, row, d, half, rev);
            if (half == 0) {
                A.uz[row] = u;
                n_acc++;
            }
        } else {
            load_ctiles<TD>(zc, a.z, rrow, d, half, rev);
        }
        if (active) {
            if (half == 0) {
                if (a.masks_out) a.masks_out[(int64_t)s * n + row] = accept ? 1 : 0;
                if (a.log_ratio_out) a.log_ratio_out[(int64_t)s * n + row] = lr;
            }
            if (A.sample_row) store_ctiles<TD>(zc, A.sample_row, row, d, half, rev);
        }
        if (a.stats.sum_x) {  // sums over the 16 chains of the wave (lanes of one lane group), kept per wave in LDS
#pragma unroll
            for (int m = 0; m < TD; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float zv = active ? zc[m][t] : 0.f;
                    double v1 = (double)zv, v2 = (double)zv * (double)zv;
                    for (int k = 1; k < 16; k <<= 1) {
                        v1 += __shfl_xor(v1, k, kWave);
                        v2 += __shfl_xor(v2, k, kWave);
                    }
                    if (col == 0) {
                        const int pos = 16 * m + 4 * half + t;
                        const int c = rev ? d - 1 - pos : pos;  // logical latent coordinate
                        red[wave * (2 * d + 2) + c] += v1;
                        red[wave * (2 * d + 2) + d + c] += v2;
                    }
                }
        }
        }  // leapfrog steps
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
        double* out = a.stats.scratch + (size_t)blockIdx.x * (2 * dp + kStatTail);
        for (int i = threadIdx.x; i < 2 * dp + kStatTail; i += kMfmaBlock) {
            double v = 0.0;
            int srci = -1;
            if (i < dp) srci = i < d ? i : -1;
            else if (i < 2 * dp) srci = (i - dp) < d ? d + (i - dp) : -1;
            else if (i == 2 * dp) srci = 2 * d;
            else if (i == 2 * dp + 1) srci = 2 * d + 1;
            if (srci >= 0)
                for (int w = 0; w < kMfmaWaves; ++w) v += red[w * (2 * d + 2) + srci];
            out[i] = v;
        }
    }
}

// initial U~(z), grad at the state (also used after the caller changed z)
template <int TD, int TH, int NHL>
static int launch_grad(const NfmcRealNVP& f, const NfmcPotential& pot, const float* z, int64_t n, float* u, float* g,
                       hipStream_t st) {
    const int64_t tiles = (n + kMfmaChains - 1) / kMfmaChains;
    const int grid = (int)(tiles < kMaxGrid ? tiles : kMaxGrid);
    auto kern = neutra_grad_mfma_kernel<TD, TH, NHL>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfmaLdsBytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kMfmaBlock), kMfmaLdsBytes, st, f, pot, z, n, u, g, tiles);
    return 0;
}

template <int TD, int TH, int NHL>
static int run_hmc(const NfmcNeutraHmcArgs& a, float* scratch, hipStream_t st) {
    constexpr int d = 16 * TD;
    const int64_t n = a.n;
    LeapArgs A;
    A.a = a;
    A.p = scratch;
    A.gz = A.p + n * d;
    A.uz = A.gz + n * d;
    A.h0 = A.uz + n;
    A.ck = A.h0 + n;
    int rc = launch_grad<TD, TH, NHL>(a.flow, a.pot, a.z, n, A.uz, A.gz, st);
    if (rc) return rc;
    const int64_t tiles = (n + kMfmaChains - 1) / kMfmaChains;
    const int grid = (int)(tiles < kCkMaxGrid ? tiles : kCkMaxGrid);   // one checkpoint area per workgroup slot
    const int dp = padded_d(d);
    auto kern = neutra_leapfrog_mfma_kernel<TD, TH, NHL>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLeapLdsBytes);
    if (e != hipSuccess) return (int)e;
    int countdown = a.samples.countdown, srow = a.samples.row;   // one launch per transition: the store cursor runs here
    for (int s = 0; s < a.n_steps; ++s) {
        A.step = s;
        A.sample_row = nullptr;
        if (a.samples.base) {
            if (countdown > 0) {
                --countdown;
            } else {
                A.sample_row = a.samples.base + (int64_t)srow * n * d;
                srow = srow + 1 == a.samples.ring_rows ? 0 : srow + 1;
                countdown = a.samples.stride - 1;
            }
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kMfmaBlock), kLeapLdsBytes, st, A, tiles, dp);
        if (a.stats.sum_x) {
            hipLaunchKernelGGL(stats_finish_kernel<true>, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, st,
                               a.stats.scratch, grid, dp, d, a.stats, (unsigned long long)n);
        }
    }
    hipError_t le = hipGetLastError();
    return le == hipSuccess ? 0 : (int)le;
}

}  // namespace nfmc

using namespace nfmc;

#define NFMC_MFMA_DISPATCH(TDV, THV, NHLV, CALL)                                        \
    if (TDV == 4 && THV == 4 && NHLV == 1) { constexpr int TD = 4, TH = 4, NHL = 1; CALL; }      \
    else if (TDV == 4 && THV == 4 && NHLV == 2) { constexpr int TD = 4, TH = 4, NHL = 2; CALL; } \
    else if (TDV == 4 && THV == 8 && NHLV == 1) { constexpr int TD = 4, TH = 8, NHL = 1; CALL; } \
    else if (TDV == 4 && THV == 8 && NHLV == 2) { constexpr int TD = 4, TH = 8, NHL = 2; CALL; } \
    else if (TDV == 8 && THV == 4 && NHLV == 1) { constexpr int TD = 8, TH = 4, NHL = 1; CALL; } \
    else if (TDV == 8 && THV == 4 && NHLV == 2) { constexpr int TD = 8, TH = 4, NHL = 2; CALL; } \
    else if (TDV == 8 && THV == 8 && NHLV == 1) { constexpr int TD = 8, TH = 8, NHL = 1; CALL; } \
    else if (TDV == 8 && THV == 8 && NHLV == 2) { constexpr int TD = 8, TH = 8, NHL = 2; CALL; } \
    else return NFMC_EUNSUPPORTED;

// shapes the matrix-core path covers
int nfmc::nfmc_mfma_supported(int32_t d, int32_t n_hidden, int32_t n_hidden_layers) {
    return (d == 64 || d == 128) && n_hidden > 32 && n_hidden <= 128 && n_hidden_layers >= 1 && n_hidden_layers <= 2;
}

int nfmc::nfmc_neutra_potential_grad_mfma_f32(const NfmcRealNVP* flow, const NfmcPotential* pot, const float* z,
                                                   int64_t n, float* u_out, float* grad_out, nfmc_stream_t stream) {
    if (!flow || !pot || !z || n <= 0) return NFMC_EINVAL;
    if (!nfmc_mfma_supported(flow->d, flow->n_hidden, flow->n_hidden_layers))   // d = 256 / 512: the streamed kernel, else no kernel
        return nfmc_neutra_potential_grad_wide_f32(flow, pot, z, n, u_out, grad_out, stream);
    const int td = flow->d / 16, th = nfmc_realnvp_padded_hidden(flow->n_hidden) / 16, nhl = flow->n_hidden_layers;
    int rc = 0;
    NFMC_MFMA_DISPATCH(td, th, nhl, rc = (launch_grad<TD, TH, NHL>(*flow, *pot, z, n, u_out, grad_out, (hipStream_t)stream)))
    if (rc) return rc;
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

int nfmc::nfmc_neutra_hmc_steps_mfma_f32(const NfmcNeutraHmcArgs* args, float* scratch, int64_t scratch_bytes,
                                              nfmc_stream_t stream) {
    if (!args || !scratch) return NFMC_EINVAL;
    const NfmcNeutraHmcArgs& a = *args;
    if (!nfmc_mfma_supported(a.flow.d, a.flow.n_hidden, a.flow.n_hidden_layers))   // the other multiples of 32: composed (mfma_wide.hip)
        return nfmc_neutra_hmc_steps_wide_f32(args, scratch, scratch_bytes, stream);
    if (scratch_bytes < nfmc_neutra_scratch_bytes(a.n, a.flow.d, a.flow.n_hidden, a.flow.n_hidden_layers, a.flow.n_coupling))
        return NFMC_ESCRATCH;
    const int td = a.flow.d / 16, th = nfmc_realnvp_padded_hidden(a.flow.n_hidden) / 16, nhl = a.flow.n_hidden_layers;
    int rc = 0;
    NFMC_MFMA_DISPATCH(td, th, nhl, rc = (run_hmc<TD, TH, NHL>(a, scratch, (hipStream_t)stream)))
    return rc;
}
