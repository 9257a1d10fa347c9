// f1 for WIDE conditioners (33 <= n_hidden <= 128 at d = 64 / 128: C4's flow) on the matrix cores: the batch loss of
// `Flow.fit` (maximum likelihood, jump.py:139-151,193-201) or `Flow.variational_fit` (reverse KL, imh.py:67-72,
// neutra.py:84-91) and its gradient with respect to EVERY parameter, in one launch; the fold / AdamW / bookkeeping kernel of
// fit_kernels.hip follows.  Round 3 trained these flows with eager torch autograd (C4's warmup: 200 epochs of ~50 launches).
//
// Built on the layout and GEMM scheme of mfma_device.hpp / mfma_flow.hpp (16 rows per wave on the N axis of
// v_mfma_f32_16x16x4_f32, 8 waves = 128 rows per workgroup, weights through two alternating LDS images):
//   forward sweep   the flow pass of the loss (ML: x -> z; reverse KL: z -> x = f^-1(z)) with what the backward sweep needs of
//                   every coupling layer -- both hidden activation tile sets, alpha, beta -- written to a checkpoint area of
//                   the resident wave (mfma_flow.hpp: CkLayout), as NeuTra's trajectory kernel does;
//   backward sweep  per layer the elementwise backward of the affine map, then the transposed products W3^T, Wh^T, W1^T of
//                   neutra_mfma.hip's reverse sweep for the INPUT gradient -- and, new here, the WEIGHT gradients:
//   dW = delta act^T   is a sum over ROWS of outer products, i.e. a GEMM whose K axis is the batch.  The rows of a wave sit on
//                   the N axis of every tile, so both operands are transposed through LDS: all eight waves write their
//                   (element x 16 rows) tiles of delta into weight image 0 and of the activations into image 1 -- image row
//                   = element, image column = row of the workgroup's 128 -- and every wave then owns output tiles
//                   dW[16 it .., 16 kt ..] = sum over 128 rows: 32 MFMAs per output tile, both fragments one 16-byte LDS read
//                   per four k-steps (the k axis is permuted the same way on both sides).  Bias gradients are the row sums of
//                   the delta image.  Three barriers per product; the weight pipeline resumes on either image afterwards.
//   layout          the trainable vector IS the matrix-core weight blob -- every matrix in both orientations -- so the
//                   sampling kernels read each step in place; the gradient of a weight is written to both of its slots,
//                   AdamW (elementwise, deterministic) keeps the two copies bitwise equal.
// Per-workgroup partial gradients go to the caller's slab (`+=` over the workgroup's row tiles in tile order); loss sums to
// its tail.  No atomics: a run repeats bit for bit.
#include "fit_mfma.hpp"
#include "mfma_flow.hpp"

namespace nfmc {

// offsets of a coupling layer's pieces inside its blob (mfma_device.hpp: mfma_layer)
struct MOff {
    int64_t W1, W1T, b1, Wh, WhT, bh, W3, W3T, b3;
};
__host__ __device__ inline MOff mfma_offsets(int d, int hp, int n_hl) {
    const int64_t da = d / 2, db = d - d / 2;
    MOff o;
    o.W1 = 0;
    o.W1T = o.W1 + hp * da;
    o.b1 = o.W1T + da * hp;
    int64_t p = o.b1 + hp;
    o.Wh = o.WhT = o.bh = -1;
    if (n_hl > 1) {
        o.Wh = p;
        o.WhT = o.Wh + (int64_t)hp * hp;
        o.bh = o.WhT + (int64_t)hp * hp;
        p = o.bh + hp;
    }
    o.W3 = p;
    o.W3T = o.W3 + 2 * db * hp;
    o.b3 = o.W3T + (int64_t)hp * 2 * db;
    return o;
}

// sum over the 16 rows of a wave that share a lane group (lanes of one DPP row); every lane gets the sum
__device__ __forceinline__ float rows16_sum(float v) { return group_allreduce<16>(v); }

// ---- dW = delta act^T over the workgroup's 128 rows.  `delta`: TI tiles (C layout), `act`: TK tiles.
//   emit(R, C, v)   receives element (R, C) of the product, R = delta element, C = activation element
//   emit_b(R, v)    receives the row sum of delta element R (the bias gradient)
template <int TI, int TK, class Emit, class EmitB>
__device__ __forceinline__ void dw_phase(const f32x4 (&delta)[TI], const f32x4 (&act)[TK], float* lds, int wave, int col,
                                         int half, Emit emit, EmitB emit_b) {
    constexpr int ld = 132;
    float* const img0 = lds;
    float* const img1 = lds + kImgFloats;
    __syncthreads();   // every wave is done with the weight images
#pragma unroll
    for (int m = 0; m < TI; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) img0[(16 * m + 4 * half + r) * ld + 16 * wave + col] = delta[m][r];
#pragma unroll
    for (int m = 0; m < TK; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) img1[(16 * m + 4 * half + r) * ld + 16 * wave + col] = act[m][r];
    __syncthreads();
    for (int t = wave; t < TI * TK; t += kMfmaWaves) {   // wave-uniform
        const int it = t / TK, kt = t - it * TK;
        const float* arow = img0 + (16 * it + col) * ld + 4 * half;
        const float* brow = img1 + (16 * kt + col) * ld + 4 * half;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < kMfmaChains / 16; ++ks) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 16 * ks);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + 16 * ks);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], b4[r], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) emit(16 * it + 4 * half + r, 16 * kt + col, acc[r]);
    }
    if (wave < TI) {   // bias: row sums of the delta image, rows 16 wave .. 16 wave + 15
        const float* arow = img0 + (16 * wave + col) * ld + 4 * half;
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < kMfmaChains / 16; ++ks) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 16 * ks);
            s += (a4[0] + a4[1]) + (a4[2] + a4[3]);
        }
        s = chain_sum(s);   // the four lane groups hold four quarters of the row
        if (half == 0) emit_b(16 * wave + col, s);
    }
    __syncthreads();   // the images are free again
}

// One coupling layer going BACKWARD through the sweep, with the weight gradients.  x: the layer's OUTPUT in the sweep's
// direction (RKL: v_b = (y_b - beta) / alpha; ML: z_b = alpha x_b + beta) -> its input; g: dL/d(output) -> dL/d(input).
// `vm`: 1 for a row of the batch, 0 for the padding rows of the last tile (their g is zero on entry).
template <int TD, int TH, int NHL, bool REV, bool RKL>
__device__ __forceinline__ void coupling_fit_backward(f32x4 (&x)[TD], f32x4 (&g)[TD], const MLayer& L, const MOff& o, int64_t L0,
                                                      float mscale, float vm, WeightPipe& wp, float* lds, int wave, int col,
                                                      int half, float* ck, float* P, bool first) {
    constexpr int TS = TD / 2, SRC0 = REV ? TS : 0, TGT0 = REV ? 0 : TS, D2 = 8 * TD, hp = 16 * TH, d_a = D2;
    using CL = CkLayout<TD, TH, NHL>;
    auto put = [&](int64_t idx, float v) { P[idx] = first ? v : P[idx] + v; };
    f32x4 du[2 * TS];   // [0, TS): d/d(u_alpha) of the target tiles, [TS, 2 TS): d/d(u_beta)
    f32x4 hl[TH];
#pragma unroll
    for (int mt = 0; mt < TS; ++mt) {
        const f32x4 al = *ck_tile(ck, CL::kAlpha + mt), be = *ck_tile(ck, CL::kBeta + mt);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float alpha = al[t], ra = __builtin_amdgcn_rcpf(alpha);
            if constexpr (RKL) {
                const float y = x[TGT0 + mt][t];                       // v_b
                const float gv = g[TGT0 + mt][t] * ra;                 // dL/dy_b
                const float d_alpha = fmaf(-gv, y, vm * ra);           // loss contains +log alpha
                du[mt][t] = 0.5f * d_alpha * (alpha - mscale);
                du[TS + mt][t] = -0.5f * gv;
                g[TGT0 + mt][t] = gv;
                x[TGT0 + mt][t] = fmaf(alpha, y, be[t]);               // y_b, the inverse layer's input
            } else {
                const float gz = g[TGT0 + mt][t];
                const float xb = (x[TGT0 + mt][t] - be[t]) * ra;       // the forward layer's input
                const float d_alpha = fmaf(gz, xb, -vm * ra);          // loss contains -log alpha
                du[mt][t] = 0.5f * d_alpha * (alpha - mscale);
                du[TS + mt][t] = 0.5f * gz;
                g[TGT0 + mt][t] = gz * alpha;
                x[TGT0 + mt][t] = xb;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < TH; ++m) hl[m] = *ck_tile(ck, CL::kHl + m);
    // ---- dW3 (2 d_b, hp), its transpose, db3: delta = [du; dv] at TARGET POSITIONS (reversed layers: logical t = D2-1-pos)
    dw_phase<2 * TS, TH>(
        du, hl, lds, wave, col, half,
        [&](int R, int C, float v) {
            const int blk = R / D2, pos = R - blk * D2, t = blk * D2 + (REV ? D2 - 1 - pos : pos);
            put(L0 + o.W3 + (int64_t)t * hp + C, v);
            put(L0 + o.W3T + (int64_t)C * (2 * D2) + t, v);
        },
        [&](int R, float v) {
            const int blk = R / D2, pos = R - blk * D2;
            put(L0 + o.b3 + blk * D2 + (REV ? D2 - 1 - pos : pos), v);
        });
    // ---- dL/dh_last = W3^T [du; dv], through tanh of the last hidden layer
    f32x4 dh[TH];
    {
        wp.template stage<2 * D2, 1, D2, 1, hp>(L.W3T, false, REV, nullptr, 0, false);
        const float* img = wp.img();
        f32x4 dus[TS], dvs[TS];
#pragma unroll
        for (int mt = 0; mt < TS; ++mt) {
            dus[mt] = du[mt];
            dvs[mt] = du[TS + mt];
        }
        gemm_phase<TS, 2 * TH>(
            [&](int i) { return img + (16 * (2 * (i >> 2) + (i & 1)) + col) * (2 * D2 + 4) + 4 * half + ((i >> 1) & 1) * D2; },
            [&](int i) {
                if (((i >> 1) & 1) == 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) dh[2 * (i >> 2) + (i & 1)][t] = 0.f;
                }
            },
            [&](int i) -> f32x4& { return dh[2 * (i >> 2) + (i & 1)]; },
            [&](int i) -> const f32x4(&)[TS] { return ((i >> 1) & 1) ? dvs : dus; },
            [&](int i) {
                if (((i >> 1) & 1) == 0) return;
                const int mo = 2 * (i >> 2) + (i & 1);
#pragma unroll
                for (int t = 0; t < 4; ++t) dh[mo][t] *= (1.f - hl[mo][t] * hl[mo][t]);
            });
    }
    if constexpr (NHL > 1) {
        f32x4 h1k[TH];
#pragma unroll
        for (int mo = 0; mo < TH; ++mo) h1k[mo] = *ck_tile(ck, CL::kH1 + mo);
        // ---- dWh (out, in), its transpose, dbh
        dw_phase<TH, TH>(
            dh, h1k, lds, wave, col, half,
            [&](int R, int C, float v) {
                put(L0 + o.Wh + (int64_t)R * hp + C, v);
                put(L0 + o.WhT + (int64_t)C * hp + R, v);
            },
            [&](int R, float v) { put(L0 + o.bh + R, v); });
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
#pragma unroll
        for (int mo = 0; mo < TH; ++mo)
#pragma unroll
            for (int t = 0; t < 4; ++t) dh[mo][t] = hl[mo][t] * (1.f - h1k[mo][t] * h1k[mo][t]);
    }
    // ---- dW1 (hp, d_a), its transpose, db1: the activations are the SOURCE tiles (reversed layers: logical j = d_a-1-pos)
    {
        f32x4 src[TS];
#pragma unroll
        for (int ms = 0; ms < TS; ++ms) src[ms] = x[SRC0 + ms];
        dw_phase<TH, TS>(
            dh, src, lds, wave, col, half,
            [&](int R, int C, float v) {
                const int j = REV ? d_a - 1 - C : C;
                put(L0 + o.W1 + (int64_t)R * d_a + j, v);
                put(L0 + o.W1T + (int64_t)j * hp + R, v);
            },
            [&](int R, float v) { put(L0 + o.b1 + R, v); });
    }
    // ---- dL/d(source) += W1^T dh
    wp.template stage<hp, D2, 1, 1, D2>(L.W1T, REV, false, nullptr, 0, false);
    const float* img = wp.img();
    gemm_phase<TH, TS>([&](int ms) { return img + (16 * ms + col) * (hp + 4) + 4 * half; }, [&](int) {},
                       [&](int ms) -> f32x4& { return g[SRC0 + ms]; },
                       [&](int) -> const f32x4(&)[TH] { return dh; }, [&](int) {});
}

template <int TD, int TH, int NHL, bool RKL>
__global__ void __launch_bounds__(kMfmaBlock, 2) fit_mfma_kernel(FitMfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (a.run_state && (a.run_state[3] != 0.f || a.run_state[4] != 0.f)) return;   // the run has ended
    constexpr int d = 16 * TD, hp = 16 * TH;
    using CL = CkLayout<TD, TH, NHL>;
    const NfmcRealNVP& f = a.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, half = lane >> 4;
    const bool rev_last = (f.n_coupling & 1) != 0;
    const float log1m = __logf(1.f - f.min_scale);
    const MOff o = mfma_offsets(d, hp, NHL);
    float* const red = lds + kMfmaStatOffset;   // [8 waves][2 d] elementwise-affine partial sums, then [8][4] loss tails
    float* const tails = red + kMfmaWaves * 2 * d;
    float* const ck = a.ck + ((size_t)blockIdx.x * kMfmaWaves + __builtin_amdgcn_readfirstlane(wave)) * f.n_coupling * CL::kLayerFloats +
                      4 * lane;
    float* const P = a.partial + (int64_t)blockIdx.x * a.pstride;
    WeightPipe wp{lds, 0};
    bool first = true;
    float loss_acc = 0.f, rows_acc = 0.f, vloss_acc = 0.f, vrows_acc = 0.f;

    // sum over the workgroup's rows of two per-element quantities (C layout) -> the slab entries of one ElementwiseAffine layer
    auto ea_flush = [&](const f32x4 (&as)[TD], const f32x4 (&at)[TD], int which, bool logical) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < TD; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float s1 = rows16_sum(as[m][t]), s2 = rows16_sum(at[m][t]);
                if (col == 0) {
                    red[wave * 2 * d + 16 * m + 4 * half + t] = s1;
                    red[wave * 2 * d + d + 16 * m + 4 * half + t] = s2;
                }
            }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * d; i += kMfmaBlock) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < kMfmaWaves; ++w) v += red[w * 2 * d + i];
            const int pos = i < d ? i : i - d;
            const int c = logical && rev_last ? d - 1 - pos : pos;
            const int64_t idx = a.ea_off + (int64_t)(which + (i < d ? 0 : 1)) * a.d4 + c;
            P[idx] = first ? v : P[idx] + v;
        }
    };

    for (int64_t tile = blockIdx.x; tile < a.tiles + a.vtiles; tile += gridDim.x) {
        const bool is_val = tile >= a.tiles;                         // uniform over the workgroup
        const float* src = is_val ? a.xv : a.x;
        const int64_t nrows = is_val ? a.nv : a.n;
        const int64_t row = (is_val ? tile - a.tiles : tile) * kMfmaChains + wave * 16 + col;
        const bool active = row < nrows;
        const int64_t rrow = active ? row : nrows - 1;
        const float vm = active ? 1.f : 0.f;
        f32x4 x[TD], g[TD];
        // ML rows are data in physical order; reverse-KL rows are latents, array column c at tile position (rev_last ? d-1-c : c)
        load_ctiles<TD>(x, src, rrow, d, half, RKL && rev_last);
        float ldp = 0.f, ss = 0.f;
        if constexpr (!RKL) {
#pragma unroll
            for (int m = 0; m < TD; ++m) {   // EA0
                const f32x4 ls = vec_tile(f.ea0_log_scale, m, half), sh = vec_tile(f.ea0_shift, m, half);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    x[m][t] = fmaf(fast_exp(ls[t]), x[m][t], sh[t]);
                    ldp += ls[t];
                }
            }
            for (int l = 0; l < f.n_coupling; ++l) {
                const MLayer L = mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL);
                float* ckl = ck + (size_t)l * CL::kLayerFloats;
                if ((l & 1) == 0) ldp += coupling_c<TD, TH, NHL, true, false, true>(x, L, f.min_scale, log1m, wp, col, half, ckl);
                else ldp += coupling_c<TD, TH, NHL, false, false, true>(x, L, f.min_scale, log1m, wp, col, half, ckl);
            }
#pragma unroll
            for (int m = 0; m < TD; ++m) {   // EA1 (logical latent coordinates)
                const f32x4 ls = rev_last ? vec_tile_rev(f.ea1_log_scale, m, half, d) : vec_tile(f.ea1_log_scale, m, half);
                const f32x4 sh = rev_last ? vec_tile_rev(f.ea1_shift, m, half, d) : vec_tile(f.ea1_shift, m, half);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    x[m][t] = fmaf(fast_exp(ls[t]), x[m][t], sh[t]);
                    ldp += ls[t];
                    ss = fmaf(x[m][t], x[m][t], ss);
                }
            }
            const float li = 0.5f * chain_sum(ss) + 0.5f * (float)d * kLog2Pi - chain_sum(ldp);
            if (active && half == 0) {
                if (is_val) {
                    vloss_acc += li;
                    vrows_acc += 1.f;
                } else {
                    loss_acc += li;
                    rows_acc += 1.f;
                }
            }
            if (is_val) continue;   // validation rows: loss only
#pragma unroll
            for (int m = 0; m < TD; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) g[m][t] = vm * x[m][t];   // dL/dz of 0.5 |z|^2
            // ---- last ElementwiseAffine backward: z = e^s y + t
            {
                f32x4 as[TD], at[TD];
#pragma unroll
                for (int m = 0; m < TD; ++m) {
                    const f32x4 ls = rev_last ? vec_tile_rev(f.ea1_log_scale, m, half, d) : vec_tile(f.ea1_log_scale, m, half);
                    const f32x4 sh = rev_last ? vec_tile_rev(f.ea1_shift, m, half, d) : vec_tile(f.ea1_shift, m, half);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float gz = g[m][t], zc = x[m][t] - sh[t];
                        as[m][t] = fmaf(gz, zc, -vm);
                        at[m][t] = gz;
                        g[m][t] = gz * fast_exp(ls[t]);
                        x[m][t] = zc * fast_exp(-ls[t]);
                    }
                }
                ea_flush(as, at, 2, true);
            }
            for (int l = f.n_coupling - 1; l >= 0; --l) {
                const MLayer L = mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL);
                float* ckl = ck + (size_t)l * CL::kLayerFloats;
                const int64_t L0 = (int64_t)l * f.layer_stride;
                if ((l & 1) == 0)
                    coupling_fit_backward<TD, TH, NHL, true, false>(x, g, L, o, L0, f.min_scale, vm, wp, lds, wave, col, half, ckl, P, first);
                else
                    coupling_fit_backward<TD, TH, NHL, false, false>(x, g, L, o, L0, f.min_scale, vm, wp, lds, wave, col, half, ckl, P, first);
            }
            // ---- first ElementwiseAffine backward: registers hold y = e^s x + t and dL/dy
            {
                f32x4 as[TD], at[TD];
#pragma unroll
                for (int m = 0; m < TD; ++m) {
                    const f32x4 sh = vec_tile(f.ea0_shift, m, half);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        as[m][t] = fmaf(g[m][t], x[m][t] - sh[t], -vm);
                        at[m][t] = g[m][t];
                    }
                }
                ea_flush(as, at, 0, false);
            }
        } else {
#pragma unroll
            for (int m = 0; m < TD; ++m) {   // EA1^-1
                const f32x4 ls = rev_last ? vec_tile_rev(f.ea1_log_scale, m, half, d) : vec_tile(f.ea1_log_scale, m, half);
                const f32x4 sh = rev_last ? vec_tile_rev(f.ea1_shift, m, half, d) : vec_tile(f.ea1_shift, m, half);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    ss = fmaf(x[m][t], x[m][t], ss);
                    x[m][t] = (x[m][t] - sh[t]) * fast_exp(-ls[t]);
                    ldp -= ls[t];
                }
            }
            for (int l = f.n_coupling - 1; l >= 0; --l) {
                const MLayer L = mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL);
                float* ckl = ck + (size_t)l * CL::kLayerFloats;
                if ((l & 1) == 0) ldp += coupling_c<TD, TH, NHL, true, true, true>(x, L, f.min_scale, log1m, wp, col, half, ckl);
                else ldp += coupling_c<TD, TH, NHL, false, true, true>(x, L, f.min_scale, log1m, wp, col, half, ckl);
            }
#pragma unroll
            for (int m = 0; m < TD; ++m) {   // EA0^-1
                const f32x4 ls = vec_tile(f.ea0_log_scale, m, half), sh = vec_tile(f.ea0_shift, m, half);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    x[m][t] = (x[m][t] - sh[t]) * fast_exp(-ls[t]);
                    ldp -= ls[t];
                }
            }
            // loss_i = log N(z) - logdet_inverse + U(x); dL/dx = grad U
            const float u = potential_value_grad_c<TD>(x, g, a.pot, half, lane);
            const float li = -0.5f * chain_sum(ss) - 0.5f * (float)d * kLog2Pi - chain_sum(ldp) + u;
            if (active && half == 0) {
                loss_acc += li;
                rows_acc += 1.f;
            }
#pragma unroll
            for (int m = 0; m < TD; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) g[m][t] *= vm;
            // ---- first ElementwiseAffine inverted, backward: x = (y - t) e^-s, -logdet_inverse contains +s
            {
                f32x4 as[TD], at[TD];
#pragma unroll
                for (int m = 0; m < TD; ++m) {
                    const f32x4 ls = vec_tile(f.ea0_log_scale, m, half), sh = vec_tile(f.ea0_shift, m, half);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float gx = g[m][t], xv = x[m][t];
                        const float gy = gx * fast_exp(-ls[t]);
                        as[m][t] = fmaf(-gx, xv, vm);
                        at[m][t] = -gy;
                        g[m][t] = gy;
                        x[m][t] = fmaf(fast_exp(ls[t]), xv, sh[t]);
                    }
                }
                ea_flush(as, at, 0, false);
            }
            for (int l = 0; l < f.n_coupling; ++l) {
                const MLayer L = mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL);
                float* ckl = ck + (size_t)l * CL::kLayerFloats;
                const int64_t L0 = (int64_t)l * f.layer_stride;
                if ((l & 1) == 0)
                    coupling_fit_backward<TD, TH, NHL, true, true>(x, g, L, o, L0, f.min_scale, vm, wp, lds, wave, col, half, ckl, P, first);
                else
                    coupling_fit_backward<TD, TH, NHL, false, true>(x, g, L, o, L0, f.min_scale, vm, wp, lds, wave, col, half, ckl, P, first);
            }
            // ---- last ElementwiseAffine inverted, backward: registers hold v = (z - t) e^-s and dL/dv (logical coordinates)
            {
                f32x4 as[TD], at[TD];
#pragma unroll
                for (int m = 0; m < TD; ++m) {
                    const f32x4 ls = rev_last ? vec_tile_rev(f.ea1_log_scale, m, half, d) : vec_tile(f.ea1_log_scale, m, half);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        as[m][t] = fmaf(-g[m][t], x[m][t], vm);
                        at[m][t] = -g[m][t] * fast_exp(-ls[t]);
                    }
                }
                ea_flush(as, at, 2, true);
            }
        }
        first = false;
    }
    // ---- losses and row counts: lanes of lane group 0 hold one row each; rows of a wave, then the eight waves, in order
    loss_acc = rows16_sum(loss_acc);
    rows_acc = rows16_sum(rows_acc);
    vloss_acc = rows16_sum(vloss_acc);
    vrows_acc = rows16_sum(vrows_acc);
    __syncthreads();
    if (lane == 0) {
        tails[wave * 4 + 0] = loss_acc;
        tails[wave * 4 + 1] = rows_acc;
        tails[wave * 4 + 2] = vloss_acc;
        tails[wave * 4 + 3] = vrows_acc;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kMfmaWaves; ++w) v += tails[w * 4 + threadIdx.x];
        P[a.n_params + threadIdx.x] = v;
    }
}

template <int TD, int TH, int NHL>
static int fit_mfma_go(bool rkl, const FitMfmaArgs& a, int grid, hipStream_t st) {
    hipError_t e;
    if (rkl) {
        auto kern = fit_mfma_kernel<TD, TH, NHL, true>;
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfmaLdsBytes);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kMfmaBlock), kMfmaLdsBytes, st, a);
    } else {
        auto kern = fit_mfma_kernel<TD, TH, NHL, false>;
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMfmaLdsBytes);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kMfmaBlock), kMfmaLdsBytes, st, a);
    }
    return 0;
}

int64_t fit_mfma_ck_floats(int d, int hp, int n_hl, int n_coupling, int grid) {
    const int64_t th = hp / 16, td = d / 16;
    const int64_t layer_floats = ((n_hl > 1 ? 2 : 1) * th + td) * 256;   // CkLayout::kLayerFloats
    return (int64_t)grid * kMfmaWaves * n_coupling * layer_floats;
}

int fit_mfma_grid(int64_t n, int64_t nv) {
    const int64_t tiles = (n + kMfmaChains - 1) / kMfmaChains + (nv + kMfmaChains - 1) / kMfmaChains;
    return (int)(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
}

int fit_mfma_launch(bool rkl, const FitMfmaArgs& a, int grid, hipStream_t st) {
    const int td = a.f.d / 16, th = nfmc_realnvp_padded_hidden(a.f.n_hidden) / 16, nhl = a.f.n_hidden_layers;
    if (td == 4 && th == 4 && nhl == 1) return fit_mfma_go<4, 4, 1>(rkl, a, grid, st);
    if (td == 4 && th == 4 && nhl == 2) return fit_mfma_go<4, 4, 2>(rkl, a, grid, st);
    if (td == 4 && th == 8 && nhl == 1) return fit_mfma_go<4, 8, 1>(rkl, a, grid, st);
    if (td == 4 && th == 8 && nhl == 2) return fit_mfma_go<4, 8, 2>(rkl, a, grid, st);
    if (td == 8 && th == 4 && nhl == 1) return fit_mfma_go<8, 4, 1>(rkl, a, grid, st);
    if (td == 8 && th == 4 && nhl == 2) return fit_mfma_go<8, 4, 2>(rkl, a, grid, st);
    if (td == 8 && th == 8 && nhl == 1) return fit_mfma_go<8, 8, 1>(rkl, a, grid, st);
    if (td == 8 && th == 8 && nhl == 2) return fit_mfma_go<8, 8, 2>(rkl, a, grid, st);
    return NFMC_EUNSUPPORTED;
}

}  // namespace nfmc
