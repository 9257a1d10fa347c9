// The matrix-core path for WIDE EVENTS: d = 256 and d = 512 -- and every other multiple of 32 that the register-resident
// kernels do not serve -- with RealNVP conditioners of width 33..128:
//   K3 / K4   bijection.forward / inverse, Flow.log_prob / sample             (jump.py:205,218; imh.py:214,221)
//   K5        the adjusted potential U~(z) = U(f^-1(z)) - logdet_inv(z) and its gradient   (neutra.py:58-68)
// The reference has no shape limit at these call sites; the register-resident kernels of flow_mfma.hip / neutra_mfma.hip
// hold the state (and gradient) tiles of a chain in VGPRs and stop at d = 128.
//
// Same GEMM scheme as mfma_device.hpp (v_mfma_f32_16x16x4_f32, weights = A from an LDS image, 16 chains of a wave = N,
// every per-chain vector in the C layout), but the state x and the gradient g of the wave's chains are STREAMED: they live
// in a scratch slab in HBM / L2 in tile-position order, and a GEMM phase loads the 16-byte tiles it needs (B operands of
// the first conditioner layer, target tiles in the affine epilogues, gradient accumulators of the last transposed
// product) and stores what it changed.  Only the conditioner's hidden activations (<= 3 x 8 tiles) stay in registers, so
// the register budget does not depend on d.  A lane only ever reads back tiles it wrote itself.
//
// The operands no longer fit an LDS image whole (W3 is 2 d_b x hp: 256-512 rows), so they are staged in slices of 128
// source or target coordinates:
//   W1 (hp x d_a)     column slices, accumulated into the same hidden tiles               d_a / 128 images
//   W3 (2 d_b x hp)   groups of 4 target tiles: their 64 alpha rows + 64 beta rows        d_b / 64 images
//   W3^T              the same groups, in the second image; the reverse sweep FUSES the W3 product, the affine backward
//                     and the W3^T product per target tile, so du / dv never exist as whole vectors
//   W1^T (d_a x hp)   row slices, accumulating into gradient tiles streamed through the epilogue
// The stagings alternate between the two images, one barrier each (WideCtx::buf; the reverse sweep's W3 / W3^T groups take
// both images behind a full barrier): a wave stages GEMM k's operand right after its GEMM k - 1, under the late waves'
// MFMAs.  Round 4: 0.756 -> 0.745 ms for the d = 256 gradient, 0.279 -> 0.272 ms forward, 0.503 -> 0.480 ms for the d = 512
// inverse -- as in the register-resident kernels, the barriers were not the main loss.  The reverse sweep recomputes the
// hidden stack (no checkpoints).
#include "mfma_flow.hpp"

namespace nfmc {

constexpr int kWideSlice = 128;                 // source / target coordinates per weight image
// 16-byte staging pieces a thread has in flight (stage_any's kBatch).  A 128 x 128 image is 8 pieces per thread: with ALL of
// them requested before the first is stored a staging is one L2 round trip instead of eight.  Round 3 gave the gradient
// kernels what 256 registers left without scratch (1 or 2); measured in round 4 (tools/probe_wide.py, d = 256, H = 128 x 2,
// NFMC_WIDE_*_SB variants on one box) the round trips cost more than the spills that the deeper batch brings:
//   gradient kernel      SB 1: 0.749 ms (225 VGPR, no scratch)   4: 0.704 (29 spilled)   8: 0.695 (16 spilled)
//   one-launch leapfrog  SB 1: 3.42 ms per trajectory (45 spilled)   4: 3.21 (109)   8: 3.34 (101)
//   forward / inverse    SB 4: 0.270 / 0.275 ms   8: 0.267 / 0.271 (no scratch either way)
#ifndef NFMC_WIDE_FLOW_SB
#define NFMC_WIDE_FLOW_SB 8
#endif
#ifndef NFMC_WIDE_GRAD_SB
#define NFMC_WIDE_GRAD_SB 8
#endif
#ifndef NFMC_WIDE_LEAP_SB
#define NFMC_WIDE_LEAP_SB 4
#endif
constexpr int kWideFlowSB = NFMC_WIDE_FLOW_SB, kWideGradSB = NFMC_WIDE_GRAD_SB, kWideLeapSB = NFMC_WIDE_LEAP_SB;
constexpr size_t kWideLdsBytes = (size_t)(2 * kImgFloats + 2 * kVecFloats) * sizeof(float);

__device__ __forceinline__ f32x4 rev4(const f32x4 a) {
    f32x4 t;
    t[0] = a[3], t[1] = a[2], t[2] = a[1], t[3] = a[0];
    return t;
}

// img[r * ld + c] = W[srow(r) * ldw + scol(c)] for r < Rv, c < Kv and 0 elsewhere (r < R, c < K = 4 << k4log), where
// srow(r) = rrev ? row0 - r : row0 + r and scol(c) = crev ? col0 - c : col0 + c (row0 / col0: the source index of image
// index 0).  All threads of the workgroup; Kv, ldw and the first source column of every 4-piece are multiples of 4 and W is
// 16-byte aligned.
template <int kBatch>
__device__ __forceinline__ void stage_any(float* __restrict__ img, int ld, const float* __restrict__ W, int ldw, int R, int k4log, int Rv,
                                          int Kv, int row0, bool rrev, int col0, bool crev) {
    const int k4 = 1 << k4log, N = R << k4log;
    // The weights sit in L2: a copy is latency, not bandwidth.  kBatch 16-byte pieces per thread are requested before the
    // first is stored (as a plain loop every piece was load -> wait -> store: 8 dependent round trips per 128 x 128 image);
    // kBatch per kernel: kWideFlowSB / kWideGradSB / kWideLeapSB.
    for (int base = threadIdx.x; base < N; base += kBatch * kMfmaBlock) {
        f32x4 v[kBatch];
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            const int idx = base + b * kMfmaBlock;
            const int r = idx >> k4log, c = (idx & (k4 - 1)) << 2;
            v[b][0] = v[b][1] = v[b][2] = v[b][3] = 0.f;
            if (idx < N && r < Rv && c < Kv) {
                const float* src = W + (size_t)(rrev ? row0 - r : row0 + r) * ldw + (crev ? col0 - c - 3 : col0 + c);
                v[b] = *reinterpret_cast<const f32x4*>(src);
            }
        }
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            const int idx = base + b * kMfmaBlock;
            const int r = idx >> k4log, c = (idx & (k4 - 1)) << 2;
            if (idx < N) *reinterpret_cast<f32x4*>(img + r * ld + c) = crev ? rev4(v[b]) : v[b];
        }
    }
}

__device__ __forceinline__ int log2i(int v) { return 31 - __builtin_clz(v); }

struct WideCtx {
    float *img0, *img1, *vec0;   // LDS
    // The stagings ALTERNATE between the two weight images (and the two bias vectors): staging k writes the image GEMM
    // k - 2 read, and a wave passes the barrier that closes staging k - 1 only after it has finished GEMM k - 2, so one
    // barrier per staging is enough (the first version staged everything into image 0 and needed a second barrier in
    // front of each staging).  A phase that uses BOTH images (the reverse sweep's W3 / W3^T groups, the statistics that
    // borrow the image memory) opens with a full barrier, after which either image may be written.
    mutable int buf;
    float* xs;                   // this lane's chain in the state slab: element (16 m + 4 q + t) of tile m at xs[16 m + t]
    float* gs;                   // the same in the gradient slab
    int d, D2, TS, nslice, ngroup;   // D2 = d / 2 = 16 TS; slices of 128 source coordinates, groups of 64 target coordinates (the last may be partial)
    int col, q;
    float mscale, log1m;
};

__device__ __forceinline__ f32x4 tile_ld(const float* p, int m) { return *reinterpret_cast<const f32x4*>(p + 16 * m); }
__device__ __forceinline__ void tile_st(float* p, int m, const f32x4 v) { *reinterpret_cast<f32x4*>(p + 16 * m) = v; }

// A pass over the TD tiles of a streamed vector, four tiles at a time: the four loads are requested before the first
// tile is used (a plain loop was load -> wait -> use -> store per tile: one exposed memory round trip per tile, 16-32 per
// pass, and the passes between the GEMM phases were most of the first version's time).
template <class Load, class Body>
__device__ __forceinline__ void wide_tiles(int TD, Load load, Body body) {
    for (int m0 = 0; m0 < TD; m0 += 4) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (m0 + j < TD) v[j] = load(m0 + j);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (m0 + j < TD) body(m0 + j, v[j]);
    }
}

__device__ __forceinline__ void wide_ctx_init(WideCtx& c, float* lds, const NfmcRealNVP& f, int lane) {
    c.img0 = lds;
    c.img1 = lds + kImgFloats;
    c.vec0 = lds + 2 * kImgFloats;
    c.buf = 0;
    c.d = f.d, c.D2 = f.d / 2, c.TS = f.d / 32, c.nslice = (f.d / 2 + kWideSlice - 1) / kWideSlice, c.ngroup = (f.d / 2 + 63) / 64;
    c.col = lane & 15, c.q = lane >> 4;
    c.mscale = f.min_scale;
    c.log1m = __logf(1.f - f.min_scale);
    c.gs = nullptr;
}

// the image and bias vector the next staging writes
__device__ __forceinline__ void wide_next(const WideCtx& c, float*& img, float*& vec) {
    c.buf ^= 1;
    img = c.buf ? c.img1 : c.img0;
    vec = c.vec0 + (c.buf ? kVecFloats : 0);
}

__device__ __forceinline__ float wide_sum_squares(const WideCtx& c) {
    float ss = 0.f;
    wide_tiles(c.d / 16, [&](int m) { return tile_ld(c.xs, m); },
               [&](int, const f32x4& v) {
#pragma unroll
                   for (int t = 0; t < 4; ++t) ss = fmaf(v[t], v[t], ss);
               });
    return chain_sum(ss);
}

// slab row -> (n, d) array; `rev`: the array is in LOGICAL latent order and tile position p holds logical d - 1 - p
__device__ __forceinline__ void wide_store_row(const float* src, float* __restrict__ dst_row, int d, int q, bool rev) {
    wide_tiles(d / 16, [&](int m) { return tile_ld(src, m); },
               [&](int m, const f32x4& v) {
                   const int p0 = 16 * m + 4 * q;
                   if (!rev) *reinterpret_cast<f32x4*>(dst_row + p0) = v;
                   else *reinterpret_cast<f32x4*>(dst_row + (d - 4 - p0)) = rev4(v);
               });
}

// h1 = tanh(W1 x_src + b1), hl = the last hidden layer's activations (h1 itself with one hidden layer)
template <int TH, int NHL, int SB>
__device__ __forceinline__ void wide_hidden(const WideCtx& c, const MLayer& L, bool REV, f32x4 (&h1)[TH], f32x4 (&h2)[TH]) {
    constexpr int hp = 16 * TH;
    const float* xsrc = c.xs + 16 * (REV ? c.D2 / 16 : 0);
    for (int ks = 0; ks < c.nslice; ++ks) {
        float *img, *vec;
        wide_next(c, img, vec);
        // source positions [128 ks, 128 ks + 128) of the half; a reversed layer's position p is column D2 - 1 - p
        stage_any<SB>(img, kWideSlice + 4, L.W1, c.D2, hp, 5, hp, c.D2 - kWideSlice * ks, 0, false,
                  REV ? c.D2 - 1 - kWideSlice * ks : kWideSlice * ks, REV);
        if (ks == 0)
            for (int i = threadIdx.x; i < hp; i += kMfmaBlock) vec[i] = L.b1[i];
        f32x4 src[8];
#pragma unroll
        for (int ms = 0; ms < 8; ++ms) {
            if (8 * ks + ms < c.TS) src[ms] = tile_ld(xsrc, 8 * ks + ms);
            else src[ms][0] = src[ms][1] = src[ms][2] = src[ms][3] = 0.f;   // past the half (d_a not a multiple of 128)
        }
        __syncthreads();
        const int col = c.col, q = c.q;
        const bool first = ks == 0, last = ks + 1 == c.nslice;
        gemm_phase<8, TH>([&](int mo) { return img + (16 * mo + col) * (kWideSlice + 4) + 4 * q; },
                          [&](int mo) {
                              if (first) h1[mo] = vec_tile(vec, mo, q);
                          },
                          [&](int mo) -> f32x4& { return h1[mo]; }, [&](int) -> const f32x4(&)[8] { return src; },
                          [&](int mo) {
                              if (last) h1[mo] = tanh4(h1[mo]);
                          });
    }
    if constexpr (NHL > 1) {
        float *img, *vec;
        wide_next(c, img, vec);
        stage_any<SB>(img, hp + 4, L.Wh, hp, hp, log2i(hp / 4), hp, hp, 0, false, 0, false);
        for (int i = threadIdx.x; i < hp; i += kMfmaBlock) vec[i] = L.bh[i];
        __syncthreads();
        const int col = c.col, q = c.q;
        gemm_phase<TH, TH>([&](int mo) { return img + (16 * mo + col) * (hp + 4) + 4 * q; },
                           [&](int mo) { h2[mo] = vec_tile(vec, mo, q); }, [&](int mo) -> f32x4& { return h2[mo]; },
                           [&](int) -> const f32x4(&)[TH] { return h1; }, [&](int mo) { h2[mo] = tanh4(h2[mo]); });
    }
}

// the alpha and beta rows (and biases) of target group gq -> rows [0, 64) and [64, 128) of `img`, vec[0, 128)
template <int TH, int SB>
__device__ __forceinline__ void wide_stage_w3(const WideCtx& c, const MLayer& L, bool REV, int gq, float* img, float* vec) {
    constexpr int hp = 16 * TH;
    const int rv = c.D2 - 64 * gq;                            // target positions [64 gq, 64 gq + 64) of the half that exist
    const int r0 = REV ? c.D2 - 1 - 64 * gq : 64 * gq;        // a reversed layer's target position p is output row D2 - 1 - p
    stage_any<SB>(img, hp + 4, L.W3, hp, 64, log2i(hp / 4), rv, hp, r0, REV, 0, false);
    stage_any<SB>(img + 64 * (hp + 4), hp + 4, L.W3, hp, 64, log2i(hp / 4), rv, hp, c.D2 + r0, REV, 0, false);
    for (int i = threadIdx.x; i < 128; i += kMfmaBlock) {
        const int r = i & 63, blk = i >> 6;
        vec[i] = r < rv ? L.b3[blk * c.D2 + (REV ? r0 - r : r0 + r)] : 0.f;
    }
}

// ---- one coupling layer on the streamed state.  INVERSE: v_b = (y_b - beta) / alpha, else z_b = alpha x_b + beta.
// Returns this lane's share of the layer's logdet in THAT direction.
template <int TH, int NHL, bool INVERSE, int SB>
__device__ __forceinline__ float wide_coupling(const WideCtx& c, const MLayer& L, bool REV) {
    constexpr int hp = 16 * TH;
    f32x4 h1[TH], h2[TH];
    wide_hidden<TH, NHL, SB>(c, L, REV, h1, h2);
    const f32x4(&hl)[TH] = NHL > 1 ? h2 : h1;
    float* xt = c.xs + 16 * (REV ? 0 : c.D2 / 16);
    float ld = 0.f;
    for (int gq = 0; gq < c.ngroup; ++gq) {
        float *img, *vec;
        wide_next(c, img, vec);
        wide_stage_w3<TH, SB>(c, L, REV, gq, img, vec);
        __syncthreads();
        const int col = c.col, q = c.q;
        const float mscale = c.mscale, log1m = c.log1m;
        f32x4 ua2[2], ub2[2], y2[2];
        gemm_phase<TH, 8>(
            [&](int i) { return img + (16 * ((i & 1) * 4 + (i >> 1)) + col) * (hp + 4) + 4 * q; },
            [&](int i) {
                ((i & 1) ? ub2 : ua2)[(i >> 1) & 1] = vec_tile(vec, (i & 1) * 4 + (i >> 1), q);
                if ((i & 1) == 0 && 4 * gq + (i >> 1) < c.TS) y2[(i >> 1) & 1] = tile_ld(xt, 4 * gq + (i >> 1));   // arrives under the two steps' MFMAs
            },
            [&](int i) -> f32x4& { return ((i & 1) ? ub2 : ua2)[(i >> 1) & 1]; },
            [&](int) -> const f32x4(&)[TH] { return hl; },
            [&](int i) {
                const int mt = i >> 1;
                if ((i & 1) == 0 || 4 * gq + mt >= c.TS) return;   // the pair's first step; a tile past the half
                const f32x4 ua = ua2[mt & 1], ub = ub2[mt & 1];
                f32x4 v = y2[mt & 1];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float alpha = fast_exp(fmaf(0.5f, ua[t], log1m)) + mscale;
                    if constexpr (INVERSE) {
                        v[t] = (v[t] - 0.5f * ub[t]) * __builtin_amdgcn_rcpf(alpha);
                        ld -= fast_ln(alpha);
                    } else {
                        v[t] = fmaf(alpha, v[t], 0.5f * ub[t]);
                        ld += fast_ln(alpha);
                    }
                }
                tile_st(xt, 4 * gq + mt, v);
            });
    }
    return ld;
}

// ---- reverse sweep through one inverse coupling layer on the streamed state and gradient (the arithmetic of
// coupling_inverse_backward_c in neutra_mfma.hip, without checkpoints): per target tile the W3 product, the elementwise
// backward of the affine map (state tile restored, gradient tile rescaled, du / dv) and the W3^T product into dL/dh are
// done back to back; then W_h^T, tanh' and W1^T into the source half of the gradient.
template <int TH, int NHL, int SB>
__device__ __forceinline__ void wide_coupling_backward(const WideCtx& c, const MLayer& L, bool REV) {
    constexpr int hp = 16 * TH;
    f32x4 h1[TH], h2[TH];
    wide_hidden<TH, NHL, SB>(c, L, REV, h1, h2);
    f32x4(&hl)[TH] = NHL > 1 ? h2 : h1;
    float* xt = c.xs + 16 * (REV ? 0 : c.D2 / 16);
    float* gt = c.gs + 16 * (REV ? 0 : c.D2 / 16);
    f32x4 dh[TH];
#pragma unroll
    for (int mo = 0; mo < TH; ++mo)
#pragma unroll
        for (int t = 0; t < 4; ++t) dh[mo][t] = 0.f;
    const int col = c.col, q = c.q;
    const float mscale = c.mscale, log1m = c.log1m;
    for (int gq = 0; gq < c.ngroup; ++gq) {
        __syncthreads();   // both images are written: every wave has finished the previous GEMMs
        wide_stage_w3<TH, SB>(c, L, REV, gq, c.img0, c.vec0);
        // W3^T of the same four target tiles -> image 1, a tile's 16 alpha columns next to its 16 beta columns
        for (int mt = 0; mt < 4; ++mt) {
            const int p0 = 64 * gq + 16 * mt;                      // target position of the tile's first coordinate
            const int kv = p0 < c.D2 ? 16 : 0;                     // a tile past the half: zeros
            const int c0 = REV ? c.D2 - 1 - p0 : p0;               // a reversed layer's position p is column D2 - 1 - p
            stage_any<SB>(c.img1 + 32 * mt, kWideSlice + 4, L.W3T, 2 * c.D2, hp, 2, hp, kv, 0, false, c0, REV);
            stage_any<SB>(c.img1 + 32 * mt + 16, kWideSlice + 4, L.W3T, 2 * c.D2, hp, 2, hp, kv, 0, false, c.D2 + c0, REV);
        }
        __syncthreads();
        const float* img = c.img0;
        const float* imgT = c.img1;
        const float* vec = c.vec0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (4 * gq + mt >= c.TS) break;   // uniform: the half ends inside this group
            f32x4 uab[2];
            const f32x4 y = tile_ld(xt, 4 * gq + mt), gy = tile_ld(gt, 4 * gq + mt);
            gemm_phase<TH, 2>([&](int i) { return img + (16 * (i * 4 + mt) + col) * (hp + 4) + 4 * q; },
                              [&](int i) { uab[i] = vec_tile(vec, i * 4 + mt, q); }, [&](int i) -> f32x4& { return uab[i]; },
                              [&](int) -> const f32x4(&)[TH] { return hl; }, [&](int) {});
            f32x4 duv[2], xn, gn;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float alpha = fast_exp(fmaf(0.5f, uab[0][t], log1m)) + mscale;
                const float beta = 0.5f * uab[1][t];
                const float ra = __builtin_amdgcn_rcpf(alpha);
                const float gv = gy[t] * ra;
                const float d_alpha = fmaf(-gv, y[t], ra);
                duv[0][t] = 0.5f * d_alpha * (alpha - mscale);
                duv[1][t] = -0.5f * gv;
                gn[t] = gv;
                xn[t] = fmaf(alpha, y[t], beta);
            }
            tile_st(xt, 4 * gq + mt, xn);
            tile_st(gt, 4 * gq + mt, gn);
            gemm_phase<2, TH>([&](int mo) { return imgT + (16 * mo + col) * (kWideSlice + 4) + 4 * q + 32 * mt; }, [&](int) {},
                              [&](int mo) -> f32x4& { return dh[mo]; }, [&](int) -> const f32x4(&)[2] { return duv; }, [&](int) {});
        }
    }
#pragma unroll
    for (int mo = 0; mo < TH; ++mo)
#pragma unroll
        for (int t = 0; t < 4; ++t) dh[mo][t] *= (1.f - hl[mo][t] * hl[mo][t]);
    if constexpr (NHL > 1) {
        __syncthreads();   // after the last group, which read both images
        float *img, *vec;
        wide_next(c, img, vec);
        stage_any<SB>(img, hp + 4, L.WhT, hp, hp, log2i(hp / 4), hp, hp, 0, false, 0, false);
        __syncthreads();
        gemm_phase<TH, TH>([&](int mo) { return img + (16 * mo + col) * (hp + 4) + 4 * q; },
                           [&](int mo) {
#pragma unroll
                               for (int t = 0; t < 4; ++t) hl[mo][t] = 0.f;
                           },
                           [&](int mo) -> f32x4& { return hl[mo]; }, [&](int) -> const f32x4(&)[TH] { return dh; }, [&](int) {});
#pragma unroll
        for (int mo = 0; mo < TH; ++mo)
#pragma unroll
            for (int t = 0; t < 4; ++t) dh[mo][t] = hl[mo][t] * (1.f - h1[mo][t] * h1[mo][t]);
    }
    float* gsrc = c.gs + 16 * (REV ? c.D2 / 16 : 0);
    for (int ks = 0; ks < c.nslice; ++ks) {
        if (NHL == 1 && ks == 0) __syncthreads();   // after the last group, which read both images
        float *img, *vec;
        wide_next(c, img, vec);
        stage_any<SB>(img, hp + 4, L.W1T, hp, kWideSlice, log2i(hp / 4), c.D2 - kWideSlice * ks, hp,
                  REV ? c.D2 - 1 - kWideSlice * ks : kWideSlice * ks, REV, 0, false);
        __syncthreads();
        f32x4 acc2[2];
        gemm_phase<TH, 8>([&](int ms) { return img + (16 * ms + col) * (hp + 4) + 4 * q; },
                          [&](int ms) {
                              if (8 * ks + ms < c.TS) acc2[ms & 1] = tile_ld(gsrc, 8 * ks + ms);
                          },
                          [&](int ms) -> f32x4& { return acc2[ms & 1]; }, [&](int) -> const f32x4(&)[TH] { return dh; },
                          [&](int ms) {
                              if (8 * ks + ms < c.TS) tile_st(gsrc, 8 * ks + ms, acc2[ms & 1]);
                          });
    }
}

// one ElementwiseAffine layer on every tile: inverse (x - shift) exp(-ls) [returns the sum of -ls], forward
// x exp(ls) + shift [the sum of ls], or the reverse sweep's forward with the gradient scaled by exp(-ls)
enum { kEaInverse = 0, kEaBackward = 1, kEaForward = 2 };
template <int MODE>
__device__ __forceinline__ float wide_ea(const WideCtx& c, const float* __restrict__ g_ls, const float* __restrict__ g_sh, bool rev) {
    constexpr bool BACKWARD = MODE == kEaBackward;
    float ld = 0.f;
    wide_tiles(c.d / 16, [&](int m) { return tile_ld(c.xs, m); },
               [&](int m, f32x4 x) {
                   const f32x4 ls = rev ? vec_tile_rev(g_ls, m, c.q, c.d) : vec_tile(g_ls, m, c.q);
                   const f32x4 sh = rev ? vec_tile_rev(g_sh, m, c.q, c.d) : vec_tile(g_sh, m, c.q);
                   if constexpr (BACKWARD) {
                       f32x4 g = tile_ld(c.gs, m);
#pragma unroll
                       for (int t = 0; t < 4; ++t) {
                           g[t] *= fast_exp(-ls[t]);
                           x[t] = fmaf(fast_exp(ls[t]), x[t], sh[t]);
                       }
                       tile_st(c.gs, m, g);
                   } else if constexpr (MODE == kEaForward) {
#pragma unroll
                       for (int t = 0; t < 4; ++t) {
                           x[t] = fmaf(fast_exp(ls[t]), x[t], sh[t]);
                           ld += ls[t];
                       }
                   } else {
#pragma unroll
                       for (int t = 0; t < 4; ++t) {
                           x[t] = (x[t] - sh[t]) * fast_exp(-ls[t]);
                           ld -= ls[t];
                       }
                   }
                   tile_st(c.xs, m, x);
               });
    return ld;
}

// closed-form potential and its gradient on the streamed state (potential_value_grad_c's arithmetic, tile by tile)
__device__ __forceinline__ float wide_potential_grad(const WideCtx& c, const NfmcPotential& p, int lane) {
    const int TD = c.d / 16;
    auto xtile = [&](int m) { return tile_ld(c.xs, m); };
    if (p.kind == NFMC_POT_FUNNEL) {
        const f32x4 t0 = tile_ld(c.xs, 0);
        const float x0 = __shfl(t0[0], lane & 15, kWave);   // coordinate 0 = tile 0, register 0, lane group 0
        float s = 0.f;
        wide_tiles(TD, xtile, [&](int m, const f32x4& x) {
#pragma unroll
            for (int t = 0; t < 4; ++t) s = fmaf(x[t], (m == 0 && t == 0 && c.q == 0) ? 0.f : x[t], s);
        });
        s = chain_sum(s);
        const float inv_s2 = 1.f / (p.a_scalar * p.a_scalar);
        const float e = fast_exp(-x0);
        const float hd = 0.5f * (float)(c.d - 1);
        wide_tiles(TD, xtile, [&](int m, const f32x4& x) {
            f32x4 g;
#pragma unroll
            for (int t = 0; t < 4; ++t) g[t] = x[t] * e;
            if (m == 0 && c.q == 0) g[0] = x0 * inv_s2 - 0.5f * e * s + hd;
            tile_st(c.gs, m, g);
        });
        return 0.5f * x0 * x0 * inv_s2 + 0.5f * e * s + hd * x0;
    }
    float u = 0.f;
    wide_tiles(TD, xtile, [&](int m, const f32x4& x) {
        f32x4 a, b;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = p.a_scalar;
            b[t] = p.b_scalar;
        }
        if (p.a) a = vec_tile(p.a, m, c.q);
        if (p.b) b = vec_tile(p.b, m, c.q);
        f32x4 g;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float dlt = x[t] - b[t];
            u = fmaf(a[t] * dlt, dlt, u);
            g[t] = 2.f * a[t] * dlt;
        }
        tile_st(c.gs, m, g);
    });
    return chain_sum(u);
}

// slab: (grid * 128) private rows of d floats for the state, then as many for the gradient, one row per lane group of a
// workgroup slot (lanes past n in the last tile run on a copy of row n - 1 and are never read)
template <int TH, int NHL>
__global__ void __launch_bounds__(kMfmaBlock, 2) neutra_grad_wide_kernel(NfmcRealNVP f, NfmcPotential pot, const float* __restrict__ z,
                                                                      int64_t n, float* __restrict__ u_out, float* __restrict__ grad_out,
                                                                      float* __restrict__ slab, int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int hp = 16 * TH;
    constexpr int kSB = kWideGradSB;
    const int d = f.d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool rev_last = (f.n_coupling & 1) != 0;
    WideCtx c;
    wide_ctx_init(c, lds, f, lane);
    // the lane group's private slab row belongs to the workgroup SLOT, not to the chain: the next chain tile of the grid-stride
    // loop reuses it (a tile's results have left for the output arrays by then), so the slab is bounded by the grid
    const int64_t srow = (int64_t)blockIdx.x * kMfmaChains + wave * 16 + c.col;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row = tile * kMfmaChains + wave * 16 + c.col;
        const bool active = row < n;
        const float* zr = z + (active ? row : n - 1) * d;
        c.xs = slab + srow * d + 4 * c.q;
        c.gs = slab + ((int64_t)gridDim.x * kMfmaChains + srow) * d + 4 * c.q;
        // z at tile positions in latent order (flows with an odd number of reversals: position p holds logical d - 1 - p)
        wide_tiles(d / 16, [&](int m) { return rev_last ? vec_tile_rev(zr, m, c.q, d) : vec_tile(zr, m, c.q); },
                   [&](int m, const f32x4& v) { tile_st(c.xs, m, v); });
        float ldp = wide_ea<kEaInverse>(c, f.ea1_log_scale, f.ea1_shift, rev_last);
        for (int l = f.n_coupling - 1; l >= 0; --l)
            ldp += wide_coupling<TH, NHL, true, kSB>(c, mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL), (l & 1) == 0);
        ldp += wide_ea<kEaInverse>(c, f.ea0_log_scale, f.ea0_shift, false);
        const float u = wide_potential_grad(c, pot, lane);
        wide_ea<kEaBackward>(c, f.ea0_log_scale, f.ea0_shift, false);
        for (int l = 0; l < f.n_coupling; ++l)
            wide_coupling_backward<TH, NHL, kSB>(c, mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL), (l & 1) == 0);
        wide_ea<kEaBackward>(c, f.ea1_log_scale, f.ea1_shift, rev_last);
        const float ut = u - chain_sum(ldp);
        if (active) {
            if (u_out && c.q == 0) u_out[row] = ut;
            if (grad_out) wide_store_row(c.gs, grad_out + row * d, d, c.q, rev_last);
        }
    }
}

// x -> z, logdet_forward, log_prob (realnvp_forward_mfma_kernel's contract); slab: grid * 128 private rows of d floats
template <int TH, int NHL>
__global__ void __launch_bounds__(kMfmaBlock, 2) realnvp_forward_wide_kernel(NfmcRealNVP f, const float* __restrict__ x, int64_t n,
                                                                          float* __restrict__ z, float* __restrict__ logdet,
                                                                          float* __restrict__ log_prob, float* __restrict__ slab,
                                                                          int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int hp = 16 * TH;
    const int d = f.d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool rev_last = (f.n_coupling & 1) != 0;
    WideCtx c;
    wide_ctx_init(c, lds, f, lane);
    // the lane group's private slab row belongs to the workgroup SLOT, not to the chain: the next chain tile of the grid-stride
    // loop reuses it (a tile's results have left for the output arrays by then), so the slab is bounded by the grid
    const int64_t srow = (int64_t)blockIdx.x * kMfmaChains + wave * 16 + c.col;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row = tile * kMfmaChains + wave * 16 + c.col;
        const bool active = row < n;
        const float* xr = x + (active ? row : n - 1) * d;
        c.xs = slab + srow * d + 4 * c.q;
        wide_tiles(d / 16, [&](int m) { return vec_tile(xr, m, c.q); }, [&](int m, const f32x4& v) { tile_st(c.xs, m, v); });
        float ldp = wide_ea<kEaForward>(c, f.ea0_log_scale, f.ea0_shift, false);
        for (int l = 0; l < f.n_coupling; ++l)
            ldp += wide_coupling<TH, NHL, false, kWideFlowSB>(c, mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL), (l & 1) == 0);
        ldp += wide_ea<kEaForward>(c, f.ea1_log_scale, f.ea1_shift, rev_last);
        const float ld = chain_sum(ldp);
        const float ss = wide_sum_squares(c);
        if (active) {
            if (c.q == 0) {
                if (logdet) logdet[row] = ld;
                if (log_prob) log_prob[row] = -0.5f * ss - 0.5f * (float)d * kLog2Pi + ld;
            }
            if (z) wide_store_row(c.xs, z + row * d, d, c.q, rev_last);
        }
    }
}

// z (given, or drawn from the chain's latent stream) -> x, logdet_inverse, log q(x) (realnvp_inverse_mfma_kernel's contract)
template <int TH, int NHL>
__global__ void __launch_bounds__(kMfmaBlock, 2) realnvp_inverse_wide_kernel(NfmcRealNVP f, const float* __restrict__ z, int64_t n,
                                                                          float* __restrict__ x, float* __restrict__ logdet,
                                                                          float* __restrict__ log_q, NfmcRng rng, float* __restrict__ slab,
                                                                          int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int hp = 16 * TH;
    const int d = f.d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool rev_last = (f.n_coupling & 1) != 0;
    WideCtx c;
    wide_ctx_init(c, lds, f, lane);
    // the lane group's private slab row belongs to the workgroup SLOT, not to the chain: the next chain tile of the grid-stride
    // loop reuses it (a tile's results have left for the output arrays by then), so the slab is bounded by the grid
    const int64_t srow = (int64_t)blockIdx.x * kMfmaChains + wave * 16 + c.col;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row = tile * kMfmaChains + wave * 16 + c.col;
        const bool active = row < n;
        const int64_t rrow = active ? row : n - 1;
        c.xs = slab + srow * d + 4 * c.q;
        if (z) {
            const float* zr = z + rrow * d;
            wide_tiles(d / 16, [&](int m) { return rev_last ? vec_tile_rev(zr, m, c.q, d) : vec_tile(zr, m, c.q); },
                       [&](int m, const f32x4& v) { tile_st(c.xs, m, v); });
        } else {   // Philox stream kTagLatent, one block per 4 consecutive logical coordinates (draw_latent_c's rule)
            const uint32_t gchain = (uint32_t)(rng.chain_offset + (uint64_t)rrow);
            for (int m = 0; m < d / 16; ++m) {
                const int p0 = 16 * m + 4 * c.q;
                const int blk = rev_last ? (d - 4 - p0) >> 2 : p0 >> 2;
                float w[4];
                philox_normal4(gchain, rng.step0, (uint32_t)blk, kTagLatent, (uint32_t)rng.seed, (uint32_t)(rng.seed >> 32), w);
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = rev_last ? w[3 - j] : w[j];
                tile_st(c.xs, m, v);
            }
        }
        const float ss = wide_sum_squares(c);
        float ldp = wide_ea<kEaInverse>(c, f.ea1_log_scale, f.ea1_shift, rev_last);
        for (int l = f.n_coupling - 1; l >= 0; --l)
            ldp += wide_coupling<TH, NHL, true, kWideFlowSB>(c, mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL), (l & 1) == 0);
        ldp += wide_ea<kEaInverse>(c, f.ea0_log_scale, f.ea0_shift, false);
        const float ld = chain_sum(ldp);
        if (active) {
            if (c.q == 0) {
                if (logdet) logdet[row] = ld;
                if (log_q) log_q[row] = -0.5f * ss - 0.5f * (float)d * kLog2Pi - ld;
            }
            if (x) wide_store_row(c.xs, x + row * d, d, c.q, false);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Round 4: ONE launch per NeuTra-HMC transition at these shapes too (neutra_leapfrog_mfma_kernel's structure on the streamed
// state).  A workgroup slot's slab holds THREE private rows per lane group -- position, gradient, momentum -- in tile-position
// order; a chain tile's whole trajectory runs on them: momentum draw + H0 + the first half / full step, n_leapfrog times
// {gradient of the adjusted potential on the slab (the body of neutra_grad_wide_kernel), the steps between two gradients},
// the Hamiltonian test, the masked update of (z, grad, U~) in the caller's arrays, the kept row and the statistics.  What
// the composed version (round 3: a gradient launch + an elementwise launch per leapfrog step, 2 L + 3 launches per
// transition) paid for between the gradients -- launch gaps, the ramp and tail of 2 L + 3 grids, z and the gradient through
// (n, d) arrays -- is gone; H0 and the trajectory's U~ never leave registers.
struct WideLeapArgs {
    NfmcNeutraHmcArgs a;
    float *gz, *uz;       // caller scratch: gradient and U~ at the current state (n, d) / (n)
    float* slab;          // 3 x grid x 128 rows of d floats
    int step;             // transition index within this call
    float* sample_row;    // store row this transition is kept in, or NULL
};

template <int TH, int NHL>
__global__ void __launch_bounds__(kMfmaBlock, 2) neutra_leapfrog_wide_kernel(WideLeapArgs A, int64_t tiles, int dp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int hp = 16 * TH;
    constexpr int kSB = kWideLeapSB;
    const NfmcNeutraHmcArgs& a = A.a;
    const NfmcRealNVP& f = a.flow;
    const int d = f.d, TD = f.d / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool rev = (f.n_coupling & 1) != 0;
    const int64_t n = a.n;
    const float h = a.step_size, hh = a.step_size / 2;
    const int s = A.step;
    WideCtx c;
    wide_ctx_init(c, lds, f, lane);
    const int64_t srow = (int64_t)blockIdx.x * kMfmaChains + wave * 16 + c.col;
    const int64_t slot_rows = (int64_t)gridDim.x * kMfmaChains;
    c.xs = A.slab + srow * d + 4 * c.q;
    c.gs = A.slab + (slot_rows + srow) * d + 4 * c.q;
    float* const ps = A.slab + (2 * slot_rows + srow) * d + 4 * c.q;
    double* const red = reinterpret_cast<double*>(lds);   // per-tile statistics: [8 waves][2 d] doubles over the (idle) weight images
    double* const out = a.stats.sum_x ? a.stats.scratch + (size_t)blockIdx.x * (2 * dp + kStatTail) : nullptr;
    auto mass_tile = [&](int m) {
        f32x4 one;
        one[0] = one[1] = one[2] = one[3] = 1.f;
        if (!a.inv_mass_diag) return one;
        return rev ? vec_tile_rev(a.inv_mass_diag, m, c.q, d) : vec_tile(a.inv_mass_diag, m, c.q);
    };
    uint32_t n_acc = 0, n_bad = 0;
    bool first = true;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row = tile * kMfmaChains + wave * 16 + c.col;
        const bool active = row < n;
        const int64_t rrow = active ? row : n - 1;
        const float* zr = a.z + rrow * d;
        const float* gr = A.gz + rrow * d;
        const uint32_t gchain = (uint32_t)(a.rng.chain_offset + (uint64_t)rrow);
        // ---- start of the trajectory: state and gradient into the slab (tile positions in latent order), momentum draw,
        // H0, the first half step and the first position (hmc.py:100-106, 68-70)
        float kin = 0.f;
        for (int m = 0; m < TD; ++m) {
            const int p0 = 16 * m + 4 * c.q;
            f32x4 x = rev ? vec_tile_rev(zr, m, c.q, d) : vec_tile(zr, m, c.q);
            const f32x4 g = rev ? vec_tile_rev(gr, m, c.q, d) : vec_tile(gr, m, c.q);
            float zz[4];
            if (a.rng.replay_normals) {
                const float* src = a.rng.replay_normals + ((int64_t)s * n + rrow) * d;
#pragma unroll
                for (int j = 0; j < 4; ++j) zz[j] = src[rev ? d - 1 - (p0 + j) : p0 + j];
            } else {
                const int blk = rev ? (d - 4 - p0) >> 2 : p0 >> 2;
                float w4[4];
                philox_normal4(gchain, a.rng.step0 + (uint32_t)s, (uint32_t)blk, kTagNoise, (uint32_t)a.rng.seed, (uint32_t)(a.rng.seed >> 32), w4);
#pragma unroll
                for (int j = 0; j < 4; ++j) zz[j] = rev ? w4[3 - j] : w4[j];
            }
            const f32x4 mass = mass_tile(m);
            f32x4 p;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float mm = mass[j];
                float v = zz[j] * (1.f / sqrtf(mm));   // hmc.py:100
                kin = fmaf(v * v, mm, kin);
                v = fmaf(-hh, g[j], v);                 // hmc.py:68-70
                p[j] = v;
                x[j] = fmaf(h, v * mm, x[j]);
            }
            tile_st(c.xs, m, x);
            tile_st(ps, m, p);
        }
        kin = chain_sum(kin);
        const float h0 = A.uz[rrow] + 0.5f * kin;   // hmc.py:103-106
        float u = 0.f;
        for (int l = 0; l < a.n_leapfrog; ++l) {
            // ---- U~ and its gradient at the position in the slab (neutra_grad_wide_kernel's body; the position is rebuilt)
            float ldp = wide_ea<kEaInverse>(c, f.ea1_log_scale, f.ea1_shift, rev);
            for (int ll = f.n_coupling - 1; ll >= 0; --ll)
                ldp += wide_coupling<TH, NHL, true, kSB>(c, mfma_layer(f.weights + ll * f.layer_stride, d, hp, NHL), (ll & 1) == 0);
            ldp += wide_ea<kEaInverse>(c, f.ea0_log_scale, f.ea0_shift, false);
            const float ux = wide_potential_grad(c, a.pot, lane);
            wide_ea<kEaBackward>(c, f.ea0_log_scale, f.ea0_shift, false);
            for (int ll = 0; ll < f.n_coupling; ++ll)
                wide_coupling_backward<TH, NHL, kSB>(c, mfma_layer(f.weights + ll * f.layer_stride, d, hp, NHL), (ll & 1) == 0);
            wide_ea<kEaBackward>(c, f.ea1_log_scale, f.ea1_shift, rev);
            u = ux - chain_sum(ldp);
            // ---- the closing half step and, unless this was the last, the next step's opening half and position
            const bool last = l + 1 == a.n_leapfrog;
            for (int m0 = 0; m0 < TD; m0 += 4) {
                f32x4 g4[4], p4[4], x4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (m0 + j < TD) {
                        g4[j] = tile_ld(c.gs, m0 + j);
                        p4[j] = tile_ld(ps, m0 + j);
                        if (!last) x4[j] = tile_ld(c.xs, m0 + j);
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (m0 + j < TD) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) p4[j][t] = fmaf(-hh, g4[j][t], p4[j][t]);   // hmc.py:71
                        if (!last) {
                            const f32x4 mass = mass_tile(m0 + j);
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                p4[j][t] = fmaf(-hh, g4[j][t], p4[j][t]);
                                x4[j][t] = fmaf(h, p4[j][t] * mass[t], x4[j][t]);
                            }
                            tile_st(c.xs, m0 + j, x4[j]);
                        }
                        tile_st(ps, m0 + j, p4[j]);
                    }
            }
        }
        // ---- end of the trajectory: Hamiltonian test, masked update, kept row, statistics
        bool accept = true;
        float lr = 0.f;
        if (a.adjust) {
            float k2 = 0.f;
            wide_tiles(TD, [&](int m) { return tile_ld(ps, m); },
                       [&](int m, const f32x4& p) {
                           const f32x4 mass = mass_tile(m);
#pragma unroll
                           for (int t = 0; t < 4; ++t) k2 = fmaf(p[t] * p[t], mass[t], k2);
                       });
            k2 = chain_sum(k2);
            lr = h0 - (u + 0.5f * k2);   // hmc.py:107-111
            float uni;
            if (a.rng.replay_uniforms) {
                uni = a.rng.replay_uniforms[(int64_t)s * n + rrow];
            } else {
                const uint32_t step = a.rng.step0 + (uint32_t)s;
                const uint4 r = philox4x32_10(gchain, step >> 2, 0u, kTagAccept, (uint32_t)a.rng.seed, (uint32_t)(a.rng.seed >> 32));
                uni = u32_to_uniform(pick_word(r, step & 3u));
            }
            accept = fast_ln(uni) < lr;   // hmc.py:112-113
            if (active && c.q == 0 && !(fabsf(lr) <= 3.0e38f)) n_bad++;
        }
        accept = accept && active;
        if (active && c.q == 0) {
            if (accept) {
                A.uz[row] = u;
                n_acc++;
            }
            if (a.masks_out) a.masks_out[(int64_t)s * n + row] = accept ? 1 : 0;
            if (a.log_ratio_out) a.log_ratio_out[(int64_t)s * n + row] = lr;
        }
        __syncthreads();   // the weight images are idle: their memory takes this tile's statistics
        for (int m = 0; m < TD; ++m) {
            const int p0 = 16 * m + 4 * c.q;
            f32x4 zc;
            if (accept) {
                zc = tile_ld(c.xs, m);
                const f32x4 gc = tile_ld(c.gs, m);
                if (!rev) {
                    *reinterpret_cast<f32x4*>(a.z + row * d + p0) = zc;
                    *reinterpret_cast<f32x4*>(A.gz + row * d + p0) = gc;
                } else {
                    *reinterpret_cast<f32x4*>(a.z + row * d + (d - 4 - p0)) = rev4(zc);
                    *reinterpret_cast<f32x4*>(A.gz + row * d + (d - 4 - p0)) = rev4(gc);
                }
            } else {
                zc = rev ? vec_tile_rev(zr, m, c.q, d) : vec_tile(zr, m, c.q);
            }
            if (active && A.sample_row) {
                if (!rev) *reinterpret_cast<f32x4*>(A.sample_row + row * d + p0) = zc;
                else *reinterpret_cast<f32x4*>(A.sample_row + row * d + (d - 4 - p0)) = rev4(zc);
            }
            if (a.stats.sum_x) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float zv = active ? zc[t] : 0.f;
                    double v1 = (double)zv, v2 = (double)zv * (double)zv;
                    for (int k = 1; k < 16; k <<= 1) {
                        v1 += __shfl_xor(v1, k, kWave);
                        v2 += __shfl_xor(v2, k, kWave);
                    }
                    if (c.col == 0) {
                        const int pos = p0 + t;
                        const int cc = rev ? d - 1 - pos : pos;  // logical latent coordinate
                        red[wave * 2 * d + cc] = v1;
                        red[wave * 2 * d + d + cc] = v2;
                    }
                }
            }
        }
        if (a.stats.sum_x) {
            __syncthreads();
            for (int i = threadIdx.x; i < 2 * dp; i += kMfmaBlock) {
                const int half = i / dp, cc = i - half * dp;
                double v = 0.0;
                if (cc < d)
                    for (int w8 = 0; w8 < kMfmaWaves; ++w8) v += red[w8 * 2 * d + half * d + cc];
                out[i] = first ? v : out[i] + v;
            }
        }
        __syncthreads();   // the images are free again for the next tile's stagings
        first = false;
    }
    if (a.stats.sum_x) {
        for (int m = 1; m < 16; m <<= 1) {   // counted on lane group 0 only
            n_acc += __shfl_xor(n_acc, m, kWave);
            n_bad += __shfl_xor(n_bad, m, kWave);
        }
        __syncthreads();
        if (lane == 0) {
            red[2 * wave] = (double)n_acc;
            red[2 * wave + 1] = (double)n_bad;
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            double v = 0.0;
            for (int w8 = 0; w8 < kMfmaWaves; ++w8) v += red[2 * w8 + threadIdx.x];
            out[2 * dp + threadIdx.x] = v;
        }
        if (first)   // a workgroup without a tile (never launched: grid <= tiles): keep the slab well defined
            for (int i = threadIdx.x; i < 2 * dp; i += kMfmaBlock) out[i] = 0.0;
        if (threadIdx.x >= 2 && threadIdx.x < kStatTail) out[2 * dp + threadIdx.x] = 0.0;
    }
}


// ------------------------------------------------------------------------------------------------
// Round 4: the flow-proposal Metropolis step (the jump of jump.py:205-243, the loop body of imh.py:214-249) at these shapes
// in ONE launch (flow_mh_mfma_kernel's contract on the streamed state).  Until now a wide conditioner at d = 256 / 512
// composed the transition from three launches (inverse pass, forward pass, accept / select).  The slab of a workgroup slot
// holds two private rows per lane group: the working row of the flow passes and the chain's current state.
__device__ __forceinline__ float wide_potential_value(const WideCtx& c, const float* row, const NfmcPotential& p, int lane) {
    const int TD = c.d / 16;
    auto xtile = [&](int m) { return tile_ld(row, m); };
    if (p.kind == NFMC_POT_FUNNEL) {
        const f32x4 t0 = tile_ld(row, 0);
        const float x0 = __shfl(t0[0], lane & 15, kWave);   // coordinate 0 = tile 0, register 0, lane group 0
        float s = 0.f;
        wide_tiles(TD, xtile, [&](int m, const f32x4& x) {
#pragma unroll
            for (int t = 0; t < 4; ++t) s = fmaf(x[t], (m == 0 && t == 0 && c.q == 0) ? 0.f : x[t], s);
        });
        s = chain_sum(s);
        const float inv_s2 = 1.f / (p.a_scalar * p.a_scalar);
        return 0.5f * x0 * x0 * inv_s2 + 0.5f * fast_exp(-x0) * s + 0.5f * (float)(c.d - 1) * x0;
    }
    float u = 0.f;
    wide_tiles(TD, xtile, [&](int m, const f32x4& x) {
        f32x4 a, b;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = p.a_scalar;
            b[t] = p.b_scalar;
        }
        if (p.a) a = vec_tile(p.a, m, c.q);
        if (p.b) b = vec_tile(p.b, m, c.q);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float dlt = x[t] - b[t];
            u = fmaf(a[t] * dlt, dlt, u);
        }
    });
    return chain_sum(u);
}

template <int TH, int NHL>
__global__ void __launch_bounds__(kMfmaBlock, 2) flow_mh_wide_kernel(NfmcFlowMhArgs a, float* __restrict__ slab, int64_t tiles, int dp) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int hp = 16 * TH;
    const NfmcRealNVP& f = a.flow;
    const int d = f.d, TD = f.d / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool rev = (f.n_coupling & 1) != 0;
    const int64_t n = a.n;
    const float base_c = -0.5f * (float)d * kLog2Pi;
    WideCtx c;
    wide_ctx_init(c, lds, f, lane);
    const int64_t srow = (int64_t)blockIdx.x * kMfmaChains + wave * 16 + c.col;
    c.xs = slab + srow * d + 4 * c.q;                                                   // working row of the flow passes
    float* const cs = slab + ((int64_t)gridDim.x * kMfmaChains + srow) * d + 4 * c.q;  // the chain's current state
    double* const red = reinterpret_cast<double*>(lds);   // statistics of one transition: [8 waves][2 d] doubles over the idle images
    const bool defer = a.stats.defer != 0;   // deferred: add to the caller-zeroed slab (nfmc_stats_fold_f32)
    const int slot = defer ? a.stats.tail_slot : 0;
    double* const out = a.stats.sum_x ? a.stats.scratch + (size_t)blockIdx.x * (2 * dp + kStatTail) : nullptr;
    uint32_t n_acc = 0, n_bad = 0;
    bool first = !defer;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row = tile * kMfmaChains + wave * 16 + c.col;
        const bool active = row < n;
        const int64_t rrow = active ? row : n - 1;
        const float* xr = a.x + rrow * d;
        wide_tiles(TD, [&](int m) { return vec_tile(xr, m, c.q); },
                   [&](int m, const f32x4& v) {
                       tile_st(cs, m, v);
                       tile_st(c.xs, m, v);
                   });
        float u_x = wide_potential_value(c, cs, a.pot, lane);                       // jump.py:212 / imh.py:224
        float f_x;
        if (a.logq_cached) {
            f_x = a.logq[rrow];
        } else {                                                                    // flow.log_prob(x): jump.py:218 / imh.py:214
            float ldp = wide_ea<kEaForward>(c, f.ea0_log_scale, f.ea0_shift, false);
            for (int l = 0; l < f.n_coupling; ++l)
                ldp += wide_coupling<TH, NHL, false, kWideFlowSB>(c, mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL), (l & 1) == 0);
            ldp += wide_ea<kEaForward>(c, f.ea1_log_scale, f.ea1_shift, rev);
            f_x = -0.5f * wide_sum_squares(c) + base_c + chain_sum(ldp);
        }
        StoreCursor keep(a.samples);
        const uint32_t gchain = (uint32_t)(a.rng.chain_offset + (uint64_t)rrow);
        for (int s = 0; s < a.n_steps; ++s) {
            // ---- flow.sample: the latent (Philox stream kTagLatent, one block per 4 logical coordinates; or replayed), x' = f^-1(z)
            float ss = 0.f;
            for (int m = 0; m < TD; ++m) {
                const int p0 = 16 * m + 4 * c.q;
                f32x4 v;
                if (a.rng.replay_normals) {
                    const float* src = a.rng.replay_normals + ((int64_t)s * n + rrow) * d;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = src[rev ? d - 1 - (p0 + j) : p0 + j];
                } else {
                    const int blk = rev ? (d - 4 - p0) >> 2 : p0 >> 2;
                    float w4[4];
                    philox_normal4(gchain, a.rng.step0 + (uint32_t)s, (uint32_t)blk, kTagLatent, (uint32_t)a.rng.seed,
                                   (uint32_t)(a.rng.seed >> 32), w4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = rev ? w4[3 - j] : w4[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) ss = fmaf(v[j], v[j], ss);
                tile_st(c.xs, m, v);
            }
            ss = chain_sum(ss);
            float ldp = wide_ea<kEaInverse>(c, f.ea1_log_scale, f.ea1_shift, rev);
            for (int l = f.n_coupling - 1; l >= 0; --l)
                ldp += wide_coupling<TH, NHL, true, kWideFlowSB>(c, mfma_layer(f.weights + l * f.layer_stride, d, hp, NHL), (l & 1) == 0);
            ldp += wide_ea<kEaInverse>(c, f.ea0_log_scale, f.ea0_shift, false);
            const float f_xp = -0.5f * ss + base_c - chain_sum(ldp);
            const float u_xp = wide_potential_value(c, c.xs, a.pot, lane);            // jump.py:213 / imh.py:225
            const float lr = (-u_xp) - (-u_x) + f_x - f_xp;                           // util.py:392
            bool accept = true;
            if (a.adjusted) {
                float u;
                if (a.rng.replay_uniforms) {
                    u = a.rng.replay_uniforms[(int64_t)s * n + rrow];
                } else {
                    const uint4 r = philox4x32_10(gchain, a.rng.step0 + (uint32_t)s, 0u, kTagJump, (uint32_t)a.rng.seed, (uint32_t)(a.rng.seed >> 32));
                    u = u32_to_uniform(r.x);
                }
                accept = fast_ln(u) < lr;                                             // jump.py:225 / imh.py:229-230
                if (active && c.q == 0 && !(fabsf(lr) <= 3.0e38f)) n_bad++;
            }
            accept = accept && active;
            if (accept) {                                                             // jump.py:231 / imh.py:232-233
                f_x = f_xp;
                u_x = u_xp;
                if (c.q == 0) n_acc++;
            }
            float* kept = keep.next(n * (int64_t)d);
            if (active && c.q == 0) {
                if (a.masks_out) a.masks_out[(int64_t)s * n + row] = accept ? 1 : 0;
                if (a.log_ratio_out) a.log_ratio_out[(int64_t)s * n + row] = lr;
            }
            __syncthreads();   // the weight images are idle: their memory takes this transition's statistics
            for (int m = 0; m < TD; ++m) {
                f32x4 xc;
                if (accept) {
                    xc = tile_ld(c.xs, m);
                    tile_st(cs, m, xc);
                } else {
                    xc = tile_ld(cs, m);
                }
                if (active && kept) *reinterpret_cast<f32x4*>(kept + row * d + 16 * m + 4 * c.q) = xc;
                if (a.stats.sum_x) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float xv = active ? xc[t] : 0.f;
                        double v1 = (double)xv, v2 = (double)xv * (double)xv;
                        for (int k = 1; k < 16; k <<= 1) {
                            v1 += __shfl_xor(v1, k, kWave);
                            v2 += __shfl_xor(v2, k, kWave);
                        }
                        if (c.col == 0) {
                            const int cc = 16 * m + 4 * c.q + t;   // x-space: position = coordinate
                            red[wave * 2 * d + cc] = v1;
                            red[wave * 2 * d + d + cc] = v2;
                        }
                    }
                }
            }
            if (a.stats.sum_x) {
                __syncthreads();
                for (int i = threadIdx.x; i < 2 * dp; i += kMfmaBlock) {
                    const int hf = i / dp, cc = i - hf * dp;
                    double v = 0.0;
                    if (cc < d)
                        for (int w8 = 0; w8 < kMfmaWaves; ++w8) v += red[w8 * 2 * d + hf * d + cc];
                    out[i] = first ? v : out[i] + v;
                }
            }
            __syncthreads();   // the images are free again for the next staging
            first = false;
        }
        if (active) {
            float* xo = a.x + row * d;
            wide_tiles(TD, [&](int m) { return tile_ld(cs, m); },
                       [&](int m, const f32x4& v) { *reinterpret_cast<f32x4*>(xo + 16 * m + 4 * c.q) = v; });
            if (c.q == 0) a.logq[row] = f_x;
        }
    }
    if (a.stats.sum_x) {
        for (int m = 1; m < 16; m <<= 1) {   // counted on lane group 0 only
            n_acc += __shfl_xor(n_acc, m, kWave);
            n_bad += __shfl_xor(n_bad, m, kWave);
        }
        __syncthreads();
        if (lane == 0) {
            red[2 * wave] = (double)n_acc;
            red[2 * wave + 1] = (double)n_bad;
        }
        __syncthreads();
        if (threadIdx.x < kStatTail) {
            double v = 0.0;
            const int k = (int)threadIdx.x - slot;
            if (k == 0 || k == 1)
                for (int w8 = 0; w8 < kMfmaWaves; ++w8) v += red[2 * w8 + k];
            out[2 * dp + threadIdx.x] = defer ? out[2 * dp + threadIdx.x] + v : v;
        }
    }
}

}  // namespace nfmc

using namespace nfmc;

int nfmc::nfmc_mfma_wide_supported(int32_t d, int32_t n_hidden, int32_t n_hidden_layers) {
    // whole 16-coordinate tiles per half; d = 64 / 128 belong to the register-resident kernels (nfmc_mfma_supported)
    return d >= 32 && d <= 512 && d % 32 == 0 && d != 64 && d != 128 && n_hidden > 32 && n_hidden <= 128 && n_hidden_layers >= 1 &&
           n_hidden_layers <= 2;
}

// The scratch slab of a launch comes from the caller (NfmcRealNVP.scratch; SURVEY 8b: the library never allocates): one
// private row of d floats per lane group of a workgroup SLOT -- at most kCkMaxGrid slots whatever n -- times `copies`
// (1: state; 2: state and gradient).
static int wide_grid(int64_t n) {
    const int64_t tiles = (n + kMfmaChains - 1) / kMfmaChains;
    return (int)(tiles < kCkMaxGrid ? tiles : kCkMaxGrid);
}
int64_t nfmc::nfmc_wide_slab_floats(int64_t n, int32_t d, int32_t copies) {
    return n > 0 ? (int64_t)copies * wide_grid(n) * kMfmaChains * d : 0;
}
static int wide_slab(float** slab, const NfmcRealNVP* flow, int64_t n, int copies) {
    const int64_t need = nfmc_wide_slab_floats(n, flow->d, copies) * (int64_t)sizeof(float);
    if (!flow->scratch || flow->scratch_bytes < need || (((uintptr_t)flow->scratch) & 15u)) return NFMC_ESCRATCH;
    *slab = flow->scratch;
    return NFMC_OK;
}

#define NFMC_WIDE_DISPATCH(KERNEL, ...)                                                                                        \
    {                                                                                                                           \
        const int th = nfmc_realnvp_padded_hidden(flow->n_hidden) / 16, nhl = flow->n_hidden_layers;                            \
        hipError_t e = hipSuccess;                                                                                              \
        if (th == 4 && nhl == 1) NFMC_WIDE_LAUNCH(KERNEL, 4, 1, __VA_ARGS__)                                                    \
        else if (th == 4 && nhl == 2) NFMC_WIDE_LAUNCH(KERNEL, 4, 2, __VA_ARGS__)                                               \
        else if (th == 8 && nhl == 1) NFMC_WIDE_LAUNCH(KERNEL, 8, 1, __VA_ARGS__)                                               \
        else if (th == 8 && nhl == 2) NFMC_WIDE_LAUNCH(KERNEL, 8, 2, __VA_ARGS__)                                               \
        else rc = NFMC_EUNSUPPORTED;                                                                                            \
        if (rc == NFMC_OK && e != hipSuccess) rc = (int)e;                                                                      \
        if (rc == NFMC_OK && (e = hipGetLastError()) != hipSuccess) rc = (int)e;                                                \
    }
#define NFMC_WIDE_LAUNCH(KERNEL, THV, NHLV, ...)                                                                       \
    {                                                                                                                   \
        auto kern = KERNEL<THV, NHLV>;                                                                                  \
        /* per call: the attribute belongs to the (kernel, device) pair, and a process may drive several devices */     \
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideLdsBytes);     \
        if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3(grid), dim3(kMfmaBlock), kWideLdsBytes, st, __VA_ARGS__);    \
    }

int nfmc::nfmc_realnvp_forward_wide_f32(const NfmcRealNVP* flow, const float* x, int64_t n, float* z, float* logdet, float* log_prob,
                                             nfmc_stream_t stream) {
    if (!flow || !x || n <= 0) return NFMC_EINVAL;
    if (!nfmc_mfma_wide_supported(flow->d, flow->n_hidden, flow->n_hidden_layers)) return NFMC_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int64_t tiles = (n + kMfmaChains - 1) / kMfmaChains;
    const int grid = wide_grid(n);
    float* slab = nullptr;
    int rc = wide_slab(&slab, flow, n, 1);
    if (rc) return rc;
    NFMC_WIDE_DISPATCH(realnvp_forward_wide_kernel, *flow, x, n, z, logdet, log_prob, slab, tiles)
    return rc;
}

int nfmc::nfmc_realnvp_inverse_wide_f32(const NfmcRealNVP* flow, const float* z, int64_t n, float* x, float* logdet, float* log_q,
                                             const NfmcRng* rng, nfmc_stream_t stream) {
    if (!flow || n <= 0 || (!z && !rng)) return NFMC_EINVAL;
    if (!nfmc_mfma_wide_supported(flow->d, flow->n_hidden, flow->n_hidden_layers)) return NFMC_EUNSUPPORTED;
    NfmcRng r = {};
    if (rng) r = *rng;
    hipStream_t st = (hipStream_t)stream;
    const int64_t tiles = (n + kMfmaChains - 1) / kMfmaChains;
    const int grid = wide_grid(n);
    float* slab = nullptr;
    int rc = wide_slab(&slab, flow, n, 1);
    if (rc) return rc;
    NFMC_WIDE_DISPATCH(realnvp_inverse_wide_kernel, *flow, z, n, x, logdet, log_q, r, slab, tiles)
    return rc;
}

// the gradient kernel on a slab the caller holds (2 x grid x 128 x d floats)
static int grad_wide_launch(const NfmcRealNVP* flow, const NfmcPotential* pot, const float* z, int64_t n, float* u_out, float* grad_out,
                            float* slab, hipStream_t st) {
    const int64_t tiles = (n + kMfmaChains - 1) / kMfmaChains;
    const int grid = wide_grid(n);
    int rc = NFMC_OK;
    NFMC_WIDE_DISPATCH(neutra_grad_wide_kernel, *flow, *pot, z, n, u_out, grad_out, slab, tiles)
    return rc;
}

static int grad_wide_check(const NfmcRealNVP* flow, const NfmcPotential* pot, const float* z, int64_t n, const float* grad_out) {
    if (!flow || !pot || !z || n <= 0) return NFMC_EINVAL;
    if (!nfmc_mfma_wide_supported(flow->d, flow->n_hidden, flow->n_hidden_layers)) return NFMC_EUNSUPPORTED;
    if (pot->kind != NFMC_POT_QUADRATIC && pot->kind != NFMC_POT_FUNNEL) return NFMC_EUNSUPPORTED;
    if ((((uintptr_t)z) & 15u) || (((uintptr_t)grad_out) & 15u)) return NFMC_EUNSUPPORTED;   // 16-byte tile IO
    return NFMC_OK;
}

int nfmc::nfmc_neutra_potential_grad_wide_f32(const NfmcRealNVP* flow, const NfmcPotential* pot, const float* z, int64_t n,
                                                   float* u_out, float* grad_out, nfmc_stream_t stream) {
    int rc = grad_wide_check(flow, pot, z, n, grad_out);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* slab = nullptr;
    rc = wide_slab(&slab, flow, n, 2);   // state and gradient
    if (rc) return rc;
    return grad_wide_launch(flow, pot, z, n, u_out, grad_out, slab, st);
}

// scratch of the fused trajectory: the gradient at the state (n d floats), U~ at the state (n, rounded up to a multiple of
// 4 floats), then the slab: position, gradient and momentum rows of the workgroup slots
int64_t nfmc::nfmc_neutra_wide_scratch_floats(int64_t n, int32_t d) {
    return n * (int64_t)d + (n + 3) / 4 * 4 + nfmc_wide_slab_floats(n, d, 3);
}

int nfmc::nfmc_neutra_hmc_steps_wide_f32(const NfmcNeutraHmcArgs* args, float* scratch, int64_t scratch_bytes, nfmc_stream_t stream) {
    if (!args || !scratch) return NFMC_EINVAL;
    const NfmcNeutraHmcArgs& a = *args;
    const int d = a.flow.d;
    const int64_t n = a.n;
    if (!nfmc_mfma_wide_supported(d, a.flow.n_hidden, a.flow.n_hidden_layers)) return NFMC_EUNSUPPORTED;
    if (scratch_bytes < nfmc_neutra_wide_scratch_floats(n, d) * (int64_t)sizeof(float)) return NFMC_ESCRATCH;
    if ((((uintptr_t)a.z) | ((uintptr_t)scratch) | ((uintptr_t)a.inv_mass_diag) | ((uintptr_t)a.samples.base)) & 15u)
        return NFMC_EUNSUPPORTED;   // 16-byte tiles
    hipStream_t st = (hipStream_t)stream;
    WideLeapArgs A;
    A.a = a;
    A.gz = scratch;
    A.uz = A.gz + n * d;
    A.slab = A.uz + (n + 3) / 4 * 4;
    const int dp = padded_d(d);
    const int64_t tiles = (n + kMfmaChains - 1) / kMfmaChains;
    const int grid = wide_grid(n);
    if (a.stats.sum_x && a.stats.scratch_bytes < (int64_t)grid * (2 * dp + kStatTail) * (int64_t)sizeof(double)) return NFMC_ESCRATCH;
    int rc = grad_wide_check(&a.flow, &a.pot, a.z, n, A.gz);
    if (rc) return rc;
    // U~ and its gradient at the state (also after the caller changed z), on the first two copies of the slab
    rc = grad_wide_launch(&a.flow, &a.pot, a.z, n, A.uz, A.gz, A.slab, st);
    const NfmcRealNVP* flow = &a.flow;
    int countdown = a.samples.countdown, srow = a.samples.row;   // one launch per transition: the store cursor runs here
    for (int s = 0; s < a.n_steps && rc == NFMC_OK; ++s) {
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
        NFMC_WIDE_DISPATCH(neutra_leapfrog_wide_kernel, A, tiles, dp)
        if (rc == NFMC_OK && a.stats.sum_x)
            hipLaunchKernelGGL(stats_finish_kernel<true>, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, st, a.stats.scratch, grid, dp, d,
                               a.stats, (unsigned long long)n);
    }
    const hipError_t le = hipGetLastError();
    if (rc) return rc;
    return le == hipSuccess ? NFMC_OK : (int)le;
}

int nfmc::nfmc_flow_mh_steps_wide_f32(const NfmcFlowMhArgs& a, nfmc_stream_t stream, int* grid_out, int* dp_out) {
    const NfmcRealNVP* flow = &a.flow;
    const int d = a.flow.d;
    if (!nfmc_mfma_wide_supported(d, a.flow.n_hidden, a.flow.n_hidden_layers)) return NFMC_EUNSUPPORTED;
    if ((((uintptr_t)a.x) | ((uintptr_t)a.samples.base)) & 15u) return NFMC_EUNSUPPORTED;   // 16-byte tiles
    hipStream_t st = (hipStream_t)stream;
    const int64_t tiles = (a.n + kMfmaChains - 1) / kMfmaChains;
    const int grid = wide_grid(a.n), dp = padded_d(d);
    if (a.stats.sum_x && a.stats.scratch_bytes < (int64_t)grid * (2 * dp + kStatTail) * (int64_t)sizeof(double)) return NFMC_ESCRATCH;
    if (check_defer(a.stats, dp, d)) return NFMC_EINVAL;
    float* slab = nullptr;
    int rc = wide_slab(&slab, flow, a.n, 2);   // working row + current state: what nfmc_flow_scratch_bytes(.., with_gradient = 1) sizes
    if (rc) return rc;
    NFMC_WIDE_DISPATCH(flow_mh_wide_kernel, a, slab, tiles, dp)
    if (rc) return rc;
    *grid_out = grid;
    *dp_out = dp;
    return NFMC_OK;
}
