// RealNVP on the VALU path (conditioner width H <= 32): one chain per lane, the chain's row of the
// (64 x d) wave tile lives in LDS, conditioner weights are wave-uniform and arrive through the scalar
// cache (s_load + SGPR operands of v_fmac), so the MLP costs no LDS bandwidth and no cross-lane traffic.
//
// Spec: DESIGN.md "RealNVP spec" / oracle/flow.py.  Weight blob per coupling layer (logical,
// un-permuted coordinates; HP = n_hidden rounded up to a multiple of 4, padding rows/cols are zero):
//     W1T (d_a, HP) | b1 (HP) | [WhT (HP_in, HP_out) | bh (HP)] x (n_hl-1) | W3 (2 d_b, HP) | b3 (2 d_b)
// ReversePermutation is never materialised: layer l (0-based) sees logical coordinate j at physical
// position (l even ? d-1-j : j) because l+1 reversals precede it.
#pragma once

#include "common.hpp"

namespace nfmc {

constexpr float kLog2Pi = 1.8378770664093453f;

__device__ __forceinline__ float fast_tanh(float v) {
    // 1 - 2 / (e^{2v} + 1); exact limits at +-inf, abs err ~1e-7
    const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * v);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// LDS row stride (floats) for a d-wide tile: multiple of 4 with stride/4 odd, so the 16 lanes of a
// ds_read_b128 group fall on 16 distinct 4-bank slots.
__host__ __device__ inline int tile_stride(int d) {
    int s = (d + 3) & ~3;
    if (((s >> 2) & 1) == 0) s += 4;
    return s;
}

struct FlowGeom {
    int d, d_a, d_b, n_hl, n_coupling;
    int64_t layer_stride;
    float log1m, m;
    int n_bins;      // 0 = affine coupling, 8 = rational-quadratic spline coupling
    int out_rows;    // rows of the last conditioner layer: 2 d_b (affine) or (3 n_bins - 1) d_b (spline)
    float bound;     // spline bound B
};

__device__ __forceinline__ FlowGeom make_geom(const NfmcRealNVP& f) {
    FlowGeom g;
    g.d = f.d;
    g.d_a = f.d / 2;
    g.d_b = f.d - g.d_a;
    g.n_hl = f.n_hidden_layers;
    g.n_coupling = f.n_coupling;
    g.layer_stride = f.layer_stride;
    g.m = f.min_scale;
    g.log1m = __logf(1.f - f.min_scale);
    g.n_bins = f.n_bins;
    g.out_rows = f.n_bins > 0 ? (3 * f.n_bins - 1) * g.d_b : 2 * g.d_b;
    g.bound = f.spline_bound;
    return g;
}

// physical position of logical coordinate j in coupling layer with reversal flag `rev`
__device__ __forceinline__ int phys(int j, int d, bool rev) { return rev ? d - 1 - j : j; }

// Conditioner hidden stack: h = tanh(... tanh(W1 x_a + b1) ...) for this lane's chain.
template <int HP>
__device__ __forceinline__ void conditioner_hidden(const float* __restrict__ row, const float* __restrict__ W,
                                                   const FlowGeom& g, bool rev, float (&h)[HP]) {
    const float* b1 = W + (int64_t)g.d_a * HP;
#pragma unroll
    for (int k = 0; k < HP; ++k) h[k] = b1[k];
    for (int j = 0; j < g.d_a; ++j) {
        const float xj = row[phys(j, g.d, rev)];
        const float* w = W + (int64_t)j * HP;
#pragma unroll
        for (int k = 0; k < HP; ++k) h[k] = fmaf(w[k], xj, h[k]);
    }
#pragma unroll
    for (int k = 0; k < HP; ++k) h[k] = fast_tanh(h[k]);
    const float* Wh = b1 + HP;
    for (int l = 1; l < g.n_hl; ++l) {
        float t[HP];
        const float* bh = Wh + HP * HP;
#pragma unroll
        for (int k = 0; k < HP; ++k) t[k] = bh[k];
#pragma unroll
        for (int i = 0; i < HP; ++i) {
#pragma unroll
            for (int k = 0; k < HP; ++k) t[k] = fmaf(Wh[i * HP + k], h[i], t[k]);
        }
#pragma unroll
        for (int k = 0; k < HP; ++k) h[k] = fast_tanh(t[k]);
        Wh = bh + HP;
    }
}

__device__ __forceinline__ const float* w3_of(const float* __restrict__ W, const FlowGeom& g, int HP) {
    return W + (int64_t)g.d_a * HP + HP + (int64_t)(g.n_hl - 1) * (HP * HP + HP);
}

// ---- wide conditioners (HP = 64 / 128) on shapes the matrix-core kernels do not cover (any d, any depth):
// same one-chain-per-lane scheme, but the HP accumulators are the only register array; the activations of
// the previous layer sit in an LDS buffer hbuf[HP][64 lanes] (lane-contiguous, conflict free) and are
// streamed one per HP fused multiply-adds.  Blob layout of the wide path (mfma_device.hpp):
//   W1 (HP,d_a) | W1T (d_a,HP) | b1 | [Wh | WhT | bh] x (n_hl-1) | W3 (2d_b,HP) | W3T | b3
template <int HP>
__device__ __forceinline__ void conditioner_hidden_wide(const float* __restrict__ row, const float* __restrict__ W,
                                                        const FlowGeom& g, bool rev, float (&h)[HP],
                                                        float* __restrict__ hbuf, int lane) {
    const float* W1T = W + (int64_t)HP * g.d_a;
    const float* b1 = W1T + (int64_t)g.d_a * HP;
#pragma unroll
    for (int k = 0; k < HP; ++k) h[k] = b1[k];
    for (int j = 0; j < g.d_a; ++j) {
        const float xj = row[phys(j, g.d, rev)];
        const float* w = W1T + (int64_t)j * HP;
#pragma unroll
        for (int k = 0; k < HP; ++k) h[k] = fmaf(w[k], xj, h[k]);
    }
    const float* p = b1 + HP;
    for (int l = 1; l < g.n_hl; ++l) {
        const float* WhT = p + HP * HP;
        const float* bh = WhT + HP * HP;
#pragma unroll
        for (int k = 0; k < HP; ++k) hbuf[k * 64 + lane] = fast_tanh(h[k]);
#pragma unroll
        for (int k = 0; k < HP; ++k) h[k] = bh[k];
        for (int i = 0; i < HP; ++i) {
            const float hi = hbuf[i * 64 + lane];
            const float* w = WhT + i * HP;
#pragma unroll
            for (int k = 0; k < HP; ++k) h[k] = fmaf(w[k], hi, h[k]);
        }
        p = bh + HP;
    }
#pragma unroll
    for (int k = 0; k < HP; ++k) h[k] = fast_tanh(h[k]);
}

__device__ __forceinline__ const float* w3_of_wide(const float* __restrict__ W, const FlowGeom& g, int HP) {
    return W + (int64_t)2 * HP * g.d_a + HP + (int64_t)(g.n_hl - 1) * (2 * HP * HP + HP);
}

// One coupling layer applied in place to this lane's row.  Returns the layer's log|det| contribution
// (forward: +sum log alpha; inverse: -sum log alpha).
// ---- rational-quadratic spline with K = 8 bins on [-B, B] (oracle/flow.py: rqs_params / rqs_apply; Durkan et al.
// 2019), one coordinate.  `raw` = this coordinate's 3K - 1 conditioner outputs (widths | heights | derivatives).
// The bin is found by a scan that carries the bin's corner values (no indexed register arrays).
constexpr int kRqsBins = 8;
constexpr float kRqsMinBin = 1e-3f, kRqsMinDeriv = 1e-3f;

template <bool INVERSE>
__device__ __forceinline__ float rqs_coordinate(float v, const float (&raw)[3 * kRqsBins - 1], float B, float& ld) {
    constexpr int K = kRqsBins;
    float w[K], h[K];
    float mw = raw[0], mh = raw[K];
#pragma unroll
    for (int k = 1; k < K; ++k) {
        mw = fmaxf(mw, raw[k]);
        mh = fmaxf(mh, raw[K + k]);
    }
    float sw = 0.f, sh = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        w[k] = fast_exp(raw[k] - mw);
        h[k] = fast_exp(raw[K + k] - mh);
        sw += w[k];
        sh += h[k];
    }
    const float nw = (1.f - K * kRqsMinBin) * __builtin_amdgcn_rcpf(sw), nh = (1.f - K * kRqsMinBin) * __builtin_amdgcn_rcpf(sh);
    const bool inside = v >= -B && v <= B;
    const float vc = fminf(fmaxf(v, -B), B);
    // scan: knots cw_k / ch_k, derivatives d_k (d_0 = d_K = 1); keep the corners of the bin that holds vc
    float cw = -B, ch = -B, dk = 1.f;
    float x0 = -B, x1 = B, y0 = -B, y1 = B, d0 = 1.f, d1 = 1.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float cwn = k == K - 1 ? B : fmaf(2.f * B, fmaf(nw, w[k], kRqsMinBin), cw);
        const float chn = k == K - 1 ? B : fmaf(2.f * B, fmaf(nh, h[k], kRqsMinBin), ch);
        float dn = 1.f;
        if (k < K - 1) {
            const float u = raw[2 * K + k];   // softplus(u) = max(u, 0) + log(1 + exp(-|u|))
            dn = kRqsMinDeriv + fmaxf(u, 0.f) + fast_ln(1.f + fast_exp(-fabsf(u)));
        }
        const bool here = k == 0 || (INVERSE ? vc >= ch : vc >= cw);   // bins are visited in order: the last hit wins
        x0 = here ? cw : x0;
        x1 = here ? cwn : x1;
        y0 = here ? ch : y0;
        y1 = here ? chn : y1;
        d0 = here ? dk : d0;
        d1 = here ? dn : d1;
        cw = cwn;
        ch = chn;
        dk = dn;
    }
    const float bw = x1 - x0, bh = y1 - y0;
    const float s = bh * __builtin_amdgcn_rcpf(bw);
    const float dd = d0 + d1 - 2.f * s;
    float th, out;
    if (INVERSE) {
        const float dy = vc - y0;
        const float a = fmaf(dy, dd, bh * (s - d0));
        const float b = fmaf(-dy, dd, bh * d0);
        const float c = -s * dy;
        th = 2.f * c * __builtin_amdgcn_rcpf(-b - __builtin_amdgcn_sqrtf(fmaf(b, b, -4.f * a * c)));
        out = fmaf(th, bw, x0);
    } else {
        th = (vc - x0) * __builtin_amdgcn_rcpf(bw);
        const float t1 = th * (1.f - th);
        out = fmaf(bh * fmaf(s * th, th, d0 * t1), __builtin_amdgcn_rcpf(fmaf(dd, t1, s)), y0);
    }
    const float t1 = th * (1.f - th), om = 1.f - th;
    const float den = fmaf(dd, t1, s);
    const float l = fast_ln(s * s * fmaf(d1 * th, th, fmaf(2.f * s, t1, d0 * om * om))) - 2.f * fast_ln(den);
    ld += inside ? l : 0.f;
    return inside ? out : v;
}

// Reverse mode through one spline coordinate of an INVERSE coupling layer (NeuTra's reverse sweep).  The layer computed
// y = F^-1(v; theta) with F the forward spline, and contributed l = log F'(y; theta) to L = U(x) - logdet_inverse.
// Given y (the layer's OUTPUT), gy = dL/dy from downstream and the coordinate's raw conditioner outputs, returns
//     v  = F(y; theta)                                 the layer input, rebuilt with the forward map (no root finding)
//     q  = dL/dv = (gy + dl/dy) / F'(y)                implicit-function theorem on F(y; theta) = v
//     draw[j] = dL/draw_j = dl/dtheta - q dF/dtheta    chained through the bin's six knot values to the 3K - 1 raw
//                                                      outputs (softmax widths / heights, softplus derivatives)
// All partials are those of the FORWARD formulas of rqs_coordinate (hand-written adjoints of its expression graph).
__device__ __forceinline__ void rqs_inverse_backward(float y, float gy, const float (&raw)[3 * kRqsBins - 1], float B,
                                                     float (&draw)[3 * kRqsBins - 1], float& v_out, float& q_out) {
    constexpr int K = kRqsBins;
    constexpr float scale = 1.f - K * kRqsMinBin;
#pragma unroll
    for (int j = 0; j < 3 * K - 1; ++j) draw[j] = 0.f;
    if (!(y >= -B && y <= B)) {   // identity tails: dL/dv = dL/dy, no parameter gradient
        v_out = y;
        q_out = gy;
        return;
    }
    float smw[K], smh[K];
    float mw = raw[0], mh = raw[K];
#pragma unroll
    for (int k = 1; k < K; ++k) {
        mw = fmaxf(mw, raw[k]);
        mh = fmaxf(mh, raw[K + k]);
    }
    float sw = 0.f, sh = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        smw[k] = fast_exp(raw[k] - mw);
        smh[k] = fast_exp(raw[K + k] - mh);
        sw += smw[k];
        sh += smh[k];
    }
    const float rsw = __builtin_amdgcn_rcpf(sw), rsh = __builtin_amdgcn_rcpf(sh);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        smw[k] *= rsw;   // softmax
        smh[k] *= rsh;
    }
    // scan for the bin of y (forward direction: by the width knots), keeping its corners, its index and the
    // cumulative softmax mass below each of its two width / height knots
    float cw = -B, ch = -B, dk = 1.f, cmw = 0.f, cmh = 0.f;
    float x0 = -B, x1 = B, y0 = -B, y1 = B, d0 = 1.f, d1 = 1.f, cw0 = 0.f, cw1 = 1.f, chm0 = 0.f, chm1 = 1.f;
    int bin = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float cwn = k == K - 1 ? B : fmaf(2.f * B, fmaf(scale, smw[k], kRqsMinBin), cw);
        const float chn = k == K - 1 ? B : fmaf(2.f * B, fmaf(scale, smh[k], kRqsMinBin), ch);
        float dn = 1.f;
        if (k < K - 1) {
            const float u = raw[2 * K + k];
            dn = kRqsMinDeriv + fmaxf(u, 0.f) + fast_ln(1.f + fast_exp(-fabsf(u)));
        }
        const bool here = k == 0 || y >= cw;
        x0 = here ? cw : x0;
        x1 = here ? cwn : x1;
        y0 = here ? ch : y0;
        y1 = here ? chn : y1;
        d0 = here ? dk : d0;
        d1 = here ? dn : d1;
        cw0 = here ? cmw : cw0;
        cw1 = here ? cmw + smw[k] : cw1;
        chm0 = here ? cmh : chm0;
        chm1 = here ? cmh + smh[k] : chm1;
        bin = here ? k : bin;
        cw = cwn;
        ch = chn;
        dk = dn;
        cmw += smw[k];
        cmh += smh[k];
    }
    // forward value and log-derivative at y
    const float bw = x1 - x0, bh = y1 - y0;
    const float rbw = __builtin_amdgcn_rcpf(bw);
    const float s = bh * rbw;
    const float th = (y - x0) * rbw, om = 1.f - th, t1 = th * om;
    const float dd = d0 + d1 - 2.f * s;
    const float N = fmaf(s * th, th, d0 * t1);
    const float num = bh * N;
    const float den = fmaf(dd, t1, s);
    const float rden = __builtin_amdgcn_rcpf(den);
    const float A = fmaf(d1 * th, th, fmaf(2.f * s, t1, d0 * om * om));
    const float rA = __builtin_amdgcn_rcpf(A);
    v_out = fmaf(num, rden, y0);
    const float Fp = s * s * A * rden * rden;                                  // F'(y) = exp(l)
    const float dA_dth = 2.f * (d1 * th + s * (1.f - 2.f * th) - d0 * om);
    const float dl_dy = (dA_dth * rA - 2.f * dd * (1.f - 2.f * th) * rden) * rbw;
    const float q = (gy + dl_dy) * __builtin_amdgcn_rcpf(Fp);
    q_out = q;
    // adjoints of (F, l) with F-bar = -q, l-bar = 1, down to the six knot values
    const float Fb = -q;
    float y0b = Fb, y1b = 0.f, x0b = 0.f, x1b = 0.f, d0b = 0.f, d1b = 0.f;
    const float numb = Fb * rden;
    float denb = -Fb * num * rden * rden - 2.f * rden;
    float sb = 2.f * __builtin_amdgcn_rcpf(s);
    const float Ab = rA;
    d1b += Ab * th * th;
    sb += Ab * 2.f * t1;
    float t1b = Ab * 2.f * s;
    d0b += Ab * om * om;
    float thb = Ab * 2.f * d1 * th;
    float omb = Ab * 2.f * d0 * om;
    sb += denb;
    const float ddb = denb * t1;
    t1b += denb * dd;
    float bhb = numb * N;
    const float Nb = numb * bh;
    sb += Nb * th * th;
    thb += Nb * 2.f * s * th;
    d0b += Nb * t1;
    t1b += Nb * d0;
    d0b += ddb;
    d1b += ddb;
    sb -= 2.f * ddb;
    thb += t1b * om;
    omb += t1b * th;
    thb -= omb;
    x0b -= thb * rbw;
    float bwb = -thb * th * rbw;
    bhb += sb * rbw;
    bwb -= sb * s * rbw;
    y1b += bhb;
    y0b -= bhb;
    x1b += bwb;
    x0b -= bwb;
    // knots -> raw outputs.  Knot k (1 <= k <= K - 1) = -B + 2B sum_{i<k} (scale softmax_i + min): the end knots are constants
    if (bin == 0) {
        x0b = 0.f;
        y0b = 0.f;
        d0b = 0.f;
    }
    if (bin == K - 1) {
        x1b = 0.f;
        y1b = 0.f;
        d1b = 0.f;
    }
    const float cs = 2.f * B * scale;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const float below0 = j < bin ? 1.f : 0.f, below1 = j <= bin ? 1.f : 0.f;
        draw[j] = cs * smw[j] * (x0b * (below0 - cw0) + x1b * (below1 - cw1));
        draw[K + j] = cs * smh[j] * (y0b * (below0 - chm0) + y1b * (below1 - chm1));
    }
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {   // derivative knot j + 1 = min + softplus(raw[2K + j]); d softplus = sigmoid
        const float sg = __builtin_amdgcn_rcpf(1.f + fast_exp(-raw[2 * K + j]));
        draw[2 * K + j] = (j + 1 == bin ? d0b : 0.f) * sg + (j == bin ? d1b : 0.f) * sg;
    }
}

// spline variant of the target loop of coupling_apply: W3 rows are target-major, (3K-1) per target
template <int HP, bool INVERSE>
__device__ __forceinline__ float coupling_targets_rqs(float* __restrict__ row, const float* __restrict__ W3,
                                                      const float* __restrict__ b3, const float (&h)[HP],
                                                      const FlowGeom& g, bool rev) {
    constexpr int P = 3 * kRqsBins - 1;
    float ld = 0.f;
    for (int t = 0; t < g.d_b; ++t) {
        float raw[P];
        const float* wr = W3 + (int64_t)t * P * HP;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            float u = b3[t * P + q];
#pragma unroll
            for (int k = 0; k < HP; ++k) u = fmaf(wr[q * HP + k], h[k], u);
            raw[q] = u;
        }
        const int p = phys(g.d_a + t, g.d, rev);
        row[p] = rqs_coordinate<INVERSE>(row[p], raw, g.bound, ld);
    }
    return INVERSE ? -ld : ld;
}

template <int HP, bool INVERSE>
__device__ __forceinline__ float coupling_apply(float* __restrict__ row, const float* __restrict__ W,
                                                const FlowGeom& g, bool rev, float* __restrict__ hbuf = nullptr) {
    float h[HP];
    const float* W3;
    const float* b3;
    if constexpr (HP > 32) {
        conditioner_hidden_wide<HP>(row, W, g, rev, h, hbuf, (int)(threadIdx.x & 63));
        W3 = w3_of_wide(W, g, HP);
        b3 = W3 + (int64_t)4 * g.d_b * HP;  // W3 | W3T | b3
    } else {
        conditioner_hidden<HP>(row, W, g, rev, h);
        W3 = w3_of(W, g, HP);
        b3 = W3 + (int64_t)g.out_rows * HP;
        if (g.n_bins > 0) return coupling_targets_rqs<HP, INVERSE>(row, W3, b3, h, g, rev);
    }
    float ld = 0.f;
    for (int t = 0; t < g.d_b; ++t) {
        float ua = b3[t], ub = b3[g.d_b + t];
        const float* wa = W3 + (int64_t)t * HP;
        const float* wb = W3 + (int64_t)(g.d_b + t) * HP;
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            ua = fmaf(wa[k], h[k], ua);
            ub = fmaf(wb[k], h[k], ub);
        }
        const float alpha = fast_exp(fmaf(0.5f, ua, g.log1m)) + g.m;
        const float beta = 0.5f * ub;
        const int p = phys(g.d_a + t, g.d, rev);
        const float v = row[p];
        ld += fast_ln(alpha);
        row[p] = INVERSE ? (v - beta) * __builtin_amdgcn_rcpf(alpha) : fmaf(alpha, v, beta);
    }
    return INVERSE ? -ld : ld;
}

// x -> z in place; returns logdet_forward.
template <int HP>
__device__ __forceinline__ float flow_forward_row(float* __restrict__ row, const NfmcRealNVP& f, const FlowGeom& g,
                                                  float* __restrict__ hbuf = nullptr) {
    float ld = 0.f;
    for (int c = 0; c < g.d; ++c) {
        const float ls = f.ea0_log_scale[c];
        row[c] = fmaf(fast_exp(ls), row[c], f.ea0_shift[c]);
        ld += ls;
    }
    for (int l = 0; l < g.n_coupling; ++l)
        ld += coupling_apply<HP, false>(row, f.weights + l * g.layer_stride, g, (l & 1) == 0, hbuf);
    // the last ElementwiseAffine acts on logical coordinates: physical p <-> logical (odd #reversals ? d-1-p : p)
    const bool rev = (g.n_coupling & 1) != 0;
    for (int c = 0; c < g.d; ++c) {
        const int p = phys(c, g.d, rev);
        const float ls = f.ea1_log_scale[c];
        row[p] = fmaf(fast_exp(ls), row[p], f.ea1_shift[c]);
        ld += ls;
    }
    return ld;
}

// z -> x in place; returns logdet_inverse.
template <int HP>
__device__ __forceinline__ float flow_inverse_row(float* __restrict__ row, const NfmcRealNVP& f, const FlowGeom& g,
                                                  float* __restrict__ hbuf = nullptr) {
    float ld = 0.f;
    const bool rev_last = (g.n_coupling & 1) != 0;
    for (int c = 0; c < g.d; ++c) {
        const int p = phys(c, g.d, rev_last);
        const float ls = f.ea1_log_scale[c];
        row[p] = (row[p] - f.ea1_shift[c]) * fast_exp(-ls);
        ld -= ls;
    }
    for (int l = g.n_coupling - 1; l >= 0; --l)
        ld += coupling_apply<HP, true>(row, f.weights + l * g.layer_stride, g, (l & 1) == 0, hbuf);
    for (int c = 0; c < g.d; ++c) {
        const float ls = f.ea0_log_scale[c];
        row[c] = (row[c] - f.ea0_shift[c]) * fast_exp(-ls);
        ld -= ls;
    }
    return ld;
}

// The tile of the final z (forward) / initial z (inverse) is in LOGICAL order of the last layer when the
// number of reversals is odd; these helpers map a logical latent coordinate to its tile column.
__device__ __forceinline__ int latent_col(int c, const FlowGeom& g) { return phys(c, g.d, (g.n_coupling & 1) != 0); }

// Potential value for a whole row held by one lane.
__device__ __forceinline__ float potential_row(const float* __restrict__ row, const NfmcPotential& p, int d) {
    if (p.kind == NFMC_POT_FUNNEL) {
        const float x0 = row[0];
        float s = 0.f;
        for (int c = 1; c < d; ++c) s = fmaf(row[c], row[c], s);
        const float inv_s2 = 1.f / (p.a_scalar * p.a_scalar);
        return 0.5f * x0 * x0 * inv_s2 + 0.5f * fast_exp(-x0) * s + 0.5f * (float)(d - 1) * x0;
    }
    float u = 0.f;
    for (int c = 0; c < d; ++c) {
        const float a = p.a ? p.a[c] : p.a_scalar;
        const float t = row[c] - (p.b ? p.b[c] : p.b_scalar);
        u = fmaf(a * t, t, u);
    }
    return u;
}

// grad of the closed-form potential for a whole row, written to grow (physical order == x order)
__device__ __forceinline__ float potential_value_grad_row(const float* __restrict__ row, float* __restrict__ grow,
                                                          const NfmcPotential& p, int d) {
    if (p.kind == NFMC_POT_FUNNEL) {
        const float x0 = row[0];
        float s = 0.f;
        for (int c = 1; c < d; ++c) s = fmaf(row[c], row[c], s);
        const float inv_s2 = 1.f / (p.a_scalar * p.a_scalar);
        const float e = fast_exp(-x0);
        const float hd = 0.5f * (float)(d - 1);
        grow[0] = x0 * inv_s2 - 0.5f * e * s + hd;
        for (int c = 1; c < d; ++c) grow[c] = row[c] * e;
        return 0.5f * x0 * x0 * inv_s2 + 0.5f * e * s + hd * x0;
    }
    float u = 0.f;
    for (int c = 0; c < d; ++c) {
        const float a = p.a ? p.a[c] : p.a_scalar;
        const float t = row[c] - (p.b ? p.b[c] : p.b_scalar);
        u = fmaf(a * t, t, u);
        grow[c] = 2.f * a * t;
    }
    return u;
}

// Coalesced wave-tile IO: rows r0 .. r0+63 of a row-major (n, d) array <-> tile[64][stride].
// `rev` stores column c of the array at tile column d-1-c (latent side of a flow with an odd number of
// reversals, see latent_col).
__device__ __forceinline__ void tile_load(float* __restrict__ tile, int stride, const float* __restrict__ src,
                                          int64_t r0, int64_t n, int d, bool rev = false) {
    const int lane = threadIdx.x & 63;
    const int64_t rows = n - r0 < 64 ? n - r0 : 64;
    const int total = (int)rows * d;
    const float* s = src + r0 * d;
    for (int i = lane; i < 64 * d; i += kWave) {
        const int r = i / d, c = i - r * d;
        tile[r * stride + (rev ? d - 1 - c : c)] = i < total ? s[i] : 0.f;  // rows beyond n: zeros
    }
}

__device__ __forceinline__ void tile_store(const float* __restrict__ tile, int stride, float* __restrict__ dst,
                                           int64_t r0, int64_t n, int d, bool rev = false) {
    const int lane = threadIdx.x & 63;
    const int64_t rows = n - r0 < 64 ? n - r0 : 64;
    const int total = (int)rows * d;
    float* o = dst + r0 * d;
    for (int i = lane; i < total; i += kWave) {
        const int r = i / d, c = i - r * d;
        o[i] = tile[r * stride + (rev ? d - 1 - c : c)];
    }
}

}  // namespace nfmc
