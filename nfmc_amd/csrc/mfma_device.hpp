// RealNVP with WIDE conditioners (33 <= n_hidden <= 128) on the matrix cores.
//
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 flop/clk/SIMD) with the roles
//     A = weights   lane l holds A[i = 32*mo + (l & 31)][k]            (LDS image, ds_read_b128 = 4 k-steps)
//     B = activations, chains on the N axis: lane l holds B[k][chain = l & 31], k chosen by l >> 5
// A wave owns 32 chains.  Every per-chain vector (state, gradient, hidden activations) lives in the MFMA
// C/D layout ("C layout"): tile m, register t, half h = l >> 5 hold element
//     r = 32 m + (t & 3) + 8 (t >> 2) + 4 h        of chain l & 31.
// With that layout the accumulator tile of one layer IS the B operand of the next: register t of tile m
// pairs rows (R, R+4) across the two lane halves, so k-step t multiplies with A columns (R, R+4), which
// is exactly what one 16-byte LDS read per lane delivers for four consecutive t.  No activation ever moves
// between lanes or through LDS; only weights are staged (once per GEMM per workgroup of 128 chains).
//
// Weight blob per coupling layer for this path (both orientations, so the transposed products of the
// reverse sweep also read rows): HP = 64 or 128, d = 64 or 128 (d_a = d_b = d/2):
//     W1 (HP,d_a) | W1T (d_a,HP) | b1 (HP) | [Wh (HP,HP) | WhT (HP,HP) | bh (HP)] x (n_hl-1) | W3 (2d_b,HP) | W3T (HP,2d_b) | b3 (2d_b)
#pragma once

#include "flow_device.hpp"

namespace nfmc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kMfmaBlock = 256;      // 4 waves x 32 chains
constexpr int kMfmaChains = 128;     // chains per workgroup

struct MLayer {
    const float *W1, *W1T, *b1, *Wh, *WhT, *bh, *W3, *W3T, *b3;
};

__host__ __device__ inline int64_t mfma_layer_floats(int d, int hp, int n_hl) {
    const int64_t da = d / 2, db = d - d / 2;
    return 2 * (int64_t)hp * da + hp + (int64_t)(n_hl - 1) * (2 * (int64_t)hp * hp + hp) + 4 * db * hp + 2 * db;
}

__device__ __forceinline__ MLayer mfma_layer(const float* base, int d, int hp, int n_hl) {
    const int da = d / 2, db = d - da;
    MLayer L;
    L.W1 = base;
    L.W1T = L.W1 + hp * da;
    L.b1 = L.W1T + da * hp;
    const float* p = L.b1 + hp;
    L.Wh = L.WhT = L.bh = nullptr;
    if (n_hl > 1) {
        L.Wh = p;
        L.WhT = L.Wh + hp * hp;
        L.bh = L.WhT + hp * hp;
        p = L.bh + hp;
    }
    L.W3 = p;
    L.W3T = L.W3 + 2 * db * hp;
    L.b3 = L.W3T + hp * 2 * db;
    return L;
}

// ---- staging: global (rows x K, row-major) -> LDS image [rows][K+4]; optional reversal of rows / columns
// inside blocks (a reversed coupling layer sees logical coordinate j at physical position d-1-j).
__device__ __forceinline__ void stage_matrix(float* __restrict__ img, const float* __restrict__ W, int rows, int K,
                                             bool rev_rows, int rblk, bool rev_cols, int cblk) {
    const int ld = K + 4, k4 = K >> 2;
    for (int idx = threadIdx.x; idx < rows * k4; idx += kMfmaBlock) {
        const int r = idx / k4, c = (idx - r * k4) << 2;
        const f32x4 v = *reinterpret_cast<const f32x4*>(W + (size_t)r * K + c);
        const int rr = rev_rows ? (r / rblk) * rblk + (rblk - 1 - (r % rblk)) : r;
        if (!rev_cols) {
            *reinterpret_cast<f32x4*>(img + rr * ld + c) = v;
        } else {
            const int cc = (c / cblk) * cblk + (cblk - 4 - (c % cblk));
            f32x4 w;
            w[0] = v[3];
            w[1] = v[2];
            w[2] = v[1];
            w[3] = v[0];
            *reinterpret_cast<f32x4*>(img + rr * ld + cc) = w;
        }
    }
}

__device__ __forceinline__ void stage_vector(float* __restrict__ dst, const float* __restrict__ b, int len, bool rev,
                                             int blk) {
    for (int i = threadIdx.x; i < len; i += kMfmaBlock) dst[rev ? (i / blk) * blk + (blk - 1 - (i % blk)) : i] = b[i];
}

// ---- one 32-row output tile: acc += A[32*mo .. +31][0 .. 32*TK) x act
template <int TK>
__device__ __forceinline__ void gemm_tile(f32x16& acc, const float* __restrict__ arow, const f32x16 (&act)[TK]) {
#pragma unroll
    for (int mk = 0; mk < TK; ++mk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 32 * mk + 8 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], act[mk][4 * q + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], act[mk][4 * q + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], act[mk][4 * q + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], act[mk][4 * q + 3], acc, 0, 0, 0);
        }
    }
}

// bias / parameter tile in C layout from an LDS (or global) vector: element 32*mo + 8q + 4h + j -> reg 4q+j
__device__ __forceinline__ f32x16 vec_tile(const float* __restrict__ v, int mo, int half) {
    f32x16 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(v + 32 * mo + 8 * q + 4 * half);
        t[4 * q + 0] = a[0];
        t[4 * q + 1] = a[1];
        t[4 * q + 2] = a[2];
        t[4 * q + 3] = a[3];
    }
    return t;
}

// same from a global vector indexed by LOGICAL coordinate, when the tile position p holds logical d-1-p
__device__ __forceinline__ f32x16 vec_tile_rev(const float* __restrict__ v, int mo, int half, int d) {
    f32x16 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int p0 = 32 * mo + 8 * q + 4 * half;
        const f32x4 a = *reinterpret_cast<const f32x4*>(v + (d - 4 - p0));
        t[4 * q + 0] = a[3];
        t[4 * q + 1] = a[2];
        t[4 * q + 2] = a[1];
        t[4 * q + 3] = a[0];
    }
    return t;
}

// ---- (n, d) row-major <-> C-layout tiles of this lane's chain.  `rev`: the array is in LOGICAL latent order
// and tile position p holds logical d-1-p (flows with an odd number of reversals, see latent_col).
template <int TD>
__device__ __forceinline__ void load_ctiles(f32x16 (&x)[TD], const float* __restrict__ base, int64_t row, int d,
                                            int half, bool rev) {
    const float* r = base + row * d;
#pragma unroll
    for (int m = 0; m < TD; ++m) x[m] = rev ? vec_tile_rev(r, m, half, d) : vec_tile(r, m, half);
}

template <int TD>
__device__ __forceinline__ void store_ctiles(const f32x16 (&x)[TD], float* __restrict__ base, int64_t row, int d,
                                             int half, bool rev) {
    float* r = base + row * d;
#pragma unroll
    for (int m = 0; m < TD; ++m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p0 = 32 * m + 8 * q + 4 * half;
            f32x4 a;
            if (!rev) {
                a[0] = x[m][4 * q + 0];
                a[1] = x[m][4 * q + 1];
                a[2] = x[m][4 * q + 2];
                a[3] = x[m][4 * q + 3];
                *reinterpret_cast<f32x4*>(r + p0) = a;
            } else {
                a[3] = x[m][4 * q + 0];
                a[2] = x[m][4 * q + 1];
                a[1] = x[m][4 * q + 2];
                a[0] = x[m][4 * q + 3];
                *reinterpret_cast<f32x4*>(r + (d - 4 - p0)) = a;
            }
        }
    }
}

__device__ __forceinline__ float sum16(const f32x16& v) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += v[t];
    return s;
}

__device__ __forceinline__ float pair_sum(float v) { return v + __shfl_xor(v, 32, kWave); }  // both halves of a chain

__device__ __forceinline__ f32x16 tanh16(f32x16 v) {
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = fast_tanh(v[t]);
    return v;
}

// LDS carve-up (floats): two weight images of 128 x 132 and a 256-float vector area
constexpr int kImgFloats = 128 * 132;
constexpr int kVecFloats = 256;
constexpr int kMfmaStatDoubles = 4 * (2 * 128 + 2);
constexpr size_t kMfmaLdsBytes = (size_t)(2 * kImgFloats + kVecFloats) * sizeof(float) + kMfmaStatDoubles * sizeof(double);

// entry points of neutra_mfma.hip used by the C ABI in neutra_kernels.hip / flow_kernels.hip
int nfmc_mfma_supported(int32_t d, int32_t n_hidden, int32_t n_hidden_layers);
int nfmc_neutra_potential_grad_mfma_f32(const NfmcRealNVP* flow, const NfmcPotential* pot, const float* z, int64_t n,
                                        float* u_out, float* grad_out, nfmc_stream_t stream);
int nfmc_neutra_hmc_steps_mfma_f32(const NfmcNeutraHmcArgs* args, float* scratch, int64_t scratch_bytes,
                                   nfmc_stream_t stream);

// Conditioner hidden stack in C layout: h1 = tanh(W1 x_src + b1), h2 = tanh(Wh h1 + bh).  Workgroup-collective
// (stages weights, barriers inside).  `src` are the TD/2 source tiles of the layer input.
template <int TS, int TH, int NHL>
__device__ __forceinline__ void hidden_stack(const f32x16 (&src)[TS], f32x16 (&h1)[TH], f32x16 (&h2)[TH],
                                             const MLayer& L, int hp, int d_a, bool rev, float* img0, float* vec,
                                             int col, int half) {
    __syncthreads();
    stage_matrix(img0, L.W1, hp, d_a, false, 1, rev, d_a);
    stage_vector(vec, L.b1, hp, false, 1);
    __syncthreads();
#pragma unroll
    for (int mo = 0; mo < TH; ++mo) {
        h1[mo] = vec_tile(vec, mo, half);
        gemm_tile<TS>(h1[mo], img0 + (32 * mo + col) * (d_a + 4) + 4 * half, src);
        h1[mo] = tanh16(h1[mo]);
    }
    if constexpr (NHL > 1) {
        __syncthreads();
        stage_matrix(img0, L.Wh, hp, hp, false, 1, false, 1);
        stage_vector(vec, L.bh, hp, false, 1);
        __syncthreads();
#pragma unroll
        for (int mo = 0; mo < TH; ++mo) {
            h2[mo] = vec_tile(vec, mo, half);
            gemm_tile<TH>(h2[mo], img0 + (32 * mo + col) * (hp + 4) + 4 * half, h1);
            h2[mo] = tanh16(h2[mo]);
        }
    }
}

}  // namespace nfmc
