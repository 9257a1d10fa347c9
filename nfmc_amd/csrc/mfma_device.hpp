// RealNVP with WIDE conditioners (33 <= n_hidden <= 128) on the matrix cores.
//
// v_mfma_f32_16x16x4_f32 (exact fp32; 32 cycles, 1024 MAC -> 64 flop/clk/SIMD, the fp32 peak) with the roles
//     A = weights   lane l holds A[i = 16*mo + (l & 15)][k]            (LDS image, one ds_read_b128 = 4 k-steps)
//     B = activations, chains on the N axis: lane l holds B[k][chain = l & 15], k chosen by q = l >> 4
// A wave owns 16 chains; a workgroup is 8 waves = 128 chains, two waves per SIMD (<= 256 VGPRs), so one wave's
// LDS / barrier / elementwise phases hide behind the other's MFMAs (the 32-chain-per-wave first version ran at
// one wave per SIMD with spills: 55 % of wave cycles in s_waitcnt/s_barrier, MfmaUtil 24 %, profiles/).
// Every per-chain vector (state, gradient, hidden activations) lives in the MFMA C/D layout ("C layout"):
// tile m (16 elements), register r, lane group q hold element
//     e = 16 m + 4 q + r        of chain l & 15.
// With that layout the accumulator tile of one layer IS the B operand of the next: k-step r of tile m pairs
// the four lane groups with k = 16 m + 4 q + r, so the A operand of lane (i, q) for r = 0..3 is the four
// consecutive weights A[i][16 m + 4 q .. +3]: one 16-byte LDS read.  No activation ever moves between lanes
// or through LDS; only weights are staged (once per GEMM per workgroup of 128 chains).
//
// Weight blob per coupling layer for this path (both orientations, so the transposed products of the
// reverse sweep also read rows): HP = 64 or 128, d = 64 or 128 (d_a = d_b = d/2):
//     W1 (HP,d_a) | W1T (d_a,HP) | b1 (HP) | [Wh (HP,HP) | WhT (HP,HP) | bh (HP)] x (n_hl-1) | W3 (2d_b,HP) | W3T (HP,2d_b) | b3 (2d_b)
#pragma once

#include "flow_device.hpp"

namespace nfmc {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kMfmaBlock = 512;      // 8 waves x 16 chains
constexpr int kMfmaWaves = kMfmaBlock / 64;
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
// All extents are powers of two known at compile time, so the index arithmetic is shifts and masks (with
// run-time extents each element cost several integer divisions: as much VALU time as the GEMM it fed).
template <int K, int RBLK, int CBLK>
__device__ __forceinline__ void stage_matrix(float* __restrict__ img, const float* __restrict__ W, int rows,
                                             bool rev_rows, bool rev_cols) {
    constexpr int ld = K + 4, k4 = K >> 2;
    static_assert((K & (K - 1)) == 0 && (RBLK & (RBLK - 1)) == 0 && (CBLK & (CBLK - 1)) == 0, "power-of-two extents");
    // The copy addresses depend only on the thread index, so LICM would hoist the address arithmetic of EVERY
    // staging call of the kernel above the tile loop and keep hundreds of VGPRs live through all the GEMMs
    // (measured: ~2 KB of scratch per lane).  An opaque copy of the index pins the arithmetic to the call.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    for (int idx = tid; idx < rows * k4; idx += kMfmaBlock) {
        const int r = idx / k4, c = (idx % k4) << 2;
        const f32x4 v = *reinterpret_cast<const f32x4*>(W + (size_t)r * K + c);
        const int rr = rev_rows ? (r / RBLK) * RBLK + (RBLK - 1 - (r % RBLK)) : r;
        if (!rev_cols) {
            *reinterpret_cast<f32x4*>(img + rr * ld + c) = v;
        } else {
            const int cc = (c / CBLK) * CBLK + (CBLK - 4 - (c % CBLK));
            f32x4 w;
            w[0] = v[3];
            w[1] = v[2];
            w[2] = v[1];
            w[3] = v[0];
            *reinterpret_cast<f32x4*>(img + rr * ld + cc) = w;
        }
    }
}

template <int BLK>
__device__ __forceinline__ void stage_vector(float* __restrict__ dst, const float* __restrict__ b, int len, bool rev) {
    static_assert((BLK & (BLK - 1)) == 0, "power-of-two block");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // as in stage_matrix
    for (int i = tid; i < len; i += kMfmaBlock) dst[rev ? (i / BLK) * BLK + (BLK - 1 - (i % BLK)) : i] = b[i];
}

// ---- one 16-row output tile: acc += A[16*mo .. +15][0 .. 16*TK) x act ; arow = img + (16*mo + col)*ld + 4*q
template <int TK>
__device__ __forceinline__ void gemm_tile(f32x4& acc, const float* __restrict__ arow, const f32x4 (&act)[TK]) {
    // All A fragments of the tile are read first; the fence between the reads and the MFMAs lets the scheduler
    // move the NEXT tile's reads above this tile's MFMAs (one tile of look-ahead, TK*4 extra registers) but no
    // further: without any fence it hoists the reads of every tile of the unrolled GEMM and spills.
    f32x4 a[TK];
#pragma unroll
    for (int mk = 0; mk < TK; ++mk) a[mk] = *reinterpret_cast<const f32x4*>(arow + 16 * mk);
#ifndef NFMC_MFMA_FENCE_END
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int mk = 0; mk < TK; ++mk) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mk][0], act[mk][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mk][1], act[mk][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mk][2], act[mk][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mk][3], act[mk][3], acc, 0, 0, 0);
    }
#ifdef NFMC_MFMA_FENCE_END
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// tile of a vector in C layout: elements 16*mo + 4*q + (0..3)
__device__ __forceinline__ f32x4 vec_tile(const float* __restrict__ v, int mo, int q) {
    return *reinterpret_cast<const f32x4*>(v + 16 * mo + 4 * q);
}

// same from a vector indexed by LOGICAL coordinate when tile position p holds logical d-1-p
__device__ __forceinline__ f32x4 vec_tile_rev(const float* __restrict__ v, int mo, int q, int d) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(v + (d - 4 - (16 * mo + 4 * q)));
    f32x4 t;
    t[0] = a[3];
    t[1] = a[2];
    t[2] = a[1];
    t[3] = a[0];
    return t;
}

// ---- (n, d) row-major <-> C-layout tiles of this lane's chain.  `rev`: the array is in LOGICAL latent order
// and tile position p holds logical d-1-p (flows with an odd number of reversals, see latent_col).
template <int TD>
__device__ __forceinline__ void load_ctiles(f32x4 (&x)[TD], const float* __restrict__ base, int64_t row, int d, int q,
                                            bool rev) {
    const float* r = base + row * d;
#pragma unroll
    for (int m = 0; m < TD; ++m) x[m] = rev ? vec_tile_rev(r, m, q, d) : vec_tile(r, m, q);
}

template <int TD>
__device__ __forceinline__ void store_ctiles(const f32x4 (&x)[TD], float* __restrict__ base, int64_t row, int d,
                                             int q, bool rev) {
    float* r = base + row * d;
#pragma unroll
    for (int m = 0; m < TD; ++m) {
        const int p0 = 16 * m + 4 * q;
        if (!rev) {
            *reinterpret_cast<f32x4*>(r + p0) = x[m];
        } else {
            f32x4 a;
            a[3] = x[m][0];
            a[2] = x[m][1];
            a[1] = x[m][2];
            a[0] = x[m][3];
            *reinterpret_cast<f32x4*>(r + (d - 4 - p0)) = a;
        }
    }
}

// sum over the four lanes (q = 0..3) that share a chain
__device__ __forceinline__ float chain_sum(float v) {
    v += __shfl_xor(v, 16, kWave);
    v += __shfl_xor(v, 32, kWave);
    return v;
}

__device__ __forceinline__ f32x4 tanh4(f32x4 v) {
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = fast_tanh(v[t]);
    return v;
}

// LDS carve-up: two weight images of 128 x 132 floats, a 256-float vector area, per-wave statistics
constexpr int kImgFloats = 128 * 132;
constexpr int kVecFloats = 256;          // one bias / vector area per image
constexpr int kMfmaStatDoubles = kMfmaWaves * (2 * 128 + 2);
constexpr size_t kMfmaLdsBytes = (size_t)(2 * kImgFloats + 2 * kVecFloats) * sizeof(float) + kMfmaStatDoubles * sizeof(double);

// entry points of neutra_mfma.hip used by the C ABI in neutra_kernels.hip
int nfmc_mfma_supported(int32_t d, int32_t n_hidden, int32_t n_hidden_layers);
int nfmc_neutra_potential_grad_mfma_f32(const NfmcRealNVP* flow, const NfmcPotential* pot, const float* z, int64_t n,
                                        float* u_out, float* grad_out, nfmc_stream_t stream);
int nfmc_neutra_hmc_steps_mfma_f32(const NfmcNeutraHmcArgs* args, float* scratch, int64_t scratch_bytes,
                                   nfmc_stream_t stream);
// entry points of flow_mfma.hip used by the C ABI in flow_kernels.hip (shapes: nfmc_mfma_supported)
int nfmc_realnvp_forward_mfma_f32(const NfmcRealNVP* f, const float* x, int64_t n, float* z, float* logdet,
                                  float* log_prob, nfmc_stream_t stream);
int nfmc_realnvp_inverse_mfma_f32(const NfmcRealNVP* f, const float* z, int64_t n, float* x, float* logdet,
                                  float* log_q, const NfmcRng* rng, nfmc_stream_t stream);
int nfmc_flow_mh_steps_mfma_f32(const NfmcFlowMhArgs& a, nfmc_stream_t stream, int* grid_out, int* dp_out);

// ---- the weight pipeline: two LDS images (+ their vector areas) used STRICTLY ALTERNATELY, one per GEMM.
// A wave stages the operand of GEMM k right after it finished GEMM k-1, into the image GEMM k-2 read; every
// wave passed the barrier that closed staging k-1 only after it had finished GEMM k-2, so nobody still reads
// that image: ONE barrier per GEMM (after its staging), and the staging loads of early waves overlap the
// MFMAs of late ones.  The alternation must never be broken (also across layers, sweeps and chain tiles).
struct WeightPipe {
    float* lds;
    int buf;
    __device__ __forceinline__ float* img() const { return lds + buf * kImgFloats; }
    __device__ __forceinline__ float* vec() const { return lds + 2 * kImgFloats + buf * kVecFloats; }
    // stage matrix W (rows x K) and, if given, a vector, into the next image; closes with the barrier
    template <int K, int RBLK, int CBLK, int VBLK>
    __device__ __forceinline__ void stage(const float* W, int rows, bool rev_rows, bool rev_cols, const float* v,
                                          int vlen, bool vrev) {
        buf ^= 1;
        stage_matrix<K, RBLK, CBLK>(img(), W, rows, rev_rows, rev_cols);
        if (v) stage_vector<VBLK>(vec(), v, vlen, vrev);
        __syncthreads();
    }
};

// Conditioner hidden stack in C layout: h1 = tanh(W1 x_src + b1), h2 = tanh(Wh h1 + bh).  Workgroup-collective
// (two steps of the weight pipeline).  `src` are the source tiles of the layer input (TS = d_a / 16).
template <int TS, int TH, int NHL>
__device__ __forceinline__ void hidden_stack(const f32x4 (&src)[TS], f32x4 (&h1)[TH], f32x4 (&h2)[TH], const MLayer& L,
                                             bool rev, WeightPipe& wp, int col, int q) {
    constexpr int hp = 16 * TH, d_a = 16 * TS;
    wp.template stage<d_a, 1, d_a, 1>(L.W1, hp, false, rev, L.b1, hp, false);
    {
        const float* img = wp.img();
        const float* vec = wp.vec();
#pragma unroll
        for (int mo = 0; mo < TH; ++mo) {
            h1[mo] = vec_tile(vec, mo, q);
            gemm_tile<TS>(h1[mo], img + (16 * mo + col) * (d_a + 4) + 4 * q, src);
            h1[mo] = tanh4(h1[mo]);
        }
    }
    if constexpr (NHL > 1) {
        wp.template stage<hp, 1, 1, 1>(L.Wh, hp, false, false, L.bh, hp, false);
        const float* img = wp.img();
        const float* vec = wp.vec();
#pragma unroll
        for (int mo = 0; mo < TH; ++mo) {
            h2[mo] = vec_tile(vec, mo, q);
            gemm_tile<TH>(h2[mo], img + (16 * mo + col) * (hp + 4) + 4 * q, h1);
            h2[mo] = tanh4(h2[mo]);
        }
    }
}

}  // namespace nfmc
