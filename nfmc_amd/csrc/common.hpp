// Shared device helpers for the nfmc gfx950 kernels: Philox4x32-10, Box-Muller on the hardware
// transcendentals, chain-group reductions, closed-form potentials, deterministic statistics.
// gfx950 only: wave = 64 lanes, no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nfmc_hip.h"

namespace nfmc {

constexpr int kWave = 64;
constexpr int kBlock = 256;          // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxGrid = 2048;       // 256 CUs x 8 workgroups; larger problems grid-stride (1024 measured 4 % slower).
                                     // Also the row count of the statistics slab the fold kernel walks.
#ifndef NFMC_WPE
#define NFMC_WPE 1
#endif
constexpr int kStatTail = 4;         // per-workgroup scratch tail: accepted, nonfinite, jump accepted, jump nonfinite

// RNG stream tags (oracle/philox.py)
constexpr uint32_t kTagNoise = 0, kTagAccept = 1, kTagLatent = 2, kTagJump = 3;

// ------------------------------------------------------------------------------------------------
// Philox4x32-R (Salmon et al. SC'11); R = 10 is the library's stream, R = 7 an opt-in one (NfmcRng.rounds: the
// smallest round count the Random123 authors report as passing BigCrush, 30 % fewer generator instructions).  The key
// schedule is wave-uniform, so the compiler keeps the round keys in SGPRs; the 32x32->64 products become
// v_mad_u64_u32, the two xors of a word one v_bitop3_b32.
template <int R>
__device__ __forceinline__ uint4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // gfx950's three-input bit op (truth table 0x96 = a ^ b ^ c): one VOP3 instead of two dependent v_xor --
        // the compiler does not form it by itself; mala_kernel 0.327 -> 0.291 ms per launch (tools/ubench.hip:
        // v_bitop3_b32 with an SGPR key issues at 1.31x a v_fma, the xor pair at 1.5x, and the chain is one op shorter)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                uint32_t k1) {
    return philox4x32<10>(c0, c1, c2, c3, k0, k1);
}
// NfmcRng.rounds: 0 and 10 mean Philox4x32-10; 7 is the opt-in stream; anything else is an argument error
inline int rng_rounds(const NfmcRng& r) { return r.rounds == 0 ? 10 : (int)r.rounds; }
inline bool rng_rounds_ok(const NfmcRng& r, bool seven_supported) {
    return r.rounds == 0 || r.rounds == 10 || (r.rounds == 7 && seven_supported);
}
// for entry points that only have the default stream: 0 ok, NFMC_EUNSUPPORTED for the opt-in one, NFMC_EINVAL otherwise
inline int rng_default_only(const NfmcRng& r) {
    return (r.rounds == 0 || r.rounds == 10) ? NFMC_OK : (r.rounds == 7 ? NFMC_EUNSUPPORTED : NFMC_EINVAL);
}

// (0,1) uniform with 23 random bits, exact in fp32: (2 (r >> 9) + 1) 2^-24.
__device__ __forceinline__ float u32_to_uniform(uint32_t r) {
    return (float)(2u * (r >> 9) + 1u) * 0x1p-24f;
}

// Two standard normals from two words.  v_log_f32 is log2, v_cos/v_sin take revolutions, so
// R = sqrt(-2 ln2 log2 u1), angle = u2 -- no 2 pi multiply and no range reduction.
__device__ __forceinline__ void box_muller(uint32_t ra, uint32_t rb, float& za, float& zb) {
    const float u1 = fmaf((float)ra, 0x1p-32f, 0x1p-33f);
    const float u2 = (float)rb * 0x1p-32f;
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    za = rad * __builtin_amdgcn_cosf(u2);
    zb = rad * __builtin_amdgcn_sinf(u2);
}

// Four normals for coordinate block `blk` of (chain, step) on stream `tag`.
template <int R = 10>
__device__ __forceinline__ void philox_normal4(uint32_t chain, uint32_t step, uint32_t blk, uint32_t tag, uint32_t k0,
                                               uint32_t k1, float (&z)[4]) {
    const uint4 r = philox4x32<R>(chain, step, blk, tag, k0, k1);
    box_muller(r.x, r.y, z[0], z[1]);
    box_muller(r.z, r.w, z[2], z[3]);
}

__device__ __forceinline__ uint32_t pick_word(const uint4& r, uint32_t i) {
    return i == 0 ? r.x : (i == 1 ? r.y : (i == 2 ? r.z : r.w));
}

// natural log on v_log_f32 (log2): |err| <~ 1 ulp of log2 -- used for the Metropolis test log(u).
__device__ __forceinline__ float fast_ln(float v) { return 0.6931471805599453f * __builtin_amdgcn_logf(v); }
__device__ __forceinline__ float fast_exp(float v) { return __builtin_amdgcn_exp2f(1.4426950408889634f * v); }

// ------------------------------------------------------------------------------------------------
// Butterfly all-reduce over the LPC consecutive lanes that share one chain.  fp add is commutative,
// so every lane of the group ends with the bitwise same sum: the accept decision needs no broadcast.
// Steps inside a 16-lane row are DPP moves on the VALU (quad_perm for xor 1/2, row_half_mirror and
// row_mirror pair the already-uniform halves for 8 and 16 lanes); only 32/64-lane groups go through the
// LDS crossbar (ds_bpermute).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(moved);
}

// Per-lane select with the condition in an SGPR pair (mask = __ballot(cond)).  Measured on gfx950
// (tools/ubench.hip, profiles/): the VOP2 form the compiler shrinks `c ? a : b` to, v_cndmask_b32_e32 with the
// implicit vcc, issues at 6.6x a v_fma (~22 cycles per wave instruction); the VOP3 form is 1.3x.  Hot loops
// that select a whole register row on one condition (the Metropolis update) use this.
__device__ __forceinline__ float select_f32(uint64_t mask, float if_true, float if_false) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_false), "v"(if_true), "s"(mask));
    return r;
}

template <int LPC>
__device__ __forceinline__ float group_allreduce(float v) {
    if constexpr (LPC >= 2) v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]  : lane ^ 1
    if constexpr (LPC >= 4) v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]  : lane ^ 2
    if constexpr (LPC >= 8) v = dpp_add<0x141>(v);  // row_half_mirror      : i <-> 7 - i
    if constexpr (LPC >= 16) v = dpp_add<0x140>(v); // row_mirror           : i <-> 15 - i
    if constexpr (LPC >= 32) v += __shfl_xor(v, 16, kWave);
    if constexpr (LPC >= 64) v += __shfl_xor(v, 32, kWave);
    return v;
}

// ------------------------------------------------------------------------------------------------
// Reduce-scatter / all-gather of NU (4 or 8) per-lane values over the LPC >= NU lanes of a chain: the register
// flow kernels give each conditioner hidden unit to ONE lane class (unit = lane % NU) instead of evaluating the
// whole hidden stack redundantly on every lane.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// value of lane ^ 4: row_shl:4 into the even 4-lane banks, row_shr:4 into the odd ones
__device__ __forceinline__ float dpp_xor4(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, __float_as_int(v), 0x114, 0xf, 0xa, false);
    return __int_as_float(t);
}

// position r of an all-gathered register row holds unit (lane % NU) ^ unit_xor<NU>(r)
template <int NU>
__host__ __device__ constexpr int unit_xor(int r) {
    return NU == 8 ? (((r & 1) << 2) | (r & 2) | ((r >> 2) & 1)) : (((r & 1) << 1) | ((r >> 1) & 1));
}

// h[k] = this lane's partial of unit k.  Returns the sum over the chain's LPC lanes of unit (lane % NU), with the
// association of group_allreduce (pairs at lane distance 1, then 2, 4, ...), so both give bitwise the same sums.
template <int NU, int LPC>
__device__ __forceinline__ float group_reduce_scatter(const float (&h)[NU]) {
    static_assert((NU == 4 || NU == 8) && LPC >= NU, "one lane class per unit");
    constexpr uint64_t M0 = 0xAAAAAAAAAAAAAAAAull, M1 = 0xCCCCCCCCCCCCCCCCull, M2 = 0xF0F0F0F0F0F0F0F0ull;
    float a[NU / 2];
#pragma unroll
    for (int r = 0; r < NU / 2; ++r)
        a[r] = select_f32(M0, h[2 * r + 1], h[2 * r]) + dpp_mov<0xB1>(select_f32(M0, h[2 * r], h[2 * r + 1]));
    float c;
    if constexpr (NU == 8) {
        float b[2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
            b[r] = select_f32(M1, a[2 * r + 1], a[2 * r]) + dpp_mov<0x4E>(select_f32(M1, a[2 * r], a[2 * r + 1]));
        c = select_f32(M2, b[1], b[0]) + dpp_xor4(select_f32(M2, b[0], b[1]));
    } else {
        c = select_f32(M1, a[1], a[0]) + dpp_mov<0x4E>(select_f32(M1, a[0], a[1]));
        if constexpr (LPC >= 8) c += dpp_xor4(c);
    }
    if constexpr (LPC >= 16) c += dpp_mov<0x128>(c);  // row_ror:8 : lane ^ 8
    if constexpr (LPC >= 32) c += __shfl_xor(c, 16, kWave);
    if constexpr (LPC >= 64) c += __shfl_xor(c, 32, kWave);
    return c;
}

// v = value of unit (lane % NU); h[r] = value of unit (lane % NU) ^ unit_xor<NU>(r)
template <int NU>
__device__ __forceinline__ void group_all_gather(float v, float (&h)[NU]) {
    h[0] = v;
    if constexpr (NU == 8) {
        h[1] = dpp_xor4(h[0]);
        h[2] = dpp_mov<0x4E>(h[0]);
        h[3] = dpp_mov<0x4E>(h[1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[4 + r] = dpp_mov<0xB1>(h[r]);
    } else {
        h[1] = dpp_mov<0x4E>(h[0]);
        h[2] = dpp_mov<0xB1>(h[0]);
        h[3] = dpp_mov<0xB1>(h[1]);
    }
}

template <int LPC>
__device__ __forceinline__ float group_broadcast0(float v) {
    // value held by the group's first lane
    return __shfl(v, (int)(threadIdx.x & 63) & ~(LPC - 1), kWave);
}

// Sum over the lanes of a wave that hold the SAME coordinates of DIFFERENT chains (stride LPC), in fp32: the
// addends are each lane's fp32 partial sums over the launch's transitions, at most 64 of them meet here, and the
// result is widened to fp64 before it joins the other waves' sums.  Steps inside a 16-lane row are DPP moves, the
// two across rows go through ds_bpermute: for LPC = 8 one DPP + two permutes per value instead of the six permutes
// and three v_add_f64 of a double butterfly (the epilogue was ~25 % of a one-tile-per-wave flow-MH launch).
template <int LPC>
__device__ __forceinline__ float cross_chain_reduce(float v) {
    if constexpr (LPC <= 1) v = dpp_add<0xB1>(v);
    if constexpr (LPC <= 2) v = dpp_add<0x4E>(v);
    if constexpr (LPC <= 4) v += dpp_xor4(v);
    if constexpr (LPC <= 8) v += dpp_mov<0x128>(v);   // row_ror:8 : lane ^ 8 within the row
    if constexpr (LPC <= 16) v += __shfl_xor(v, 16, kWave);
    if constexpr (LPC <= 32) v += __shfl_xor(v, 32, kWave);
    return v;
}

// ------------------------------------------------------------------------------------------------
// Register layout of a chain ("interleaved 4-blocks"): lane g of the chain's LPC lanes holds, in register i,
// coordinate 4 * ((i / 4) * LPC + g) + (i % 4): 4-coordinate blocks dealt round-robin over the lanes.
// One register quad = one Philox block = one 16-byte global access; consecutive lanes touch consecutive
// 16-byte pieces of the row (coalesced); with d = CPL * LPC, quad q of every lane lies in the q-th 1/(CPL/4)
// of the coordinates, which flow_b.hpp uses to give coupling layers compile-time source/target roles.
template <int CPL, int LPC>
__device__ __forceinline__ int coord_of(int g, int i) {
    return 4 * ((i >> 2) * LPC + g) + (i & 3);
}

// ------------------------------------------------------------------------------------------------
// Potentials.  `term` is the coordinate's share of U (U = group sum of terms), `grad` dU/dx_c.
// Coordinates beyond d carry a = 0 / x = 0 so they contribute exactly zero.
template <int CPL, int LPC, bool FAST>
struct QuadraticPot {
    // U = sum a_c (x_c - b_c)^2
    static constexpr bool kQuadratic = true;
    float a_s, b_s;
    float a[FAST ? 1 : CPL], b[FAST ? 1 : CPL];
    struct Ctx {};

    __device__ __forceinline__ void init(const NfmcPotential& p, int g, int d) {
        a_s = p.a_scalar;
        b_s = p.b_scalar;
        if constexpr (!FAST) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = coord_of<CPL, LPC>(g, i);
                const bool ok = c < d;
                a[i] = ok ? (p.a ? p.a[c] : p.a_scalar) : 0.f;
                b[i] = ok ? (p.b ? p.b[c] : p.b_scalar) : 0.f;
            }
        }
    }
    __device__ __forceinline__ Ctx prepare(const float (&)[CPL], int, int) const { return Ctx{}; }
    __device__ __forceinline__ float aa(int i) const { return FAST ? a_s : a[FAST ? 0 : i]; }
    __device__ __forceinline__ float bb(int i) const { return FAST ? b_s : b[FAST ? 0 : i]; }
    __device__ __forceinline__ float grad(const Ctx&, int i, float x) const { return 2.f * aa(i) * (x - bb(i)); }
    __device__ __forceinline__ float term(const Ctx&, int i, float x) const {
        const float t = x - bb(i);
        return aa(i) * t * t;
    }
};

template <int CPL, int LPC, bool FAST>
struct FunnelPot {
    // U = x0^2/(2 s^2) + sum_{i>=1} [ x_i^2 e^{-x0} / 2 + x0 / 2 ]
    static constexpr bool kQuadratic = false;
    float inv_s2, half_dm1;
    bool lead;          // this lane holds coordinate 0 in register 0
    float valid[CPL];   // 1 for real coordinates, 0 for padding
    struct Ctx {
        float x0, e, s;  // x_0, exp(-x_0), sum_{i>=1} x_i^2
    };

    __device__ __forceinline__ void init(const NfmcPotential& p, int g, int d) {
        inv_s2 = 1.f / (p.a_scalar * p.a_scalar);
        half_dm1 = 0.5f * (float)(d - 1);
        lead = (g == 0);
#pragma unroll
        for (int i = 0; i < CPL; ++i) valid[i] = coord_of<CPL, LPC>(g, i) < d ? 1.f : 0.f;
    }
    __device__ __forceinline__ Ctx prepare(const float (&x)[CPL], int, int) const {
        Ctx c;
        c.x0 = group_broadcast0<LPC>(x[0]);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) s = fmaf(x[i], (lead && i == 0) ? 0.f : x[i], s);
        c.s = group_allreduce<LPC>(s);
        c.e = fast_exp(-c.x0);
        return c;
    }
    __device__ __forceinline__ float grad(const Ctx& c, int i, float x) const {
        const float g0 = c.x0 * inv_s2 - 0.5f * c.e * c.s + half_dm1;
        return (lead && i == 0) ? g0 : x * c.e * valid[i];
    }
    __device__ __forceinline__ float term(const Ctx& c, int i, float x) const {
        const float t0 = 0.5f * c.x0 * c.x0 * inv_s2 + half_dm1 * c.x0;
        return (lead && i == 0) ? t0 : 0.5f * c.e * x * x;
    }
};

// ------------------------------------------------------------------------------------------------
// Row IO: register quad q of lane g <-> the 16 bytes at coordinate 4 * (q * LPC + g) of the row.
template <int CPL, int LPC, bool VEC>
__device__ __forceinline__ void load_row(const float* __restrict__ base, int64_t row, int d, int g, bool active,
                                         float (&x)[CPL]) {
    if constexpr (VEC) {
        const float4* p = reinterpret_cast<const float4*>(base + row * d) + g;
#pragma unroll
        for (int q = 0; q < CPL / 4; ++q) {
            float4 v = active ? p[q * LPC] : make_float4(0.f, 0.f, 0.f, 0.f);
            x[4 * q] = v.x;
            x[4 * q + 1] = v.y;
            x[4 * q + 2] = v.z;
            x[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = coord_of<CPL, LPC>(g, i);
            x[i] = (active && c < d) ? base[row * d + c] : 0.f;
        }
    }
}

template <int CPL, int LPC, bool VEC>
__device__ __forceinline__ void store_row(float* __restrict__ base, int64_t row, int d, int g, bool active,
                                          const float (&x)[CPL]) {
    if constexpr (VEC) {
        if (active) {
            float4* p = reinterpret_cast<float4*>(base + row * d) + g;
#pragma unroll
            for (int q = 0; q < CPL / 4; ++q)
                p[q * LPC] = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = coord_of<CPL, LPC>(g, i);
            if (active && c < d) base[row * d + c] = x[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Kept states (NfmcSampleStore): which transitions of a launch are kept, and in which ring row, is wave-uniform
// bookkeeping on scalars -- a countdown to the next kept transition and the row it goes to, advanced without divisions.
struct StoreCursor {
    float* base;
    int stride, countdown, ring, row;
    __device__ __forceinline__ explicit StoreCursor(const NfmcSampleStore& s)
        : base(s.base), stride(s.stride), countdown(s.countdown), ring(s.ring_rows), row(s.row) {}
    // the store row (pointer to its n*d floats) the CURRENT transition is kept in, or nullptr; advances to the next one
    __device__ __forceinline__ float* next(int64_t nd) {
        if (!base) return nullptr;
        if (countdown > 0) {
            --countdown;
            return nullptr;
        }
        float* p = base + (int64_t)row * nd;
        row = row + 1 == ring ? 0 : row + 1;
        countdown = stride - 1;
        return p;
    }
};
// the same for transition t of the call, out of order (the data-parallel IMH replay writes runs of equal states)
__device__ __forceinline__ float* store_row_of(const NfmcSampleStore& s, int t, int64_t nd) {
    const int k = t - s.countdown;
    if (!s.base || k < 0 || k % s.stride != 0) return nullptr;
    return s.base + (int64_t)((s.row + k / s.stride) % s.ring_rows) * nd;
}
// f(row pointer) for every KEPT transition t of the call in [t0, t1), in order: one division for the run instead of three
// per transition (the replay writes a state that lasted c steps into every kept row of those c: with thinning most of
// the c lookups found nothing, 0.8 ms of a 1.1 ms replay at the C2 shape with every 250th state kept)
template <class F>
__device__ __forceinline__ void store_rows_in(const NfmcSampleStore& s, int t0, int t1, int64_t nd, F f) {
    if (!s.base) return;
    int k0 = t0 - s.countdown;
    if (k0 < 0) k0 = 0;
    const int j = (k0 + s.stride - 1) / s.stride;             // first kept index at or after t0
    int row = (s.row + j) % s.ring_rows;
    for (int t = s.countdown + j * s.stride; t < t1; t += s.stride) {
        f(s.base + (int64_t)row * nd);
        row = row + 1 == s.ring_rows ? 0 : row + 1;
    }
}
// the store descriptor of the NEXT launch after one that offered k transitions (host side of StoreCursor)
inline void store_advance(NfmcSampleStore& s, int k) {
    if (!s.base) return;
    if (k <= s.countdown) {
        s.countdown -= k;
        return;
    }
    const int rest = k - s.countdown;                       // from the launch's first kept transition to its end
    const int kept = (rest + s.stride - 1) / s.stride;
    s.row = (s.row + kept) % s.ring_rows;
    s.countdown = s.stride - 1 - (rest - ((kept - 1) * s.stride + 1));
}
inline bool store_ok(const NfmcSampleStore& s) {
    return !s.base || (s.stride >= 1 && s.countdown >= 0 && s.countdown < s.stride && s.ring_rows >= 1 && s.row >= 0 &&
                       s.row < s.ring_rows);
}

// ------------------------------------------------------------------------------------------------
// Deterministic statistics.  Each lane carries fp32 partial sums over the <= 512 steps of one call;
// they are widened to fp64, reduced over the chains of the wave by shuffles, over the waves of the
// workgroup through LDS, written to the workgroup's slot of the scratch slab, and a second tiny
// kernel folds the slab into the accumulators in a fixed order (no atomics: run-to-run bitwise equal).
template <int CPL, int LPC>
__device__ __forceinline__ void block_stats_flush(const float (&sx)[CPL], const float (&sxx)[CPL], uint32_t accepted,
                                                  uint32_t nonfinite, const NfmcStats& st,
                                                  uint32_t jump_accepted = 0, uint32_t jump_nonfinite = 0,
                                                  const double* __restrict__ extra = nullptr) {
    // extra: 2 * DP sums another kernel prepared for this workgroup, added to its slab (imh_parallel.hip)
    double* __restrict__ scratch = st.scratch;
    const bool defer = st.defer != 0;
    const int slot = defer ? st.tail_slot : 0;   // deferred jumps book their counts in the jump slots
    constexpr int DP = CPL * LPC;
    __shared__ double red[kWavesPerBlock][2 * DP + kStatTail];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const double a = (double)cross_chain_reduce<LPC>(sx[i]);
        const double b = (double)cross_chain_reduce<LPC>(sxx[i]);
        if (lane < LPC) {
            red[wave][coord_of<CPL, LPC>(g, i)] = a;
            red[wave][DP + coord_of<CPL, LPC>(g, i)] = b;
        }
    }
    if (lane == 0) {
        red[wave][2 * DP + 0] = slot == 0 ? (double)accepted : 0.0;
        red[wave][2 * DP + 1] = slot == 0 ? (double)nonfinite : 0.0;
        red[wave][2 * DP + 2] = slot == 0 ? (double)jump_accepted : (double)accepted;
        red[wave][2 * DP + 3] = slot == 0 ? (double)jump_nonfinite : (double)nonfinite;
    }
    __syncthreads();
    double* out = scratch + (size_t)blockIdx.x * (2 * DP + kStatTail);
    for (int t = threadIdx.x; t < 2 * DP + kStatTail; t += kBlock) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) s += red[w][t];
        if (extra && t < 2 * DP) s += extra[t];
        out[t] = defer ? out[t] + s : s;   // one owner thread per (workgroup, column): no race, fixed order
    }
}

// scratch (nblocks, 2*dp + kStatTail) -> stats (+=).  Workgroup b owns 32 columns; its 256 threads are
// 32 columns x 8 row-slices (loads coalesced across columns, 8 independent chains per column), and the 8
// partials of a column are added in slice order, so the result does not depend on timing.
constexpr int kFinishCols = 32, kFinishSlices = 32, kFinishBlock = kFinishCols * kFinishSlices;
template <bool ZERO>
static __device__ __forceinline__ double take(double* p) {
    const double v = *p;
    if (ZERO) *p = 0.0;
    return v;
}

// ZERO: the slabs are zeroed as they are read, so the scratch is all-zero again after every fold.  Every launch site
// uses it: a fold-per-call launch leaves nothing behind that a later DEFERRED launch (which adds to the slabs) could
// pick up when the two modes are mixed within one run.
template <bool ZERO = false>
static __global__ void __launch_bounds__(kFinishBlock) stats_finish_kernel(double* __restrict__ scratch,
                                                                           int nblocks, int dp, int d, NfmcStats st,
                                                                           unsigned long long attempted,
                                                                           unsigned long long* jump_counters = nullptr,
                                                                           unsigned long long jump_attempted = 0) {
    __shared__ double part[kFinishSlices][kFinishCols];
    const int width = 2 * dp + kStatTail;
    const int col = threadIdx.x % kFinishCols, slice = threadIdx.x / kFinishCols;
    const int t = blockIdx.x * kFinishCols + col;
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;  // four loads in flight per thread, combined in fixed order
    if (t < width) {
        int b = slice;
        for (; b + 3 * kFinishSlices < nblocks; b += 4 * kFinishSlices) {
            p0 += take<ZERO>(scratch + (size_t)b * width + t);
            p1 += take<ZERO>(scratch + (size_t)(b + kFinishSlices) * width + t);
            p2 += take<ZERO>(scratch + (size_t)(b + 2 * kFinishSlices) * width + t);
            p3 += take<ZERO>(scratch + (size_t)(b + 3 * kFinishSlices) * width + t);
        }
        for (; b < nblocks; b += kFinishSlices) p0 += take<ZERO>(scratch + (size_t)b * width + t);
    }
    part[slice][col] = (p0 + p1) + (p2 + p3);
    __syncthreads();
    if (slice == 0 && t < width) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < kFinishSlices; ++k) s += part[k][col];
        if (t < dp) {
            if (t < d) st.sum_x[t] += s;
        } else if (t < 2 * dp) {
            if (t - dp < d) st.sum_x2[t - dp] += s;
        } else if (t == 2 * dp) {
            st.counters[NFMC_CNT_ACCEPTED] += (unsigned long long)(s + 0.5);
            st.counters[NFMC_CNT_ATTEMPTED] += attempted;
        } else if (t == 2 * dp + 1) {
            st.counters[NFMC_CNT_NONFINITE] += (unsigned long long)(s + 0.5);
        } else if (t == 2 * dp + 2 && jump_counters) {
            jump_counters[NFMC_CNT_ACCEPTED] += (unsigned long long)(s + 0.5);
            jump_counters[NFMC_CNT_ATTEMPTED] += jump_attempted;
        } else if (t == 2 * dp + 3 && jump_counters) {
            jump_counters[NFMC_CNT_NONFINITE] += (unsigned long long)(s + 0.5);
        }
    }
}

inline int stats_finish_grid(int dp) { return (2 * dp + kStatTail + kFinishCols - 1) / kFinishCols; }

inline int64_t stats_scratch_doubles(int dp) { return (int64_t)kMaxGrid * (2 * dp + kStatTail); }

inline int padded_d(int d);

// Deferred statistics (NfmcStats.defer): every kernel of a run must lay its slabs out with the same width,
// 2 * padded_d(d) + kStatTail, and the scratch must hold kMaxGrid of them (nfmc_stats_fold_f32 reads them all).
inline int check_defer(const NfmcStats& s, int dp, int d) {
    if (!s.sum_x || !s.defer) return 0;
    if (dp != padded_d(d) || (s.tail_slot != 0 && s.tail_slot != 2)) return -1;
    if (s.scratch_bytes < stats_scratch_doubles(dp) * (int64_t)sizeof(double)) return -1;
    return 0;
}

inline int padded_d(int d) {
    int p = 4;
    while (p < d) p <<= 1;
    return p;
}

#define NFMC_HIP_CHECK_LAUNCH()                \
    do {                                       \
        hipError_t e_ = hipGetLastError();     \
        if (e_ != hipSuccess) return (int)e_;  \
    } while (0)

}  // namespace nfmc
