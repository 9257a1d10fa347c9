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

// Workgroup shape: 8 waves of 16 chains (128 chains per workgroup), two LDS weight images used alternately (ONE barrier per
// GEMM), one workgroup per CU.  Measured alternatives, not kept: two 4-wave workgroups per CU with one image each (round 2:
// 3.72 vs 3.50 ms, every workgroup re-stages every operand for half as many chains); 4 waves with TWO 16-chain tiles each,
// one wave per SIMD with the whole 512-entry register file, the next image copied by LDS-DMA woven into the MFMA stream
// (round 3: 3.0-3.4 vs 2.28 ms -- tools/experiments/r03_c4_two_tiles_per_wave.patch).
#ifndef NFMC_STAGE_BATCH
#define NFMC_STAGE_BATCH 4
#endif
#ifndef NFMC_FRAG_CHUNK
#define NFMC_FRAG_CHUNK 4
#endif
constexpr int kMfmaWaves = 8;
constexpr int kMfmaImages = 2;
constexpr int kMfmaBlock = 64 * kMfmaWaves;
constexpr int kMfmaChains = 16 * kMfmaWaves;     // chains per workgroup
// Threads that stage weight images.  Of the two waves of a SIMD the OLDER (waves 0..3 of an 8-wave workgroup) wins the
// issue arbitration, finishes every GEMM phase first and then waits for its partner (timeline: tools/trace_c4.py: at the
// C4 shape the old wave needs ~14.5k cycles for a 256-MFMA phase, the young one ~21.7k, the last third of it alone on
// the SIMD).  So the whole copy belongs to the old waves -- it runs under the partner's MFMAs -- and the young waves
// go from their last MFMA straight to the barrier: C4 2.66 -> 2.47 ms per trajectory.
constexpr int kStageThreads = kMfmaBlock / 2;
#ifndef NFMC_CK_MAX_GRID
#define NFMC_CK_MAX_GRID 256
#endif
constexpr int kCkMaxGrid = NFMC_CK_MAX_GRID;     // workgroup slots of the NeuTra trajectory kernel (1 per CU: 256 vs 512 measured +0.7 %, half the scratch); more chain tiles grid-stride

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
template <int K, int RBLK, int CBLK, int ROWS>
__device__ __forceinline__ void stage_matrix(float* __restrict__ img, const float* __restrict__ W, bool rev_rows,
                                             bool rev_cols) {
    constexpr int ld = K + 4, k4 = K >> 2, N = ROWS * k4;
    static_assert((K & (K - 1)) == 0 && (RBLK & (RBLK - 1)) == 0 && (CBLK & (CBLK - 1)) == 0, "power-of-two extents");
    constexpr int kStagers = kStageThreads;       // threads that copy (see kStageThreads)
    static_assert(N % kStagers == 0, "whole passes of the staging threads");
    constexpr int IT = N / kStagers;              // 128-bit pieces per thread: 8 for a 128 x 128 matrix and 512 threads
    constexpr int BATCH = IT < NFMC_STAGE_BATCH ? IT : NFMC_STAGE_BATCH;   // loads in flight per thread (4 registers each)
    // The copy addresses depend only on the thread index, so LICM would hoist the address arithmetic of EVERY
    // staging call of the kernel above the tile loop and keep hundreds of VGPRs live through all the GEMMs
    // (measured: ~2 KB of scratch per lane).  An opaque copy of the index pins the arithmetic to the call.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    if constexpr (kStagers < kMfmaBlock) {
        if (__builtin_amdgcn_readfirstlane(tid) >= kStagers) return;
    }
    // The weights sit in L2, so a staging is latency, not bandwidth: as a plain loop every piece was load -> wait ->
    // store (8 dependent L2 round trips per 128 x 128 operand, ~10k cycles in which both waves of every SIMD sat idle:
    // the largest single loss of the first version, MFMA pipe busy 59 %).  Loads are issued BATCH at a time before
    // the first store.
    // Index arithmetic once per call, in unsigned 32 bits: the staging threads are a whole number of image rows
    // (kStagers % k4 == 0), so a thread keeps its column and walks down RSTEP rows per piece -- the global offset and
    // the LDS offset of piece j are the thread's base plus a compile-time constant.  (With signed `idx / k4`,
    // `idx % k4` and 64-bit addresses per piece the copy cost ~16 VALU instructions per 16-byte piece: more VALU work
    // per gradient on the staging waves than all the elementwise epilogues together.)
    static_assert(kStagers % k4 == 0, "staging threads cover whole rows");
    constexpr uint32_t RSTEP = kStagers / k4;
    const uint32_t t = (uint32_t)tid;
    const uint32_t c = (t % (uint32_t)k4) << 2, r0 = t / (uint32_t)k4;          // r0 < RSTEP
    const uint32_t g0 = r0 * (uint32_t)K + c;
    uint32_t cc = c;
    if (rev_cols) cc = (c / CBLK) * CBLK + (CBLK - 4 - (c % CBLK));
#pragma unroll
    for (int b0 = 0; b0 < IT; b0 += BATCH) {
        f32x4 v[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {   // 32-bit BYTE offset from the uniform base: scalar-base addressing, one add per piece
            const uint32_t gb = (g0 << 2) + (uint32_t)((b0 + b) * RSTEP * K * 4);
            v[b] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(W) + gb);
        }
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            constexpr uint32_t kR = RBLK;
            const uint32_t rj = (uint32_t)(b0 + b) * RSTEP;                        // compile-time after unrolling
            uint32_t rr;
            if constexpr (kR % RSTEP == 0) {
                // reversal inside blocks of RBLK rows: (rj + r0) % RBLK = rj % RBLK + r0 (no carry: r0 < RSTEP | RBLK)
                rr = rev_rows ? (rj / kR) * kR + (kR - 1 - (rj % kR)) - r0 : rj + r0;
            } else {
                const uint32_t r = rj + r0;
                rr = rev_rows ? (r / kR) * kR + (kR - 1 - (r % kR)) : r;
            }
            f32x4 w = v[b];
            if (rev_cols) {
                w[0] = v[b][3];
                w[1] = v[b][2];
                w[2] = v[b][1];
                w[3] = v[b][0];
            }
            *reinterpret_cast<f32x4*>(img + (rr * (uint32_t)ld + cc)) = w;
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

#ifdef NFMC_TRACE_STEPS
// Diagnostic build (tools/trace_c4.py, with NFMC_TRACE): per-step marks of workgroup 0 of the trajectory kernel, one
// stream per wave behind the phase marks (30 = a step's reads begin, 31 = its MFMAs are issued, the epilogue follows).
// The kernel arms it (pointers + magic in static LDS); other kernels of the build leave the magic unset or stale -- the
// buffer they would then write to is the still-allocated trace area.
__shared__ unsigned long long* g_step_buf[8];
__shared__ int g_step_cnt[8];
__shared__ unsigned g_step_magic;
__device__ __forceinline__ void step_mark(int id) {
    if ((threadIdx.x & 63) == 0 && g_step_magic == 0x5A17C0DEu) {
        const int w = threadIdx.x >> 6;
        unsigned long long* b = g_step_buf[w];
        const int k = g_step_cnt[w];
        if (b && k < 8192) {
            b[k] = ((unsigned long long)id << 48) | (__builtin_amdgcn_s_memtime() & 0xFFFFFFFFFFFFull);
            g_step_cnt[w] = k + 1;
        }
    }
}
#endif

// ---- a GEMM phase = NSTEP steps, each: read this lane's A fragments of one 16-row output block from the LDS image,
// 4 * TK MFMAs into one accumulator quad, an elementwise epilogue.
//   row(i)   LDS address of the A fragments of step i (this lane's row / column group)
//   init(i)  set up the accumulator of step i        acc(i)  the accumulator (f32x4&)
//   act(i)   the B operand tiles (f32x4[TK])          fin(i)  the elementwise epilogue of step i
// Order: reads(i), MFMAs(i), epilogue(i), one step at a time.  Round 2 measured three other orders of the same work at the
// C4 shape (tools/ab_c4.sh, all bitwise equal in their results; 3.53 ms per trajectory for this one at the time):
// epilogue(i - 1) woven between the MFMAs of step i (sched_group_barrier) 3.54; reads(i + 1) issued before MFMAs(i) with two
// fragment sets 3.60; steps in pairs with two interleaved accumulators 3.66.  None beat the plain order and they were
// removed: ablations of the same build (no fragment reads: 3.43 ms; no MFMAs and no reads: 0.87 ms) and a register-only
// MFMA loop (153-156 TFLOP/s on one dependent accumulator per wave, tools/ubench_mfma.hip) showed that neither LDS latency
// nor the accumulation chain idles the matrix pipe.  What does (timeline of tools/trace_c4.py): of the two
// waves of a SIMD the older wins the issue arbitration, finishes a phase early, stages the next image and waits at the
// barrier; the younger runs the last third of the phase alone at ~75-80 % MFMA density.  (Round 1 had read this as "the
// barriers keep the eight waves in lock-step"; the timeline showed they are a third of a phase apart.)
template <int TK>
__device__ __forceinline__ void frag_load(f32x4 (&a)[TK], const float* __restrict__ arow) {
#pragma unroll
    for (int mk = 0; mk < TK; ++mk) a[mk] = *reinterpret_cast<const f32x4*>(arow + 16 * mk);
}
template <int TK, int NSTEP, class Row, class Init, class Acc, class Act, class Fin>
__device__ __forceinline__ void gemm_phase(Row row, Init init, Acc acc, Act act, Fin fin) {
    // A step's A fragments are read in chunks of at most kFragChunk k-tiles (4 registers each): the scheduler may move the
    // NEXT chunk's reads above this chunk's MFMAs (the fence stops anything further), so the fragments cost
    // 2 x 4 x kFragChunk registers at the peak -- 32 instead of 64 for an 8-tile step: the kernel lives at the 256-register
    // limit of two waves per SIMD and spills cost more than the shorter look-ahead (16 MFMAs = 512 cycles still cover an
    // LDS read).
    constexpr int CH = TK < NFMC_FRAG_CHUNK ? TK : NFMC_FRAG_CHUNK;
    static_assert(TK % CH == 0, "whole chunks");
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {   // one step at a time: reads, then MFMAs, then epilogue
#ifdef NFMC_TRACE_STEPS
        step_mark(30);
#endif
        init(i);
#pragma unroll
        for (int c0 = 0; c0 < TK; c0 += CH) {
            f32x4 a[CH];
            frag_load<CH>(a, row(i) + 16 * c0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mk = 0; mk < CH; ++mk)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc(i) = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mk][r], act(i)[c0 + mk][r], acc(i), 0, 0, 0);
        }
#ifdef NFMC_TRACE_STEPS
        step_mark(31);
#endif
        fin(i);
    }
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
constexpr int kMfmaStatOffset = kMfmaImages * (kImgFloats + kVecFloats);   // floats before the per-wave statistics
constexpr size_t kMfmaLdsBytes = (size_t)kMfmaStatOffset * sizeof(float) + kMfmaStatDoubles * sizeof(double);

// entry points of neutra_mfma.hip used by the C ABI in neutra_kernels.hip
int nfmc_mfma_supported(int32_t d, int32_t n_hidden, int32_t n_hidden_layers);
int nfmc_neutra_potential_grad_mfma_f32(const NfmcRealNVP* flow, const NfmcPotential* pot, const float* z, int64_t n,
                                        float* u_out, float* grad_out, nfmc_stream_t stream);
int nfmc_neutra_hmc_steps_mfma_f32(const NfmcNeutraHmcArgs* args, float* scratch, int64_t scratch_bytes,
                                   nfmc_stream_t stream);
// entry points of mfma_wide.hip: d = 256 / 512 with the state streamed through a scratch slab
int nfmc_mfma_wide_supported(int32_t d, int32_t n_hidden, int32_t n_hidden_layers);
int nfmc_neutra_potential_grad_wide_f32(const NfmcRealNVP* flow, const NfmcPotential* pot, const float* z, int64_t n,
                                        float* u_out, float* grad_out, nfmc_stream_t stream);
int64_t nfmc_neutra_wide_scratch_floats(int64_t n, int32_t d);
int64_t nfmc_wide_slab_floats(int64_t n, int32_t d, int32_t copies);
int nfmc_neutra_hmc_steps_wide_f32(const NfmcNeutraHmcArgs* args, float* scratch, int64_t scratch_bytes, nfmc_stream_t stream);
int nfmc_realnvp_forward_wide_f32(const NfmcRealNVP* flow, const float* x, int64_t n, float* z, float* logdet, float* log_prob,
                                  nfmc_stream_t stream);
int nfmc_realnvp_inverse_wide_f32(const NfmcRealNVP* flow, const float* z, int64_t n, float* x, float* logdet, float* log_q,
                                  const NfmcRng* rng, nfmc_stream_t stream);
int nfmc_flow_mh_steps_wide_f32(const NfmcFlowMhArgs& a, nfmc_stream_t stream, int* grid_out, int* dp_out);
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
#ifdef NFMC_TRACE
    // Diagnostic build (tools/trace_c4.py): workgroup 0 writes (id << 48 | s_memtime) marks, one stream per wave
    unsigned long long* tr = nullptr;
    __device__ __forceinline__ void mark(int id) {
        if (tr) {
            if ((threadIdx.x & 63) == 0) *tr = ((unsigned long long)id << 48) | (__builtin_amdgcn_s_memtime() & 0xFFFFFFFFFFFFull);
            ++tr;
        }
    }
#else
    __device__ __forceinline__ void mark(int) {}
#endif
    __device__ __forceinline__ float* img() const { return lds + buf * kImgFloats; }
    __device__ __forceinline__ float* vec() const { return lds + kMfmaImages * kImgFloats + buf * kVecFloats; }
    // stage matrix W (rows x K) and, if given, a vector, into the next image; closes with the barrier
    template <int K, int RBLK, int CBLK, int VBLK, int ROWS>
    __device__ __forceinline__ void stage(const float* W, bool rev_rows, bool rev_cols, const float* v, int vlen,
                                          bool vrev) {
        mark(1);
        buf ^= 1;
        stage_matrix<K, RBLK, CBLK, ROWS>(img(), W, rev_rows, rev_cols);
        if (v) stage_vector<VBLK>(vec(), v, vlen, vrev);
        mark(2);
        __syncthreads();
        mark(3);
    }
};

// Conditioner hidden stack in C layout: h1 = tanh(W1 x_src + b1), h2 = tanh(Wh h1 + bh).  Workgroup-collective
// (two steps of the weight pipeline).  `src` are the source tiles of the layer input (TS = d_a / 16).
template <int TS, int TH, int NHL>
__device__ __forceinline__ void hidden_stack(const f32x4 (&src)[TS], f32x4 (&h1)[TH], f32x4 (&h2)[TH], const MLayer& L,
                                             bool rev, WeightPipe& wp, int col, int q) {
    constexpr int hp = 16 * TH, d_a = 16 * TS;
    wp.template stage<d_a, 1, d_a, 1, hp>(L.W1, false, rev, L.b1, hp, false);
    {
        const float* img = wp.img();
        const float* vec = wp.vec();
        gemm_phase<TS, TH>([&](int mo) { return img + (16 * mo + col) * (d_a + 4) + 4 * q; },
                           [&](int mo) { h1[mo] = vec_tile(vec, mo, q); },
                           [&](int mo) -> f32x4& { return h1[mo]; },
                           [&](int) -> const f32x4(&)[TS] { return src; },
                           [&](int mo) { h1[mo] = tanh4(h1[mo]); });
    }
    if constexpr (NHL > 1) {
        wp.template stage<hp, 1, 1, 1, hp>(L.Wh, false, false, L.bh, hp, false);
        const float* img = wp.img();
        const float* vec = wp.vec();
        gemm_phase<TH, TH>([&](int mo) { return img + (16 * mo + col) * (hp + 4) + 4 * q; },
                           [&](int mo) { h2[mo] = vec_tile(vec, mo, q); },
                           [&](int mo) -> f32x4& { return h2[mo]; },
                           [&](int) -> const f32x4(&)[TH] { return h1; },
                           [&](int mo) { h2[mo] = tanh4(h2[mo]);                            });
    }
}

}  // namespace nfmc
