// FixedIMH as a data-parallel problem (imh.py:200-255).  An independence sampler's proposals do not depend on the
// chain's state, so a run of k transitions of n chains is
//   A  k*n independent proposal evaluations  x' = f^-1(z),  f' = log q(x'),  u' = U(x')      (imh_eval_kernel)
//   B  one cheap sequential scan per chain over (u', f') with the accept uniforms               (imh_scan_kernel)
//   C  one replay of the ACCEPTED proposals, weighted by how long each stayed the state, for the moments, the
//      sample store and the final state                                                            (imh_replay_kernel)
// instead of k dependent transitions per chain.  Few chains (the reference's default is 100) no longer leave the GPU
// idle: A and C fill it with k*n work items.  Noise streams, arithmetic and results are those of
// nfmc_flow_mh_steps_f32 (same Philox counters per (chain, step); states and accept masks bit for bit, moments up to
// the order of summation).  Register-layout flow kernels (flow_b.hpp): conditioners of width <= 8.
#pragma once

#include "flow_b.hpp"

namespace nfmc {

constexpr int32_t kLastFlag = 1 << 30;   // replay: set in the dwell time of the proposal that is a chain's final state
constexpr int kScanBlock = 28;           // steps per accept-mask word: two blocks of records in flight per lane + the mask stores stay under the 63
                                         // operations the memory counter tracks (at 32 the compiler drains the queue every iteration)
// accept-mask words per chain: an even number of blocks (the scan works on two at a time)
__host__ __device__ inline int imh_words(int k) { return ((k + 2 * kScanBlock - 1) / (2 * kScanBlock)) * 2; }

struct ImhWork {
    // step-major (k, n): row r = s * n + i.  The scan runs one chain per LANE, so the 64 lanes of a wave read 64
    // consecutive words of one step; the proposal and replay kernels only need consecutive rows to be cheap to
    // enumerate.  (Rounds 1-2 scanned one chain per WAVE over chain-major arrays: 173 us at C2 with a quarter of the
    // proposals accepted, 600 us with the fitted flow's 86 %, sequential per acceptance while 63 lanes idled.)
    float4* rec;     // (k, n) {U(x'), log q(x'), log of the accept uniform of (chain, step), -}: one 16-byte load per step
    uint32_t* bits;  // (words, n) accept masks: bit b of word q of chain i = proposal (kScanBlock q + b, i) was accepted.  The replay
                     // reads a proposal's dwell time (the steps it stayed the state) off them: the distance to the next set bit
    int32_t* dwell0; // (n)    the steps the initial state lasted (the first accepted step, k when none)
    float* x0;       // (n, d) copy of the initial states
    double* esum;    // (grid of the proposal kernel, 2 * dp) its workgroups' sums of x' and x'^2 over ALL finite proposals
    unsigned long long* visit;   // [0] rows the replay visits when it sums the accepted proposals, [1] when it corrects esum
};

// first row of a wave's 64 -> (step, chain): one 32-bit division per 64 rows (a 64-bit one is ~180 VALU instructions
// on gfx950; every launch the host makes has n * k < 2^31, `small`)
struct RowBase {
    int64_t i;
    int s;
};
__device__ __forceinline__ RowBase row_base(int64_t r0, int64_t n, bool small) {
    RowBase b;
    if (small) {
        const uint32_t q = (uint32_t)r0 / (uint32_t)n;
        b.s = (int)q;
        b.i = (int64_t)((uint32_t)r0 - q * (uint32_t)n);
    } else {
        b.s = (int)(r0 / n);
        b.i = r0 - (int64_t)b.s * n;
    }
    return b;
}
// row r0 + off (off < 64) -> (chain i, step s)
__device__ __forceinline__ void split_row(const RowBase& b, int off, int64_t n, int64_t& i, int& s) {
    i = b.i + off;
    s = b.s;
    if (n >= 64) {
        if (i >= n) i -= n, ++s;
    } else {
        const uint32_t q = (uint32_t)i / (uint32_t)n;
        s += (int)q;
        i -= (int64_t)q * n;
    }
}

// proposal (s, i): latent from the chain's stream, inverse pass, log q and potential.  All lanes of the row group.
template <int CPL, int LPC, int HP, class FlowT, class PotT>
__device__ __forceinline__ void imh_propose(float (&xp)[CPL], float& f_xp, float& u_xp, const NfmcFlowMhArgs& a,
                                            const FlowT& fl, const PotT& pot, int64_t i, int s, int g, bool revl,
                                            float base_c) {
    const int d = a.flow.d;
    const uint32_t gchain = (uint32_t)(a.rng.chain_offset + (uint64_t)i);
    draw_latent<CPL, LPC, FlowT::kExact>(xp, a.rng.replay_normals ? a.rng.replay_normals + (int64_t)s * a.n * d : nullptr, a.rng.seed,
                          a.rng.step0 + (uint32_t)s, gchain, i, a.n, d, g, revl);          // flow.sample: imh.py:221
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) part = fmaf(-0.5f * xp[k], xp[k], part);
    part -= fl.inverse(xp);
    f_xp = group_allreduce<LPC>(part) + base_c;
    float up = 0.f;
    const auto ctx = pot.prepare(xp, g, d);
#pragma unroll
    for (int k = 0; k < CPL; ++k) up += pot.term(ctx, k, xp[k]);
    u_xp = group_allreduce<LPC>(up);                                                          // imh.py:225
}

__device__ __forceinline__ unsigned long long wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, kWave);
    return (unsigned long long)v;
}

// a workgroup's sums of x' and x'^2 (fp32 per lane -> fp64 over the chains of the wave and the waves of the workgroup,
// fixed order) into its slot of ImhWork::esum
template <int CPL, int LPC>
__device__ __forceinline__ void block_sums_store(const float (&sx)[CPL], const float (&sxx)[CPL], double* __restrict__ out) {
    constexpr int DP = CPL * LPC;
    __shared__ double red[kWavesPerBlock][2 * DP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const double p = (double)cross_chain_reduce<LPC>(sx[i]);
        const double q = (double)cross_chain_reduce<LPC>(sxx[i]);
        if (lane < LPC) {
            red[wave][coord_of<CPL, LPC>(g, i)] = p;
            red[wave][DP + coord_of<CPL, LPC>(g, i)] = q;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * DP; t += kBlock) {
        double v = 0.0;
#pragma unroll
        for (int wv = 0; wv < kWavesPerBlock; ++wv) v += red[wv][t];
        out[t] = v;
    }
}

// a proposal enters the proposal kernel's running sums when its potential and log density are finite (then so is x');
// the replay recomputes both bit for bit and applies the same test when it corrects those sums
__device__ __forceinline__ bool imh_summed(float u_xp, float f_xp) { return fabsf(u_xp) <= 3.0e38f && fabsf(f_xp) <= 3.0e38f; }

// SUMS: keep the running sums of x' and x'^2 for a correcting replay.  Off when the replay cannot correct (no statistics
// asked for, or a sample store, which needs every kept row anyway -- the reference's default): the kernel then needs 16
// registers less and runs at five waves per SIMD instead of four (C2 shape: 0.95 vs 1.02 ms).
template <int CPL, int LPC, int HP, template <int, int, bool> class Pot, bool FAST, int NB = 0, bool SUMS = true>
__global__ void __launch_bounds__(kBlock, (!SUMS && NB == 0 && CPL == 8 && HP == 4 && FAST) ? 5 : 1) imh_eval_kernel(NfmcFlowMhArgs a, ImhWork w, int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CPW = kWave / LPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC, cw = lane / LPC;
    const int d = a.flow.d;
    const int64_t n = a.n, total = n * (int64_t)a.n_steps;
    const bool small = total < (1ll << 31);
    using Flow = FlowB<CPL, LPC, HP, false, (FAST && CPL >= 8), NB>;
    Flow::Img::stage(lds, a.flow, kBlock);
    __syncthreads();
    Flow fl;
    fl.init(lds, a.flow, g);
    Pot<CPL, LPC, FAST> pot;
    pot.init(a.pot, g, d);
    const bool revl = (a.flow.n_coupling & 1) != 0;
    const float base_c = -0.5f * (float)d * kLog2Pi;
    float sx[SUMS ? CPL : 1], sxx[SUMS ? CPL : 1];
#pragma unroll
    for (int q = 0; q < (SUMS ? CPL : 1); ++q) sx[q] = sxx[q] = 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) w.visit[0] = w.visit[1] = 0ull;   // the scan (next on the stream) counts into them
    // a wave owns 64 consecutive rows per tile: first every lane draws the accept uniform of ONE row (one Philox call
    // per row instead of one per lane and row: 8 % of the kernel at LPC = 8), then LPC passes evaluate CPW rows each
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t r0 = (tile * kWavesPerBlock + wave) * kWave;
        if (r0 >= total) continue;                                    // wave-uniform
        const RowBase rb0 = row_base(r0, n, small);
        float logu_lane = 0.f;
        {
            const int64_t r = r0 + lane;
            if (r < total) {
                int64_t i;
                int s;
                split_row(rb0, lane, n, i, s);
                float uu;   // the accept uniform of (chain i, step s): imh.py:229
                if (a.rng.replay_uniforms) {
                    uu = a.rng.replay_uniforms[r];
                } else {
                    const uint4 rnd = philox4x32_10((uint32_t)(a.rng.chain_offset + (uint64_t)i), a.rng.step0 + (uint32_t)s, 0u,
                                                    kTagJump, (uint32_t)a.rng.seed, (uint32_t)(a.rng.seed >> 32));
                    uu = u32_to_uniform(rnd.x);
                }
                logu_lane = fast_ln(uu);
            }
        }
        for (int sub = 0; sub < LPC; ++sub) {
            const int64_t rb = r0 + (int64_t)sub * CPW;
            if (rb >= total) break;                                   // wave-uniform
            const int64_t r = rb + cw;
            const bool active = r < total;
            int64_t i;
            int s;
            split_row(rb0, active ? sub * CPW + cw : (int)(total - 1 - r0), n, i, s);
            float xp[CPL], f_xp, u_xp;
            imh_propose<CPL, LPC, HP>(xp, f_xp, u_xp, a, fl, pot, i, s, g, revl, base_c);
            const float logu_row = __shfl(logu_lane, sub * CPW + cw, kWave);   // from the lane that drew this row's uniform
            if (active && g == 0) w.rec[r] = make_float4(u_xp, f_xp, logu_row, 0.f);
            if constexpr (SUMS) {
                if (active && imh_summed(u_xp, f_xp)) {
#pragma unroll
                    for (int q = 0; q < CPL; ++q) {
                        sx[q] += xp[q];
                        sxx[q] = fmaf(xp[q], xp[q], sxx[q]);
                    }
                }
            }
        }
    }
    if constexpr (SUMS) block_sums_store<CPL, LPC>(sx, sxx, w.esum + (size_t)blockIdx.x * (2 * CPL * LPC));
}

// One LANE per chain: the Metropolis scan over the k proposals (imh.py:223-233).  Everything expensive (proposal,
// uniform, logarithm) was done in parallel by imh_eval_kernel; what is left per step is the log ratio in the sequential
// kernel's order of operations, the test, two selects and one bit of the chain's accept mask.
//
// Memory: the records of a block of kScanBlock steps are requested one whole block of arithmetic before their use (two
// register rings, A and B, in turn).  The requests and the waits are written out (inline `global_load_dwordx3` +
// `s_waitcnt vmcnt(N)`): left to the compiler, the same loop either waited for ALL requests at the top of every
// iteration, the one just issued included (rings live across the back edge), or had its loads sunk to just before
// their use.  N counts what may still be in flight behind the ring that is needed: the other ring's kScanBlock
// requests and the one mask store between them.  With the optional per-step outputs (OUT) the stores in between are
// not counted here and the wait is for everything.
//
// Rounds 1-2 scanned one chain per WAVE (173 us at C2 with a quarter of the proposals accepted, 600 us with the
// fitted flow's 86 %); a lane per chain with one dwell-time store per step and compiler-scheduled loads was 136-203 us.
typedef float ScanRec __attribute__((ext_vector_type(3)));   // {u', f', log uniform}

struct ScanState {
    float u_x, f_x;
    int first;                 // first accepted step, -1: none yet
    uint32_t prev_top;         // the previous word's top bit
    uint32_t n_acc, n_bad, n_pairs;   // n_pairs: accepted proposals replaced after ONE step (the rows the correcting replay skips)
};

// rows of ImhWork::rec: the scan requests up to two blocks past the last word's and never looks at what it got there
__host__ __device__ inline int64_t imh_rec_rows(int k) { return (int64_t)(imh_words(k) + 2) * kScanBlock; }

__device__ __forceinline__ void scan_request(ScanRec (&ring)[kScanBlock], const float4* __restrict__ rec_blk, uint32_t lane_off, int64_t n) {
    const char* row = (const char*)rec_blk;      // wave-uniform: an SGPR pair, bumped by one row per request
#pragma unroll
    for (int j = 0; j < kScanBlock; ++j) {
        asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(ring[j]) : "v"(lane_off), "s"(row) : "memory");
        row += n * (int64_t)sizeof(float4);
    }
}

// the ring's requests have landed when at most N younger memory operations are in flight (they complete in order)
template <int N>
__device__ __forceinline__ void scan_arrived(ScanRec (&ring)[kScanBlock]) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#pragma unroll
    for (int j = 0; j < kScanBlock; ++j) asm volatile("" : "+v"(ring[j]));   // every use of the ring follows the wait
}

template <bool OUT>
__device__ __forceinline__ void scan_block(const ScanRec (&ring)[kScanBlock], ScanState& st, const NfmcFlowMhArgs& a, uint32_t* __restrict__ bits,
                                           int blk, int k, int64_t n, int64_t ic) {
    uint32_t word = 0u;
#pragma unroll
    for (int j = 0; j < kScanBlock; ++j) {
        const int s = blk * kScanBlock + j;
        const bool live = s < k;                                          // uniform
        const ScanRec p = ring[j];
        const float lr = (-p.x) - (-st.u_x) + st.f_x - p.y;              // util.py:392
        const bool acc = live && p.z < lr;                                // imh.py:229-230; NaN -> reject
        st.n_bad += live && !(fabsf(lr) <= 3.0e38f) ? 1u : 0u;
        if (OUT) {
            const int64_t ro = (int64_t)(live ? s : k - 1) * n + ic;
            if (live && a.masks_out) a.masks_out[ro] = acc ? 1 : 0;
            if (live && a.log_ratio_out) a.log_ratio_out[ro] = lr;
        }
        word |= acc ? (1u << j) : 0u;
        st.u_x = acc ? p.x : st.u_x;
        st.f_x = acc ? p.y : st.f_x;
    }
    bits[(int64_t)blk * n + ic] = word;   // the ONE compiler-visible memory operation of a block (scan_arrived counts it)
    st.n_acc += (uint32_t)__popc(word);
    st.n_pairs += (uint32_t)__popc(word & (word >> 1)) + (st.prev_top & word & 1u);
    st.prev_top = word >> (kScanBlock - 1);
    st.first = (st.first < 0 && word != 0u) ? blk * kScanBlock + (__ffs((int)word) - 1) : st.first;
}

template <bool OUT>
__global__ void __launch_bounds__(kWave) imh_scan_kernel(NfmcFlowMhArgs a, ImhWork w) {
    const int64_t n = a.n;
    const int d = a.flow.d;
    const int k = a.n_steps;
    const int64_t i = (int64_t)blockIdx.x * kWave + threadIdx.x;
    const bool valid = i < n;
    const int64_t ic = valid ? i : n - 1;   // lanes past the last chain redo chain n - 1 and store the same words (no branch below)
    ScanState st;
    st.u_x = potential_row(w.x0 + ic * d, a.pot, d);   // imh.py:224
    st.f_x = a.logq[ic];                                // imh.py:214 (filled by the caller when not cached)
    st.first = -1;
    st.prev_top = st.n_acc = st.n_bad = st.n_pairs = 0u;
    const int words = imh_words(k);
    const float4* __restrict__ rec0 = w.rec + (int64_t)blockIdx.x * kWave;          // this wave's 64 chains, row 0
    const uint32_t lane_off = (uint32_t)(ic - (int64_t)blockIdx.x * kWave) * (uint32_t)sizeof(float4);
    const int64_t blk_rows = (int64_t)kScanBlock * n;
    constexpr int kBehind = OUT ? 0 : kScanBlock + 1;   // the other ring's requests + one mask store
    ScanRec ring_a[kScanBlock], ring_b[kScanBlock];
    __builtin_amdgcn_s_waitcnt(0);                      // nothing of the prologue in flight: the counts below start from zero
    scan_request(ring_a, rec0, lane_off, n);
    scan_request(ring_b, rec0 + blk_rows, lane_off, n);
    scan_arrived<0>(ring_a);
    scan_arrived<0>(ring_b);
    scan_block<OUT>(ring_a, st, a, w.bits, 0, k, n, ic);
    for (int blk = 0; blk < words; blk += 2) {
        scan_request(ring_a, rec0 + (blk + 2) * blk_rows, lane_off, n);
        scan_block<OUT>(ring_b, st, a, w.bits, blk + 1, k, n, ic);
        scan_request(ring_b, rec0 + (blk + 3) * blk_rows, lane_off, n);
        scan_arrived<kBehind>(ring_a);
        scan_block<OUT>(ring_a, st, a, w.bits, blk + 2, k, n, ic);      // past the end: no live step, a zero word
        scan_arrived<OUT ? 0 : 1>(ring_b);
    }
    w.dwell0[ic] = st.first < 0 ? k : st.first;
    a.logq[ic] = st.f_x;                                              // imh.py:233
    if (!valid) st.n_acc = st.n_bad = st.n_pairs = 0u;
    // integer counters: atomic adds are exact, the totals do not depend on the order
    const unsigned long long acc_w = wave_sum_u32(st.n_acc), bad_w = wave_sum_u32(st.n_bad), pair_w = wave_sum_u32(st.n_pairs);
    const unsigned long long chains_w = wave_sum_u32(valid ? 1u : 0u);
    if (threadIdx.x == 0) {
        if (a.stats.counters) {
            if (acc_w) atomicAdd(a.stats.counters + NFMC_CNT_ACCEPTED, acc_w);
            if (bad_w) atomicAdd(a.stats.counters + NFMC_CNT_NONFINITE, bad_w);
        }
        // rows a summing replay visits: the accepted proposals; a correcting one: all but those that lasted one step
        if (acc_w) atomicAdd(w.visit + 0, acc_w);
        atomicAdd(w.visit + 1, chains_w * (unsigned long long)k - pair_w);
    }
}

// The two words of accept masks a lane of the replay needs for its row (the row's step and the next block's), loaded
// together.  (Requesting them one tile ahead, behind the flow passes of the tile in hand, cost three registers, a
// wave of occupancy, and gained nothing: 477 us either way.)
struct ImhLook {
    uint32_t w0, w1;
};

// more than kScanBlock rejections after an accepted step: rare, out of line (its registers would cost the replay a wave
// of occupancy)
__device__ __attribute__((noinline)) int imh_dwell_far(const uint32_t* __restrict__ bits, int64_t i, int s, int q, int k, int64_t n) {
    const int words = (k + kScanBlock - 1) / kScanBlock;
    for (int q2 = q + 2; q2 < words; ++q2) {
        const uint32_t w2 = bits[(int64_t)q2 * n + i];
        if (w2) return q2 * kScanBlock + (__ffs((int)w2) - 1) - s;
    }
    return (k - s) | kLastFlag;
}

// dwell time of proposal (s, i): 0 when it was rejected, else the distance to the chain's next accepted step (to the
// end, flagged, when there is none)
__device__ __forceinline__ int imh_dwell(const ImhLook& l, const uint32_t* __restrict__ bits, int64_t i, int s, int k, int64_t n) {
    const int q = s / kScanBlock, b = s - q * kScanBlock;
    if (((l.w0 >> b) & 1u) == 0u) return 0;
    const unsigned long long rest = (((unsigned long long)l.w1 << kScanBlock) | l.w0) >> (b + 1);
    if (rest) return __ffsll((long long)rest);
    return imh_dwell_far(bits, i, s, q, k, n);
}

// Replay of proposals for the moments, the sample store and the final states.  Two ways, chosen per call from the
// scan's counts (the same for every workgroup, and a function of the chains' data only, so a run repeats bit for bit):
//   SUM      visit the ACCEPTED proposals, weight = steps each stayed the state (few acceptances: few rows);
//   CORRECT  start from the proposal kernel's sums over ALL proposals and visit only those whose weight is not 1 --
//            rejected (-1), kept for c > 1 steps (c - 1) -- plus every chain's final state (many acceptances: few rows).
// With a sample store every kept step needs its row, so SUM is used.
// The aligned 8-coordinates-per-lane layouts with narrow conditioners sit at 129-131 registers, one past four waves per
// SIMD; held to 128 (at most 16 bytes of scratch) they run 10 % faster (C2: 477 -> 430 us).  The other layouts would spill.
template <int CPL, int LPC, int HP, template <int, int, bool> class Pot, bool FAST, int NB = 0>
__global__ void __launch_bounds__(kBlock, (CPL == 8 && HP == 4 && FAST && NB == 0) ? 4 : 1) imh_replay_kernel(NfmcFlowMhArgs a, ImhWork w, int64_t tiles, int eval_grid) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CPW = kWave / LPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC, cw = lane / LPC;
    const int d = a.flow.d;
    const int64_t n = a.n, total = n * (int64_t)a.n_steps + n;   // proposals, then the n initial states
    const bool small = total < (1ll << 31);
    const bool correct = a.stats.sum_x && !a.samples.base && w.visit[1] < w.visit[0];
    using Flow = FlowB<CPL, LPC, HP, false, (FAST && CPL >= 8), NB>;
    Flow::Img::stage(lds, a.flow, kBlock);
    __syncthreads();
    Flow fl;
    fl.init(lds, a.flow, g);
    Pot<CPL, LPC, FAST> pot;
    pot.init(a.pot, g, d);
    const bool revl = (a.flow.n_coupling & 1) != 0;
    const float base_c = -0.5f * (float)d * kLog2Pi;
    float sx[CPL], sxx[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) sx[q] = sxx[q] = 0.f;
    // a wave looks at 64 consecutive rows at a time (their dwell times off the accept masks) and runs the flow pass
    // only for the rows it has to visit, CPW of them per pass: most 64-row chunks need one or two passes.
    // Round 4 measured two ways of filling the passes better (at C2 a look finds ~17 rows: 2 full passes + 1 of one row,
    // 2.6 passes per look on average where 2.1 would do) -- both parity-green, both SLOWER, neither kept (tools/ab_c2.sh,
    // rocprofv3, the 1000-transition call: 0.450 ms for this loop):
    //   a sliding 64-row window per wave over a contiguous row range that only runs full passes and moves to the first row
    //   still to be served: 0.678 ms (a third more looks, each an exposed load round trip);
    //   256 rows per look (four mask-word pairs per lane in one round trip, passes filled across the four windows: 2.2 passes
    //   per 64 rows, a quarter of the looks): 0.488 ms (four permutes and a rank search per pass, 16 spilled registers at the
    //   128-register cap).
    // The passes are not what the kernel waits for: at 43 % VALU-busy the issue time of its ~600-instruction passes is
    // ~0.22 ms; the rest is the dependent chain of a pass with four waves per SIMD to cover it.
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t r0 = (tile * kWavesPerBlock + wave) * kWave;
        if (r0 >= total) continue;                                // wave-uniform
        const RowBase rb0 = row_base(r0, n, small);
        int c_lane = 0;
        bool want = false;
        {
            const int64_t r = r0 + lane;
            if (r < total) {
                if (r >= total - n) {
                    c_lane = w.dwell0[r - (total - n)];
                    want = c_lane > 0;
                } else {
                    int64_t i;
                    int s;
                    split_row(rb0, lane, n, i, s);
                    ImhLook here;
                    here.w0 = w.bits[(int64_t)(s / kScanBlock) * n + i];
                    here.w1 = w.bits[(int64_t)(s / kScanBlock + 1) * n + i];   // the scan writes one (zero) word past the last
                    c_lane = imh_dwell(here, w.bits, i, s, a.n_steps, n);
                    want = correct ? c_lane != 1 : c_lane > 0;    // the flagged dwell time of a final state is never 1
                }
            }
        }
        unsigned long long todo = __ballot(want);
        while (todo) {
            unsigned long long m = todo;                      // this group's row: the cw-th set bit
            for (int t = 0; t < cw; ++t) m &= m - 1ull;
            const int bit = m ? __ffsll((long long)m) - 1 : -1;
            const int word = __shfl(c_lane, bit < 0 ? 0 : bit, kWave);
            const bool have = bit >= 0;
            const int c = have ? (word & ~kLastFlag) : 0;
            const bool is_last = have && (word & kLastFlag) != 0;
            const int64_t r = r0 + (have ? bit : 0);
            const bool initial = have && r >= total - n;
            int s = 0;
            int64_t i = 0;
            if (have) {
                if (initial) i = r - (total - n);
                else split_row(rb0, bit, n, i, s);
            }
            float xs[CPL];
            float wgt = (float)c;
            if (__ballot(have && !initial) != 0ull) {
                float f_xp, u_xp;
                imh_propose<CPL, LPC, HP>(xs, f_xp, u_xp, a, fl, pot, i, s, g, revl, base_c);
                if (correct && !initial && imh_summed(u_xp, f_xp)) wgt -= 1.f;   // the proposal kernel counted it once (an initial state sharing the pass: never)
            }
            if (initial) load_row<CPL, LPC, FAST>(w.x0, i, d, g, true, xs);
            if (have) {
                if (wgt != 0.f) {
#pragma unroll
                    for (int q = 0; q < CPL; ++q) {
                        sx[q] = fmaf(wgt, xs[q], sx[q]);
                        sxx[q] = fmaf(wgt * xs[q], xs[q], sxx[q]);
                    }
                }
                if (!initial) {
                    store_rows_in(a.samples, s, s + c, n * (int64_t)d, [&](float* kept) { store_row<CPL, LPC, FAST>(kept, i, d, g, true, xs); });
                    if (is_last) store_row<CPL, LPC, FAST>(a.x, i, d, g, true, xs);
                } else {
                    store_rows_in(a.samples, 0, c, n * (int64_t)d, [&](float* kept) { store_row<CPL, LPC, FAST>(kept, i, d, g, true, xs); });
                }
            }
            for (int t = 0; t < CPW && todo; ++t) todo &= todo - 1ull;   // CPW rows done
        }
    }
    if (a.stats.sum_x)
        block_stats_flush<CPL, LPC>(sx, sxx, 0u, 0u, a.stats, 0u, 0u,
                                    correct && (int)blockIdx.x < eval_grid ? w.esum + (size_t)blockIdx.x * (2 * CPL * LPC) : nullptr);
}

struct PCfg {
    int cpl, lpc;
};
static const PCfg kPCfgs[] = {{4, 1}, {4, 2}, {4, 4}, {4, 8}, {8, 8}, {8, 16}, {8, 32}, {8, 64}};
#define NFMC_FOR_PCFG(M) M(4, 1) M(4, 2) M(4, 4) M(4, 8) M(8, 8) M(8, 16) M(8, 32) M(8, 64)

// NB: 0 = affine couplings (this unit's instantiations: imh_parallel.hip), kRqsBins = spline couplings (imh_parallel_rqs.hip)
template <int CPL, int LPC, int HP, int NB = 0>
static int launch_imh(const NfmcFlowMhArgs& a, const ImhWork& w, hipStream_t st, int* grid_c, int* dp_out, bool dry) {
    const int64_t total_a = a.n * (int64_t)a.n_steps, total_c = total_a + a.n;
    const int64_t tiles_a = (total_a + kWavesPerBlock * kWave - 1) / (kWavesPerBlock * kWave);   // 64 rows per wave and tile
    const int64_t tiles_c = (total_c + kWavesPerBlock * kWave - 1) / (kWavesPerBlock * kWave);   // 64 rows per wave look
    const int grid_a = (int)(tiles_a < kMaxGrid ? tiles_a : kMaxGrid);
    const int gc = (int)(tiles_c < kMaxGrid ? tiles_c : kMaxGrid);
    const int dp = CPL * LPC;
    if (a.stats.sum_x && a.stats.scratch_bytes < (int64_t)gc * (2 * dp + kStatTail) * (int64_t)sizeof(double)) return NFMC_ESCRATCH;
    if (check_defer(a.stats, dp, a.flow.d)) return NFMC_EINVAL;
#define NFMC_LI(POT, F)                                                                                           \
    {                                                                                                             \
        const size_t lds = (size_t)FlowImage<CPL, LPC, HP, (F && CPL >= 8), NB>::total_floats(a.flow.n_hidden_layers,  \
                                                                                       a.flow.n_coupling) * sizeof(float); \
        if (lds > 120 * 1024) return NFMC_EUNSUPPORTED;                                                           \
        if (dry) return 0;                                                                                        \
        const bool sums = a.stats.sum_x != nullptr && a.samples.base == nullptr;   /* the replay may correct */          \
        auto ka = sums ? imh_eval_kernel<CPL, LPC, HP, POT, F, NB, true> : imh_eval_kernel<CPL, LPC, HP, POT, F, NB, false>; \
        auto kc = imh_replay_kernel<CPL, LPC, HP, POT, F, NB>;                                                        \
        if (lds > 48 * 1024) {                                                                                    \
            hipError_t e = hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                                   \
            e = hipFuncSetAttribute((const void*)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
            if (e != hipSuccess) return (int)e;                                                                   \
        }                                                                                                         \
        hipLaunchKernelGGL(ka, dim3(grid_a), dim3(kBlock), lds, st, a, w, tiles_a);                               \
        if (a.masks_out || a.log_ratio_out)                                                                       \
            hipLaunchKernelGGL(imh_scan_kernel<true>, dim3((unsigned)((a.n + kWave - 1) / kWave)), dim3(kWave), 0, st, a, w); \
        else   /* one lane per chain */                                                                           \
            hipLaunchKernelGGL(imh_scan_kernel<false>, dim3((unsigned)((a.n + kWave - 1) / kWave)), dim3(kWave), 0, st, a, w); \
        hipLaunchKernelGGL(kc, dim3(gc), dim3(kBlock), lds, st, a, w, tiles_c, grid_a);                           \
    }
    const int d = a.flow.d;
    const bool fast = d == dp && (d % 4) == 0 && a.pot.a == nullptr && a.pot.b == nullptr && (((uintptr_t)a.x) & 15u) == 0 &&
                      (((uintptr_t)w.x0) & 15u) == 0 && (!a.samples.base || (((uintptr_t)a.samples.base) & 15u) == 0);
    if (a.pot.kind == NFMC_POT_FUNNEL) NFMC_LI(FunnelPot, false)
    else if (fast) NFMC_LI(QuadraticPot, true)
    else NFMC_LI(QuadraticPot, false)
#undef NFMC_LI
    *grid_c = gc;
    *dp_out = dp;
    return 0;
}

// spline couplings: the same three kernels with FlowB<..., NB = 8> (instantiated in imh_parallel_rqs.hip: a unit of its own,
// the spline flow pass is ~350 instructions per coordinate and 48 layouts x potentials of it compile for minutes)
int launch_imh_rqs(int cpl, int lpc, int hp, const NfmcFlowMhArgs& a, const ImhWork& w, hipStream_t st, int* grid_c, int* dp_out, bool dry);

}  // namespace nfmc
