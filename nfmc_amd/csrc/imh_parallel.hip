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
#include "flow_b.hpp"

namespace nfmc {

struct ImhWork {
    // chain-major (n, k): a chain's k proposals are contiguous, so the scan's 64 lanes read 64 consecutive words
    // (step-major made every lane of the scan touch its own cache line: 284 us of a 1.8 ms run at n = 8192, k = 1000)
    float* u;        // (n, k) U(x')
    float* f;        // (n, k) log q(x')
    float* logu;     // (n, k) log of the accept uniform of (chain, step)
    int32_t* dwell;  // (n, k) number of steps proposal (i, s) was the state of chain i (0: rejected)
    int32_t* dwell0; // (n)    the same for the initial state
    int32_t* last;   // (n)    step whose proposal is the final state, -1: the initial state
    float* x0;       // (n, d) copy of the initial states
};

// row r of the chain-major work arrays -> (chain i, step s).  A 64-bit division is ~180 VALU instructions on
// gfx950 (a quarter of the proposal loop); every launch the host makes has n * k < 2^31 (`small`).
__device__ __forceinline__ void split_row(int64_t r, int k, bool small, int64_t& i, int& s) {
    if (small) {
        const uint32_t q = (uint32_t)r / (uint32_t)k;
        i = q;
        s = (int)((uint32_t)r - q * (uint32_t)k);
    } else {
        i = r / k;
        s = (int)(r - i * k);
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

template <int CPL, int LPC, int HP, template <int, int, bool> class Pot, bool FAST>
__global__ void __launch_bounds__(kBlock) imh_eval_kernel(NfmcFlowMhArgs a, ImhWork w, int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CPW = kWave / LPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC, cw = lane / LPC;
    const int d = a.flow.d;
    const int64_t n = a.n, total = n * (int64_t)a.n_steps;
    const bool small = total < (1ll << 31);
    using Flow = FlowB<CPL, LPC, HP, false, (FAST && CPL >= 8)>;
    Flow::Img::stage(lds, a.flow, kBlock);
    __syncthreads();
    Flow fl;
    fl.init(lds, a.flow, g);
    Pot<CPL, LPC, FAST> pot;
    pot.init(a.pot, g, d);
    const bool revl = (a.flow.n_coupling & 1) != 0;
    const float base_c = -0.5f * (float)d * kLog2Pi;
    // a wave owns 64 consecutive rows per tile: first every lane draws the accept uniform of ONE row (one Philox call
    // per row instead of one per lane and row: 8 % of the kernel at LPC = 8), then LPC passes evaluate CPW rows each
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t r0 = (tile * kWavesPerBlock + wave) * kWave;
        {
            const int64_t r = r0 + lane;
            if (r < total) {
                int64_t i;
                int s;
                split_row(r, a.n_steps, small, i, s);
                float uu;   // the accept uniform of (chain i, step s): imh.py:229
                if (a.rng.replay_uniforms) {
                    uu = a.rng.replay_uniforms[(int64_t)s * n + i];
                } else {
                    const uint4 rnd = philox4x32_10((uint32_t)(a.rng.chain_offset + (uint64_t)i), a.rng.step0 + (uint32_t)s, 0u,
                                                    kTagJump, (uint32_t)a.rng.seed, (uint32_t)(a.rng.seed >> 32));
                    uu = u32_to_uniform(rnd.x);
                }
                w.logu[r] = fast_ln(uu);
                w.dwell[r] = 0;
            }
        }
        for (int sub = 0; sub < LPC; ++sub) {
            const int64_t rb = r0 + (int64_t)sub * CPW;
            if (rb >= total) break;                                   // wave-uniform
            const int64_t r = rb + cw;
            const bool active = r < total;
            int64_t i;
            int s;
            split_row(active ? r : total - 1, a.n_steps, small, i, s);
            float xp[CPL], f_xp, u_xp;
            imh_propose<CPL, LPC, HP>(xp, f_xp, u_xp, a, fl, pot, i, s, g, revl, base_c);
            if (active && g == 0) {
                w.u[r] = u_xp;
                w.f[r] = f_xp;
            }
        }
    }
}

// One WAVE per chain: the Metropolis scan over the k proposals (imh.py:223-233).  Everything expensive (proposal,
// uniform, logarithm) was done in parallel by imh_eval_kernel.  The scan is sequential only through acceptances: the
// 64 lanes test the next 64 steps against the current state at once, the first accepting lane (ballot + ffs) becomes
// the state and the scan resumes right after it, on the same 64 steps held in registers -- k / 64 memory round trips per
// chain, plus a few dozen cycles per acceptance.
__global__ void __launch_bounds__(256) imh_scan_kernel(NfmcFlowMhArgs a, ImhWork w) {
    const int64_t n = a.n;
    const int d = a.flow.d;
    const int k = a.n_steps;
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    unsigned long long acc_total = 0, bad_total = 0;   // wave-uniform
    for (int64_t i = wave0; i < n; i += nwaves) {
        float u_x = potential_row(w.x0 + i * d, a.pot, d);   // imh.py:224 (every lane: uniform addresses)
        float f_x = a.logq[i];                                // imh.py:214 (filled by the caller when not cached)
        int cur = -1;
        // A window of 64 steps is read ONCE (and the next one requested ahead); every acceptance inside it re-tests the
        // remaining lanes against the new state from registers.  The first version re-read the window shifted past each
        // acceptance and wrote one scattered dwell word per acceptance (~600 per chain at C2's acceptance rate): 215 us
        // of a 1.5 ms call; 173 us now.  What is left is instruction issue: 8192 waves x ~600 acceptances x ~30
        // instructions of a loop that is sequential per chain.
        float nu, nf, nl;   // the next window, requested one window ahead
        {
            const int64_t r0 = i * k + (lane < k ? lane : k - 1);
            nu = w.u[r0], nf = w.f[r0], nl = w.logu[r0];
        }
        for (int base = 0; base < k; base += 64) {
            const int s = base + lane;
            const bool valid = s < k;
            const int sv = valid ? s : k - 1;
            const int64_t r = i * k + sv, ro = (int64_t)sv * n + i;   // work arrays (n, k); outputs (k, n)
            const float pu = nu, pf = nf, pl = nl;
            if (base + 64 < k) {
                const int sn = base + 64 + lane;
                const int64_t rn = i * k + (sn < k ? sn : k - 1);
                nu = w.u[rn], nf = w.f[rn], nl = w.logu[rn];
            }
            int start = 0;                                            // lanes below are decided already
            int dw = 0;   // dwell time of this lane's proposal when it is accepted AND replaced within this window
            while (start < 64) {
                const float lr = (-pu) - (-u_x) + f_x - pf;          // util.py:392
                const bool open = valid && lane >= start;
                const bool acc = open && pl < lr;                     // imh.py:229-230; NaN -> reject
                const bool bad = open && !(fabsf(lr) <= 3.0e38f);
                const unsigned long long am = __ballot(acc), bm = __ballot(bad);
                const int j = am ? __ffsll((long long)am) - 1 : 63;  // lanes start..j are decided by this round
                const unsigned long long decided = j == 63 ? ~0ull : ((2ull << j) - 1ull);
                bad_total += (unsigned long long)__popcll(bm & decided);
                if (open && lane <= j) {
                    if (a.masks_out) a.masks_out[ro] = (acc && lane == j) ? 1 : 0;
                    if (a.log_ratio_out) a.log_ratio_out[ro] = lr;
                }
                if (!am) break;
                const int s_acc = base + j;
                // steps the previous state stayed: into the owning lane's register when that proposal sits in this window
                // (written with the window, one coalesced store), else one store by lane 0 -- at most one per window
                // instead of one scattered 4-byte store per acceptance (~600 per chain at C2)
                if (cur >= base) {
                    if (lane == cur - base) dw = s_acc - cur;
                } else if (lane == 0) {
                    if (cur < 0) w.dwell0[i] = s_acc;                    // steps before the first acceptance
                    else w.dwell[i * k + cur] = s_acc - cur;             // steps this proposal stayed the state
                }
                // j is wave-uniform (from the ballot): v_readlane, not a ds_bpermute round trip through the LDS per acceptance
                u_x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pu), j));
                f_x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pf), j));
                cur = s_acc;
                acc_total++;
                start = j + 1;
            }
            // single writer per element: the proposal that is still the state at the end of the window gets its dwell time
            // from lane 0 later (a later window's `else` branch above, or the tail below); every other element of the
            // window -- rejected (0) or accepted and replaced inside it (dw) -- is written here, one coalesced store
            if (valid && !(cur >= base && lane == cur - base)) w.dwell[r] = dw;
        }
        if (lane == 0) {
            if (cur < 0) w.dwell0[i] = k;
            else w.dwell[i * k + cur] = k - cur;
            w.last[i] = cur;
            a.logq[i] = f_x;                                              // imh.py:233
        }
    }
    // integer counters: atomic adds are exact, the totals do not depend on the order
    if (lane == 0 && a.stats.counters) {
        if (acc_total) atomicAdd(a.stats.counters + NFMC_CNT_ACCEPTED, acc_total);
        if (bad_total) atomicAdd(a.stats.counters + NFMC_CNT_NONFINITE, bad_total);
    }
}

// Replay of the proposals that were accepted (and of the initial states), weighted by their dwell times.
template <int CPL, int LPC, int HP, template <int, int, bool> class Pot, bool FAST>
__global__ void __launch_bounds__(kBlock) imh_replay_kernel(NfmcFlowMhArgs a, ImhWork w, int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CPW = kWave / LPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC, cw = lane / LPC;
    const int d = a.flow.d;
    const int64_t n = a.n, total = n * (int64_t)a.n_steps + n;   // proposals, then the n initial states
    const bool small = total < (1ll << 31);
    using Flow = FlowB<CPL, LPC, HP, false, (FAST && CPL >= 8)>;
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
    // a wave looks at 64 consecutive rows at a time (one coalesced load of their dwell counts) and runs the flow pass
    // only for the rows that were ever a chain's state, CPW of them per pass: at the usual acceptance rates almost
    // every 64-row chunk is skipped after the load (8 rows per look cost 151 us at n = 8192, k = 1000)
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t r0 = (tile * kWavesPerBlock + wave) * kWave;
        int c_lane = 0;
        {
            const int64_t r = r0 + lane;
            if (r < total) c_lane = r >= total - n ? w.dwell0[r - (total - n)] : w.dwell[r];
        }
        unsigned long long todo = __ballot(c_lane > 0);
        while (todo) {
            unsigned long long m = todo;                      // this group's row: the cw-th set bit
            for (int t = 0; t < cw; ++t) m &= m - 1ull;
            const int bit = m ? __ffsll((long long)m) - 1 : -1;
            const int c = __shfl(c_lane, bit < 0 ? 0 : bit, kWave) * (bit >= 0 ? 1 : 0);
            const int64_t r = r0 + (bit < 0 ? 0 : bit);
            const bool initial = bit >= 0 && r >= total - n;
            int s = 0;
            int64_t i = 0;
            if (bit >= 0) {
                if (initial) i = r - (total - n);
                else split_row(r, a.n_steps, small, i, s);
            }
            float xs[CPL];
            if (__ballot(!initial && c > 0) != 0ull) {
                float f_xp, u_xp;
                imh_propose<CPL, LPC, HP>(xs, f_xp, u_xp, a, fl, pot, i, s, g, revl, base_c);
            }
            if (initial) load_row<CPL, LPC, FAST>(w.x0, i, d, g, true, xs);
            if (c > 0) {
                const float cf = (float)c;
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    sx[q] = fmaf(cf, xs[q], sx[q]);
                    sxx[q] = fmaf(cf * xs[q], xs[q], sxx[q]);
                }
                if (!initial) {
                    if (a.samples.base)
                        for (int t = s; t < s + c; ++t)
                            if (float* kept = store_row_of(a.samples, t, n * (int64_t)d)) store_row<CPL, LPC, FAST>(kept, i, d, g, true, xs);
                    if (w.last[i] == s) store_row<CPL, LPC, FAST>(a.x, i, d, g, true, xs);
                } else if (a.samples.base) {
                    for (int t = 0; t < c; ++t)
                        if (float* kept = store_row_of(a.samples, t, n * (int64_t)d)) store_row<CPL, LPC, FAST>(kept, i, d, g, true, xs);
                }
            }
            for (int t = 0; t < CPW && todo; ++t) todo &= todo - 1ull;   // CPW rows done
        }
    }
    if (a.stats.sum_x) block_stats_flush<CPL, LPC>(sx, sxx, 0u, 0u, a.stats);
}

struct PCfg {
    int cpl, lpc;
};
static const PCfg kPCfgs[] = {{4, 1}, {4, 2}, {4, 4}, {4, 8}, {8, 8}, {8, 16}, {8, 32}, {8, 64}};
#define NFMC_FOR_PCFG(M) M(4, 1) M(4, 2) M(4, 4) M(4, 8) M(8, 8) M(8, 16) M(8, 32) M(8, 64)

template <int CPL, int LPC, int HP>
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
        const size_t lds = (size_t)FlowImage<CPL, LPC, HP, (F && CPL >= 8)>::total_floats(a.flow.n_hidden_layers,  \
                                                                                       a.flow.n_coupling) * sizeof(float); \
        if (lds > 120 * 1024) return NFMC_EUNSUPPORTED;                                                           \
        if (dry) return 0;                                                                                        \
        auto ka = imh_eval_kernel<CPL, LPC, HP, POT, F>;                                                          \
        auto kc = imh_replay_kernel<CPL, LPC, HP, POT, F>;                                                        \
        if (lds > 48 * 1024) {                                                                                    \
            hipError_t e = hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                                   \
            e = hipFuncSetAttribute((const void*)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
            if (e != hipSuccess) return (int)e;                                                                   \
        }                                                                                                         \
        hipLaunchKernelGGL(ka, dim3(grid_a), dim3(kBlock), lds, st, a, w, tiles_a);                               \
        const int gs = (int)((a.n + 3) / 4 < 4096 ? (a.n + 3) / 4 : 4096);   /* one wave per chain */              \
        hipLaunchKernelGGL(imh_scan_kernel, dim3(gs), dim3(256), 0, st, a, w);                                    \
        hipLaunchKernelGGL(kc, dim3(gc), dim3(kBlock), lds, st, a, w, tiles_c);                                   \
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

}  // namespace nfmc

using namespace nfmc;

static int64_t imh_work_floats(int64_t n, int32_t d, int32_t k) { return 4 * n * (int64_t)k + 2 * n + n * (int64_t)d + 8; }

extern "C" int64_t nfmc_imh_parallel_work_bytes(int64_t n, int32_t d, int32_t n_steps) {
    if (n <= 0 || d <= 0 || n_steps <= 0) return 0;
    return imh_work_floats(n, d, n_steps) * 4 + 64;
}

static int imh_parallel_run(const NfmcFlowMhArgs* args, void* work, int64_t work_bytes, nfmc_stream_t stream, bool dry);

extern "C" int nfmc_imh_parallel_f32(const NfmcFlowMhArgs* args, void* work, int64_t work_bytes, nfmc_stream_t stream) {
    if (!work) return NFMC_EINVAL;
    return imh_parallel_run(args, work, work_bytes, stream, false);
}

// NFMC_OK when nfmc_imh_parallel_f32 has a kernel for `args` (validates, launches nothing), else the error it would return
extern "C" int nfmc_imh_parallel_supported_f32(const NfmcFlowMhArgs* args) {
    return imh_parallel_run(args, nullptr, 0, nullptr, true);
}

static int imh_parallel_run(const NfmcFlowMhArgs* args, void* work, int64_t work_bytes, nfmc_stream_t stream, bool dry) {
    if (!args) return NFMC_EINVAL;
    NfmcFlowMhArgs a = *args;
    const NfmcRealNVP& f = a.flow;
    if (!a.x || !a.logq || a.n <= 0 || a.n_steps <= 0 || !a.adjusted || !store_ok(a.samples)) return NFMC_EINVAL;
    if (int rr = rng_default_only(a.rng)) return rr;
    if (!f.ea0_log_scale || !f.ea0_shift || !f.ea1_log_scale || !f.ea1_shift || (f.n_coupling > 0 && !f.weights)) return NFMC_EINVAL;
    if (f.d < 2 || f.d > 512 || a.n_steps > NFMC_IMH_PARALLEL_MAX_STEPS) return NFMC_ESHAPE;
    if (f.n_hidden <= 0 || f.n_hidden > 8 || f.n_hidden_layers <= 0 || f.n_bins != 0) return NFMC_EUNSUPPORTED;
    if (a.pot.kind != NFMC_POT_QUADRATIC && a.pot.kind != NFMC_POT_FUNNEL) return NFMC_EUNSUPPORTED;
    if (a.stats.sum_x && (!a.stats.sum_x2 || !a.stats.counters || !a.stats.scratch)) return NFMC_EINVAL;
    if (a.stats.sum_x && a.stats.defer && a.stats.tail_slot != 0) return NFMC_EINVAL;
    if ((a.rng.replay_normals != nullptr) != (a.rng.replay_uniforms != nullptr)) return NFMC_EINVAL;
    if (!dry && work_bytes < nfmc_imh_parallel_work_bytes(a.n, f.d, a.n_steps)) return NFMC_ESCRATCH;
    hipStream_t st = (hipStream_t)stream;
    const int64_t kn = a.n * (int64_t)a.n_steps;
    ImhWork w;
    w.u = (float*)work;
    w.f = w.u + kn;
    w.logu = w.f + kn;
    w.dwell = (int32_t*)(w.logu + kn);
    w.dwell0 = w.dwell + kn;
    w.last = w.dwell0 + a.n;
    w.x0 = (float*)(w.last + a.n);
    w.x0 += (4 - ((4 * kn + 2 * a.n) & 3)) & 3;   // keep the copy of the states 16-byte aligned (vector IO)
    if (!dry) {
        if (!a.logq_cached) {   // flow.log_prob(x0): imh.py:214
            const int rc = nfmc_realnvp_forward_f32(&f, a.x, a.n, nullptr, nullptr, a.logq, stream);
            if (rc) return rc;
        }
        hipError_t e = hipMemcpyAsync(w.x0, a.x, (size_t)a.n * f.d * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    const int d = f.d, hp = f.n_hidden <= 4 ? 4 : 8;
    PCfg c = {0, 0};
    for (const PCfg& k : kPCfgs) {   // smallest capacity; CPL = 4 first (more lanes per row for small batches)
        if (k.cpl * k.lpc < d) continue;
        if (c.cpl == 0 || k.cpl * k.lpc < c.cpl * c.lpc) c = k;
    }
    if (!c.cpl) return NFMC_EUNSUPPORTED;
    int rc = NFMC_EUNSUPPORTED, grid = 0, dp = 0;
#define M(CPL, LPC)                   \
    if (c.cpl == CPL && c.lpc == LPC) \
        rc = hp == 4 ? launch_imh<CPL, LPC, 4>(a, w, st, &grid, &dp, dry) : launch_imh<CPL, LPC, 8>(a, w, st, &grid, &dp, dry);
    NFMC_FOR_PCFG(M)
#undef M
    if (rc || dry) return rc;
    NFMC_HIP_CHECK_LAUNCH();
    if (a.stats.sum_x && !a.stats.defer) {
        hipLaunchKernelGGL(stats_finish_kernel<true>, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, st, a.stats.scratch, grid, dp, d,
                           a.stats, (unsigned long long)a.n * (unsigned long long)a.n_steps);
        NFMC_HIP_CHECK_LAUNCH();
    }
    return NFMC_OK;
}
