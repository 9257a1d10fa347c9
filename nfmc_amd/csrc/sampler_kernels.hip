// K1/K2 (+K6, K7): fused Langevin (MALA/ULA) and HMC/UHMC transitions for closed-form potentials.
//
// Replaces, per transition, the reference's eager-op sequence
//   Langevin.propose  nfmc/algorithms/sampling/mcmc/langevin.py:61-122
//   HMC.propose       nfmc/algorithms/sampling/mcmc/hmc.py:61-77,96-126
//   masked update, counters, streaming moments, sample store   mcmc/base.py:74-90, sampling/base.py:75-95,234-259
//
// Layout: a chain is spread over LPC consecutive lanes, lane g holding the CPL contiguous
// coordinates g*CPL.. (16-byte vector IO).  The state stays in VGPRs for all n_steps of a call, so
// HBM sees one read and one write of (n, d) per call (plus one write per step if samples are kept);
// the kernel is VALU-bound (Philox + Box-Muller + ~25 flop per coordinate), not HBM-bound.
// One butterfly reduction per transition produces the log acceptance ratio in every lane of the
// group; the accept count comes from a wave ballot.
#include "common.hpp"

namespace nfmc {

template <int CPL, int LPC, bool FAST>
struct MassCoef {
    // Langevin: c1 = -h/m^2, c2 = sqrt(2h)/m, hA = h/m^2, invA = m^2 ; HMC: rs = 1/sqrt(m), m
    float c1_s, c2_s, hA_s;
    float c1[FAST ? 1 : CPL], c2[FAST ? 1 : CPL], hA[FAST ? 1 : CPL], invA[FAST ? 1 : CPL];
    float m[FAST ? 1 : CPL], rs[FAST ? 1 : CPL];

    __device__ __forceinline__ void init(float h, float sqrt2h, const float* __restrict__ imd, int g, int d) {
        c1_s = -h;
        c2_s = sqrt2h;
        hA_s = h;
        if constexpr (!FAST) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = coord_of<CPL, LPC>(g, i);
                const bool ok = c < d;
                const float mm = (ok && imd) ? imd[c] : 1.f;
                const float A = 1.f / (mm * mm);
                c1[i] = ok ? (-h) / (mm * mm) : 0.f;
                c2[i] = ok ? sqrt2h / mm : 0.f;
                hA[i] = ok ? h * A : 0.f;
                invA[i] = ok ? 1.f / A : 0.f;
                m[i] = ok ? mm : 0.f;
                rs[i] = ok ? 1.f / sqrtf(mm) : 0.f;
            }
        }
    }
    __device__ __forceinline__ float C1(int i) const { return FAST ? c1_s : c1[FAST ? 0 : i]; }
    __device__ __forceinline__ float C2(int i) const { return FAST ? c2_s : c2[FAST ? 0 : i]; }
    __device__ __forceinline__ float HA(int i) const { return FAST ? hA_s : hA[FAST ? 0 : i]; }
    __device__ __forceinline__ float IA(int i) const { return FAST ? 1.f : invA[FAST ? 0 : i]; }
    __device__ __forceinline__ float M(int i) const { return FAST ? 1.f : m[FAST ? 0 : i]; }
    __device__ __forceinline__ float RS(int i) const { return FAST ? 1.f : rs[FAST ? 0 : i]; }
};

// noise for this lane's CPL coordinates of (chain, step): native Philox or replay from HBM
template <int CPL, int LPC>
__device__ __forceinline__ void draw_normals(const NfmcRng& rng, uint32_t tag, uint32_t gchain, int64_t row, int64_t n,
                                             int d, int g, int s, float (&e)[CPL]) {
    if (rng.replay_normals) {
        const float* p = rng.replay_normals + ((int64_t)s * n + row) * d;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = coord_of<CPL, LPC>(g, i);
            e[i] = (row < n && c < d) ? p[c] : 0.f;
        }
    } else {
        const uint32_t k0 = (uint32_t)rng.seed, k1 = (uint32_t)(rng.seed >> 32);
#pragma unroll
        for (int b = 0; b < CPL / 4; ++b) {
            float z[4];
            philox_normal4(gchain, rng.step0 + (uint32_t)s, (uint32_t)(b * LPC + g), tag, k0, k1, z);
            e[4 * b] = z[0];
            e[4 * b + 1] = z[1];
            e[4 * b + 2] = z[2];
            e[4 * b + 3] = z[3];
        }
    }
}

struct AcceptUniform {
    uint4 r;
    __device__ __forceinline__ float draw(const NfmcRng& rng, uint32_t gchain, int64_t row, int64_t n, int s) {
        if (rng.replay_uniforms) return row < n ? rng.replay_uniforms[(int64_t)s * n + row] : 0.5f;
        const uint32_t step = rng.step0 + (uint32_t)s;
        if (s == 0 || (step & 3u) == 0u)
            r = philox4x32_10(gchain, step >> 2, 0u, kTagAccept, (uint32_t)rng.seed, (uint32_t)(rng.seed >> 32));
        return u32_to_uniform(pick_word(r, step & 3u));
    }
};

// ------------------------------------------------------------------------------------------------
template <int CPL, int LPC, template <int, int, bool> class Pot, bool FAST>
__global__ void __launch_bounds__(kBlock, NFMC_WPE) mala_kernel(NfmcMalaArgs a, float sqrt2h, int64_t tiles) {
    constexpr int CPW = kWave / LPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC, cw = lane / LPC;
    const int d = a.d;
    const int64_t n = a.n;
    const float h = a.step_size;
    const float inv4h = 1.f / (4.f * h);

    MassCoef<CPL, LPC, FAST> mc;
    mc.init(h, sqrt2h, a.inv_mass_diag, g, d);
    Pot<CPL, LPC, FAST> pot;
    pot.init(a.pot, g, d);

    float sx[CPL], sxx[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) sx[i] = sxx[i] = 0.f;
    uint32_t n_acc = 0, n_bad = 0;
    const unsigned long long leaders = LPC == 64 ? 1ull : (LPC == 32 ? 0x0000000100000001ull
                                       : LPC == 16 ? 0x0001000100010001ull
                                       : LPC == 8 ? 0x0101010101010101ull
                                       : LPC == 4 ? 0x1111111111111111ull
                                       : LPC == 2 ? 0x5555555555555555ull : ~0ull);

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row = (tile * kWavesPerBlock + wave) * CPW + cw;
        const bool active = row < n;
        const uint32_t gchain = (uint32_t)(a.rng.chain_offset + (uint64_t)row);
        float x[CPL];
        load_row<CPL, LPC, FAST>(a.x, row, d, g, active, x);
        AcceptUniform au;

        for (int s = 0; s < a.n_steps; ++s) {
            float e[CPL], xp[CPL];
            draw_normals<CPL, LPC>(a.rng, kTagNoise, gchain, row, n, d, g, s, e);
            bool accept = true;
            float lr = 0.f;
            if constexpr (Pot<CPL, LPC, FAST>::kQuadratic) {
                // U = sum a (x-b)^2: with t = x - b, t' = x' - b the reference's ratio (langevin.py:88-105)
                //   (u - u') + [q(x'|x) - q(x|x')]   collapses term by term to   a^2 (h/m^2) (t^2 - t'^2)
                // (expand tf = d + 2 a hA t, tb = -d + 2 a hA t', d = t' - t; invA hA = h): same value, 7
                // instead of 17 VALU instructions per coordinate; checked against the golden vectors.
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    const float t = x[i] - pot.bb(i);
                    xp[i] = fmaf(mc.C2(i), e[i], fmaf(mc.C1(i) * (2.f * pot.aa(i)), t, x[i]));  // langevin.py:74-76
                    const float tp = xp[i] - pot.bb(i);
                    lr = fmaf(pot.aa(i) * pot.aa(i) * mc.HA(i) * (t - tp), t + tp, lr);
                }
            } else {
                const auto ctx = pot.prepare(x, g, d);
#pragma unroll
                for (int i = 0; i < CPL; ++i)
                    xp[i] = fmaf(mc.C2(i), e[i], fmaf(mc.C1(i), pot.grad(ctx, i, x[i]), x[i]));  // langevin.py:74-76
                if (a.adjust) {
                    const auto ctxp = pot.prepare(xp, g, d);
#pragma unroll
                    for (int i = 0; i < CPL; ++i) {
                        const float gj = pot.grad(ctx, i, x[i]), gp = pot.grad(ctxp, i, xp[i]);
                        const float tf = (xp[i] - x[i]) + mc.HA(i) * gj;  // q(x'|x)  langevin.py:31-42
                        const float tb = (x[i] - xp[i]) + mc.HA(i) * gp;  // q(x|x')
                        lr += (pot.term(ctx, i, x[i]) - pot.term(ctxp, i, xp[i])) +
                              inv4h * mc.IA(i) * (tf * tf - tb * tb);
                    }
                }
            }
            if (a.adjust) {
                lr = group_allreduce<LPC>(lr);
                const float u = au.draw(a.rng, gchain, row, n, s);
                accept = fast_ln(u) < lr;  // NaN -> reject (langevin.py:106)
                n_bad += (uint32_t)__popcll(__ballot(active && !(fabsf(lr) <= 3.0e38f)) & leaders);
            }
            accept = accept && active;
            n_acc += (uint32_t)__popcll(__ballot(accept) & leaders);
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                x[i] = accept ? xp[i] : x[i];  // mcmc/base.py:77
                sx[i] += x[i];
                sxx[i] = fmaf(x[i], x[i], sxx[i]);
            }
            if (a.samples) store_row<CPL, LPC, FAST>(a.samples + (int64_t)s * n * d, row, d, g, active, x);
            if (g == 0 && active) {
                if (a.masks_out) a.masks_out[(int64_t)s * n + row] = accept ? 1 : 0;
                if (a.log_ratio_out) a.log_ratio_out[(int64_t)s * n + row] = lr;
            }
        }
        store_row<CPL, LPC, FAST>(a.x, row, d, g, active, x);
    }
    if (a.stats.sum_x) block_stats_flush<CPL, LPC>(sx, sxx, n_acc, n_bad, a.stats.scratch);
}

// ------------------------------------------------------------------------------------------------
template <int CPL, int LPC, template <int, int, bool> class Pot, bool FAST>
__global__ void __launch_bounds__(kBlock, NFMC_WPE) hmc_kernel(NfmcHmcArgs a, int64_t tiles) {
    constexpr int CPW = kWave / LPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC, cw = lane / LPC;
    const int d = a.d;
    const int64_t n = a.n;
    const float h = a.step_size, hh = a.step_size / 2;

    MassCoef<CPL, LPC, FAST> mc;
    mc.init(h, 0.f, a.inv_mass_diag, g, d);
    Pot<CPL, LPC, FAST> pot;
    pot.init(a.pot, g, d);

    float sx[CPL], sxx[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) sx[i] = sxx[i] = 0.f;
    uint32_t n_acc = 0, n_bad = 0;
    const unsigned long long leaders = LPC == 64 ? 1ull : (LPC == 32 ? 0x0000000100000001ull
                                       : LPC == 16 ? 0x0001000100010001ull
                                       : LPC == 8 ? 0x0101010101010101ull
                                       : LPC == 4 ? 0x1111111111111111ull
                                       : LPC == 2 ? 0x5555555555555555ull : ~0ull);

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t row = (tile * kWavesPerBlock + wave) * CPW + cw;
        const bool active = row < n;
        const uint32_t gchain = (uint32_t)(a.rng.chain_offset + (uint64_t)row);
        float x[CPL];
        load_row<CPL, LPC, FAST>(a.x, row, d, g, active, x);
        AcceptUniform au;

        for (int s = 0; s < a.n_steps; ++s) {
            float p[CPL], q[CPL];
            draw_normals<CPL, LPC>(a.rng, kTagNoise, gchain, row, n, d, g, s, p);
            float dh = 0.f;  // this lane's share of H0 - H1
            {
                const auto ctx = pot.prepare(x, g, d);
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    p[i] *= mc.RS(i);  // hmc.py:100
                    q[i] = x[i];
                    dh += pot.term(ctx, i, x[i]) + 0.5f * (p[i] * p[i] * mc.M(i));  // hmc.py:103-106
                }
            }
            for (int l = 0; l < a.n_leapfrog; ++l) {  // hmc.py:67-71, both half steps kept separate
                const auto c0 = pot.prepare(q, g, d);
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    p[i] = fmaf(-hh, pot.grad(c0, i, q[i]), p[i]);
                    q[i] = fmaf(h, p[i] * mc.M(i), q[i]);
                }
                const auto c1 = pot.prepare(q, g, d);
#pragma unroll
                for (int i = 0; i < CPL; ++i) p[i] = fmaf(-hh, pot.grad(c1, i, q[i]), p[i]);
            }
            bool accept = true;
            float lr = 0.f;
            if (a.adjust) {
                const auto ctx = pot.prepare(q, g, d);
#pragma unroll
                for (int i = 0; i < CPL; ++i) dh -= pot.term(ctx, i, q[i]) + 0.5f * (p[i] * p[i] * mc.M(i));  // :107-110
                lr = group_allreduce<LPC>(dh);
                const float u = au.draw(a.rng, gchain, row, n, s);
                accept = fast_ln(u) < lr;  // hmc.py:111-113
                n_bad += (uint32_t)__popcll(__ballot(active && !(fabsf(lr) <= 3.0e38f)) & leaders);
            }
            accept = accept && active;
            n_acc += (uint32_t)__popcll(__ballot(accept) & leaders);
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                x[i] = accept ? q[i] : x[i];
                sx[i] += x[i];
                sxx[i] = fmaf(x[i], x[i], sxx[i]);
            }
            if (a.samples) store_row<CPL, LPC, FAST>(a.samples + (int64_t)s * n * d, row, d, g, active, x);
            if (g == 0 && active) {
                if (a.masks_out) a.masks_out[(int64_t)s * n + row] = accept ? 1 : 0;
                if (a.log_ratio_out) a.log_ratio_out[(int64_t)s * n + row] = lr;
            }
        }
        store_row<CPL, LPC, FAST>(a.x, row, d, g, active, x);
    }
    if (a.stats.sum_x) block_stats_flush<CPL, LPC>(sx, sxx, n_acc, n_bad, a.stats.scratch);
}

// ------------------------------------------------------------------------------------------------
// host side: configuration choice and launch
struct Cfg {
    int cpl, lpc;
};

// (CPL, LPC) instantiated below, ordered by capacity CPL*LPC.
static const Cfg kCfgs[] = {{4, 1}, {4, 2}, {4, 4}, {4, 8}, {8, 8}, {4, 16}, {16, 4}, {8, 16}, {16, 8}, {8, 32}, {16, 16}, {8, 64}, {16, 32}, {16, 64}};

static Cfg choose_cfg(int d, bool fast_ok) {
    // NFMC_SAMPLER_CFG="cpl,lpc" overrides (tuning)
    if (const char* e = getenv("NFMC_SAMPLER_CFG")) {
        int c = 0, l = 0;
        if (sscanf(e, "%d,%d", &c, &l) == 2)
            for (const Cfg& k : kCfgs)
                if (k.cpl == c && k.lpc == l && c * l >= d) return k;
    }
    Cfg best = {0, 0};
    for (const Cfg& k : kCfgs) {
        if (k.cpl * k.lpc < d) continue;
        // smallest capacity wins; kCfgs lists equal capacities in order of measured preference
        // (CPL = 8 keeps 4 waves/SIMD resident: 14.6 vs 12.8 G chain-steps/s at n=65536, d=64)
        if (best.cpl == 0 || k.cpl * k.lpc < best.cpl * best.lpc) best = k;
    }
    return best;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

template <class Args>
static int check_common(const Args* a) {
    if (!a || !a->x) return NFMC_EINVAL;
    if (a->n <= 0 || a->d <= 0 || a->n_steps <= 0) return NFMC_EINVAL;
    if (a->n_steps > NFMC_MAX_STEPS_PER_CALL) return NFMC_ESHAPE;
    if (a->d > 1024) return NFMC_ESHAPE;
    if (!(a->step_size > 0.f)) return NFMC_EINVAL;
    if (a->pot.kind != NFMC_POT_QUADRATIC && a->pot.kind != NFMC_POT_FUNNEL) return NFMC_EUNSUPPORTED;
    if (((uintptr_t)a->x & 3u) != 0) return NFMC_EALIGN;
    if ((a->rng.replay_normals == nullptr) != (a->rng.replay_uniforms == nullptr) && a->adjust) return NFMC_EINVAL;
    if (a->stats.sum_x && (!a->stats.sum_x2 || !a->stats.counters || !a->stats.scratch)) return NFMC_EINVAL;
    return NFMC_OK;
}

template <class Args>
static bool fast_path(const Args* a, const Cfg& c) {
    return a->d == c.cpl * c.lpc && a->inv_mass_diag == nullptr && a->pot.a == nullptr && a->pot.b == nullptr &&
           aligned16(a->x) && (!a->samples || aligned16(a->samples));
}

#define NFMC_FOR_CFG(M)                                                                                             \
    M(4, 1) M(4, 2) M(4, 4) M(4, 8) M(4, 16) M(8, 8) M(16, 4) M(8, 16) M(16, 8) M(8, 32) M(16, 16) M(8, 64) M(16, 32) \
        M(16, 64)

template <int CPL, int LPC>
static int launch_mala(const NfmcMalaArgs& a, bool fast, int64_t tiles, int grid, float sqrt2h, hipStream_t st) {
    const bool funnel = a.pot.kind == NFMC_POT_FUNNEL;
    if (funnel) {
        if (fast) hipLaunchKernelGGL((mala_kernel<CPL, LPC, FunnelPot, true>), dim3(grid), dim3(kBlock), 0, st, a, sqrt2h, tiles);
        else hipLaunchKernelGGL((mala_kernel<CPL, LPC, FunnelPot, false>), dim3(grid), dim3(kBlock), 0, st, a, sqrt2h, tiles);
    } else {
        if (fast) hipLaunchKernelGGL((mala_kernel<CPL, LPC, QuadraticPot, true>), dim3(grid), dim3(kBlock), 0, st, a, sqrt2h, tiles);
        else hipLaunchKernelGGL((mala_kernel<CPL, LPC, QuadraticPot, false>), dim3(grid), dim3(kBlock), 0, st, a, sqrt2h, tiles);
    }
    return NFMC_OK;
}

template <int CPL, int LPC>
static int launch_hmc(const NfmcHmcArgs& a, bool fast, int64_t tiles, int grid, hipStream_t st) {
    const bool funnel = a.pot.kind == NFMC_POT_FUNNEL;
    if (funnel) {
        if (fast) hipLaunchKernelGGL((hmc_kernel<CPL, LPC, FunnelPot, true>), dim3(grid), dim3(kBlock), 0, st, a, tiles);
        else hipLaunchKernelGGL((hmc_kernel<CPL, LPC, FunnelPot, false>), dim3(grid), dim3(kBlock), 0, st, a, tiles);
    } else {
        if (fast) hipLaunchKernelGGL((hmc_kernel<CPL, LPC, QuadraticPot, true>), dim3(grid), dim3(kBlock), 0, st, a, tiles);
        else hipLaunchKernelGGL((hmc_kernel<CPL, LPC, QuadraticPot, false>), dim3(grid), dim3(kBlock), 0, st, a, tiles);
    }
    return NFMC_OK;
}

}  // namespace nfmc

using namespace nfmc;

extern "C" int64_t nfmc_stats_scratch_bytes(int32_t d) {
    if (d <= 0 || d > 1024) return 0;
    return stats_scratch_doubles(padded_d(d)) * (int64_t)sizeof(double);
}

extern "C" int nfmc_mala_steps_f32(const NfmcMalaArgs* args, nfmc_stream_t stream) {
    int rc = check_common(args);
    if (rc) return rc;
    NfmcMalaArgs a = *args;
    hipStream_t st = (hipStream_t)stream;
    Cfg c = choose_cfg(a.d, true);
    if (!c.cpl) return NFMC_ESHAPE;
    bool fast = fast_path(&a, c);
    if (!fast) c = choose_cfg(a.d, false);
    fast = fast_path(&a, c);
    const int dp = c.cpl * c.lpc;
    const int cpw = kWave / c.lpc;
    const int64_t tiles = (a.n + (int64_t)kWavesPerBlock * cpw - 1) / ((int64_t)kWavesPerBlock * cpw);
    const int grid = (int)(tiles < kMaxGrid ? tiles : kMaxGrid);
    if (a.stats.sum_x && a.stats.scratch_bytes < (int64_t)grid * (2 * dp + kStatTail) * (int64_t)sizeof(double))
        return NFMC_ESCRATCH;
    const float sqrt2h = (float)sqrt(2.0 * (double)a.step_size);  // math.sqrt(2*step_size), langevin.py:75
#define M(CPL, LPC) \
    if (c.cpl == CPL && c.lpc == LPC) rc = launch_mala<CPL, LPC>(a, fast, tiles, grid, sqrt2h, st);
    NFMC_FOR_CFG(M)
#undef M
    NFMC_HIP_CHECK_LAUNCH();
    if (a.stats.sum_x) {
        hipLaunchKernelGGL(stats_finish_kernel, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, st, a.stats.scratch, grid, dp, a.d, a.stats,
                           (unsigned long long)a.n * (unsigned long long)a.n_steps);
        NFMC_HIP_CHECK_LAUNCH();
    }
    return rc;
}

extern "C" int nfmc_hmc_steps_f32(const NfmcHmcArgs* args, nfmc_stream_t stream) {
    int rc = check_common(args);
    if (rc) return rc;
    if (args->n_leapfrog <= 0) return NFMC_EINVAL;
    NfmcHmcArgs a = *args;
    hipStream_t st = (hipStream_t)stream;
    Cfg c = choose_cfg(a.d, true);
    if (!c.cpl) return NFMC_ESHAPE;
    bool fast = fast_path(&a, c);
    if (!fast) c = choose_cfg(a.d, false);
    fast = fast_path(&a, c);
    const int dp = c.cpl * c.lpc;
    const int cpw = kWave / c.lpc;
    const int64_t tiles = (a.n + (int64_t)kWavesPerBlock * cpw - 1) / ((int64_t)kWavesPerBlock * cpw);
    const int grid = (int)(tiles < kMaxGrid ? tiles : kMaxGrid);
    if (a.stats.sum_x && a.stats.scratch_bytes < (int64_t)grid * (2 * dp + kStatTail) * (int64_t)sizeof(double))
        return NFMC_ESCRATCH;
#define M(CPL, LPC) \
    if (c.cpl == CPL && c.lpc == LPC) rc = launch_hmc<CPL, LPC>(a, fast, tiles, grid, st);
    NFMC_FOR_CFG(M)
#undef M
    NFMC_HIP_CHECK_LAUNCH();
    if (a.stats.sum_x) {
        hipLaunchKernelGGL(stats_finish_kernel, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, st, a.stats.scratch, grid, dp, a.d, a.stats,
                           (unsigned long long)a.n * (unsigned long long)a.n_steps);
        NFMC_HIP_CHECK_LAUNCH();
    }
    return rc;
}
