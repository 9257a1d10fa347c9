// Flow-proposal Metropolis transitions (jump of JumpNFMC.sample, jump.py:205-243; loop body of
// FixedIMH.sample, imh.py:220-249) for narrow conditioners (HP <= 8) in the samplers' register layout:
// LPC lanes per chain, state and proposal in VGPRs, flow weights in one LDS image per workgroup
// (flow_b.hpp).  Same skeleton and statistics path as mala_kernel.
#include "flow_b.hpp"

namespace nfmc {

// DIAG = false is the production instantiation: no replayed noise, no sample store, no mask / log-ratio outputs --
// the branches on those pointers (and the scalar registers that carry them through the tile loop: the DIAG kernel
// spills SGPRs into VGPR lanes there) are compiled out.  The host picks it when all of those arguments are NULL.
// NB = 8: rational-quadratic spline couplings ('c-rqnsf') on the same skeleton (round 3; before, spline flows ran the jump on
// the one-chain-per-lane kernel of flow_kernels.hip only).
template <int CPL, int LPC, int HP, template <int, int, bool> class Pot, bool FAST, bool DIAG, int RR = 10, int NB = 0>
#ifndef NFMC_FLOWB_WPE
#define NFMC_FLOWB_WPE 1
#endif
__global__ void __launch_bounds__(kBlock, NFMC_FLOWB_WPE) flow_mh_b_kernel(NfmcFlowMhArgs a, int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CPW = kWave / LPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC, cw = lane / LPC;
    const int d = a.flow.d;
    const int64_t n = a.n;
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
        float u_x;
        {
            const auto ctx = pot.prepare(x, g, d);
            float up = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) up += pot.term(ctx, i, x[i]);
            u_x = group_allreduce<LPC>(up);                                  // jump.py:212 / imh.py:224
        }
        float f_x;
        if (a.logq_cached) {
            f_x = active ? a.logq[row] : 0.f;
        } else {                                                             // flow.log_prob(x): jump.py:218 / imh.py:214
            float w[CPL];
#pragma unroll
            for (int i = 0; i < CPL; ++i) w[i] = x[i];
            float part = fl.forward(w);
#pragma unroll
            for (int i = 0; i < CPL; ++i) part = fmaf(-0.5f * w[i], w[i], part);
            f_x = group_allreduce<LPC>(part) + base_c;
        }
        StoreCursor keep(a.samples);
        for (int s = 0; s < a.n_steps; ++s) {
            float xp[CPL];
            draw_latent<CPL, LPC, FAST, RR>(xp, (DIAG && a.rng.replay_normals) ? a.rng.replay_normals + (int64_t)s * n * d : nullptr, a.rng.seed,
                                  a.rng.step0 + (uint32_t)s, gchain, row, n, d, g, revl);  // flow.sample: jump.py:205 / imh.py:221
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) part = fmaf(-0.5f * xp[i], xp[i], part);
            part -= fl.inverse(xp);
            const float f_xp = group_allreduce<LPC>(part) + base_c;
            float up = 0.f;
            {
                const auto ctx = pot.prepare(xp, g, d);
#pragma unroll
                for (int i = 0; i < CPL; ++i) up += pot.term(ctx, i, xp[i]);
            }
            const float u_xp = group_allreduce<LPC>(up);                     // jump.py:213 / imh.py:225
            const float lr = (-u_xp) - (-u_x) + f_x - f_xp;                  // util.py:392
            bool accept = true;
            if (a.adjusted) {
                float u;
                if (DIAG && a.rng.replay_uniforms) {
                    u = active ? a.rng.replay_uniforms[(int64_t)s * n + row] : 0.5f;
                } else {
                    const uint4 r = philox4x32<RR>(gchain, a.rng.step0 + (uint32_t)s, 0u, kTagJump, (uint32_t)a.rng.seed,
                                                  (uint32_t)(a.rng.seed >> 32));
                    u = u32_to_uniform(r.x);
                }
                accept = fast_ln(u) < lr;                                     // jump.py:225 / imh.py:229-230
                n_bad += (uint32_t)__popcll(__ballot(active && !(fabsf(lr) <= 3.0e38f)) & leaders);
            }
            accept = accept && active;
            const uint64_t am = __ballot(accept);
            n_acc += (uint32_t)__popcll(am & leaders);
            f_x = select_f32(am, f_xp, f_x);
            u_x = select_f32(am, u_xp, u_x);
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                x[i] = select_f32(am, xp[i], x[i]);                                // jump.py:231 / imh.py:232-233
                sx[i] += x[i];
                sxx[i] = fmaf(x[i], x[i], sxx[i]);
            }
            if constexpr (DIAG) {
                if (float* kept = keep.next(n * d)) store_row<CPL, LPC, FAST>(kept, row, d, g, active, x);
            }
            if (DIAG && g == 0 && active) {
                if (a.masks_out) a.masks_out[(int64_t)s * n + row] = accept ? 1 : 0;
                if (a.log_ratio_out) a.log_ratio_out[(int64_t)s * n + row] = lr;
            }
        }
        store_row<CPL, LPC, FAST>(a.x, row, d, g, active, x);
        if (g == 0 && active) a.logq[row] = f_x;
    }
    if (a.stats.sum_x) block_stats_flush<CPL, LPC>(sx, sxx, n_acc, n_bad, a.stats);
}

// The production instantiation with TWO chains per lane group (FlowB::forward2 / inverse2: every weight row read from
// LDS feeds both chains).  Exact-fit quadratic targets, no diagnostics; chains 2p and 2p + 1 share a lane group.  Per
// chain the arithmetic and the random stream are those of flow_mh_b_kernel, so the results are bitwise the same.
template <int CPL, int LPC, int HP, int RR = 10>
__global__ void __launch_bounds__(kBlock, 1) flow_mh_b2_kernel(NfmcFlowMhArgs a, int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CPW = kWave / LPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC, cw = lane / LPC;
    const int d = a.flow.d;
    const int64_t n = a.n;
    using Flow = FlowB<CPL, LPC, HP, false, true>;
    Flow::Img::stage(lds, a.flow, kBlock);
    __syncthreads();
    Flow fl;
    fl.init(lds, a.flow, g);
    QuadraticPot<CPL, LPC, true> pot;
    pot.init(a.pot, g, d);
    const bool revl = (a.flow.n_coupling & 1) != 0;
    const float base_c = -0.5f * (float)d * kLog2Pi;

    float sx[CPL], sxx[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) sx[i] = sxx[i] = 0.f;
    uint32_t n_acc = 0, n_bad = 0;
    const unsigned long long leaders = LPC == 64 ? 1ull : (LPC == 32 ? 0x0000000100000001ull
                                       : LPC == 16 ? 0x0001000100010001ull : 0x0101010101010101ull);
    static_assert(LPC >= 8, "two-chain layouts: 8 .. 64 lanes per chain");

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t ra = 2 * ((tile * kWavesPerBlock + wave) * CPW + cw), rb = ra + 1;
        const bool act_a = ra < n, act_b = rb < n;
        const uint32_t gca = (uint32_t)(a.rng.chain_offset + (uint64_t)ra), gcb = gca + 1u;
        float xa[CPL], xb[CPL];
        load_row<CPL, LPC, true>(a.x, ra, d, g, act_a, xa);
        load_row<CPL, LPC, true>(a.x, rb, d, g, act_b, xb);
        float u_a, u_b;
        {
            const auto ca = pot.prepare(xa, g, d);
            const auto cb = pot.prepare(xb, g, d);
            float pa = 0.f, pb = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                pa += pot.term(ca, i, xa[i]);
                pb += pot.term(cb, i, xb[i]);
            }
            u_a = group_allreduce<LPC>(pa);
            u_b = group_allreduce<LPC>(pb);
        }
        float f_a, f_b;
        if (a.logq_cached) {
            f_a = act_a ? a.logq[ra] : 0.f;
            f_b = act_b ? a.logq[rb] : 0.f;
        } else {
            float wa[CPL], wb[CPL];
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                wa[i] = xa[i];
                wb[i] = xb[i];
            }
            float pa, pb;
            fl.forward2(wa, wb, pa, pb);
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                pa = fmaf(-0.5f * wa[i], wa[i], pa);
                pb = fmaf(-0.5f * wb[i], wb[i], pb);
            }
            f_a = group_allreduce<LPC>(pa) + base_c;
            f_b = group_allreduce<LPC>(pb) + base_c;
        }
        for (int s = 0; s < a.n_steps; ++s) {
            float za[CPL], zb[CPL];
            draw_latent<CPL, LPC, true, RR>(za, nullptr, a.rng.seed, a.rng.step0 + (uint32_t)s, gca, ra, n, d, g, revl);
            draw_latent<CPL, LPC, true, RR>(zb, nullptr, a.rng.seed, a.rng.step0 + (uint32_t)s, gcb, rb, n, d, g, revl);
            float pa = 0.f, pb = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                pa = fmaf(-0.5f * za[i], za[i], pa);
                pb = fmaf(-0.5f * zb[i], zb[i], pb);
            }
            float la, lb;
            fl.inverse2(za, zb, la, lb);
            pa -= la;
            pb -= lb;
            const float f_pa = group_allreduce<LPC>(pa) + base_c, f_pb = group_allreduce<LPC>(pb) + base_c;
            float qa = 0.f, qb = 0.f;
            {
                const auto ca = pot.prepare(za, g, d);
                const auto cb = pot.prepare(zb, g, d);
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    qa += pot.term(ca, i, za[i]);
                    qb += pot.term(cb, i, zb[i]);
                }
            }
            const float u_pa = group_allreduce<LPC>(qa), u_pb = group_allreduce<LPC>(qb);
            const float lra = (-u_pa) - (-u_a) + f_a - f_pa, lrb = (-u_pb) - (-u_b) + f_b - f_pb;   // util.py:392
            bool acc_a = true, acc_b = true;
            if (a.adjusted) {
                const uint32_t step = a.rng.step0 + (uint32_t)s;
                const uint4 r0 = philox4x32<RR>(gca, step, 0u, kTagJump, (uint32_t)a.rng.seed, (uint32_t)(a.rng.seed >> 32));
                const uint4 r1 = philox4x32<RR>(gcb, step, 0u, kTagJump, (uint32_t)a.rng.seed, (uint32_t)(a.rng.seed >> 32));
                acc_a = fast_ln(u32_to_uniform(r0.x)) < lra;                     // jump.py:225 / imh.py:229-230
                acc_b = fast_ln(u32_to_uniform(r1.x)) < lrb;
                n_bad += (uint32_t)__popcll(__ballot(act_a && !(fabsf(lra) <= 3.0e38f)) & leaders);
                n_bad += (uint32_t)__popcll(__ballot(act_b && !(fabsf(lrb) <= 3.0e38f)) & leaders);
            }
            acc_a = acc_a && act_a;
            acc_b = acc_b && act_b;
            const uint64_t ma = __ballot(acc_a), mb = __ballot(acc_b);
            n_acc += (uint32_t)__popcll(ma & leaders) + (uint32_t)__popcll(mb & leaders);
            f_a = select_f32(ma, f_pa, f_a);
            f_b = select_f32(mb, f_pb, f_b);
            u_a = select_f32(ma, u_pa, u_a);
            u_b = select_f32(mb, u_pb, u_b);
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                xa[i] = select_f32(ma, za[i], xa[i]);                            // jump.py:231 / imh.py:232-233
                xb[i] = select_f32(mb, zb[i], xb[i]);
                sx[i] += xa[i];
                sxx[i] = fmaf(xa[i], xa[i], sxx[i]);
                sx[i] += xb[i];
                sxx[i] = fmaf(xb[i], xb[i], sxx[i]);
            }
        }
        store_row<CPL, LPC, true>(a.x, ra, d, g, act_a, xa);
        store_row<CPL, LPC, true>(a.x, rb, d, g, act_b, xb);
        if (g == 0 && act_a) a.logq[ra] = f_a;
        if (g == 0 && act_b) a.logq[rb] = f_b;
    }
    if (a.stats.sum_x) block_stats_flush<CPL, LPC>(sx, sxx, n_acc, n_bad, a.stats);
}

struct BCfg {
    int cpl, lpc;
};
static const BCfg kBCfgs[] = {{4, 1}, {4, 2}, {4, 4}, {4, 8}, {8, 8}, {4, 16}, {8, 16}, {4, 32}, {8, 32}, {4, 64}, {8, 64}};

#define NFMC_FOR_BCFG(M) M(4, 1) M(4, 2) M(4, 4) M(4, 8) M(8, 8) M(4, 16) M(8, 16) M(4, 32) M(8, 32) M(4, 64) M(8, 64)

// diagnostics (replayed noise, sample store, mask / log-ratio outputs) run on the DIAG instantiation
static bool wants_diag(const NfmcFlowMhArgs& a) {
    return a.rng.replay_normals || a.rng.replay_uniforms || a.samples.base || a.masks_out || a.log_ratio_out;
}

template <int CPL, int LPC, int HP>
static int launch_b2(const NfmcFlowMhArgs& a, int64_t tiles, int grid, hipStream_t st, bool dry) {
    if constexpr (CPL >= 8 && LPC >= 8) {
        const size_t lds = (size_t)FlowImage<CPL, LPC, HP, true>::total_floats(a.flow.n_hidden_layers, a.flow.n_coupling) * sizeof(float);
        if (lds > 120 * 1024) return NFMC_EUNSUPPORTED;
        if (dry) return 0;
        auto kern = rng_rounds(a.rng) == 7 ? flow_mh_b2_kernel<CPL, LPC, HP, 7> : flow_mh_b2_kernel<CPL, LPC, HP, 10>;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, a, tiles);
        return 0;
    } else {
        return NFMC_EUNSUPPORTED;
    }
}

// spline couplings: one instantiation per (layout, width, potential, exact-fit), diagnostics compiled in, default stream
template <int CPL, int LPC, int HP>
static int launch_b_rqs(const NfmcFlowMhArgs& a, bool fast, int64_t tiles, int grid, hipStream_t st, bool dry) {
#define NFMC_LBR(POT, F)                                                                                          \
    {                                                                                                             \
        const size_t lds = (size_t)FlowImage<CPL, LPC, HP, (F && CPL >= 8), kRqsBins>::total_floats(a.flow.n_hidden_layers, \
                                                                                                 a.flow.n_coupling) * sizeof(float); \
        if (lds > 120 * 1024) return NFMC_EUNSUPPORTED;                                                           \
        if (rng_rounds(a.rng) == 7) return NFMC_EUNSUPPORTED;                                                     \
        if (dry) return 0;                                                                                        \
        auto kern = flow_mh_b_kernel<CPL, LPC, HP, POT, F, true, 10, kRqsBins>;                                   \
        if (lds > 48 * 1024) {                                                                                    \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                               (int)lds);                                                         \
            if (e != hipSuccess) return (int)e;                                                                   \
        }                                                                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, a, tiles);                                    \
    }
    if (a.pot.kind == NFMC_POT_FUNNEL) NFMC_LBR(FunnelPot, false)
    else if (fast) NFMC_LBR(QuadraticPot, true)
    else NFMC_LBR(QuadraticPot, false)
#undef NFMC_LBR
    return 0;
}

template <int CPL, int LPC, int HP>
static int launch_b(const NfmcFlowMhArgs& a, bool fast, int64_t tiles, int grid, hipStream_t st, bool dry) {
#define NFMC_LB(POT, F)                                                                                         \
    {                                                                                                           \
        const size_t lds = (size_t)FlowImage<CPL, LPC, HP, (F && CPL >= 8)>::total_floats(a.flow.n_hidden_layers, \
                                                                                       a.flow.n_coupling) * sizeof(float); \
        if (lds > 120 * 1024) return NFMC_EUNSUPPORTED;                                                         \
        if (dry) return 0;                                                                                      \
        const bool diag = !F || wants_diag(a);                                                                  \
        const bool r7 = rng_rounds(a.rng) == 7;   /* opt-in Philox4x32-7 stream: exact-fit kernels only */          \
        if (r7 && !F) return NFMC_EUNSUPPORTED;                                                                 \
        auto kern = r7 ? (diag ? flow_mh_b_kernel<CPL, LPC, HP, POT, F, true, (F ? 7 : 10)>                        \
                               : flow_mh_b_kernel<CPL, LPC, HP, POT, F, (F ? false : true), (F ? 7 : 10)>)         \
                       : (diag ? flow_mh_b_kernel<CPL, LPC, HP, POT, F, true>                                      \
                               : flow_mh_b_kernel<CPL, LPC, HP, POT, F, (F ? false : true)>);                      \
        if (lds > 48 * 1024) {                                                                                  \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                               (int)lds);                                                       \
            if (e != hipSuccess) return (int)e;                                                                 \
        }                                                                                                       \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, a, tiles);                                  \
    }
    if (a.pot.kind == NFMC_POT_FUNNEL) NFMC_LB(FunnelPot, false)
    else if (fast) NFMC_LB(QuadraticPot, true)
    else NFMC_LB(QuadraticPot, false)
#undef NFMC_LB
    return 0;
}

// Returns NFMC_EUNSUPPORTED when this path does not cover the request (caller falls back to flow_mh_kernel).
// dry: decide only, launch nothing (nfmc_flow_mh_supported_f32).
int flow_mh_b_launch(const NfmcFlowMhArgs& a, hipStream_t st, int* grid_out, int* dp_out, bool dry) {
    const int d = a.flow.d;
    const int hp = a.flow.n_hidden <= 4 ? 4 : 8;
    if (a.flow.n_hidden > 8 || d > 512) return NFMC_EUNSUPPORTED;
    // smallest capacity; at equal capacity CPL = 8 (exact-fit layouts read only their source / target rows;
    // IMH d = 64, 1000 transitions: 3.23 vs 3.64 ms at n = 8192, 4.2 vs 6.5 ms at n = 16384) except for very
    // few chains, where twice the lanes per chain win (n = 4096: 3.30 vs 3.40 ms)
    BCfg c = {0, 0};
    const int want_cpl = a.n <= 4096 ? 4 : 8;
    for (const BCfg& k : kBCfgs) {
        if (k.cpl * k.lpc < d) continue;
        if (c.cpl == 0 || k.cpl * k.lpc < c.cpl * c.lpc || (k.cpl * k.lpc == c.cpl * c.lpc && k.cpl == want_cpl)) c = k;
    }
    if (const char* e = getenv("NFMC_FLOWB_CFG")) {  // "cpl,lpc" override (tuning)
        int cc = 0, ll = 0;
        if (sscanf(e, "%d,%d", &cc, &ll) == 2)
            for (const BCfg& k : kBCfgs)
                if (k.cpl == cc && k.lpc == ll && cc * ll >= d) c = k;
    }
    if (!c.cpl) return NFMC_EUNSUPPORTED;
    const int dp = c.cpl * c.lpc;
    const bool fast = d == dp && (d % 4) == 0 && a.pot.a == nullptr && a.pot.b == nullptr &&
                      (((uintptr_t)a.x) & 15u) == 0 && (!a.samples.base || (((uintptr_t)a.samples.base) & 15u) == 0);
    // Two chains per lane group (flow_mh_b2_kernel): exact-fit quadratic targets without diagnostics, where the LDS
    // reads of the weight image bound the one-chain kernel (d >= 256; NFMC_FLOWB_DUAL=0/1 overrides for tuning).
    const bool rqs = a.flow.n_bins != 0;
    if (rqs && (a.flow.n_bins != kRqsBins || !(a.flow.spline_bound > 0.f))) return NFMC_EUNSUPPORTED;
    bool dual = fast && c.cpl >= 8 && c.lpc >= 8 && a.pot.kind != NFMC_POT_FUNNEL && !wants_diag(a) && dp >= 256;
    if (const char* e = getenv("NFMC_FLOWB_DUAL"))
        dual = atoi(e) != 0 && fast && c.cpl >= 8 && c.lpc >= 8 && a.pot.kind != NFMC_POT_FUNNEL && !wants_diag(a);
    dual = dual && !rqs;
    const int cpw = (kWave / c.lpc) * (dual ? 2 : 1);
    const int64_t tiles = (a.n + (int64_t)kWavesPerBlock * cpw - 1) / ((int64_t)kWavesPerBlock * cpw);
    // Persistent workgroups: the weight image is staged once per workgroup and every wave pays a fixed prologue /
    // statistics epilogue, so fewer, longer-lived workgroups amortise both, while more of them hide the latency of a
    // transition's dependent chain.  Measured (rocprofv3, tools/ab_jump.sh): C3 jump (65536 x 64) 26.1 / 23.1 / 25.4 us
    // at 512 / 1024 / 2048 workgroups; C5 jump (32768 x 256, image 35 KB) 61.6 / 65.6 / 73.9 us.
    int gcap = dp <= 128 ? 1024 : 512;
    if (const char* e = getenv("NFMC_FLOWB_GRID")) gcap = atoi(e) > 0 ? atoi(e) : gcap;
    const int grid = (int)(tiles < gcap ? tiles : gcap);
    if (a.stats.sum_x && a.stats.scratch_bytes < (int64_t)grid * (2 * dp + kStatTail) * (int64_t)sizeof(double))
        return NFMC_ESCRATCH;
    if (check_defer(a.stats, dp, d)) return NFMC_EINVAL;
    int rc = NFMC_EUNSUPPORTED;
#define M(CPL, LPC)                                                                      \
    if (c.cpl == CPL && c.lpc == LPC)                                                    \
        rc = rqs ? (hp == 4 ? launch_b_rqs<CPL, LPC, 4>(a, fast, tiles, grid, st, dry) : launch_b_rqs<CPL, LPC, 8>(a, fast, tiles, grid, st, dry)) \
           : dual ? (hp == 4 ? launch_b2<CPL, LPC, 4>(a, tiles, grid, st, dry) : launch_b2<CPL, LPC, 8>(a, tiles, grid, st, dry)) \
                  : (hp == 4 ? launch_b<CPL, LPC, 4>(a, fast, tiles, grid, st, dry) : launch_b<CPL, LPC, 8>(a, fast, tiles, grid, st, dry));
    NFMC_FOR_BCFG(M)
#undef M
    *grid_out = grid;
    *dp_out = dp;
    return rc;
}

}  // namespace nfmc
