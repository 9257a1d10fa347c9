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
#pragma once

#include "flow_b.hpp"

namespace nfmc {

template <int CPL, int LPC, bool FAST>
struct MassCoef {
    // Langevin: c1 = -h/m^2, c2 = sqrt(2h)/m, hA = h/m^2, invA = m^2 ; HMC: rs = 1/sqrt(m), m
    // Random-walk proposals (mh.py:51-55): x' = x + m eps, symmetric => c1 = 0, c2 = m and no q terms.
    float c1_s, c2_s, hA_s, cg_s, kap_s;
    float c1[FAST ? 1 : CPL], c2[FAST ? 1 : CPL], hA[FAST ? 1 : CPL], invA[FAST ? 1 : CPL];
    float m[FAST ? 1 : CPL], rs[FAST ? 1 : CPL], cg[FAST ? 1 : CPL], kap[FAST ? 1 : CPL];

    // coefficients of the closed-form quadratic-potential transition: x' = x + cg t + c2 eps,
    // log r = sum kap (t^2 - t'^2);  MALA: cg = -2 a h/m^2, kap = a^2 h/m^2;  random walk: cg = 0, kap = a
    template <class PotT>
    __device__ __forceinline__ void init_quadratic(const PotT& pot, bool rw) {
        cg_s = rw ? 0.f : c1_s * (2.f * pot.aa(0));
        kap_s = rw ? pot.aa(0) : pot.aa(0) * pot.aa(0) * hA_s;
        if constexpr (!FAST) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                cg[i] = rw ? 0.f : c1[i] * (2.f * pot.aa(i));
                kap[i] = rw ? pot.aa(i) : pot.aa(i) * pot.aa(i) * hA[i];
            }
        }
    }
    __device__ __forceinline__ float CG(int i) const { return FAST ? cg_s : cg[FAST ? 0 : i]; }
    __device__ __forceinline__ float KAP(int i) const { return FAST ? kap_s : kap[FAST ? 0 : i]; }

    __device__ __forceinline__ void init(float h, float sqrt2h, const float* __restrict__ imd, int g, int d,
                                         bool rw = false) {
        c1_s = rw ? 0.f : -h;
        c2_s = rw ? 1.f : sqrt2h;
        hA_s = h;
        if constexpr (!FAST) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = coord_of<CPL, LPC>(g, i);
                const bool ok = c < d;
                const float mm = (ok && imd) ? imd[c] : 1.f;
                const float A = 1.f / (mm * mm);
                c1[i] = (ok && !rw) ? (-h) / (mm * mm) : 0.f;
                c2[i] = ok ? (rw ? mm : sqrt2h / mm) : 0.f;
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
template <int CPL, int LPC, int R = 10>
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
            philox_normal4<R>(gchain, rng.step0 + (uint32_t)s, (uint32_t)(b * LPC + g), tag, k0, k1, z);
            e[4 * b] = z[0];
            e[4 * b + 1] = z[1];
            e[4 * b + 2] = z[2];
            e[4 * b + 3] = z[3];
        }
    }
}

template <int R = 10>
struct AcceptUniformR {
    uint4 r;
    __device__ __forceinline__ float draw(const NfmcRng& rng, uint32_t gchain, int64_t row, int64_t n, int s) {
        if (rng.replay_uniforms) return row < n ? rng.replay_uniforms[(int64_t)s * n + row] : 0.5f;
        const uint32_t step = rng.step0 + (uint32_t)s;
        if (s == 0 || (step & 3u) == 0u)
            r = philox4x32<R>(gchain, step >> 2, 0u, kTagAccept, (uint32_t)rng.seed, (uint32_t)(rng.seed >> 32));
        return u32_to_uniform(pick_word(r, step & 3u));
    }
};

// ------------------------------------------------------------------------------------------------
// Device-side copy of NfmcJumpTail (the header's struct is a host pointer target).
struct JumpDev {
    NfmcRealNVP flow;
    int adjusted;
    const float* replay_latent;
    const float* replay_uniform;
    uint8_t* mask_out;
    float* log_ratio_out;
};

// One flow-proposal Metropolis jump on the registers of the chain (jump.py:205-243).  Returns accept.
// The flow code lifts a fused kernel to ~208 VGPRs (occupancy 5 -> 2 for the whole launch), which costs the inner
// transitions more than the separate flow-MH launch: the tail is exercised by the tests but off by default.
template <int CPL, int LPC, int HP, class FlowT, class PotT>
__device__ __forceinline__ bool jump_once(float (&x)[CPL], const FlowT& fl, const PotT& pot,
                                          const JumpDev& j, uint64_t seed, uint32_t step, uint32_t gchain, int64_t row,
                                          int64_t n, int d, int g, bool active, float& lr_out, bool& bad) {
    const bool revl = (j.flow.n_coupling & 1) != 0;
    float part;  // this lane's share of  -u(x) ... assembled so that ONE butterfly gives log alpha
    {
        const auto ctx = pot.prepare(x, g, d);
        float w[CPL];
        part = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            part += pot.term(ctx, i, x[i]);       // + u(x)                      jump.py:212
            w[i] = x[i];
        }
        part += fl.forward(w);                    // + log q(x) = logdet_fwd - z^2/2 (+c)   jump.py:218
#pragma unroll
        for (int i = 0; i < CPL; ++i) part = fmaf(-0.5f * w[i], w[i], part);
    }
    float xp[CPL];
    draw_latent<CPL, LPC>(xp, j.replay_latent, seed, step, gchain, row, n, d, g, revl);  // flow.sample  jump.py:205
#pragma unroll
    for (int i = 0; i < CPL; ++i) part = fmaf(0.5f * xp[i], xp[i], part);               // - log q(x') ...
    part += fl.inverse(xp);                                                               // ... = z'^2/2 + logdet_inv (-c)
    {
        const auto ctx = pot.prepare(xp, g, d);
#pragma unroll
        for (int i = 0; i < CPL; ++i) part -= pot.term(ctx, i, xp[i]);                   // - u(x')     jump.py:213
    }
    const float lr = group_allreduce<LPC>(part);                                          // util.py:392
    bool accept = true;
    bad = false;
    if (j.adjusted) {
        float u;
        if (j.replay_uniform) {
            u = active ? j.replay_uniform[row] : 0.5f;
        } else {
            const uint4 r = philox4x32_10(gchain, step, 0u, kTagJump, (uint32_t)seed, (uint32_t)(seed >> 32));
            u = u32_to_uniform(r.x);
        }
        accept = fast_ln(u) < lr;                                                         // jump.py:225
        bad = active && !(fabsf(lr) <= 3.0e38f);
    }
    accept = accept && active;
    const uint64_t am = __ballot(accept);
#pragma unroll
    for (int i = 0; i < CPL; ++i) x[i] = select_f32(am, xp[i], x[i]);                    // jump.py:231
    lr_out = lr;
    return accept;
}

// ------------------------------------------------------------------------------------------------
template <int CPL, int LPC, template <int, int, bool> class Pot, bool FAST, int JHP, int RR = 10>
__global__ void __launch_bounds__(kBlock, NFMC_WPE) mala_kernel(NfmcMalaArgs a, float sqrt2h, int64_t tiles, JumpDev jd) {
    extern __shared__ __attribute__((aligned(16))) float flow_lds[];
    constexpr int CPW = kWave / LPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC, cw = lane / LPC;
    const int d = a.d;
    const int64_t n = a.n;
    // warmup: the step size lives in device memory, where the controller of the previous call left it (NfmcTune)
    const double h_dev = a.tune.state ? a.tune.state[NFMC_TUNE_STEP_SIZE] : 0.0;
    const float h = a.tune.state ? (float)h_dev : a.step_size;
    if (a.tune.state) sqrt2h = (float)sqrt(2.0 * h_dev);   // math.sqrt(2 * step_size) of the fp64 step, langevin.py:75
    const bool adjust = (a.adjust & 1) != 0, rw = (a.adjust & 2) != 0;
    const float inv4h = rw ? 0.f : 1.f / (4.f * h);

    MassCoef<CPL, LPC, FAST> mc;
    mc.init(h, sqrt2h, a.inv_mass_diag, g, d, rw);
    Pot<CPL, LPC, FAST> pot;
    pot.init(a.pot, g, d);
    if constexpr (Pot<CPL, LPC, FAST>::kQuadratic) mc.init_quadratic(pot, rw);
    FlowB<CPL, LPC, JHP == 0 ? 4 : JHP, true, (FAST && CPL >= 8 && JHP > 0)> fl;
    if constexpr (JHP > 0) {
        decltype(fl)::Img::stage(flow_lds, jd.flow, kBlock);
        __syncthreads();
        fl.init(flow_lds, jd.flow, g);
    }

    float sx[CPL], sxx[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) sx[i] = sxx[i] = 0.f;
    uint32_t n_acc = 0, n_bad = 0, j_acc = 0, j_bad = 0;
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
        AcceptUniformR<RR> au;
        StoreCursor keep(a.samples);
        float sq = 0.f, sq_prop = 0.f;   // FAST quadratic: this lane's share of |x|^2 (current state / proposal)
        if constexpr (Pot<CPL, LPC, FAST>::kQuadratic && FAST) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) sq = fmaf(x[i], x[i], sq);
        }

        for (int s = 0; s < a.n_steps; ++s) {
            float e[CPL], xp[CPL];
            draw_normals<CPL, LPC, RR>(a.rng, kTagNoise, gchain, row, n, d, g, s, e);
            bool accept = true;
            float lr = 0.f;
            if constexpr (Pot<CPL, LPC, FAST>::kQuadratic && FAST) {
                // FAST: scalar a, b = 0, unit mass.  The ratio below, a^2 h (sum_j x_j^2 - sum_j x'_j^2), needs only
                // this lane's share of |x'|^2: the share of |x|^2 is carried from the previous transition (`sq`,
                // replaced by the proposal's on acceptance).  3 VALU instructions per coordinate for proposal + ratio.
                float sp0 = 0.f, sp1 = 0.f;
#pragma unroll
                for (int i = 0; i < CPL; i += 2) {
                    xp[i] = fmaf(mc.C2(i), e[i], fmaf(mc.CG(i), x[i], x[i]));              // langevin.py:74-76 / mh.py:55
                    xp[i + 1] = fmaf(mc.C2(i + 1), e[i + 1], fmaf(mc.CG(i + 1), x[i + 1], x[i + 1]));
                    sp0 = fmaf(xp[i], xp[i], sp0);
                    sp1 = fmaf(xp[i + 1], xp[i + 1], sp1);
                }
                sq_prop = sp0 + sp1;
                lr = mc.KAP(0) * (sq - sq_prop);
            } else if constexpr (Pot<CPL, LPC, FAST>::kQuadratic) {
                // U = sum a (x-b)^2: with t = x - b, t' = x' - b the reference's ratio (langevin.py:88-105)
                //   (u - u') + [q(x'|x) - q(x|x')]   collapses term by term to   a^2 (h/m^2) (t^2 - t'^2)
                // (expand tf = d + 2 a hA t, tb = -d + 2 a hA t', d = t' - t; invA hA = h): same value, 7
                // instead of 17 VALU instructions per coordinate; checked against the golden vectors.
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    const float t = x[i] - pot.bb(i);
                    xp[i] = fmaf(mc.C2(i), e[i], fmaf(mc.CG(i), t, x[i]));  // langevin.py:74-76 / mh.py:55
                    const float tp = xp[i] - pot.bb(i);
                    lr = fmaf(mc.KAP(i) * (t - tp), t + tp, lr);
                }
            } else {
                const auto ctx = pot.prepare(x, g, d);
#pragma unroll
                for (int i = 0; i < CPL; ++i)
                    xp[i] = fmaf(mc.C2(i), e[i], fmaf(mc.C1(i), pot.grad(ctx, i, x[i]), x[i]));  // langevin.py:74-76
                if (adjust) {
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
            if (adjust) {
                lr = group_allreduce<LPC>(lr);
                const float u = au.draw(a.rng, gchain, row, n, s);
                accept = fast_ln(u) < lr;  // NaN -> reject (langevin.py:106, mh.py:59)
                n_bad += (uint32_t)__popcll(__ballot(active && !(fabsf(lr) <= 3.0e38f)) & leaders);
            }
            accept = accept && active;
            const uint64_t am = __ballot(accept);
            n_acc += (uint32_t)__popcll(am & leaders);
            if constexpr (Pot<CPL, LPC, FAST>::kQuadratic && FAST) sq = select_f32(am, sq_prop, sq);
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                x[i] = select_f32(am, xp[i], x[i]);  // mcmc/base.py:77
                sx[i] += x[i];
                sxx[i] = fmaf(x[i], x[i], sxx[i]);
            }
            if (float* kept = keep.next(n * d)) store_row<CPL, LPC, FAST>(kept, row, d, g, active, x);
            if (g == 0 && active) {
                if (a.masks_out) a.masks_out[(int64_t)s * n + row] = accept ? 1 : 0;
                if (a.log_ratio_out) a.log_ratio_out[(int64_t)s * n + row] = lr;
            }
        }
        if constexpr (JHP > 0) {
            float lr;
            bool bad;
            fl.launder();
            const bool acc = jump_once<CPL, LPC, JHP>(x, fl, pot, jd, a.rng.seed, a.rng.step0 + (uint32_t)a.n_steps, gchain,
                                                      row, n, d, g, active, lr, bad);
            j_acc += (uint32_t)__popcll(__ballot(acc) & leaders);
            j_bad += (uint32_t)__popcll(__ballot(bad) & leaders);
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                sx[i] += x[i];
                sxx[i] = fmaf(x[i], x[i], sxx[i]);
            }
            if (float* kept = keep.next(n * d)) store_row<CPL, LPC, FAST>(kept, row, d, g, active, x);
            if (g == 0 && active) {
                if (jd.mask_out) jd.mask_out[row] = acc ? 1 : 0;
                if (jd.log_ratio_out) jd.log_ratio_out[row] = lr;
            }
        }
        store_row<CPL, LPC, FAST>(a.x, row, d, g, active, x);
    }
    if (a.stats.sum_x) block_stats_flush<CPL, LPC>(sx, sxx, n_acc, n_bad, a.stats, j_acc, j_bad);
}

// ------------------------------------------------------------------------------------------------
template <int CPL, int LPC, template <int, int, bool> class Pot, bool FAST, int JHP, int RR = 10>
__global__ void __launch_bounds__(kBlock, NFMC_WPE) hmc_kernel(NfmcHmcArgs a, int64_t tiles, JumpDev jd) {
    extern __shared__ __attribute__((aligned(16))) float flow_lds[];
    constexpr int CPW = kWave / LPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane % LPC, cw = lane / LPC;
    const int d = a.d;
    const int64_t n = a.n;
    const float h = a.tune.state ? (float)a.tune.state[NFMC_TUNE_STEP_SIZE] : a.step_size, hh = h / 2;

    MassCoef<CPL, LPC, FAST> mc;
    mc.init(h, 0.f, a.inv_mass_diag, g, d);
    Pot<CPL, LPC, FAST> pot;
    pot.init(a.pot, g, d);
    FlowB<CPL, LPC, JHP == 0 ? 4 : JHP, true, (FAST && CPL >= 8 && JHP > 0)> fl;
    if constexpr (JHP > 0) {
        decltype(fl)::Img::stage(flow_lds, jd.flow, kBlock);
        __syncthreads();
        fl.init(flow_lds, jd.flow, g);
    }

    float sx[CPL], sxx[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) sx[i] = sxx[i] = 0.f;
    uint32_t n_acc = 0, n_bad = 0, j_acc = 0, j_bad = 0;
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
        AcceptUniformR<RR> au;
        StoreCursor keep(a.samples);

        for (int s = 0; s < a.n_steps; ++s) {
            float p[CPL], q[CPL];
            draw_normals<CPL, LPC, RR>(a.rng, kTagNoise, gchain, row, n, d, g, s, p);
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
            // hmc.py:67-71.  The reference's trajectory is L x [half step of p, full step of q, half step of p]; the closing
            // half step of one leapfrog step and the opening half step of the next use the SAME gradient (q has not moved),
            // so they are applied as one full step of p -- the textbook leapfrog, one gradient evaluation per position, the
            // same trajectory up to one rounding of p per step (the reference rounds twice).  Round 3: 3 -> 2 fused
            // multiply-adds per coordinate and leapfrog step on the exact-fit path (C5: hmc_kernel 100 -> 83 us per launch of 5 trajectories).
            if constexpr (Pot<CPL, LPC, FAST>::kQuadratic && FAST) {
                // scalar a, b = 0, unit mass: (h/2) grad U(q) = (h/2 * 2a) q
                const float cq = hh * (2.f * pot.aa(0)), cq2 = 2.f * cq;
                const int L = a.n_leapfrog;
                if (L > 0) {
#pragma unroll
                    for (int i = 0; i < CPL; ++i) p[i] = fmaf(-cq, q[i], p[i]);
                    for (int l = 0; l + 1 < L; ++l) {
#pragma unroll
                        for (int i = 0; i < CPL; ++i) {
                            q[i] = fmaf(h, p[i], q[i]);
                            p[i] = fmaf(-cq2, q[i], p[i]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < CPL; ++i) {
                        q[i] = fmaf(h, p[i], q[i]);
                        p[i] = fmaf(-cq, q[i], p[i]);
                    }
                }
            } else {
                const int L = a.n_leapfrog;
                if (L > 0) {
                    const auto c0 = pot.prepare(q, g, d);
#pragma unroll
                    for (int i = 0; i < CPL; ++i) p[i] = fmaf(-hh, pot.grad(c0, i, q[i]), p[i]);
                    for (int l = 0; l < L; ++l) {
#pragma unroll
                        for (int i = 0; i < CPL; ++i) q[i] = fmaf(h, p[i] * mc.M(i), q[i]);
                        const auto c1 = pot.prepare(q, g, d);
                        const float hs = l + 1 < L ? h : hh;
#pragma unroll
                        for (int i = 0; i < CPL; ++i) p[i] = fmaf(-hs, pot.grad(c1, i, q[i]), p[i]);
                    }
                }
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
            const uint64_t am = __ballot(accept);
            n_acc += (uint32_t)__popcll(am & leaders);
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                x[i] = select_f32(am, q[i], x[i]);
                sx[i] += x[i];
                sxx[i] = fmaf(x[i], x[i], sxx[i]);
            }
            if (float* kept = keep.next(n * d)) store_row<CPL, LPC, FAST>(kept, row, d, g, active, x);
            if (g == 0 && active) {
                if (a.masks_out) a.masks_out[(int64_t)s * n + row] = accept ? 1 : 0;
                if (a.log_ratio_out) a.log_ratio_out[(int64_t)s * n + row] = lr;
            }
        }
        if constexpr (JHP > 0) {
            float lr;
            bool bad;
            fl.launder();
            const bool acc = jump_once<CPL, LPC, JHP>(x, fl, pot, jd, a.rng.seed, a.rng.step0 + (uint32_t)a.n_steps, gchain,
                                                      row, n, d, g, active, lr, bad);
            j_acc += (uint32_t)__popcll(__ballot(acc) & leaders);
            j_bad += (uint32_t)__popcll(__ballot(bad) & leaders);
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                sx[i] += x[i];
                sxx[i] = fmaf(x[i], x[i], sxx[i]);
            }
            if (float* kept = keep.next(n * d)) store_row<CPL, LPC, FAST>(kept, row, d, g, active, x);
            if (g == 0 && active) {
                if (jd.mask_out) jd.mask_out[row] = acc ? 1 : 0;
                if (jd.log_ratio_out) jd.log_ratio_out[row] = lr;
            }
        }
        store_row<CPL, LPC, FAST>(a.x, row, d, g, active, x);
    }
    if (a.stats.sum_x) block_stats_flush<CPL, LPC>(sx, sxx, n_acc, n_bad, a.stats, j_acc, j_bad);
}


// ------------------------------------------------------------------------------------------------
// host-side launchers, one set per JHP (instantiated in sampler_j*.hip so the variants compile in parallel)
struct Cfg {
    int cpl, lpc;
};

template <int CPL, int LPC, int JHP>
int launch_mala_cfg(const NfmcMalaArgs& a, const JumpDev& jd, bool fast, int64_t tiles, int grid, float sqrt2h,
                    hipStream_t st) {
    size_t lds = 0;
    if constexpr (JHP > 0)
        lds = (size_t)FlowImage<CPL, LPC, JHP>::total_floats(jd.flow.n_hidden_layers, jd.flow.n_coupling) * sizeof(float);
    if (lds > 120 * 1024) return NFMC_EUNSUPPORTED;
#define NFMC_L(POT, F)                                                                                            \
    {                                                                                                             \
        auto kern = mala_kernel<CPL, LPC, POT, F, JHP>;                                                           \
        if (lds > 48 * 1024) {                                                                                    \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                                   \
        }                                                                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, a, sqrt2h, tiles, jd);                        \
    }
    if (rng_rounds(a.rng) == 7) {   // opt-in Philox4x32-7 stream: the exact-fit quadratic kernel without a jump tail
        if constexpr (JHP == 0) {
            if (fast && a.pot.kind == NFMC_POT_QUADRATIC) {
                auto kern = mala_kernel<CPL, LPC, QuadraticPot, true, 0, 7>;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, a, sqrt2h, tiles, jd);
                return NFMC_OK;
            }
        }
        return NFMC_EUNSUPPORTED;
    }
    if (a.pot.kind == NFMC_POT_FUNNEL) {
        if (fast) NFMC_L(FunnelPot, true) else NFMC_L(FunnelPot, false)
    } else {
        if (fast) NFMC_L(QuadraticPot, true) else NFMC_L(QuadraticPot, false)
    }
#undef NFMC_L
    return NFMC_OK;
}

template <int CPL, int LPC, int JHP>
int launch_hmc_cfg(const NfmcHmcArgs& a, const JumpDev& jd, bool fast, int64_t tiles, int grid, hipStream_t st) {
    size_t lds = 0;
    if constexpr (JHP > 0)
        lds = (size_t)FlowImage<CPL, LPC, JHP>::total_floats(jd.flow.n_hidden_layers, jd.flow.n_coupling) * sizeof(float);
    if (lds > 120 * 1024) return NFMC_EUNSUPPORTED;
#define NFMC_L(POT, F)                                                                                            \
    {                                                                                                             \
        auto kern = hmc_kernel<CPL, LPC, POT, F, JHP>;                                                            \
        if (lds > 48 * 1024) {                                                                                    \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                                   \
        }                                                                                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, a, tiles, jd);                                \
    }
    if (rng_rounds(a.rng) == 7) {
        if constexpr (JHP == 0) {
            if (fast && a.pot.kind == NFMC_POT_QUADRATIC) {
                auto kern = hmc_kernel<CPL, LPC, QuadraticPot, true, 0, 7>;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, a, tiles, jd);
                return NFMC_OK;
            }
        }
        return NFMC_EUNSUPPORTED;
    }
    if (a.pot.kind == NFMC_POT_FUNNEL) {
        if (fast) NFMC_L(FunnelPot, true) else NFMC_L(FunnelPot, false)
    } else {
        if (fast) NFMC_L(QuadraticPot, true) else NFMC_L(QuadraticPot, false)
    }
#undef NFMC_L
    return NFMC_OK;
}

// all sampler layouts (JHP == 0) / the layouts shared with flow_b (JHP > 0)
#define NFMC_FOR_CFG(M)                                                                                             \
    M(4, 1) M(4, 2) M(4, 4) M(4, 8) M(4, 16) M(8, 8) M(16, 4) M(8, 16) M(16, 8) M(8, 32) M(16, 16) M(8, 64) M(16, 32) \
        M(16, 64)
#define NFMC_FOR_BCFG(M) M(4, 1) M(4, 2) M(4, 4) M(4, 8) M(8, 8) M(8, 16) M(8, 32) M(8, 64)

// defined in sampler_mala_j*.hip / sampler_hmc_j*.hip
int launch_mala_j0(const NfmcMalaArgs&, const JumpDev&, Cfg, bool, int64_t, int, float, hipStream_t);
int launch_mala_j4(const NfmcMalaArgs&, const JumpDev&, Cfg, bool, int64_t, int, float, hipStream_t);
int launch_mala_j8(const NfmcMalaArgs&, const JumpDev&, Cfg, bool, int64_t, int, float, hipStream_t);
int launch_hmc_j0(const NfmcHmcArgs&, const JumpDev&, Cfg, bool, int64_t, int, hipStream_t);
int launch_hmc_j4(const NfmcHmcArgs&, const JumpDev&, Cfg, bool, int64_t, int, hipStream_t);
int launch_hmc_j8(const NfmcHmcArgs&, const JumpDev&, Cfg, bool, int64_t, int, hipStream_t);

}  // namespace nfmc
