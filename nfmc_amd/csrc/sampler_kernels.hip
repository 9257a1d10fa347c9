// K1/K2 (+K6, K7): C entry points of the fused Langevin / HMC samplers.  Kernels: sampler_impl.hpp; the
// template instantiations live in sampler_{mala,hmc}_j{0,4,8}.hip (j = conditioner width of the optional
// jump tail, 0 = none) so they compile in parallel.
#include "sampler_impl.hpp"

namespace nfmc {

// (CPL, LPC) layouts, ordered by capacity CPL*LPC; equal capacities in order of measured preference
// (CPL = 8 keeps 4 waves/SIMD resident).
static const Cfg kCfgs[] = {{4, 1}, {4, 2}, {4, 4}, {4, 8}, {8, 8}, {4, 16}, {16, 4}, {8, 16}, {16, 8}, {8, 32}, {16, 16}, {8, 64}, {16, 32}, {16, 64}};
static const Cfg kBCfgs[] = {{4, 1}, {4, 2}, {4, 4}, {4, 8}, {8, 8}, {8, 16}, {8, 32}, {8, 64}};  // jump-tail variants

static Cfg choose_cfg(int d, bool with_jump) {
    if (const char* e = getenv("NFMC_SAMPLER_CFG")) {  // "cpl,lpc" override (tuning)
        int c = 0, l = 0;
        if (sscanf(e, "%d,%d", &c, &l) == 2) {
            const Cfg* list = with_jump ? kBCfgs : kCfgs;
            const int len = with_jump ? (int)(sizeof(kBCfgs) / sizeof(Cfg)) : (int)(sizeof(kCfgs) / sizeof(Cfg));
            for (int i = 0; i < len; ++i)
                if (list[i].cpl == c && list[i].lpc == l && c * l >= d) return list[i];
        }
    }
    Cfg best = {0, 0};
    if (with_jump) {
        for (const Cfg& k : kBCfgs)
            if (k.cpl * k.lpc >= d && (best.cpl == 0 || k.cpl * k.lpc < best.cpl * best.lpc)) best = k;
    } else {
        for (const Cfg& k : kCfgs)
            if (k.cpl * k.lpc >= d && (best.cpl == 0 || k.cpl * k.lpc < best.cpl * best.lpc)) best = k;
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
    if (!store_ok(a->samples)) return NFMC_EINVAL;
    if (!rng_rounds_ok(a->rng, true)) return NFMC_EINVAL;
    if ((a->rng.replay_normals == nullptr) != (a->rng.replay_uniforms == nullptr) && (a->adjust & 1)) return NFMC_EINVAL;
    if (a->stats.sum_x && (!a->stats.sum_x2 || !a->stats.counters || !a->stats.scratch)) return NFMC_EINVAL;
    if (a->jump) {
        const NfmcJumpTail& j = *a->jump;
        if (j.flow.d != a->d || !j.counters) return NFMC_EINVAL;
        if (!j.flow.ea0_log_scale || !j.flow.ea0_shift || !j.flow.ea1_log_scale || !j.flow.ea1_shift) return NFMC_EINVAL;
        if (j.flow.n_coupling > 0 && !j.flow.weights) return NFMC_EINVAL;
        if (j.flow.n_hidden <= 0 || j.flow.n_hidden_layers <= 0) return NFMC_EINVAL;
        if (j.flow.n_hidden > 8 || a->d > 512 || j.flow.n_bins != 0) return NFMC_EUNSUPPORTED;
        if (!a->stats.sum_x) return NFMC_EINVAL;  // the jump counters travel through the statistics slab
        if (j.adjusted && (j.replay_latent != nullptr) != (j.replay_uniform != nullptr)) return NFMC_EINVAL;
    }
    return NFMC_OK;
}

template <class Args>
static bool fast_path(const Args* a, const Cfg& c) {
    // the FAST kernels assume a scalar potential with b = 0 (the carried |x|^2 of the Langevin ratio)
    return a->d == c.cpl * c.lpc && a->inv_mass_diag == nullptr && a->pot.a == nullptr && a->pot.b == nullptr &&
           (a->pot.kind != NFMC_POT_QUADRATIC || a->pot.b_scalar == 0.f) && aligned16(a->x) &&
           (!a->samples.base || aligned16(a->samples.base));
}

// ------------------------------------------------------------------------------------------------
// Warmup: statistics fold + tuning controller in one launch of ONE workgroup (NfmcTune).  The column totals of the
// call's per-workgroup partials are folded into the run's accumulators like stats_finish_kernel<true> does and kept in
// the tuning state; then the same workgroup runs the controller: mass-diagonal update over the coordinates, dual
// averaging of the step size on thread 0.  (A first version spread the fold over several workgroups and let the last
// one to arrive -- ticket counter, device-scope fences -- run the controller: on this multi-XCD part a device-scope
// release writes back the whole L2 of the XCD, 25-30 us per controller update with the sampler's 32 MB of state dirty
// in it.  Tuning launches therefore use at most kTuneGrid workgroups, so that one workgroup folds their slabs in a
// few load round trips.)
constexpr int kTuneGrid = 256;   // one load round trip of the folding workgroup (two need more than its 128 VGPRs per thread)
template <int NCHUNK>   // round trips: slabs / 256
__global__ void __launch_bounds__(kFinishBlock) tune_finish_kernel(double* __restrict__ scratch, int nblocks, int dp, int d,
                                                                   NfmcStats st, NfmcTune tn, unsigned long long attempted) {
    __shared__ double part[kFinishSlices][kFinishCols];
    const int width = 2 * dp + kStatTail;
    const int col = threadIdx.x % kFinishCols, slice = threadIdx.x / kFinishCols;
    double* __restrict__ totals = tn.state + NFMC_TUNE_WORDS;
    // The slabs were written by workgroups on all eight XCDs, so every dependent load round trip of this workgroup goes
    // through memory: a thread therefore issues ALL its loads of kGroups column groups -- 8 rows each, what kTuneGrid
    // workgroups leave per thread -- before it adds anything (5 groups cover d <= 64 in one round trip).
    constexpr int kGroups = 5, kRows = 8;   // 40 fp64 loads in flight per thread (1024 threads: 128 VGPRs each)
    for (int t0 = 0; t0 < width; t0 += kGroups * kFinishCols) {
        double acc[kGroups];
#pragma unroll
        for (int gi = 0; gi < kGroups; ++gi) acc[gi] = 0.0;
#pragma unroll
        for (int r0 = 0; r0 < NCHUNK * kRows * kFinishSlices; r0 += kRows * kFinishSlices) {   // 256 slabs per round trip
            double v[kGroups][kRows];
#pragma unroll
            for (int gi = 0; gi < kGroups; ++gi) {
                const int t = t0 + gi * kFinishCols + col;
#pragma unroll
                for (int u = 0; u < kRows; ++u) {
                    const int b = r0 + slice + u * kFinishSlices;
                    v[gi][u] = (t < width && b < nblocks) ? scratch[(size_t)b * width + t] : 0.0;
                }
            }
#pragma unroll
            for (int gi = 0; gi < kGroups; ++gi) {
                const int t = t0 + gi * kFinishCols + col;
#pragma unroll
                for (int u = 0; u < kRows; ++u) acc[gi] += v[gi][u];   // row order
#pragma unroll
                for (int u = 0; u < kRows; ++u) {
                    const int b = r0 + slice + u * kFinishSlices;
                    if (t < width && b < nblocks) scratch[(size_t)b * width + t] = 0.0;
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // one chunk's 40 loads in flight at a time (128 VGPRs per thread)
        }
#pragma unroll
        for (int gi = 0; gi < kGroups; ++gi) {
            const int t = t0 + gi * kFinishCols + col;
            const double p0 = acc[gi];
            __syncthreads();   // part[] of the previous column group has been consumed
            part[slice][col] = p0;
            __syncthreads();
            if (slice == 0 && t < width) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < kFinishSlices; ++k) s += part[k][col];
                totals[t] = s;
                if (t < dp) {
                    if (t < d) st.sum_x[t] += s;
                } else if (t < 2 * dp) {
                    if (t - dp < d) st.sum_x2[t - dp] += s;
                } else if (t == 2 * dp) {
                    st.counters[NFMC_CNT_ACCEPTED] += (unsigned long long)(s + 0.5);
                    st.counters[NFMC_CNT_ATTEMPTED] += attempted;
                } else if (t == 2 * dp + 1) {
                    st.counters[NFMC_CNT_NONFINITE] += (unsigned long long)(s + 0.5);
                }
            }
        }
    }
    __syncthreads();   // totals[] written by this workgroup's own threads: workgroup scope is enough
    const double n_tot = (double)attempted;
    if (tn.tune_inv_mass_diag && tn.inv_mass_diag && n_tot > 1.0) {   // mcmc/base.py:146-151
        const double beta = tn.state[NFMC_TUNE_IMD_ADJUSTMENT];
        for (int j = threadIdx.x; j < d; j += kFinishBlock) {
            const double sx = totals[j], sxx = totals[dp + j];
            // torch.var (unbiased) from one-pass sums: for chains far from the origin relative to their spread the
            // subtraction cancels (fp32 per-lane partials of x^2), so the result is held at >= 0 -- the next launch takes
            // 1 / inv_mass_diag^2 and its square root
            double var = (sxx - sx * sx / n_tot) / (n_tot - 1.0);
            var = var > 0.0 ? var : 0.0;
            tn.inv_mass_diag[j] = (float)(beta * var + (1.0 - beta) * (double)tn.inv_mass_diag[j]);
        }
    }
    if (threadIdx.x == 0 && tn.tune_step_size) {                                   // mcmc/base.py:153-161, tuning.py:22-38
        const double acc = totals[2 * dp];
        const double err = tn.state[NFMC_TUNE_TARGET] - acc / n_tot;
        const double S = tn.state[NFMC_TUNE_ERROR_SUM] + err;
        const double it = tn.state[NFMC_TUNE_ITERATION];
        const double log_raw = tn.state[NFMC_TUNE_ANCHOR] - S / (sqrt(it) * tn.state[NFMC_TUNE_GAMMA]);
        const double w = pow(it, -tn.state[NFMC_TUNE_KAPPA]);
        const double log_smooth = w * log_raw + (1.0 - w) * tn.state[NFMC_TUNE_LOG_SMOOTH];
        tn.state[NFMC_TUNE_ERROR_SUM] = S;
        tn.state[NFMC_TUNE_LOG_RAW] = log_raw;
        tn.state[NFMC_TUNE_LOG_SMOOTH] = log_smooth;
        tn.state[NFMC_TUNE_ITERATION] = it + 1.0;
        tn.state[NFMC_TUNE_STEP_SIZE] = exp(log_smooth);
    }
}

// Round 4: the tuning launches no longer run at a quarter of the machine.  With kTuneGrid workgroups a one-transition launch
// of 65536 x 64 chains is 1024 waves, one per SIMD, each walking eight chain tiles one after the other (26 us); with the full
// grid it is ~8 us, but leaves up to 2032 slabs -- which kTuneFoldWgs workgroups first fold 128 at a time (every thread has all
// its loads in flight at once, as in tune_finish_kernel) into partial slabs at the END of the scratch, and tune_finish_kernel
// then folds those.  Three launches per controller update instead of two, ~30 us instead of 47 (profiles/r04_warmup_probe.txt).
constexpr int kTuneFoldSlabs = 128, kTuneFoldWgs = 16;
__global__ void __launch_bounds__(kFinishBlock) tune_fold_kernel(double* __restrict__ scratch, int nblocks, int width,
                                                                 double* __restrict__ partial) {
    __shared__ double part[kFinishSlices][kFinishCols];
    const int col = threadIdx.x % kFinishCols, slice = threadIdx.x / kFinishCols;
    const int b0 = blockIdx.x * kTuneFoldSlabs;
    double* __restrict__ dst = partial + (size_t)blockIdx.x * width;
    constexpr int kGroups = 5, kRows = kTuneFoldSlabs / kFinishSlices;   // 20 fp64 loads in flight per thread
    for (int t0 = 0; t0 < width; t0 += kGroups * kFinishCols) {
        double v[kGroups][kRows];
#pragma unroll
        for (int gi = 0; gi < kGroups; ++gi) {
            const int t = t0 + gi * kFinishCols + col;
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const int b = b0 + slice + u * kFinishSlices;
                v[gi][u] = (t < width && b < nblocks) ? scratch[(size_t)b * width + t] : 0.0;
            }
        }
#pragma unroll
        for (int gi = 0; gi < kGroups; ++gi) {
            const int t = t0 + gi * kFinishCols + col;
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < kRows; ++u) acc += v[gi][u];   // row order
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const int b = b0 + slice + u * kFinishSlices;
                if (t < width && b < nblocks) scratch[(size_t)b * width + t] = 0.0;
            }
            __syncthreads();   // part[] of the previous column group has been consumed
            part[slice][col] = acc;
            __syncthreads();
            if (slice == 0 && t < width) {
                double sum = 0.0;
#pragma unroll
                for (int k = 0; k < kFinishSlices; ++k) sum += part[k][col];
                dst[t] = sum;
            }
        }
    }
}

// grid of a tuning launch and whether its slabs are folded in two levels (the scratch then holds the partial slabs behind
// the last workgroup's)
static int tune_grid(int64_t tiles, int dp, int64_t scratch_bytes, bool* two_level) {
    const int64_t width = 2 * dp + kStatTail;
    *two_level = tiles > kTuneGrid && scratch_bytes >= (int64_t)kMaxGrid * width * (int64_t)sizeof(double) && !getenv("NFMC_TUNE_ONE_LEVEL");
    const int cap = *two_level ? kMaxGrid - kTuneFoldWgs : kTuneGrid;
    return (int)(tiles < cap ? tiles : cap);
}

// the fold + controller of one update behind a tuning launch of `grid` workgroups
static void tune_update(const NfmcStats& stats, const NfmcTune& tune, int grid, bool two_level, int dp, int d, unsigned long long attempted,
                        hipStream_t st) {
    const int width = 2 * dp + kStatTail;
    if (two_level) {
        double* partial = stats.scratch + (size_t)(kMaxGrid - kTuneFoldWgs) * width;
        const int wgs = (grid + kTuneFoldSlabs - 1) / kTuneFoldSlabs;
        hipLaunchKernelGGL(tune_fold_kernel, dim3(wgs), dim3(kFinishBlock), 0, st, stats.scratch, grid, width, partial);
        hipLaunchKernelGGL(tune_finish_kernel<1>, dim3(1), dim3(kFinishBlock), 0, st, partial, wgs, dp, d, stats, tune, attempted);
    } else {
        hipLaunchKernelGGL(tune_finish_kernel<1>, dim3(1), dim3(kFinishBlock), 0, st, stats.scratch, grid, dp, d, stats, tune, attempted);
    }
}

template <class Args>
static int check_tune(const Args& a) {
    if (!a.tune.state) return NFMC_OK;
    if (!a.stats.sum_x || a.stats.defer || a.jump) return NFMC_EINVAL;   // the controller rides on the per-call fold
    if (a.tune.tune_inv_mass_diag && (!a.tune.inv_mass_diag || a.tune.inv_mass_diag != a.inv_mass_diag)) return NFMC_EINVAL;
    return NFMC_OK;
}

static JumpDev jump_dev(const NfmcJumpTail* j) {
    JumpDev jd = {};
    if (j) {
        jd.flow = j->flow;
        jd.adjusted = j->adjusted;
        jd.replay_latent = j->replay_latent;
        jd.replay_uniform = j->replay_uniform;
        jd.mask_out = j->mask_out;
        jd.log_ratio_out = j->log_ratio_out;
    }
    return jd;
}

}  // namespace nfmc

using namespace nfmc;

extern "C" int64_t nfmc_stats_scratch_bytes(int32_t d) {
    if (d <= 0 || d > 1024) return 0;
    return stats_scratch_doubles(padded_d(d)) * (int64_t)sizeof(double);
}

extern "C" int64_t nfmc_tune_state_doubles(int32_t d) {
    if (d <= 0 || d > 1024) return 0;
    return NFMC_TUNE_WORDS + 2 * padded_d(d) + kStatTail;
}

extern "C" int nfmc_mala_steps_f32(const NfmcMalaArgs* args, nfmc_stream_t stream) {
    int rc = check_common(args);
    if (rc) return rc;
    NfmcMalaArgs a = *args;
    hipStream_t st = (hipStream_t)stream;
    const int jhp = a.jump ? (a.jump->flow.n_hidden <= 4 ? 4 : 8) : 0;
    const Cfg c = choose_cfg(a.d, jhp > 0);
    if (!c.cpl) return NFMC_ESHAPE;
    const bool fast = fast_path(&a, c);
    const int dp = c.cpl * c.lpc;
    const int cpw = kWave / c.lpc;
    const int64_t tiles = (a.n + (int64_t)kWavesPerBlock * cpw - 1) / ((int64_t)kWavesPerBlock * cpw);
    bool two_level = false;
    const int grid = a.tune.state ? tune_grid(tiles, dp, a.stats.scratch_bytes, &two_level) : (int)(tiles < kMaxGrid ? tiles : kMaxGrid);
    if (a.stats.sum_x && a.stats.scratch_bytes < (int64_t)grid * (2 * dp + kStatTail) * (int64_t)sizeof(double))
        return NFMC_ESCRATCH;
    if (check_defer(a.stats, dp, a.d)) return NFMC_EINVAL;
    if ((rc = check_tune(a))) return rc;
    const float sqrt2h = (float)sqrt(2.0 * (double)a.step_size);  // math.sqrt(2*step_size), langevin.py:75
    const JumpDev jd = jump_dev(a.jump);
    unsigned long long* jc = a.jump ? a.jump->counters : nullptr;
    a.jump = nullptr;  // host pointer: never dereferenced on the device
    if (a.tune.state) {
        // warmup: `every` transitions per controller update, all pairs of the call enqueued here
        const int every = (a.tune.every > 0 && a.tune.every < a.n_steps) ? a.tune.every : a.n_steps;
        const int total = a.n_steps;
        for (int s0 = 0; s0 < total; s0 += every) {
            const int k = total - s0 < every ? total - s0 : every;
            NfmcMalaArgs b = a;
            b.n_steps = k;
            b.rng.step0 = a.rng.step0 + (uint32_t)s0;
            if (a.rng.replay_normals) b.rng.replay_normals = a.rng.replay_normals + (int64_t)s0 * a.n * a.d;
            if (a.rng.replay_uniforms) b.rng.replay_uniforms = a.rng.replay_uniforms + (int64_t)s0 * a.n;
            if (a.masks_out) b.masks_out = a.masks_out + (int64_t)s0 * a.n;
            if (a.log_ratio_out) b.log_ratio_out = a.log_ratio_out + (int64_t)s0 * a.n;
            rc = launch_mala_j0(b, jd, c, fast, tiles, grid, sqrt2h, st);
            if (rc) return rc;
            tune_update(a.stats, a.tune, grid, two_level, dp, a.d, (unsigned long long)a.n * (unsigned long long)k, st);
            NFMC_HIP_CHECK_LAUNCH();
            store_advance(a.samples, k);
        }
        return NFMC_OK;
    }
    rc = jhp == 0 ? launch_mala_j0(a, jd, c, fast, tiles, grid, sqrt2h, st)
                  : (jhp == 4 ? launch_mala_j4(a, jd, c, fast, tiles, grid, sqrt2h, st)
                              : launch_mala_j8(a, jd, c, fast, tiles, grid, sqrt2h, st));
    if (rc) return rc;
    NFMC_HIP_CHECK_LAUNCH();
    if (a.stats.sum_x && !a.stats.defer) {
        hipLaunchKernelGGL(stats_finish_kernel<true>, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, st, a.stats.scratch,
                           grid, dp, a.d, a.stats, (unsigned long long)a.n * (unsigned long long)a.n_steps, jc,
                           (unsigned long long)a.n);
        NFMC_HIP_CHECK_LAUNCH();
    }
    return NFMC_OK;
}

extern "C" int nfmc_hmc_steps_f32(const NfmcHmcArgs* args, nfmc_stream_t stream) {
    int rc = check_common(args);
    if (rc) return rc;
    if (args->n_leapfrog <= 0) return NFMC_EINVAL;
    NfmcHmcArgs a = *args;
    hipStream_t st = (hipStream_t)stream;
    const int jhp = a.jump ? (a.jump->flow.n_hidden <= 4 ? 4 : 8) : 0;
    const Cfg c = choose_cfg(a.d, jhp > 0);
    if (!c.cpl) return NFMC_ESHAPE;
    const bool fast = fast_path(&a, c);
    const int dp = c.cpl * c.lpc;
    const int cpw = kWave / c.lpc;
    const int64_t tiles = (a.n + (int64_t)kWavesPerBlock * cpw - 1) / ((int64_t)kWavesPerBlock * cpw);
    bool two_level = false;
    const int grid = a.tune.state ? tune_grid(tiles, dp, a.stats.scratch_bytes, &two_level) : (int)(tiles < kMaxGrid ? tiles : kMaxGrid);
    if (a.stats.sum_x && a.stats.scratch_bytes < (int64_t)grid * (2 * dp + kStatTail) * (int64_t)sizeof(double))
        return NFMC_ESCRATCH;
    if (check_defer(a.stats, dp, a.d)) return NFMC_EINVAL;
    if ((rc = check_tune(a))) return rc;
    const JumpDev jd = jump_dev(a.jump);
    unsigned long long* jc = a.jump ? a.jump->counters : nullptr;
    a.jump = nullptr;
    if (a.tune.state) {
        const int every = (a.tune.every > 0 && a.tune.every < a.n_steps) ? a.tune.every : a.n_steps;
        const int total = a.n_steps;
        for (int s0 = 0; s0 < total; s0 += every) {
            const int k = total - s0 < every ? total - s0 : every;
            NfmcHmcArgs b = a;
            b.n_steps = k;
            b.rng.step0 = a.rng.step0 + (uint32_t)s0;
            if (a.rng.replay_normals) b.rng.replay_normals = a.rng.replay_normals + (int64_t)s0 * a.n * a.d;
            if (a.rng.replay_uniforms) b.rng.replay_uniforms = a.rng.replay_uniforms + (int64_t)s0 * a.n;
            if (a.masks_out) b.masks_out = a.masks_out + (int64_t)s0 * a.n;
            if (a.log_ratio_out) b.log_ratio_out = a.log_ratio_out + (int64_t)s0 * a.n;
            rc = launch_hmc_j0(b, jd, c, fast, tiles, grid, st);
            if (rc) return rc;
            tune_update(a.stats, a.tune, grid, two_level, dp, a.d, (unsigned long long)a.n * (unsigned long long)k, st);
            NFMC_HIP_CHECK_LAUNCH();
            store_advance(a.samples, k);
        }
        return NFMC_OK;
    }
    rc = jhp == 0 ? launch_hmc_j0(a, jd, c, fast, tiles, grid, st)
                  : (jhp == 4 ? launch_hmc_j4(a, jd, c, fast, tiles, grid, st) : launch_hmc_j8(a, jd, c, fast, tiles, grid, st));
    if (rc) return rc;
    NFMC_HIP_CHECK_LAUNCH();
    if (a.stats.sum_x && !a.stats.defer) {
        hipLaunchKernelGGL(stats_finish_kernel<true>, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, st, a.stats.scratch,
                           grid, dp, a.d, a.stats, (unsigned long long)a.n * (unsigned long long)a.n_steps, jc,
                           (unsigned long long)a.n);
        NFMC_HIP_CHECK_LAUNCH();
    }
    return NFMC_OK;
}
