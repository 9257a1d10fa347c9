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
#include "imh_parallel.hpp"

using namespace nfmc;

// smallest register layout that holds d coordinates; CPL = 4 first (more lanes per row for small batches)
static PCfg imh_layout(int d) {
    PCfg c = {0, 0};
    for (const PCfg& k : kPCfgs) {
        if (k.cpl * k.lpc < d) continue;
        if (c.cpl == 0 || k.cpl * k.lpc < c.cpl * c.lpc) c = k;
    }
    return c;
}

// work buffer: rec (imh_rec_rows(k) x n x 16 bytes), bits ((imh_words(k) + 1) x n words), dwell0 (n), pad to 16 bytes, x0 (n d floats), pad to 8,
// esum (kMaxGrid x 2 dp doubles), visit (2 counters)
struct ImhLayout {
    int64_t x0_off, esum_off, visit_off, bytes;
};
static ImhLayout imh_work_layout(int64_t n, int32_t d, int32_t k) {
    const int64_t kn = n * (int64_t)k;
    const PCfg c = imh_layout(d);
    const int64_t dp = c.cpl ? c.cpl * c.lpc : d;
    ImhLayout l;
    l.x0_off = ((4 * imh_rec_rows(k) + (int64_t)imh_words(k) + 2) * n * 4 + 15) & ~15ll;
    l.esum_off = (l.x0_off + n * (int64_t)d * 4 + 7) & ~7ll;
    l.visit_off = l.esum_off + (int64_t)kMaxGrid * 2 * dp * 8;
    l.bytes = l.visit_off + 16;
    return l;
}

extern "C" int64_t nfmc_imh_parallel_work_bytes(int64_t n, int32_t d, int32_t n_steps) {
    if (n <= 0 || d <= 0 || n_steps <= 0) return 0;
    return imh_work_layout(n, d, n_steps).bytes + 64;
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
    if (f.n_hidden <= 0 || f.n_hidden > 8 || f.n_hidden_layers <= 0) return NFMC_EUNSUPPORTED;
    if (f.n_bins != 0 && (f.n_bins != kRqsBins || !(f.spline_bound > 0.f))) return NFMC_EUNSUPPORTED;
    if (a.pot.kind != NFMC_POT_QUADRATIC && a.pot.kind != NFMC_POT_FUNNEL) return NFMC_EUNSUPPORTED;
    if (a.stats.sum_x && (!a.stats.sum_x2 || !a.stats.counters || !a.stats.scratch)) return NFMC_EINVAL;
    if (a.stats.sum_x && a.stats.defer && a.stats.tail_slot != 0) return NFMC_EINVAL;
    if ((a.rng.replay_normals != nullptr) != (a.rng.replay_uniforms != nullptr)) return NFMC_EINVAL;
    if (!dry && work_bytes < nfmc_imh_parallel_work_bytes(a.n, f.d, a.n_steps)) return NFMC_ESCRATCH;
    if (!dry && (((uintptr_t)work) & 15u) != 0) return NFMC_EINVAL;   // the layout below assumes a 16-byte aligned base
    hipStream_t st = (hipStream_t)stream;
    const int64_t kn = a.n * (int64_t)a.n_steps;
    ImhWork w;
    w.rec = (float4*)work;
    w.bits = (uint32_t*)(w.rec + imh_rec_rows(a.n_steps) * a.n);
    w.dwell0 = (int32_t*)(w.bits + (int64_t)(imh_words(a.n_steps) + 1) * a.n);
    const ImhLayout wl = imh_work_layout(a.n, f.d, a.n_steps);
    w.x0 = (float*)((char*)work + wl.x0_off);       // 16-byte aligned (vector IO)
    w.esum = (double*)((char*)work + wl.esum_off);
    w.visit = (unsigned long long*)((char*)work + wl.visit_off);
    if (!dry) {
        if (!a.logq_cached) {   // flow.log_prob(x0): imh.py:214
            const int rc = nfmc_realnvp_forward_f32(&f, a.x, a.n, nullptr, nullptr, a.logq, stream);
            if (rc) return rc;
        }
        hipError_t e = hipMemcpyAsync(w.x0, a.x, (size_t)a.n * f.d * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    const int d = f.d, hp = f.n_hidden <= 4 ? 4 : 8;
    const PCfg c = imh_layout(d);
    if (!c.cpl) return NFMC_EUNSUPPORTED;
    int rc = NFMC_EUNSUPPORTED, grid = 0, dp = 0;
    if (f.n_bins != 0) {   // spline couplings: imh_parallel_rqs.hip
        rc = launch_imh_rqs(c.cpl, c.lpc, hp, a, w, st, &grid, &dp, dry);
    } else {
#define M(CPL, LPC)                   \
    if (c.cpl == CPL && c.lpc == LPC) \
        rc = hp == 4 ? launch_imh<CPL, LPC, 4>(a, w, st, &grid, &dp, dry) : launch_imh<CPL, LPC, 8>(a, w, st, &grid, &dp, dry);
        NFMC_FOR_PCFG(M)
#undef M
    }
    if (rc || dry) return rc;
    NFMC_HIP_CHECK_LAUNCH();
    if (a.stats.sum_x && !a.stats.defer) {
        hipLaunchKernelGGL(stats_finish_kernel<true>, dim3(stats_finish_grid(dp)), dim3(kFinishBlock), 0, st, a.stats.scratch, grid, dp, d,
                           a.stats, (unsigned long long)a.n * (unsigned long long)a.n_steps);
        NFMC_HIP_CHECK_LAUNCH();
    }
    return NFMC_OK;
}
