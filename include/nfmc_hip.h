/*
 * nfmc_hip.h -- C ABI of libnfmc_hip.so: the MI355X (gfx950) implementation of the nfmc hot path.
 *
 * The reference (davidnabergoj/nfmc) is pure Python on PyTorch and has no FFI; its two seams for
 * this path are `MCMCSampler.propose` (nfmc/algorithms/sampling/mcmc/base.py:27-34) and the
 * torchflows `Flow` methods called at jump.py:205,218, imh.py:214,221, neutra.py:60.  Each entry
 * point below names the reference code it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   ownership   every pointer is a BORROWED device allocation (e.g. torch tensor.data_ptr());
 *               the library never allocates or frees user-visible memory; scratch is caller-supplied.
 *   layout      row-major contiguous fp32 `(n, d)` state, `(n,)` per-chain vectors, u8 masks.
 *   errors      return 0 on success; negative = argument error (NFMC_E*); positive = hipError_t.
 *               No C++ exception crosses the boundary.
 *   async       every call only enqueues work on `stream` (a hipStream_t) and returns; re-entrant,
 *               no global mutable state.
 *   rng         native mode: Philox4x32-10 keyed by `seed`, counter (chain id, transition, block, stream)
 *               -- spec in oracle/philox.py; replay mode: caller-supplied normals/uniforms
 *               (non-NULL replay pointers), used for parity tests against the reference's noise.
 */
#ifndef NFMC_HIP_H
#define NFMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nfmc_stream_t; /* hipStream_t */

#define NFMC_ABI_VERSION 4

enum {
    NFMC_OK = 0,
    NFMC_EINVAL = -1,       /* NULL pointer / non-positive size */
    NFMC_ESHAPE = -2,       /* shape outside what the kernels support (see nfmc_limits) */
    NFMC_EALIGN = -3,       /* pointer not aligned as required (state rows: 4 B; 16 B enables vector IO) */
    NFMC_EUNSUPPORTED = -4, /* valid request with no kernel instantiation (e.g. unknown potential kind) */
    NFMC_ESCRATCH = -5      /* scratch buffer too small: see nfmc_stats_scratch_bytes */
};

/* ---- closed-form potentials U(x) (negative log density), evaluated with their gradient in-kernel.
 * Replaces `self.target(x)` + `torch.autograd.grad` (langevin.py:66-68,80-82; hmc.py:40-48). */
enum {
    NFMC_POT_QUADRATIC = 0, /* U = sum_j a_j (x_j - b_j)^2 ; a,b per coordinate or scalar */
    NFMC_POT_FUNNEL = 1     /* U = x_0^2/(2 s^2) + sum_{i>=1} [x_i^2 / (2 e^{x_0}) + x_0/2], s = a_scalar */
};

typedef struct {
    int32_t kind;
    int32_t reserved;
    const float* a; /* (d,) or NULL -> a_scalar */
    const float* b; /* (d,) or NULL -> b_scalar */
    float a_scalar;
    float b_scalar;
} NfmcPotential;

typedef struct {
    uint64_t seed;
    uint64_t chain_offset;        /* global id of row 0 (chains sharded over GPUs keep global ids) */
    uint32_t step0;               /* transition index of the first step of this call */
    uint32_t rounds;              /* 0 or 10: Philox4x32-10 (the library's stream).  7: Philox4x32-7, an opt-in stream with
                                     30 % fewer generator instructions (the smallest round count Random123 reports as
                                     passing BigCrush); supported by nfmc_mala_steps_f32 / nfmc_hmc_steps_f32 on their
                                     exact-fit quadratic kernels (no jump tail), by the register-layout kernels of
                                     nfmc_flow_mh_steps_f32 without diagnostics, and by nfmc_philox_*; elsewhere
                                     NFMC_EUNSUPPORTED.  oracle/philox.py carries the same parameter. */
    const float* replay_normals;  /* NULL -> native Philox; else (n_steps, n, d) */
    const float* replay_uniforms; /* NULL -> native Philox; else (n_steps, n)     */
} NfmcRng;

/* Streaming statistics, all ACCUMULATED (+=) by the kernels.  Replaces
 * `MCMCExpectation.update` for f = x, x^2 and the counters of `MCMCStatistics`
 * (nfmc/algorithms/sampling/base.py:75-95,126-161; mcmc/base.py:79-86). */
enum { NFMC_CNT_ACCEPTED = 0, NFMC_CNT_ATTEMPTED = 1, NFMC_CNT_NONFINITE = 2, NFMC_CNT_WORDS = 4 };

typedef struct {
    double* sum_x;                /* (d,)  += sum over steps and chains of x     (NULL: statistics off) */
    double* sum_x2;               /* (d,)  += ... of x^2 */
    unsigned long long* counters; /* (NFMC_CNT_WORDS,) */
    double* scratch;              /* per-workgroup partials, >= nfmc_stats_scratch_bytes(d) bytes */
    int64_t scratch_bytes;
    int32_t defer;                /* 0: every call folds its partials into sum_x / sum_x2 / counters before it
                                     returns control of the stream (one extra small kernel per call).
                                     1: the call only ADDS its per-workgroup partials to `scratch`, which the caller
                                     zeroed once; nfmc_stats_fold_f32 folds them later (one fold per sample() instead
                                     of one per launch).  Supported by K1, K2, the flow-MH kernels and K7; the NeuTra
                                     entry points return NFMC_EUNSUPPORTED. */
    int32_t tail_slot;            /* defer = 1 only: 0 = accepted / non-finite counts are MCMC counts, 2 = this call
                                     is a jump (they go to the jump counters at fold time) */
} NfmcStats;

int64_t nfmc_stats_scratch_bytes(int32_t d);

/* Fold the deferred partials in stats->scratch into sum_x, sum_x2 and counters (+= ; ATTEMPTED += attempted), the
 * jump slots into jump_counters (may be NULL; ATTEMPTED += jump_attempted), and zero the scratch again.  The
 * fold order is fixed (run-to-run bitwise equal).  d = flattened event size the partials were produced with. */
int nfmc_stats_fold_f32(const NfmcStats* stats, int32_t d, uint64_t attempted, unsigned long long* jump_counters,
                        uint64_t jump_attempted, nfmc_stream_t stream);

/* ---- RealNVP (the build's spec: DESIGN.md "RealNVP spec"; stands in for torchflows.RealNVP,
 * call sites nfmc/util.py:280-281). */
typedef struct {
    int32_t d;                /* flattened event size */
    int32_t n_coupling;       /* number of (ReversePermutation, AffineCoupling) pairs */
    int32_t n_hidden;         /* H */
    int32_t n_hidden_layers;  /* conditioner hidden layers (>= 1) */
    float min_scale;          /* m in alpha = exp(u/2 + log(1-m)) + m  (1 = additive coupling, 'nice') */
    int32_t n_bins;           /* 0: affine coupling.  8: rational-quadratic spline coupling ('c-rqnsf') with 8 bins on
                                 [-spline_bound, spline_bound], identity outside; the last conditioner layer then has
                                 (3*n_bins - 1) * d_b outputs, target-major: [t*(3K-1) + (K widths | K heights | K-1
                                 derivatives)] (spec: DESIGN.md section 4).  Spline flows run on the one-chain-per-lane
                                 kernels with n_hidden <= 32 (forward, inverse, flow-MH, and the NeuTra reverse sweep:
                                 hand-written adjoints of the spline, csrc/flow_device.hpp rqs_inverse_backward). */
    const float* ea0_log_scale; /* (d,) first ElementwiseAffine */
    const float* ea0_shift;
    const float* ea1_log_scale; /* (d,) last ElementwiseAffine */
    const float* ea1_shift;
    const float* weights;     /* per coupling layer (logical, un-permuted coordinates), HP = padded H:
                                 W1T (d_a, HP) | b1 (HP) | [WhT (HP_in, HP_out) | bh (HP)] x (n_hidden_layers-1)
                                 | W3 (2 d_b, HP) | b3 (2 d_b);  W1T/WhT are the TRANSPOSES of the PyTorch Linear
                                 weights (in, out), W3 keeps (out, in); rows [0, d_b) of W3 give u_alpha, rows
                                 [d_b, 2 d_b) give u_beta; d_a = d/2, d_b = d - d_a */
    int64_t layer_stride;     /* floats between consecutive coupling layers */
    float spline_bound;       /* B (n_bins > 0) */
    int32_t reserved;
    float* scratch;           /* workspace of the streamed matrix-core kernels (n_hidden 33..128 at d = 32 .. 512 a multiple of
                                 32 other than 64 / 128): >= nfmc_flow_scratch_bytes(flow, n, ...) bytes, 16-byte aligned, private
                                 to one stream at a time; NULL / 0 elsewhere.  The library never allocates: a call that needs it
                                 and does not find it returns NFMC_ESCRATCH */
    int64_t scratch_bytes;
} NfmcRealNVP;

/* Bytes of NfmcRealNVP.scratch that nfmc_realnvp_forward_f32 / nfmc_realnvp_inverse_f32 (with_gradient = 0) or
 * nfmc_neutra_potential_grad_f32 (with_gradient = 1) need for n rows: the state (and gradient) of the resident workgroups'
 * chains stream through it, one private row per lane group of a workgroup SLOT -- at most 256 x 128 rows of d floats (x 2
 * with the gradient) whatever n.  0 for every shape that runs register- or LDS-resident. */
int64_t nfmc_flow_scratch_bytes(const NfmcRealNVP* flow, int64_t n, int32_t with_gradient);

/* HP: n_hidden padded to the kernels' width (4, 8, 16, 32 on the VALU path; 64 or 128 on the matrix-core path,
 * whose blob also carries every matrix in both orientations: csrc/mfma_device.hpp); 0 = unsupported. */
int32_t nfmc_realnvp_padded_hidden(int32_t n_hidden);
int64_t nfmc_realnvp_layer_floats(int32_t d, int32_t n_hidden, int32_t n_hidden_layers);
/* the same for any coupling kind (n_bins as in NfmcRealNVP; 0 = affine) */
int64_t nfmc_coupling_layer_floats(int32_t d, int32_t n_hidden, int32_t n_hidden_layers, int32_t n_bins);

/* ---- Kept states.  Replaces `MCMCSamples.add` (nfmc/algorithms/sampling/base.py:234-263): the state after a
 * transition is kept iff the transition's global index is a multiple of `thinning`, and only the newest `max_samples`
 * kept states survive.  Both decisions are made BEFORE the launch, so a kernel writes only rows that will still be
 * there at the end, into a store of `ring_rows` rows of n*d floats that is never larger than max_samples rows
 * (the reference slices its host-side list after the fact).  The store is a ring: the j-th kept state of a run goes to
 * row j mod ring_rows; the caller reads it back in order from the oldest surviving row. */
typedef struct {
    float* base;        /* NULL: keep nothing */
    int32_t stride;     /* thinning: every stride-th transition is kept (>= 1) */
    int32_t countdown;  /* transitions of THIS call to skip before its first kept one: (-seen) mod stride, where seen =
                           transitions offered to the store so far */
    int32_t ring_rows;  /* rows of the store (>= 1); kept rows wrap around */
    int32_t row;        /* ring row of this call's first kept transition: ceil(seen / stride) mod ring_rows */
} NfmcSampleStore;

/* ---- Warmup tuning on the device.  Replaces `MetropolisSampler.update_kernel` + `DualAveraging.step`
 * (nfmc/algorithms/sampling/mcmc/base.py:142-161, tuning.py:15-41): after the transitions of a call, the controller
 *     m_j <- beta var_j + (1 - beta) m_j            var_j = unbiased variance of coordinate j over the call's states
 *     err = target - accepted / attempted;  S += err;  log h_raw = anchor - S / (sqrt(t) gamma);
 *     w = t^-kappa;  log h-bar <- w log h_raw + (1 - w) log h-bar;  t += 1;  h = exp(log h-bar)
 * runs inside the statistics fold of the call (its last workgroup), on state that lives in device memory: the next call
 * reads its step size and mass diagonal from there, so a warmup is a stream of launches without one host round trip.
 * One transition per call reproduces the reference update for update; n_steps = K updates once per K transitions with
 * the statistics pooled over them.  `state` words (fp64): */
enum {
    NFMC_TUNE_STEP_SIZE = 0,   /* h: read by the sampler kernel, rewritten by the controller */
    NFMC_TUNE_LOG_SMOOTH = 1,  /* log h-bar */
    NFMC_TUNE_ERROR_SUM = 2,   /* S */
    NFMC_TUNE_ITERATION = 3,   /* t */
    NFMC_TUNE_ANCHOR = 4,      /* log(10 h0) */
    NFMC_TUNE_LOG_RAW = 5,     /* log h_raw (output) */
    NFMC_TUNE_TARGET = 6,      /* target acceptance rate (0.651) */
    NFMC_TUNE_KAPPA = 7,
    NFMC_TUNE_GAMMA = 8,
    NFMC_TUNE_IMD_ADJUSTMENT = 9, /* beta */
    NFMC_TUNE_TICKET = 10,     /* internal: workgroup counter of the fold (zero between calls) */
    NFMC_TUNE_WORDS = 12       /* followed by 2 * padded_d + 4 words of column totals: nfmc_tune_state_doubles(d) in all */
};
typedef struct {
    double* state;             /* NULL: no tuning (step_size / inv_mass_diag of the call are used as given) */
    float* inv_mass_diag;      /* (d,) device, updated in place when tune_inv_mass_diag; must be the call's inv_mass_diag */
    int32_t tune_step_size;
    int32_t tune_inv_mass_diag;
    int32_t every;             /* transitions per controller update.  0 or >= n_steps: one update after the call's n_steps
                                  transitions.  Otherwise the call enqueues ceil(n_steps / every) kernel + controller
                                  pairs itself (noise, masks and the sample store advance between them), so a whole
                                  warmup is ONE call: no host work between updates */
    int32_t reserved;
} NfmcTune;
int64_t nfmc_tune_state_doubles(int32_t d);

/* Optional tail of a sampler call: after the n_steps inner transitions, ONE flow-proposal Metropolis jump
 * (jump.py:205-243: flow.sample, flow.log_prob, 2 target calls, log u < log alpha, masked update) on the
 * same registers, as transition rng.step0 + n_steps.  With a tail, `samples` is offered n_steps + 1 states and the
 * moments include the post-jump state.  Narrow conditioners only (n_hidden <= 8, d <= 512); otherwise the
 * call returns NFMC_EUNSUPPORTED and the caller issues nfmc_flow_mh_steps_f32 separately. */
typedef struct {
    NfmcRealNVP flow;
    int32_t adjusted;             /* 0: accept every proposal (adjusted_jumps=False) */
    int32_t reserved;
    unsigned long long* counters; /* (NFMC_CNT_WORDS,) jump counters: accepted / attempted / nonfinite */
    const float* replay_latent;   /* NULL -> native stream 2; else (n, d) latents */
    const float* replay_uniform;  /* NULL -> native stream 3; else (n,) */
    uint8_t* mask_out;            /* NULL or (n,) */
    float* log_ratio_out;         /* NULL or (n,) */
} NfmcJumpTail;

/* ---- K1 (+K6, K7): n_steps fused Langevin transitions.
 * Replaces `Langevin.propose` + the masked update + counters + moments of `MCMCSampler.sample`
 * (langevin.py:61-122; mcmc/base.py:74-90). */
typedef struct {
    float* x;                   /* (n, d) in/out */
    int64_t n;
    int32_t d;
    int32_t n_steps;            /* 1..NFMC_MAX_STEPS_PER_CALL */
    float step_size;
    int32_t adjust;             /* bit 0: Metropolis test (1 = MALA, 0 = ULA); bit 1: random-walk proposal
                                   x' = x + inv_mass_diag * eps instead of the Langevin drift (mh.py:51-60:
                                   3 = MH, 2 = RandomWalk) */
    const float* inv_mass_diag; /* (d,) or NULL = ones (mcmc/base.py:113-116) */
    NfmcPotential pot;
    NfmcRng rng;
    NfmcStats stats;
    NfmcSampleStore samples;    /* state after every step, thinned / windowed as described above (MCMCSamples.add) */
    uint8_t* masks_out;         /* NULL or (n_steps, n) accept masks (tests / split path) */
    float* log_ratio_out;       /* NULL or (n_steps, n) */
    const NfmcJumpTail* jump;   /* NULL or a jump to run after the inner transitions (host pointer) */
    NfmcTune tune;              /* device-side tuning (warmup); needs stats with defer = 0 and no jump tail */
} NfmcMalaArgs;

int nfmc_mala_steps_f32(const NfmcMalaArgs* args, nfmc_stream_t stream);

/* ---- K2: n_steps fused HMC trajectories.  Replaces `HMC.propose`/`hmc_trajectory` (hmc.py:61-126). */
typedef struct {
    float* x;
    int64_t n;
    int32_t d;
    int32_t n_steps;
    float step_size;
    int32_t n_leapfrog;
    int32_t adjust;             /* 1 = HMC, 0 = UHMC */
    int32_t reserved;
    const float* inv_mass_diag;
    NfmcPotential pot;
    NfmcRng rng;
    NfmcStats stats;
    NfmcSampleStore samples;
    uint8_t* masks_out;
    float* log_ratio_out;
    const NfmcJumpTail* jump;   /* as in NfmcMalaArgs */
    NfmcTune tune;              /* as in NfmcMalaArgs */
} NfmcHmcArgs;

int nfmc_hmc_steps_f32(const NfmcHmcArgs* args, nfmc_stream_t stream);

/* K4: x -> z, logdet_forward; log_prob = N(z;0,I) + logdet (either output may be NULL).
 * Replaces `bijection.forward` / `Flow.log_prob` (jump.py:218, imh.py:214). */
int nfmc_realnvp_forward_f32(const NfmcRealNVP* flow, const float* x, int64_t n, float* z, float* logdet,
                             float* log_prob, nfmc_stream_t stream);

/* K3: z -> x, logdet_inverse; z == NULL draws z ~ N(0,I) from `rng` (stream 2); log_q = log q(x).
 * Replaces `bijection.inverse` / `Flow.sample(n, return_log_prob=True)` (jump.py:205, imh.py:221, neutra.py:60). */
int nfmc_realnvp_inverse_f32(const NfmcRealNVP* flow, const float* z, int64_t n, float* x, float* logdet,
                             float* log_q, const NfmcRng* rng, nfmc_stream_t stream);

/* ---- K3+K4+K6+K7 fused: n_steps independent-MH transitions with the flow as proposal.
 * n_steps = 1, logq_cached = 0 is the jump of `JumpNFMC.sample` (jump.py:205-243);
 * n_steps = K with the cached log q(x) is `FixedIMH.sample` (imh.py:214-249). */
typedef struct {
    float* x;                 /* (n, d) in/out */
    float* logq;              /* (n,) cached log q(x): read if logq_cached, always written */
    int64_t n;
    int32_t n_steps;
    int32_t logq_cached;
    int32_t adjusted;         /* 0: accept every proposal (adjusted_jumps=False, jump.py:228-229) */
    int32_t reserved;
    NfmcRealNVP flow;
    NfmcPotential pot;
    NfmcRng rng;              /* replay_normals are the latents z (n_steps, n, d) */
    NfmcStats stats;          /* counters use the ACCEPTED/ATTEMPTED words */
    NfmcSampleStore samples;
    uint8_t* masks_out;
    float* log_ratio_out;
} NfmcFlowMhArgs;

int nfmc_flow_mh_steps_f32(const NfmcFlowMhArgs* args, nfmc_stream_t stream);

/* NFMC_OK when nfmc_flow_mh_steps_f32 has a kernel for `args` (launches nothing), NFMC_EUNSUPPORTED when not -- e.g.
 * ragged d > ~300 with a narrow conditioner whose weight image does not fit the LDS and whose wave tiles do not
 * either: the caller then composes the transition from nfmc_realnvp_inverse_f32 / nfmc_realnvp_forward_f32 and
 * nfmc_mh_accept_select_f32 (the split path every foreign flow object takes; jump.py:205-231). */
int nfmc_flow_mh_supported_f32(const NfmcFlowMhArgs* args);

/* The same run of n_steps independent-MH transitions (FixedIMH.sample, imh.py:200-255) as a data-parallel problem:
 * the proposals of an independence sampler do not depend on the state, so all n * n_steps of them are evaluated at
 * once, a per-chain scan (one lane per chain) applies the Metropolis tests, and proposals are replayed for the moments /
 * sample store / final state: the accepted ones weighted by their dwell times, or -- when more than about half are
 * accepted and no samples are stored -- only the corrections to the sums the proposal pass kept over all of them.
 * Same noise streams and results as
 * nfmc_flow_mh_steps_f32 (states and masks bit for bit; moments up to summation order).  Faster at every
 * chain count measured (3x at n = 1000, 1.15x at n = 65536, d = 64): few chains no longer leave the GPU idle, and the
 * accept uniforms are drawn once per (chain, step) instead of once per lane.  Requires adjusted = 1, n_hidden <= 8 (affine or 8-bin spline couplings), n_steps <= NFMC_IMH_PARALLEL_MAX_STEPS;
 * `work` >= nfmc_imh_parallel_work_bytes(n, d, n_steps) bytes of device scratch, 16-byte aligned. */
#define NFMC_IMH_PARALLEL_MAX_STEPS 65536
int64_t nfmc_imh_parallel_work_bytes(int64_t n, int32_t d, int32_t n_steps);
int nfmc_imh_parallel_f32(const NfmcFlowMhArgs* args, void* work, int64_t work_bytes, nfmc_stream_t stream);
/* NFMC_OK when nfmc_imh_parallel_f32 has a kernel for `args` (validates, launches nothing). */
int nfmc_imh_parallel_supported_f32(const NfmcFlowMhArgs* args);

/* ---- K5 + K2: HMC in latent space on U~(z) = U(f^-1(z)) - logdet_inv(z), gradient by a hand-written
 * VJP through the coupling stack.  Replaces `NeuTra.adjusted_target` under `HMC.propose`
 * (neutra.py:58-68,109-129; hmc.py:40-48,96-126). */
typedef struct {
    float* z;                 /* (n, d) latent state in/out */
    int64_t n;
    int32_t n_steps;
    int32_t n_leapfrog;
    float step_size;
    int32_t adjust;
    const float* inv_mass_diag;
    NfmcRealNVP flow;
    NfmcPotential pot;
    NfmcRng rng;
    NfmcStats stats;          /* moments of the latent z (reference quirk) */
    NfmcSampleStore samples;
    uint8_t* masks_out;
    float* log_ratio_out;
    float* scratch;           /* matrix-core path (n_hidden > 32) only: >= nfmc_neutra_scratch_bytes(...) bytes */
    int64_t scratch_bytes;
} NfmcNeutraHmcArgs;

/* 0 for the VALU path (n_hidden <= 32).  Matrix-core path at d = 64 / 128: 2 (n, d) tiles + 2 (n,) vectors, plus the activation
 * checkpoints of the resident workgroups (hidden activations, alpha, beta of every coupling layer: the reverse sweep
 * reads them back instead of recomputing them) -- at most 256 workgroups x 128 chains, whatever n.  At the other multiples
 * of 32 (trajectory composed from the streamed gradient kernel, csrc/mfma_wide.hip): 4 (n, d) arrays + 3 (n,) vectors + the
 * gradient kernel's own workspace (nfmc_flow_scratch_bytes with the gradient; NfmcRealNVP.scratch is not used by this call). */
int64_t nfmc_neutra_scratch_bytes(int64_t n, int32_t d, int32_t n_hidden, int32_t n_hidden_layers, int32_t n_coupling);

int nfmc_neutra_hmc_steps_f32(const NfmcNeutraHmcArgs* args, nfmc_stream_t stream);

/* Adjusted potential and its gradient alone (tests; split path).  n_hidden <= 32: every d <= 512.  n_hidden 33..128
 * (matrix cores): d = 64 / 128 register-resident, every other multiple of 32 up to 512 with the state and the gradient
 * streamed through the caller's workspace NfmcRealNVP.scratch (nfmc_flow_scratch_bytes; csrc/mfma_wide.hip); other d:
 * NFMC_EUNSUPPORTED.  nfmc_realnvp_forward_f32 / nfmc_realnvp_inverse_f32 use the same streamed kernels at those d. */
int nfmc_neutra_potential_grad_f32(const NfmcRealNVP* flow, const NfmcPotential* pot, const float* z, int64_t n,
                                   float* u_out, float* grad_out, nfmc_stream_t stream);

/* ---- split path for arbitrary Python targets (U and grad U come from torch autograd on the GPU).
 * K6: mask = log(u) < lp_t' - lp_t + lp_q - lp_q' (nfmc/util.py:382-392), x[mask] = x'[mask]
 * (mcmc/base.py:77), optional carried per-chain scalars, counters. */
typedef struct {
    float* x;                 /* (n, d) in/out */
    const float* x_prime;     /* (n, d) */
    int64_t n;
    int32_t d;
    int32_t n_carry;          /* 0..2 per-chain vectors selected with the same mask (imh.py:233) */
    const float* log_ratio;   /* (n,) or NULL = accept all */
    const float* uniforms;    /* (n,) replay uniforms or NULL -> Philox stream `rng_tag` */
    float* carry[2];
    const float* carry_prime[2];
    NfmcRng rng;
    int32_t rng_tag;          /* 1 = accept stream, 3 = jump stream */
    int32_t reserved;
    NfmcStats stats;          /* moments of the post-selection x + counters */
    uint8_t* mask_out;
} NfmcSelectArgs;

int nfmc_mh_accept_select_f32(const NfmcSelectArgs* args, nfmc_stream_t stream);

/* Langevin proposal and log acceptance ratio from externally computed U, grad U
 * (langevin.py:74-76 and :31-42,88-105). */
int nfmc_langevin_propose_f32(const float* x, const float* grad_u, const float* inv_mass_diag, float step_size,
                              int64_t n, int32_t d, const NfmcRng* rng, float* x_prime, nfmc_stream_t stream);
int nfmc_langevin_log_ratio_f32(const float* x, const float* x_prime, const float* u, const float* u_prime,
                                const float* grad_u, const float* grad_u_prime, const float* inv_mass_diag,
                                float step_size, int64_t n, int32_t d, float* log_ratio, nfmc_stream_t stream);

/* K7 alone: sum_x += sum_rows x, sum_x2 += sum_rows x^2 over a (rows, d) block. */
int nfmc_moments_update_f32(const float* x, int64_t rows, int32_t d, const NfmcStats* stats, nfmc_stream_t stream);

/* Philox normals / uniforms alone (tests pin the native stream against oracle/philox.py). */
int nfmc_philox_normals_f32(const NfmcRng* rng, int32_t tag, int64_t n, int32_t d, float* out, nfmc_stream_t stream);
int nfmc_philox_uniforms_f32(const NfmcRng* rng, int32_t tag, int64_t n, float* out, nfmc_stream_t stream);

/* ---- introspection */
#define NFMC_MAX_STEPS_PER_CALL 512
/* ---- f1: maximum-likelihood (re)fit of the RealNVP proposal on the device.
 * Replaces the torchflows `Flow.fit` epochs of jump.py:139-151 (warmup), jump.py:193-201 (refit when fit_nf) and
 * imh.py:166-170 (AdaptiveIMH).  One call = one optimiser step on one batch: the mean negative log-likelihood of the rows
 * x (n, d), its gradient with respect to every parameter (hand-written reverse sweep, csrc/fit_kernels.hip) and the AdamW
 * update (torch.optim.AdamW semantics).  The trainable vector `params` has the layout of the flow's weight blob -- coupling
 * layers (n_coupling * layer_stride floats, VALU layout above), then at `ea_off` the four ElementwiseAffine vectors
 * (ea0 log-scale, ea0 shift, ea1 log-scale, ea1 shift; d4 = d rounded up to 4 floats each) -- and `flow`'s pointers must be
 * views of it (weights = params, ea0_log_scale = params + ea_off, ...), so the sampling kernels see every step at once.
 * Shapes: affine / additive couplings (n_bins = 0), one or two hidden layers; n_hidden <= 8 at d <= 512, n_hidden 9..32 at
 * d <= 256, n_hidden 33..128 at d = 64 / 128 (nfmc_flow_fit_supported_f32); other flows are trained by the host package's torch path.
 * Conditioners of width <= 8 (every default flow) run on the row-per-wave kernel (csrc/fit_rows.hpp: up to 1024 waves per
 * launch); it needs `params` 16-byte aligned and layer_stride, ea_off multiples of 4 floats (NFMC_EALIGN otherwise). */
typedef struct {
    float lr, beta1, beta2, eps, weight_decay;
    int32_t step;             /* 1-based count of applied steps, for the bias corrections */
} NfmcAdamW;

typedef struct {
    NfmcRealNVP flow;         /* views of `params` (see above) */
    float* params;            /* (n_params) trainable vector, updated in place */
    float* adam_m;            /* (n_params) first moments, zero before the first step */
    float* adam_v;            /* (n_params) second moments */
    int64_t n_params;         /* >= ea_off + 4 * d4 */
    int64_t ea_off;           /* offset of the ElementwiseAffine vectors inside params (multiple of 4) */
    float* partial;           /* scratch, >= nfmc_flow_fit_partial_floats(n, n_params) floats, ZEROED once by the caller */
    int64_t partial_floats;
    float* status;            /* device (3): [0] mean loss of the batch at the parameters BEFORE the step;
                                 [1] 1 if the step was applied, 0 if the loss was not finite (parameters untouched);
                                 [2] mean NLL of the validation rows at the parameters BEFORE the step ([0] if none) */
    const float* x_val;       /* optional (maximum likelihood only): validation rows (n_val, d), evaluated in the same launch */
    int64_t n_val;
    float* params_prev;       /* optional (n_params): receives the parameters as they were BEFORE the step -- what the
                                 validation loss of this call belongs to (best-weights bookkeeping without a second launch) */
    float* best;              /* runs (nfmc_flow_fit_epochs_f32) with keep_best_weights: (n_params) the weights the best
                                 monitored loss so far belongs to; call 0 initialises it with the starting weights */
    float* run_state;         /* runs: device, 2 x NFMC_FIT_STATE_FLOATS floats (call c reads half c & 1, writes the other) */
    float* scratch;           /* conditioners of width 33..128 (matrix-core kernel, csrc/fit_mfma.hip): the resident waves'
                                 activation checkpoints, >= nfmc_flow_fit_workspace(...) bytes, 16-byte aligned; NULL elsewhere */
    int64_t scratch_bytes;
} NfmcFlowFit;

/* The epoch loop of `Flow.fit` / `Flow.variational_fit` (torchflows, as nfmc drives it: jump.py:139-151 early stopping +
 * keep_best_weights + ValueError on divergence; imh.py:67-72; neutra.py:84-91) with its bookkeeping ON THE DEVICE: call c of
 * a run computes the batch loss (and the validation loss) at the weights w_c, decides -- in the fold kernel -- whether the
 * monitored loss improved (best weights <- w_c with validation rows, w_{c+1} without), whether to stop early, whether the
 * run diverged (non-finite loss), and applies the AdamW step unless the run has ended; calls after the end are no-ops.
 * The host enqueues any number of calls and reads `run_state` when it wants to (at the end; every few epochs under a time
 * limit).  With validation rows the run has n_epochs + 1 calls: the last one (index n_epochs) only evaluates. */
typedef struct {
    int32_t n_epochs;                 /* optimiser steps of the run at most */
    int32_t early_stopping;           /* stop once the monitored loss has not improved for more than `threshold` epochs */
    int32_t early_stopping_threshold;
    int32_t keep_best_weights;        /* maintain NfmcFlowFit.best */
    int32_t skip_nonfinite;           /* a non-finite batch loss skips the epoch (variational fits with
                                         check_for_divergences = False) instead of ending the run as diverged */
    int32_t reserved;
} NfmcFitControl;
enum {
    NFMC_FIT_BEST_LOSS = 0,    /* best monitored loss so far (+inf before the first booked epoch) */
    NFMC_FIT_SINCE_BEST = 1,   /* booked epochs since it improved */
    NFMC_FIT_APPLIED = 2,      /* optimiser steps applied (AdamW's bias corrections use applied + 1) */
    NFMC_FIT_STOPPED = 3,      /* 1: early stopping ended the run */
    NFMC_FIT_DIVERGED = 4,     /* 1: a non-finite loss ended the run (the host package raises ValueError) */
    NFMC_FIT_LAST_LOSS = 5,    /* batch loss of the latest live call */
    NFMC_FIT_LAST_VAL = 6,     /* validation loss of the latest live call */
    NFMC_FIT_BOOKED = 7,       /* epochs whose monitored loss has been booked */
    NFMC_FIT_STATE_FLOATS = 8
};

int nfmc_flow_fit_supported_f32(const NfmcRealNVP* flow);
int64_t nfmc_flow_fit_partial_floats(int64_t n, int64_t n_params);
/* Workspace of a fit of `flow` on n batch rows (+ n_val validation rows) with a trainable vector of n_params floats:
 * returns the bytes of NfmcFlowFit.scratch (0 for conditioners of width <= 32) and stores the floats `partial` must hold
 * (one slab of n_params + 4 floats per workgroup of the gradient launch).  Conditioners of width 33..128 are trained at
 * d = 64 / 128 (the shapes of the register-resident matrix-core kernels) with the trainable vector in the matrix-core blob
 * layout (csrc/mfma_device.hpp: every matrix in both orientations; the gradient kernel writes both slots, AdamW keeps them
 * equal), layer_stride >= nfmc_realnvp_layer_floats(d, n_hidden, n_hidden_layers) and a multiple of 4. */
int64_t nfmc_flow_fit_workspace(const NfmcRealNVP* flow, int64_t n, int64_t n_val, int64_t n_params, int64_t* partial_floats);
int nfmc_flow_fit_step_f32(const NfmcFlowFit* fit, const float* x, int64_t n, const NfmcAdamW* opt, nfmc_stream_t stream);
/* The same step for the variational fit of imh.py:67-72 / neutra.py:84-91 (`Flow.variational_fit`): rows z (n, d) are
 * latents drawn from N(0, I) by the caller, the loss is the reverse KL estimate mean[log q(x) - log p(x)], x = f^-1(z),
 * with -log p = the closed-form potential `pot` (its gradient is evaluated in the kernel). */
int nfmc_flow_variational_fit_step_f32(const NfmcFlowFit* fit, const NfmcPotential* pot, const float* z, int64_t n,
                                       const NfmcAdamW* opt, nfmc_stream_t stream);
/* Calls first_call .. first_call + n_calls - 1 of a run (see NfmcFitControl), enqueued back to back: pot = NULL maximum
 * likelihood on rows x (n, d), else the variational fit on latents; call c uses the rows at x + (c - first_call) *
 * epoch_stride floats (0: the same rows every epoch; variational fits draw fresh latents per epoch).  AdamW's moments
 * count as zero at call 0 and `opt->step` is ignored (the device counts applied steps). */
int nfmc_flow_fit_epochs_f32(const NfmcFlowFit* fit, const NfmcPotential* pot, const float* x, int64_t n,
                             int64_t epoch_stride, const NfmcAdamW* opt, const NfmcFitControl* ctl, int32_t first_call,
                             int32_t n_calls, nfmc_stream_t stream);

/* nn.Parameters <-> trainable vector in ONE launch.  A piece is one parameter tensor, contiguous (rows, cols) floats at
 * `param`, whose element (r, c) lives at vec[vec_off + r * vec_row_stride + c * vec_col_stride] (the blob stores the first
 * conditioner layer transposed and every hidden width padded).  to_vector = 1 gathers the parameters into the vector,
 * 0 scatters the vector into them.  Replaces one slice copy per tensor of `Flow.fit`'s epilogue / prologue. */
#define NFMC_BLOB_MAX_PIECES 16
typedef struct {
    float* param;
    int64_t vec_off;
    int32_t rows, cols;
    int32_t vec_row_stride, vec_col_stride;
} NfmcBlobPiece;
int nfmc_flow_blob_copy_f32(float* vec, const NfmcBlobPiece* pieces, int32_t n_pieces, int32_t to_vector, nfmc_stream_t stream);

/* The shuffled split of the refit buffer (`train_val_split`, tuning.py:44-65: pooled rows shuffled by torch.randperm, cut
 * at train_pct, capped) without materialising a permutation of all N rows: out[i] = x[pi(first + i)], i < m, with pi a
 * keyed pseudo-random PERMUTATION of [0, N) (6-round balanced Feistel network on ceil(log2 N) bits + cycle walking, keys
 * from `seed` by splitmix64; nfmc_rows_sample_index evaluates pi on the host; oracle/shuffle.py restates it).  Rows
 * [0, a) and [a, a + b) of one permutation are what the reference's (train, validation) pair is in distribution: distinct
 * rows, uniformly placed.  `index_out` (optional, (m,) int64) receives the source rows. */
#define NFMC_PRP_ROUNDS 6
int nfmc_rows_sample_f32(const float* x, int64_t n, int32_t d, uint64_t seed, int64_t first, float* out, int64_t m,
                         int64_t* index_out, nfmc_stream_t stream);
int64_t nfmc_rows_sample_index(int64_t n, uint64_t seed, int64_t i);

typedef struct {
    int32_t abi_version;
    int32_t max_d_sampler;   /* largest d for nfmc_mala/hmc_steps */
    int32_t max_d_flow;      /* largest d for the RealNVP kernels */
    int32_t max_hidden_valu; /* H up to which the VALU conditioner path is used */
    int32_t max_hidden;      /* largest H (MFMA path) */
    int32_t max_steps_per_call;
} NfmcLimits;

int nfmc_limits(NfmcLimits* out);
const char* nfmc_error_string(int code);
/* sha256 (hex) of the sources and flags this library was built from (nfmc_amd/build.py: _digest): profiles record it
 * so that counter-derived figures can be tied to the code that was measured.  Static storage; never NULL. */
const char* nfmc_build_digest(void);

#ifdef __cplusplus
}
#endif
#endif /* NFMC_HIP_H */
