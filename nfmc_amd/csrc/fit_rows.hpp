// f1, round 4: the flow (re)fit's gradient kernel for narrow conditioners (HP = 4 / 8: every default flow) with the
// batch row spread over the 64 LANES of a wave instead of one row per lane.
//
// Why.  Round 3's fit_grad_kernel (fit_kernels.hip; still used for HP = 16 / 32) gives every lane a whole row: 64 rows per
// wave, so a 4096-row refit is 64 waves on a 1024-SIMD machine, each a dependent chain of ~10^4 instructions behind scalar
// weight loads (280 us per epoch at the C5 refit shape, < 1 % of any roof).  An epoch is ~0.2 GFLOP and 4 MB: the only
// thing that matters is the length of the longest dependent chain, so the row is the unit of PARALLELISM here:
//   * lanes = coordinates.  A wave owns S rows at a time (slots: S independent chains the scheduler interleaves); lane g
//     holds, per row, elements g, g + 64, ... of the two physical halves of the row (CH = 1 / 2 / 4 registers per half: d <= 128 CH).
//     A coupling layer's conditioner input is one half, its targets the other; the reversal between layers only swaps the
//     roles and reverses the ORDER in which the weight rows are read -- the state never moves across lanes.
//   * the first conditioner GEMV is a per-lane partial (CH x HP multiply-adds) + a reduce-scatter over the wave that leaves
//     lane g with hidden unit g mod HP (DPP inside a 16-lane row, two ds_bpermute across rows); bias + tanh on that one
//     unit; an all-gather by quad broadcasts returns all HP activations to every lane in QUAD-MAJOR order (own quad's units
//     first): position q holds unit q ^ (lane & 4).  The lane's weight slices are read from LDS in that order (two 16-byte
//     reads with the halves swapped for the odd quads), so nothing is ever re-ordered.
//   * weight gradients are sums over ROWS of outer products; with lanes = coordinates they are lane-local: the lane that
//     owns coordinate j accumulates dW1[j][:], dW3[t][:], db3[t] in registers over the S rows of the tile, hidden-unit
//     lanes accumulate dWh[u][:], db1[u], dbh[u].  No reductions, no atomics.  Per layer the four waves of a workgroup
//     stage their accumulators in LDS and add them in wave order into the workgroup's slab of partial gradients; the fold
//     kernel adds the slabs in slab order: every sum has a fixed association (run-to-run bitwise).
//   * no activation is stored: going backward a layer's input is rebuilt from its output and its conditioner re-evaluated
//     from the unchanged half (as in NeuTra's reverse sweep); accumulators are per layer, so any number of layers fits.
//   * the whole trainable vector (coupling blobs + the four ElementwiseAffine vectors, <= 96 KB) is staged into LDS once per
//     workgroup in one global round trip; 4 waves per workgroup, one workgroup per CU, up to 1024 waves per launch.
// Reference semantics: torchflows' Flow.fit / Flow.variational_fit as nfmc calls them (jump.py:139-151,193-201;
// imh.py:60-75,166-170; neutra.py:84-91); the arithmetic is oracle/flow.py's (tests/test_gpu_fit.py: autograd of it).
#pragma once

#include "flow_device.hpp"

namespace nfmc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kFrWaves = 4, kFrThreads = 256;
constexpr int kFitTailFloats = 4;   // per-slab tail: loss sum, rows, validation loss sum, validation rows

struct FitRowsArgs {
    NfmcRealNVP f;
    NfmcPotential pot;
    const float* x;
    int64_t n;
    const float* xv;
    int64_t nv;
    float* partial;
    int64_t pstride;
    int64_t ea_off;
    int d4;
    int64_t n_params;
    int64_t tiles4;     // train tiles of S rows, rounded up to a multiple of the waves of a workgroup
    int64_t vtiles;
    const float* params;      // the trainable vector (f's pointers are views of it)
    const float* run_state;   // optional (8): [3] early-stopped, [4] diverged -> the launch does nothing
};

template <int HP, int CH>
__host__ __device__ constexpr int fit_rows_nacc() { return 3 * CH * HP + 2 * CH + HP + 2; }
// accumulator registers a wave stages per flush round: all of them up to CH = 2, half of them at CH = 4 (d <= 512)
__host__ __device__ constexpr int fit_rows_nstage(int hp, int ch) {
    return ch <= 2 ? 3 * ch * hp + 2 * ch + hp + 2 : (3 * ch * hp + 2 * ch + hp + 2 + 1) / 2;
}

__host__ __device__ inline size_t fit_rows_lds_bytes(int64_t n_params, int hp, int ch) {
    return ((size_t)((n_params + 3) / 4 * 4) + (size_t)kFrWaves * fit_rows_nstage(hp, ch) * 64 + (size_t)kFrWaves * kFitTailFloats) * sizeof(float);
}

// ---- quad-major reduce-scatter / all-gather over the 64 lanes of a wave.  Lane g owns hidden unit u = g % HP.
// Position q of a lane's HP-vector means unit q ^ (g & 4) for HP = 8 (own quad's four units first), unit q for HP = 4.
template <int HP>
__device__ __forceinline__ int unit_of(int q, int lane) {
    return HP == 8 ? (q ^ (lane & 4)) : q;
}

template <int CTRL>
__device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// h[q] = this lane's partial of the unit at position q.  Returns the sum over all 64 lanes of unit (lane % HP).
template <int HP>
__device__ __forceinline__ float wave_reduce_scatter(const float (&h)[HP]) {
    static_assert(HP == 4 || HP == 8, "narrow conditioners");
    constexpr uint64_t M0 = 0xAAAAAAAAAAAAAAAAull, M1 = 0xCCCCCCCCCCCCCCCCull;
    float a[HP / 2];
#pragma unroll
    for (int r = 0; r < HP / 2; ++r)
        a[r] = select_f32(M0, h[2 * r + 1], h[2 * r]) + dpp_mov<0xB1>(select_f32(M0, h[2 * r], h[2 * r + 1]));
    float c = select_f32(M1, a[1], a[0]) + dpp_mov<0x4E>(select_f32(M1, a[0], a[1]));
    if constexpr (HP == 8) {
        const float c2 = select_f32(M1, a[3], a[2]) + dpp_mov<0x4E>(select_f32(M1, a[2], a[3]));   // the other quad's unit
        c += dpp_xor4(c2);
    } else {
        c += dpp_xor4(c);
    }
    c += dpp_mov<0x128>(c);            // row_ror:8 : lane ^ 8
    c += __shfl_xor(c, 16, kWave);
    c += __shfl_xor(c, 32, kWave);
    return c;
}

// v = value of unit (lane % HP)  ->  h[q] = value of unit_of(q, lane)
template <int HP>
__device__ __forceinline__ void wave_all_gather(float v, float (&h)[HP]) {
    h[0] = quad_bcast<0x00>(v);
    h[1] = quad_bcast<0x55>(v);
    h[2] = quad_bcast<0xAA>(v);
    h[3] = quad_bcast<0xFF>(v);
    if constexpr (HP == 8) {
        const float o = dpp_xor4(v);
        h[4] = quad_bcast<0x00>(o);
        h[5] = quad_bcast<0x55>(o);
        h[6] = quad_bcast<0xAA>(o);
        h[7] = quad_bcast<0xFF>(o);
    }
}

__device__ __forceinline__ float wave_sum(float v) { return group_allreduce<64>(v); }

// HP floats of an LDS row in quad-major order (16-byte reads; the odd quads take the two halves swapped)
template <int HP>
__device__ __forceinline__ void lds_row_q(const float* __restrict__ row, int lane, bool ok, float (&w)[HP]) {
    if (!ok) {
#pragma unroll
        for (int q = 0; q < HP; ++q) w[q] = 0.f;
        return;
    }
    const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
    if constexpr (HP == 8) {
        const int sw = (lane >> 2) & 1;
        const f32x4 lo = r4[sw], hi = r4[sw ^ 1];
        w[0] = lo[0]; w[1] = lo[1]; w[2] = lo[2]; w[3] = lo[3];
        w[4] = hi[0]; w[5] = hi[1]; w[6] = hi[2]; w[7] = hi[3];
    } else {
        const f32x4 lo = r4[0];
        w[0] = lo[0]; w[1] = lo[1]; w[2] = lo[2]; w[3] = lo[3];
    }
}

// ---- geometry of one coupling layer as one lane sees it
template <int HP, int CH>
struct LaneLayer {
    float w1[CH][HP];           // W1T row of the lane's source elements (zero beyond d_a)
    float wa[CH][HP], wb[CH][HP], ba[CH], bb[CH];   // W3 / b3 rows of the lane's target elements (alpha | beta)
    float whcol[HP], whrow[HP]; // WhT[unit_of(q)][u] and WhT[u][unit_of(q)]
    float b1u, bhu;
    float tmask[CH];            // 1 where the target element exists
    int jrow[CH], trow[CH];     // j * HP, t * HP of the lane's elements (-1: none)
};

struct FitOffsets {
    int w1t, b1, wht, bh, w3, b3;
};
__host__ __device__ inline FitOffsets fit_rows_offsets(int d_a, int d_b, int n_hl, int HP) {
    FitOffsets o;
    o.w1t = 0;
    o.b1 = d_a * HP;
    o.wht = o.b1 + HP;
    o.bh = o.wht + HP * HP;
    o.w3 = o.b1 + HP + (n_hl - 1) * (HP * HP + HP);
    o.b3 = o.w3 + 2 * d_b * HP;
    return o;
}

// Element e = lane + 64 i of a block.  A' = physical [0, d_b) (for odd d its last element is the middle coordinate M, a
// target in every layer); B' = physical [d_b, d) followed, for odd d, by a SHADOW of M at element d_a.
// REV = false (odd layers): sources A = A'[0, d_a) with j = e; targets B' with t = e + (d_b - d_a), the shadow has t = 0.
// REV = true (even layers): sources B = B'[0, d_a) with j = d_a - 1 - e; targets A' with t = d_b - 1 - e (M: t = 0).
template <int HP, int CH, bool REV>
__device__ __forceinline__ void load_lane_layer(LaneLayer<HP, CH>& L, const float* __restrict__ W, const FitOffsets& o,
                                                int d_a, int d_b, int n_hl, int lane) {
    const int u = lane % HP;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int e = lane + 64 * i;
        const bool sok = e < d_a;
        const int j = REV ? d_a - 1 - e : e;
        L.jrow[i] = sok ? j * HP : -1;
        lds_row_q<HP>(W + o.w1t + (sok ? j : 0) * HP, lane, sok, L.w1[i]);
        const bool tok = e < d_b;
        int t;
        if (REV) t = d_b - 1 - e;
        else t = (e == d_a) ? 0 : e + (d_b - d_a);   // e == d_a only exists for odd d (the shadow of M)
        L.trow[i] = tok ? t * HP : -1;
        L.tmask[i] = tok ? 1.f : 0.f;
        lds_row_q<HP>(W + o.w3 + (tok ? t : 0) * HP, lane, tok, L.wa[i]);
        lds_row_q<HP>(W + o.w3 + (tok ? d_b + t : 0) * HP, lane, tok, L.wb[i]);
        L.ba[i] = tok ? W[o.b3 + t] : 0.f;
        L.bb[i] = tok ? W[o.b3 + d_b + t] : 0.f;
    }
    L.b1u = W[o.b1 + u];
    if (n_hl > 1) {
        L.bhu = W[o.bh + u];
        lds_row_q<HP>(W + o.wht + u * HP, lane, true, L.whrow);
#pragma unroll
        for (int q = 0; q < HP; ++q) L.whcol[q] = W[o.wht + unit_of<HP>(q, lane) * HP + u];
    } else {
        L.bhu = 0.f;
#pragma unroll
        for (int q = 0; q < HP; ++q) L.whcol[q] = L.whrow[q] = 0.f;
    }
}

// conditioner hidden stack of one row from its source registers: own units' activations + the gathered vectors
template <int HP, int CH>
__device__ __forceinline__ void conditioner_q(const LaneLayer<HP, CH>& L, const float (&xs)[CH], int n_hl, float& h1u,
                                              float& hlu, float (&h1)[HP], float (&hl)[HP]) {
    float acc[HP];
#pragma unroll
    for (int q = 0; q < HP; ++q) acc[q] = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
        for (int q = 0; q < HP; ++q) acc[q] = fmaf(L.w1[i][q], xs[i], acc[q]);
    h1u = fast_tanh(wave_reduce_scatter<HP>(acc) + L.b1u);
    wave_all_gather<HP>(h1u, h1);
    if (n_hl > 1) {
        float p = L.bhu;
#pragma unroll
        for (int q = 0; q < HP; ++q) p = fmaf(L.whcol[q], h1[q], p);
        hlu = fast_tanh(p);
        wave_all_gather<HP>(hlu, hl);
    } else {
        hlu = h1u;
#pragma unroll
        for (int q = 0; q < HP; ++q) hl[q] = h1[q];
    }
}

template <int HP, int CH>
struct LayerAcc {
    float w1[CH][HP], wa[CH][HP], wb[CH][HP], ba[CH], bb[CH], wh[HP], b1, bh;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            ba[i] = bb[i] = 0.f;
#pragma unroll
            for (int q = 0; q < HP; ++q) w1[i][q] = wa[i][q] = wb[i][q] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < HP; ++q) wh[q] = 0.f;
        b1 = bh = 0.f;
    }
};

// One coupling layer, FORWARD direction of the sweep (ML: x -> z, the forward map; RKL: z -> x, the inverse map) applied
// to the target registers of one row; returns the lane's share of sum log alpha over its targets.
template <int HP, int CH, bool RKL>
__device__ __forceinline__ float layer_apply(const LaneLayer<HP, CH>& L, const float (&xs)[CH], float (&xt)[CH], int n_hl,
                                             float log1m, float m) {
    float h1u, hlu, h1[HP], hl[HP];
    conditioner_q<HP, CH>(L, xs, n_hl, h1u, hlu, h1, hl);
    float ld = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        float ua = L.ba[i], ub = L.bb[i];
#pragma unroll
        for (int q = 0; q < HP; ++q) {
            ua = fmaf(L.wa[i][q], hl[q], ua);
            ub = fmaf(L.wb[i][q], hl[q], ub);
        }
        const float alpha = fast_exp(fmaf(0.5f, ua, log1m)) + m;
        ld += L.tmask[i] * fast_ln(alpha);
        if constexpr (RKL) xt[i] = (xt[i] - 0.5f * ub) * __builtin_amdgcn_rcpf(alpha);
        else xt[i] = fmaf(alpha, xt[i], 0.5f * ub);
    }
    return ld;
}

// The same layer going BACKWARD through the sweep for one row.  The target registers hold the layer's OUTPUT in the sweep's
// direction (ML: z_b; RKL: v_b = (y_b - beta) / alpha) and gt its gradient; they leave holding the layer's input and its
// gradient, the source gradient gs receives the conditioner's share, the accumulators this row's outer products.
template <int HP, int CH, bool RKL>
__device__ __forceinline__ void layer_backward(const LaneLayer<HP, CH>& L, LayerAcc<HP, CH>& A, const float (&xs)[CH],
                                               float (&gs)[CH], float (&xt)[CH], float (&gt)[CH], float valid, int n_hl,
                                               float log1m, float m) {
    float h1u, hlu, h1[HP], hl[HP];
    conditioner_q<HP, CH>(L, xs, n_hl, h1u, hlu, h1, hl);
    float gh[HP];
#pragma unroll
    for (int q = 0; q < HP; ++q) gh[q] = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        float ua = L.ba[i], ub = L.bb[i];
#pragma unroll
        for (int q = 0; q < HP; ++q) {
            ua = fmaf(L.wa[i][q], hl[q], ua);
            ub = fmaf(L.wb[i][q], hl[q], ub);
        }
        const float alpha = fast_exp(fmaf(0.5f, ua, log1m)) + m;
        const float ra = __builtin_amdgcn_rcpf(alpha);
        const float vm = valid * L.tmask[i];
        float da, db;
        if constexpr (!RKL) {
            // z_b = alpha x_b + beta, loss contains -log alpha
            const float xb = (xt[i] - 0.5f * ub) * ra;
            const float gz = gt[i];
            xt[i] = xb;
            gt[i] = gz * alpha;
            const float ga = fmaf(gz, xb, -vm * ra);
            da = 0.5f * ga * (alpha - m);
            db = 0.5f * gz;
        } else {
            // v_b = (y_b - beta) / alpha, loss contains +log alpha; registers hold v, dL/dv
            const float vb = xt[i], gv = gt[i];
            xt[i] = fmaf(alpha, vb, 0.5f * ub);
            const float gy = gv * ra;
            gt[i] = gy;
            const float ga = (vm - gv * vb) * ra;
            da = 0.5f * ga * (alpha - m);
            db = -0.5f * gy;
        }
        A.ba[i] += da;
        A.bb[i] += db;
#pragma unroll
        for (int q = 0; q < HP; ++q) {
            A.wa[i][q] = fmaf(da, hl[q], A.wa[i][q]);
            A.wb[i][q] = fmaf(db, hl[q], A.wb[i][q]);
            gh[q] = fmaf(L.wa[i][q], da, fmaf(L.wb[i][q], db, gh[q]));
        }
    }
    const float dlu = wave_reduce_scatter<HP>(gh) * (1.f - hlu * hlu);
    float dfu = dlu;
    if (n_hl > 1) {
        float dl[HP];
        wave_all_gather<HP>(dlu, dl);
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < HP; ++q) {
            a = fmaf(L.whrow[q], dl[q], a);
            A.wh[q] = fmaf(h1u, dl[q], A.wh[q]);
        }
        dfu = a * (1.f - h1u * h1u);
        A.bh += dlu;
    }
    A.b1 += dfu;
    float df[HP];
    wave_all_gather<HP>(dfu, df);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < HP; ++q) {
            a = fmaf(L.w1[i][q], df[q], a);
            A.w1[i][q] = fmaf(xs[i], df[q], A.w1[i][q]);
        }
        gs[i] += a;
    }
}

template <int HP, bool RKL, int CH, int S>
__global__ void __launch_bounds__(kFrThreads) fit_rows_kernel(FitRowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (a.run_state && (a.run_state[3] != 0.f || a.run_state[4] != 0.f)) return;   // the run has ended: nothing to do
    constexpr int NACC = fit_rows_nacc<HP, CH>(), NSTAGE = fit_rows_nstage(HP, CH);
    static_assert(NSTAGE >= 4 * CH, "the elementwise-affine flush stages 4 CH registers");
    const NfmcRealNVP& f = a.f;
    const int d = f.d, d_a = d / 2, d_b = d - d_a, n_hl = f.n_hidden_layers, n_coupling = f.n_coupling;
    const bool odd = d_b != d_a;
    const float m = f.min_scale, log1m = __logf(1.f - f.min_scale);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int np4 = (int)((a.n_params + 3) / 4 * 4);
    float* const pl = lds;                                   // the trainable vector
    float* const stage = lds + np4;                          // [wave][NSTAGE][64]
    float* const tails = stage + kFrWaves * NSTAGE * 64;     // [wave][4]
    const FitOffsets o = fit_rows_offsets(d_a, d_b, n_hl, HP);
    const bool rev_last = (n_coupling & 1) != 0;
    // ---- the trainable vector into LDS: every thread issues all its 16-byte loads before its first LDS store
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.params);
        f32x4* dst = reinterpret_cast<f32x4*>(pl);
        const int n4 = np4 / 4, full4 = (int)(a.n_params / 4);
        for (int base = 0; base < n4; base += kFrThreads * 8) {
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = base + k * kFrThreads + tid;
                v[k] = i < full4 ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = base + k * kFrThreads + tid;
                if (i < n4) dst[i] = v[k];
            }
        }
        for (int i = full4 * 4 + tid; i < (int)a.n_params; i += kFrThreads) pl[i] = a.params[i];
    }
    __syncthreads();
    const float* const ea = pl + a.ea_off;
    const int d4 = a.d4;
    // ---- per-lane coordinate tables.  Register i of block A' is physical pA = e, of block B' physical pB = d_b + e (the
    // shadow of M, element d_a of B', mirrors physical d_a and takes no part in elementwise phases).
    bool okA[CH], okB[CH], shadow[CH];
    int pA[CH], pB[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int e = lane + 64 * i;
        okA[i] = e < d_b;
        okB[i] = e < d_a;
        shadow[i] = odd && e == d_a;
        pA[i] = e;
        pB[i] = shadow[i] ? d_a : d_b + e;
    }
    // sum of the elementwise log-scales (the rows' common part of the log-determinant)
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (okA[i]) lsum += ea[pA[i]] + ea[2 * d4 + (rev_last ? d - 1 - pA[i] : pA[i])];
        if (okB[i]) lsum += ea[pB[i]] + ea[2 * d4 + (rev_last ? d - 1 - pB[i] : pB[i])];
    }
    lsum = wave_sum(lsum);
    const int64_t W = (int64_t)gridDim.x * kFrWaves;
    const int64_t total = a.tiles4 + a.vtiles;
    const int64_t passes = (total + W - 1) / W;
    float* const P = a.partial + (int64_t)blockIdx.x * a.pstride;
    bool first = true;
    float loss_acc = 0.f, rows_acc = 0.f, vloss_acc = 0.f, vrows_acc = 0.f;
    float* const mystage = stage + wave * NSTAGE * 64;

    for (int64_t pass = 0; pass < passes; ++pass) {
        const int64_t tile0 = pass * W + (int64_t)blockIdx.x * kFrWaves;   // the workgroup's first tile of this pass
        const bool wg_train = tile0 < a.tiles4;                          // uniform over the workgroup (tiles4 % 4 == 0)
        const int64_t tile = tile0 + wave;
        const bool is_val = !wg_train && tile < total;
        const float* src = wg_train ? a.x : a.xv;
        const int64_t nrows = wg_train ? a.n : (is_val ? a.nv : 0);
        const int64_t r0 = (wg_train ? tile : tile - a.tiles4) * S;
        float xa[S][CH], xb[S][CH], valid[S];
        // ---- load: ML rows are x in physical order; RKL rows are latents, array column c at physical (rev_last ? d-1-c : c)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int64_t r = r0 + s;
            const bool rv = r < nrows;
            valid[s] = rv ? 1.f : 0.f;
            const float* row = src + (rv ? r : 0) * d;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int ca = (RKL && rev_last) ? d - 1 - pA[i] : pA[i];
                const int cb = (RKL && rev_last) ? d - 1 - pB[i] : pB[i];
                xa[s][i] = (rv && okA[i]) ? row[ca] : 0.f;
                xb[s][i] = (rv && (okB[i] || shadow[i])) ? row[cb] : 0.f;
            }
        }
        float ss[S], ld[S];
        // ---- forward sweep
        if constexpr (!RKL) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                ld[s] = 0.f;
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if (okA[i]) xa[s][i] = fmaf(fast_exp(ea[pA[i]]), xa[s][i], ea[d4 + pA[i]]);
                    if (okB[i] || shadow[i]) xb[s][i] = fmaf(fast_exp(ea[pB[i]]), xb[s][i], ea[d4 + pB[i]]);
                }
            }
            for (int l = 0; l < n_coupling; ++l) {
                const float* Wl = pl + (int64_t)l * f.layer_stride;
                LaneLayer<HP, CH> L;
                if ((l & 1) == 0) {
                    load_lane_layer<HP, CH, true>(L, Wl, o, d_a, d_b, n_hl, lane);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        ld[s] += layer_apply<HP, CH, false>(L, xb[s], xa[s], n_hl, log1m, m);
#pragma unroll
                        for (int i = 0; i < CH; ++i) xb[s][i] = shadow[i] ? xa[s][i] : xb[s][i];
                    }
                } else {
                    load_lane_layer<HP, CH, false>(L, Wl, o, d_a, d_b, n_hl, lane);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        ld[s] += layer_apply<HP, CH, false>(L, xa[s], xb[s], n_hl, log1m, m);
#pragma unroll
                        for (int i = 0; i < CH; ++i) xa[s][i] = shadow[i] ? xb[s][i] : xa[s][i];
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if (okA[i]) {
                        const int c = rev_last ? d - 1 - pA[i] : pA[i];
                        xa[s][i] = fmaf(fast_exp(ea[2 * d4 + c]), xa[s][i], ea[3 * d4 + c]);
                        q = fmaf(xa[s][i], xa[s][i], q);
                    }
                    if (okB[i]) {
                        const int c = rev_last ? d - 1 - pB[i] : pB[i];
                        xb[s][i] = fmaf(fast_exp(ea[2 * d4 + c]), xb[s][i], ea[3 * d4 + c]);
                        q = fmaf(xb[s][i], xb[s][i], q);
                    }
                }
                ss[s] = wave_sum(q);
                ld[s] = wave_sum(ld[s]) + lsum;
                const float li = 0.5f * ss[s] + 0.5f * (float)d * kLog2Pi - ld[s];
                if (wg_train) {
                    loss_acc += valid[s] > 0.f ? li : 0.f;
                    rows_acc += valid[s];
                } else {
                    vloss_acc += valid[s] > 0.f ? li : 0.f;
                    vrows_acc += valid[s];
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float q = 0.f;
                ld[s] = 0.f;
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if (okA[i]) {
                        const int c = rev_last ? d - 1 - pA[i] : pA[i];
                        q = fmaf(xa[s][i], xa[s][i], q);
                        xa[s][i] = (xa[s][i] - ea[3 * d4 + c]) * fast_exp(-ea[2 * d4 + c]);
                    }
                    if (okB[i] || shadow[i]) {
                        const int c = rev_last ? d - 1 - pB[i] : pB[i];
                        if (okB[i]) q = fmaf(xb[s][i], xb[s][i], q);
                        xb[s][i] = (xb[s][i] - ea[3 * d4 + c]) * fast_exp(-ea[2 * d4 + c]);
                    }
                }
                ss[s] = wave_sum(q);
            }
            for (int l = n_coupling - 1; l >= 0; --l) {
                const float* Wl = pl + (int64_t)l * f.layer_stride;
                LaneLayer<HP, CH> L;
                if ((l & 1) == 0) {
                    load_lane_layer<HP, CH, true>(L, Wl, o, d_a, d_b, n_hl, lane);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        ld[s] += layer_apply<HP, CH, true>(L, xb[s], xa[s], n_hl, log1m, m);
#pragma unroll
                        for (int i = 0; i < CH; ++i) xb[s][i] = shadow[i] ? xa[s][i] : xb[s][i];
                    }
                } else {
                    load_lane_layer<HP, CH, false>(L, Wl, o, d_a, d_b, n_hl, lane);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        ld[s] += layer_apply<HP, CH, true>(L, xa[s], xb[s], n_hl, log1m, m);
#pragma unroll
                        for (int i = 0; i < CH; ++i) xa[s][i] = shadow[i] ? xb[s][i] : xa[s][i];
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if (okA[i]) xa[s][i] = (xa[s][i] - ea[d4 + pA[i]]) * fast_exp(-ea[pA[i]]);
                    if (okB[i] || shadow[i]) xb[s][i] = (xb[s][i] - ea[d4 + pB[i]]) * fast_exp(-ea[pB[i]]);
                }
            }
        }
        if (!wg_train) continue;   // validation rows: loss only (uniform over the workgroup)

        // ---- gradient of the loss with respect to the state at the end of the forward sweep
        float ga[S][CH], gb[S][CH];
        if constexpr (!RKL) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    ga[s][i] = okA[i] ? valid[s] * xa[s][i] : 0.f;     // dL/dz of 0.5 |z|^2
                    gb[s][i] = okB[i] ? valid[s] * xb[s][i] : 0.f;
                }
        } else {
            // loss_i = log N(z) - logdet_inverse + U(x); dL/dx = grad U(x)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float u;
                if (a.pot.kind == NFMC_POT_FUNNEL) {
                    const float x0 = __shfl(xa[s][0], 0, kWave);
                    float q = 0.f;
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        if (okA[i] && pA[i] > 0) q = fmaf(xa[s][i], xa[s][i], q);
                        if (okB[i]) q = fmaf(xb[s][i], xb[s][i], q);
                    }
                    q = wave_sum(q);
                    const float inv_s2 = 1.f / (a.pot.a_scalar * a.pot.a_scalar);
                    const float e = fast_exp(-x0), hd = 0.5f * (float)(d - 1);
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        ga[s][i] = okA[i] ? (pA[i] == 0 ? x0 * inv_s2 - 0.5f * e * q + hd : xa[s][i] * e) : 0.f;
                        gb[s][i] = okB[i] ? xb[s][i] * e : 0.f;
                    }
                    u = 0.5f * x0 * x0 * inv_s2 + 0.5f * e * q + hd * x0;
                } else {
                    float q = 0.f;
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        ga[s][i] = gb[s][i] = 0.f;
                        if (okA[i]) {
                            const float aa = a.pot.a ? a.pot.a[pA[i]] : a.pot.a_scalar;
                            const float t = xa[s][i] - (a.pot.b ? a.pot.b[pA[i]] : a.pot.b_scalar);
                            q = fmaf(aa * t, t, q);
                            ga[s][i] = 2.f * aa * t;
                        }
                        if (okB[i]) {
                            const float aa = a.pot.a ? a.pot.a[pB[i]] : a.pot.a_scalar;
                            const float t = xb[s][i] - (a.pot.b ? a.pot.b[pB[i]] : a.pot.b_scalar);
                            q = fmaf(aa * t, t, q);
                            gb[s][i] = 2.f * aa * t;
                        }
                    }
                    u = wave_sum(q);
                }
                const float ldi = -(wave_sum(ld[s]) + lsum);   // logdet_inverse
                const float li = -0.5f * ss[s] - 0.5f * (float)d * kLog2Pi - ldi + u;
                loss_acc += valid[s] > 0.f ? li : 0.f;
                rows_acc += valid[s];
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    ga[s][i] *= valid[s];
                    gb[s][i] *= valid[s];
                }
            }
        }
        // ---- the sweep's first elementwise layer going backward: ML the LAST ElementwiseAffine (logical coordinates),
        // RKL the FIRST one inverted.  Accumulators: (d log-scale, d shift) per coordinate register.
        auto flush_ea = [&](const float (&as)[2 * CH], const float (&at)[2 * CH], int which, bool logical) {
            __syncthreads();   // the stage is free (previous readers are done)
#pragma unroll
            for (int k = 0; k < 2 * CH; ++k) {
                mystage[k * 64 + lane] = as[k];
                mystage[(2 * CH + k) * 64 + lane] = at[k];
            }
            __syncthreads();
            // wave w adds register w (mod 4) of both vectors over the four waves, in wave order
#pragma unroll
            for (int k = 0; k < 4 * CH; ++k) {
                if ((k & 3) != wave) continue;
                const float v = stage[(0 * NSTAGE + k) * 64 + lane] + stage[(1 * NSTAGE + k) * 64 + lane] +
                                stage[(2 * NSTAGE + k) * 64 + lane] + stage[(3 * NSTAGE + k) * 64 + lane];
                const int kk = k % (2 * CH), i = kk % CH;
                const bool blockB = kk >= CH;
                const bool ok = blockB ? okB[i] : okA[i];
                const int p = blockB ? pB[i] : pA[i];
                const int c = logical ? (rev_last ? d - 1 - p : p) : p;
                if (ok) {
                    const int64_t idx = a.ea_off + (int64_t)(which + (k >= 2 * CH ? 1 : 0)) * d4 + c;
                    P[idx] = first ? v : P[idx] + v;
                }
            }
        };
        {
            float as[2 * CH], at[2 * CH];
#pragma unroll
            for (int k = 0; k < 2 * CH; ++k) as[k] = at[k] = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if constexpr (!RKL) {
                        // z = e^s y + t
                        if (okA[i]) {
                            const int c = rev_last ? d - 1 - pA[i] : pA[i];
                            const float sc = ea[2 * d4 + c], zc = xa[s][i] - ea[3 * d4 + c], gz = ga[s][i];
                            as[i] += fmaf(gz, zc, -valid[s]);
                            at[i] += gz;
                            ga[s][i] = gz * fast_exp(sc);
                            xa[s][i] = zc * fast_exp(-sc);
                        }
                        if (okB[i]) {
                            const int c = rev_last ? d - 1 - pB[i] : pB[i];
                            const float sc = ea[2 * d4 + c], zc = xb[s][i] - ea[3 * d4 + c], gz = gb[s][i];
                            as[CH + i] += fmaf(gz, zc, -valid[s]);
                            at[CH + i] += gz;
                            gb[s][i] = gz * fast_exp(sc);
                            xb[s][i] = zc * fast_exp(-sc);
                        }
                    } else {
                        // x = (y - t) e^-s; -logdet_inverse contains +s
                        if (okA[i]) {
                            const float sc = ea[pA[i]], gx = ga[s][i], xv = xa[s][i];
                            const float gy = gx * fast_exp(-sc);
                            as[i] += fmaf(-gx, xv, valid[s]);
                            at[i] -= gy;
                            ga[s][i] = gy;
                            xa[s][i] = fmaf(fast_exp(sc), xv, ea[d4 + pA[i]]);
                        }
                        if (okB[i]) {
                            const float sc = ea[pB[i]], gx = gb[s][i], xv = xb[s][i];
                            const float gy = gx * fast_exp(-sc);
                            as[CH + i] += fmaf(-gx, xv, valid[s]);
                            at[CH + i] -= gy;
                            gb[s][i] = gy;
                            xb[s][i] = fmaf(fast_exp(sc), xv, ea[d4 + pB[i]]);
                        }
                    }
                }
            // shadow of M follows its primary (the primary is in A')
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    if (shadow[i]) {
                        xb[s][i] = xa[s][i];
                        gb[s][i] = ga[s][i];
                    }
            flush_ea(as, at, RKL ? 0 : 2, !RKL);
        }
        // ---- coupling layers going backward
        for (int li = 0; li < n_coupling; ++li) {
            const int l = RKL ? li : n_coupling - 1 - li;
            const float* Wl = pl + (int64_t)l * f.layer_stride;
            const int64_t L0 = (int64_t)l * f.layer_stride;
            LaneLayer<HP, CH> L;
            LayerAcc<HP, CH> A;
            A.zero();
            const bool rev = (l & 1) == 0;
            if (rev) {
                load_lane_layer<HP, CH, true>(L, Wl, o, d_a, d_b, n_hl, lane);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    layer_backward<HP, CH, RKL>(L, A, xb[s], gb[s], xa[s], ga[s], valid[s], n_hl, log1m, m);
#pragma unroll
                    for (int i = 0; i < CH; ++i)
                        if (shadow[i]) {
                            xb[s][i] = xa[s][i];
                            gb[s][i] = ga[s][i];
                        }
                }
            } else {
                load_lane_layer<HP, CH, false>(L, Wl, o, d_a, d_b, n_hl, lane);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    // the shadow's source-side gradient does not exist (it is a target here); its target gradient is M's
                    layer_backward<HP, CH, RKL>(L, A, xa[s], ga[s], xb[s], gb[s], valid[s], n_hl, log1m, m);
#pragma unroll
                    for (int i = 0; i < CH; ++i)
                        if (shadow[i]) {
                            xa[s][i] = xb[s][i];
                            ga[s][i] = gb[s][i];
                        }
                }
            }
            if constexpr (CH <= 2) {
                // ---- the four waves' accumulators -> LDS -> added in wave order into the workgroup's slab
                __syncthreads();
                {
                    int r = 0;
#pragma unroll
                    for (int i = 0; i < CH; ++i)
#pragma unroll
                        for (int q = 0; q < HP; ++q) mystage[(r++) * 64 + lane] = A.w1[i][q];
#pragma unroll
                    for (int i = 0; i < CH; ++i)
#pragma unroll
                        for (int q = 0; q < HP; ++q) mystage[(r++) * 64 + lane] = A.wa[i][q];
#pragma unroll
                    for (int i = 0; i < CH; ++i)
#pragma unroll
                        for (int q = 0; q < HP; ++q) mystage[(r++) * 64 + lane] = A.wb[i][q];
#pragma unroll
                    for (int i = 0; i < CH; ++i) mystage[(r++) * 64 + lane] = A.ba[i];
#pragma unroll
                    for (int i = 0; i < CH; ++i) mystage[(r++) * 64 + lane] = A.bb[i];
#pragma unroll
                    for (int q = 0; q < HP; ++q) mystage[(r++) * 64 + lane] = A.wh[q];
                    mystage[(r++) * 64 + lane] = A.b1;
                    mystage[(r++) * 64 + lane] = A.bh;
                }
                __syncthreads();
                auto folded = [&](int r) {
                    return stage[(0 * NSTAGE + r) * 64 + lane] + stage[(1 * NSTAGE + r) * 64 + lane] +
                           stage[(2 * NSTAGE + r) * 64 + lane] + stage[(3 * NSTAGE + r) * 64 + lane];
                };
                auto emit = [&](int64_t idx, float v) { P[idx] = first ? v : P[idx] + v; };
                const int u = lane % HP;
                if (wave == 0) {   // W1T rows of the lane's source elements
#pragma unroll
                    for (int i = 0; i < CH; ++i)
                        if (L.jrow[i] >= 0) {
#pragma unroll
                            for (int q = 0; q < HP; ++q)
                                emit(L0 + o.w1t + L.jrow[i] + unit_of<HP>(q, lane), folded(i * HP + q));
                        }
                } else if (wave == 1) {   // W3 alpha rows
#pragma unroll
                    for (int i = 0; i < CH; ++i)
                        if (L.trow[i] >= 0) {
#pragma unroll
                            for (int q = 0; q < HP; ++q)
                                emit(L0 + o.w3 + L.trow[i] + unit_of<HP>(q, lane), folded(CH * HP + i * HP + q));
                        }
                } else if (wave == 2) {   // W3 beta rows
#pragma unroll
                    for (int i = 0; i < CH; ++i)
                        if (L.trow[i] >= 0) {
#pragma unroll
                            for (int q = 0; q < HP; ++q)
                                emit(L0 + o.w3 + (int64_t)d_b * HP + L.trow[i] + unit_of<HP>(q, lane),
                                     folded(2 * CH * HP + i * HP + q));
                        }
                } else {   // b3, WhT rows, b1, bh
#pragma unroll
                    for (int i = 0; i < CH; ++i)
                        if (L.trow[i] >= 0) {
                            emit(L0 + o.b3 + L.trow[i] / HP, folded(3 * CH * HP + i));
                            emit(L0 + o.b3 + d_b + L.trow[i] / HP, folded(3 * CH * HP + CH + i));
                        }
                    if (lane < HP) {
                        if (n_hl > 1) {
#pragma unroll
                            for (int q = 0; q < HP; ++q)
                                emit(L0 + o.wht + u * HP + unit_of<HP>(q, lane), folded(3 * CH * HP + 2 * CH + q));
                            emit(L0 + o.bh + u, folded(3 * CH * HP + 2 * CH + HP + 1));
                        }
                        emit(L0 + o.b1 + u, folded(3 * CH * HP + 2 * CH + HP));
                    }
                }
        
            } else {
                // ---- the four waves' accumulators -> LDS -> added in wave order into the workgroup's slab.  NSTAGE registers per
                // wave and round (all of them in one round up to d = 256; two rounds at CH = 4, where staging all 114 next to the
                // staged vector would exceed the LDS); register r of a round is folded and written by wave r mod 4.
                float acc[NACC];
                {
                    int r = 0;
#pragma unroll
                    for (int i = 0; i < CH; ++i)
#pragma unroll
                        for (int q = 0; q < HP; ++q) acc[r++] = A.w1[i][q];
#pragma unroll
                    for (int i = 0; i < CH; ++i)
#pragma unroll
                        for (int q = 0; q < HP; ++q) acc[r++] = A.wa[i][q];
#pragma unroll
                    for (int i = 0; i < CH; ++i)
#pragma unroll
                        for (int q = 0; q < HP; ++q) acc[r++] = A.wb[i][q];
#pragma unroll
                    for (int i = 0; i < CH; ++i) acc[r++] = A.ba[i];
#pragma unroll
                    for (int i = 0; i < CH; ++i) acc[r++] = A.bb[i];
#pragma unroll
                    for (int q = 0; q < HP; ++q) acc[r++] = A.wh[q];
                    acc[r++] = A.b1;
                    acc[r++] = A.bh;
                }
                const int u = lane % HP;
#pragma unroll
                for (int r0 = 0; r0 < NACC; r0 += NSTAGE) {
                    __syncthreads();   // the stage is free (the previous round's readers are done)
#pragma unroll
                    for (int r = r0; r < r0 + NSTAGE && r < NACC; ++r) mystage[(r - r0) * 64 + lane] = acc[r];
                    __syncthreads();
#pragma unroll
                    for (int r = r0; r < r0 + NSTAGE && r < NACC; ++r) {
                        if (((r - r0) & 3) != wave) continue;   // wave-uniform
                        const int sl = r - r0;
                        const float v = stage[(0 * NSTAGE + sl) * 64 + lane] + stage[(1 * NSTAGE + sl) * 64 + lane] +
                                        stage[(2 * NSTAGE + sl) * 64 + lane] + stage[(3 * NSTAGE + sl) * 64 + lane];
                        int64_t idx = -1;
                        if (r < CH * HP) {                              // W1T rows of the lane's source elements
                            const int i = r / HP, q = r % HP;
                            if (L.jrow[i] >= 0) idx = L0 + o.w1t + L.jrow[i] + unit_of<HP>(q, lane);
                        } else if (r < 2 * CH * HP) {                   // W3 alpha rows of its target elements
                            const int i = (r - CH * HP) / HP, q = r % HP;
                            if (L.trow[i] >= 0) idx = L0 + o.w3 + L.trow[i] + unit_of<HP>(q, lane);
                        } else if (r < 3 * CH * HP) {                   // W3 beta rows
                            const int i = (r - 2 * CH * HP) / HP, q = r % HP;
                            if (L.trow[i] >= 0) idx = L0 + o.w3 + (int64_t)d_b * HP + L.trow[i] + unit_of<HP>(q, lane);
                        } else if (r < 3 * CH * HP + CH) {              // b3, alpha half
                            const int i = r - 3 * CH * HP;
                            if (L.trow[i] >= 0) idx = L0 + o.b3 + L.trow[i] / HP;
                        } else if (r < 3 * CH * HP + 2 * CH) {          // b3, beta half
                            const int i = r - 3 * CH * HP - CH;
                            if (L.trow[i] >= 0) idx = L0 + o.b3 + d_b + L.trow[i] / HP;
                        } else if (r < 3 * CH * HP + 2 * CH + HP) {     // WhT row of the lane's hidden unit
                            const int q = r - 3 * CH * HP - 2 * CH;
                            if (lane < HP && n_hl > 1) idx = L0 + o.wht + u * HP + unit_of<HP>(q, lane);
                        } else if (r == 3 * CH * HP + 2 * CH + HP) {    // b1
                            if (lane < HP) idx = L0 + o.b1 + u;
                        } else {                                        // bh
                            if (lane < HP && n_hl > 1) idx = L0 + o.bh + u;
                        }
                        if (idx >= 0) P[idx] = first ? v : P[idx] + v;
                    }
                }
        
            }
        }
        // ---- the sweep's last elementwise layer going backward: ML the FIRST ElementwiseAffine, RKL the LAST one inverted
        {
            float as[2 * CH], at[2 * CH];
#pragma unroll
            for (int k = 0; k < 2 * CH; ++k) as[k] = at[k] = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if constexpr (!RKL) {
                        // registers hold y = e^s x + t and dL/dy
                        if (okA[i]) {
                            as[i] += fmaf(ga[s][i], xa[s][i] - ea[d4 + pA[i]], -valid[s]);
                            at[i] += ga[s][i];
                        }
                        if (okB[i]) {
                            as[CH + i] += fmaf(gb[s][i], xb[s][i] - ea[d4 + pB[i]], -valid[s]);
                            at[CH + i] += gb[s][i];
                        }
                    } else {
                        // registers hold v = (z - t) e^-s and dL/dv (logical coordinates of the last layer)
                        if (okA[i]) {
                            const int c = rev_last ? d - 1 - pA[i] : pA[i];
                            as[i] += fmaf(-ga[s][i], xa[s][i], valid[s]);
                            at[i] = fmaf(-ga[s][i], fast_exp(-ea[2 * d4 + c]), at[i]);
                        }
                        if (okB[i]) {
                            const int c = rev_last ? d - 1 - pB[i] : pB[i];
                            as[CH + i] += fmaf(-gb[s][i], xb[s][i], valid[s]);
                            at[CH + i] = fmaf(-gb[s][i], fast_exp(-ea[2 * d4 + c]), at[CH + i]);
                        }
                    }
                }
            flush_ea(as, at, RKL ? 2 : 0, RKL);
        }
        first = false;
    }
    // ---- losses and row counts: the four waves' sums in wave order
    __syncthreads();
    if (lane == 0) {
        tails[wave * 4 + 0] = loss_acc;
        tails[wave * 4 + 1] = rows_acc;
        tails[wave * 4 + 2] = vloss_acc;
        tails[wave * 4 + 3] = vrows_acc;
    }
    __syncthreads();
    if (tid < 4) P[a.n_params + tid] = tails[tid] + tails[4 + tid] + tails[8 + tid] + tails[12 + tid];
}

}  // namespace nfmc

// One translation unit per conditioner width (build time): fit_rows_h4.hip / fit_rows_h8.hip expand this.
#define NFMC_FIT_ROWS_UNIT(HPV, NAME)                                                                                      \
    namespace nfmc {                                                                                                       \
    template <bool RKL, int CH, int S>                                                                                     \
    static int fit_rows_go(const FitRowsArgs& a, int grid, size_t lds, hipStream_t st) {                                   \
        auto kern = fit_rows_kernel<HPV, RKL, CH, S>;                                                                      \
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kFrThreads), lds, st, a);                                                \
        return 0;                                                                                                          \
    }                                                                                                                      \
    int NAME(bool rkl, int ch, int s, const FitRowsArgs& a, int grid, size_t lds, hipStream_t st) {                        \
        if (rkl) {                                                                                                         \
            if (ch == 1) return s == 1 ? fit_rows_go<true, 1, 1>(a, grid, lds, st) : fit_rows_go<true, 1, 4>(a, grid, lds, st); \
            if (ch == 2) return s == 1 ? fit_rows_go<true, 2, 1>(a, grid, lds, st) : fit_rows_go<true, 2, 4>(a, grid, lds, st); \
            return s == 1 ? fit_rows_go<true, 4, 1>(a, grid, lds, st) : fit_rows_go<true, 4, 2>(a, grid, lds, st);         \
        }                                                                                                                  \
        if (ch == 1) return s == 1 ? fit_rows_go<false, 1, 1>(a, grid, lds, st) : fit_rows_go<false, 1, 4>(a, grid, lds, st); \
        if (ch == 2) return s == 1 ? fit_rows_go<false, 2, 1>(a, grid, lds, st) : fit_rows_go<false, 2, 4>(a, grid, lds, st); \
        return s == 1 ? fit_rows_go<false, 4, 1>(a, grid, lds, st) : fit_rows_go<false, 4, 2>(a, grid, lds, st);           \
    }                                                                                                                      \
    }
