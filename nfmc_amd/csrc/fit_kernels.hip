// f1: maximum-likelihood (re)fit and variational (reverse-KL) fit of the RealNVP proposal on the device -- one optimiser
// step per call:
//   fit_grad_kernel   mean loss of the batch and its gradient with respect to EVERY parameter
//   adamw_fold_kernel fixed-order fold of the per-workgroup partial gradients + the AdamW update, in one launch
// Replaces the torchflows `Flow.fit` epochs the reference runs at jump.py:139-151 (warmup), jump.py:193-201 (refit every
// outer iteration when fit_nf), imh.py:166-170 (AdaptiveIMH's one-epoch refits) and the `Flow.variational_fit` epochs of
// imh.py:67-72 / neutra.py:84-91 (warmup of the independence and NeuTra samplers), which the round-2 build evaluated with
// eager torch autograd + torch.optim.AdamW (2.75 ms per epoch at the C5 shape: ~50 small launches forward and backward).
//
// Layout: one wave per workgroup, one batch row per lane, the wave's (64, d) tile of rows in LDS twice (state and its
// gradient) -- the one-chain-per-lane scheme of flow_device.hpp, weights wave-uniform through the scalar cache.
// No activation is stored: going backward, a coupling layer's input is rebuilt from its output (x_b = (z_b - beta) / alpha)
// and its conditioner re-evaluated from the unchanged source half, as in NeuTra's reverse sweep.
// Weight gradients are sums over rows of outer products (delta x activation).  With rows on lanes that would be one
// cross-lane reduction per weight; instead each layer has a ROW phase (lanes = rows: conditioner, affine map, input
// gradients; the per-row hidden activations and deltas go to a small LDS tile) and TRANSPOSED phases (lanes = output
// coordinates: every lane walks the 64 rows of the tile, reads the per-row values column-wise / by broadcast, re-derives the
// row's scale from the stored activations, and accumulates its own weights' gradients in registers): no reductions, no
// atomics, every partial sum in a fixed order.
// The trainable vector has the layout of the flow's weight blob (flows.py: packed): coupling layers, then the four
// ElementwiseAffine vectors; gradients and AdamW moments use the same indices.
// Round 4: conditioners of width <= 8 (every default flow) take the row-per-WAVE kernel of fit_rows.hpp (lanes = coordinates,
// up to 1024 waves per launch); the row-per-lane kernel below stays for widths 16 / 32.  The epoch loop's bookkeeping (best
// validation loss, best weights, early stopping, divergence) moved into the fold kernel, so a run of epochs is enqueued
// without a host round trip per epoch (nfmc_flow_fit_epochs_f32).
#include "fit_rows.hpp"
#include "fit_mfma.hpp"

namespace nfmc {

int fit_rows_launch_h4(bool rkl, int ch, int s, const FitRowsArgs& a, int grid, size_t lds, hipStream_t st);
int fit_rows_launch_h8(bool rkl, int ch, int s, const FitRowsArgs& a, int grid, size_t lds, hipStream_t st);

constexpr int kFitBlock = 64;
constexpr int kMfmaChainsFit = 128;   // rows per workgroup tile of the matrix-core kernel (mfma_device.hpp: kMfmaChains)
constexpr int kFitTail = kFitTailFloats;   // per-workgroup partial: [n_params] gradient sums, then loss sum, rows, validation loss sum, validation rows

__host__ __device__ inline int fit_hb_stride(int hp) { return 4 * hp + 1; }   // odd: lanes = rows write conflict-free
__host__ __device__ inline size_t fit_lds_bytes(int d, int hp, int rpw = 64) {
    return ((size_t)2 * rpw * tile_stride(d) + (size_t)rpw * fit_hb_stride(hp)) * sizeof(float);
}
// Rows per wave: 64 (every lane owns a row in the row phases).  Measured alternative at the C5 refit shape (4096 rows,
// d = 256, conditioner 6 -> 8): 16 rows per wave = 256 waves instead of 64, transposed phases four times shorter -- 325 us per
// launch against 266 us, and a fold over 256 slabs instead of 64 (72 against 31 us): the row phases, whose length does not
// depend on the number of rows, dominate (dependent chains behind scalar weight loads, one wave per CU), so fewer, fuller
// waves win.  The 16-row instantiation stays for batches of at most 16 rows (variational fits draw 1 .. 16 latents).
__host__ __device__ inline int fit_rows_per_wave(int64_t n) { return n <= 16 ? 16 : 64; }

// offsets of a coupling layer's pieces inside its blob (VALU layout, flow_device.hpp)
struct FitOff {
    int w1t, b1, wht, bh, w3, b3;
};
__device__ __forceinline__ FitOff fit_offsets(const FlowGeom& g, int HP) {
    FitOff o;
    o.w1t = 0;
    o.b1 = g.d_a * HP;
    o.wht = o.b1 + HP;
    o.bh = o.wht + HP * HP;
    o.w3 = o.b1 + HP + (g.n_hl - 1) * (HP * HP + HP);
    o.b3 = o.w3 + 2 * g.d_b * HP;
    return o;
}

// RKL = false: maximum likelihood.  Rows are data x; forward sweep x -> z, loss_i = -log N(z_i) - logdet_forward; the
//   backward sweep walks the layers last to first, rebuilding each layer's INPUT from its output.
// RKL = true: reverse KL (variational fit, imh.py:67-72 / neutra.py:84-91).  Rows are latents z ~ N(0, I); inverse sweep
//   z -> x = f^-1(z), loss_i = log q(x_i) - log p(x_i) = log N(z_i) - logdet_inverse + U(x_i) with the closed-form potential
//   U = -log p and its gradient; the backward sweep walks the layers first to last, rebuilding each inverse layer's INPUT
//   (the forward map's output) from its output.  Same phases, different elementwise formulas.
// Tiles [0, tiles) are batch rows; tiles [tiles, tiles + vtiles) are VALIDATION rows (maximum likelihood only): forward
// sweep and loss, no gradient -- they ride in the same launch, on otherwise idle CUs, instead of a second kernel.
template <int RPW>
__device__ __forceinline__ void fit_tile_load(float* __restrict__ tile, int stride, const float* __restrict__ src, int64_t r0,
                                              int64_t n, int d, bool rev) {
    const int lane = threadIdx.x & 63;
    const int64_t rows = n - r0 < RPW ? n - r0 : RPW;
    const int total = (int)rows * d;
    const float* s = src + r0 * d;
    for (int i = lane; i < RPW * d; i += kWave) {
        const int r = i / d, c = i - r * d;
        tile[r * stride + (rev ? d - 1 - c : c)] = i < total ? s[i] : 0.f;  // rows beyond n: zeros
    }
}

template <int HP, bool RKL, int RPW>
__global__ void __launch_bounds__(kFitBlock) fit_grad_kernel(NfmcRealNVP f, NfmcPotential pot, const float* __restrict__ x,
                                                             int64_t n, const float* __restrict__ xv, int64_t nv,
                                                             float* __restrict__ partial, int64_t pstride,
                                                             int64_t ea_off, int d4, int64_t n_params, int64_t tiles,
                                                             int64_t vtiles, const float* __restrict__ run_state) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (run_state && (run_state[3] != 0.f || run_state[4] != 0.f)) return;   // the run has ended
    const FlowGeom g = make_geom(f);
    const int d = g.d, stride = tile_stride(d), hs = fit_hb_stride(HP);
    const int lane = threadIdx.x;
    float* const xt = lds;                       // state tile: row r at xt + r * stride
    float* const gt = lds + RPW * stride;        // gradient of the loss with respect to the state
    float* const hb = lds + 2 * RPW * stride;    // per row: h_last | delta_last | h_first | delta_first (HP each)
    const bool rowlane = lane < RPW;             // lanes that own a row in the ROW phases
    float* const xrow = xt + lane * stride;
    float* const grow = gt + lane * stride;
    float* const P = partial + (int64_t)blockIdx.x * pstride;
    const FitOff o = fit_offsets(g, HP);
    const bool rev_last = (g.n_coupling & 1) != 0;
    bool first = true;
    float loss_acc = 0.f, rows_acc = 0.f, vloss_acc = 0.f, vrows_acc = 0.f;
    auto emit = [&](int64_t idx, float v) { P[idx] = first ? v : P[idx] + v; };
    for (int64_t tile = blockIdx.x; tile < tiles + vtiles; tile += gridDim.x) {
        if (tile >= tiles) {   // a validation tile (wave-uniform): loss only
            const int64_t r0 = (tile - tiles) * RPW;
            __syncthreads();
            fit_tile_load<RPW>(xt, stride, xv, r0, nv, d, false);
            __syncthreads();
            if (rowlane && r0 + lane < nv) {
                const float ld = flow_forward_row<HP>(xrow, f, g);
                float ss = 0.f;
                for (int c = 0; c < d; ++c) ss = fmaf(xrow[c], xrow[c], ss);
                vloss_acc += 0.5f * ss + 0.5f * (float)d * kLog2Pi - ld;
                vrows_acc += 1.f;
            }
            continue;
        }
        const int64_t r0 = tile * RPW;
        const int nvalid = (int)(n - r0 < RPW ? n - r0 : RPW);
        const bool valid = lane < nvalid;
        __syncthreads();
        fit_tile_load<RPW>(xt, stride, x, r0, n, d, RKL && rev_last);
        __syncthreads();
        if (rowlane) {
        if constexpr (!RKL) {
            // ---- forward: z = f(x) in place, loss_i = -log N(z) - logdet
            const float ld = flow_forward_row<HP>(xrow, f, g);
            float ss = 0.f;
            for (int c = 0; c < d; ++c) {
                const float z = xrow[c];
                ss = fmaf(z, z, ss);
                grow[c] = valid ? z : 0.f;           // dL/dz of 0.5 |z|^2; rows beyond the batch carry no gradient
            }
            if (valid) {
                loss_acc += 0.5f * ss + 0.5f * (float)d * kLog2Pi - ld;
                rows_acc += 1.f;
            }
        } else {
            // ---- x = f^-1(z) in place, loss_i = log N(z) - logdet_inverse + U(x); dL/dx = grad U
            float ss = 0.f;
            for (int c = 0; c < d; ++c) ss = fmaf(xrow[c], xrow[c], ss);
            const float ld = flow_inverse_row<HP>(xrow, f, g);
            const float u = potential_value_grad_row(xrow, grow, pot, d);
            if (!valid)
                for (int c = 0; c < d; ++c) grow[c] = 0.f;
            if (valid) {
                loss_acc += -0.5f * ss - 0.5f * (float)d * kLog2Pi - ld + u;
                rows_acc += 1.f;
            }
        }
        }
        __syncthreads();
        if constexpr (!RKL) {
            // ---- last ElementwiseAffine (logical coordinates), transposed: z_p = e^s y_p + t
            for (int c = lane; c < d; c += kFitBlock) {
                const int p = phys(c, d, rev_last);
                const float s = f.ea1_log_scale[c], t = f.ea1_shift[c];
                const float es = fast_exp(s), eis = fast_exp(-s);
                float as = 0.f, at = 0.f;
                for (int r = 0; r < RPW; ++r) {
                    const float gz = gt[r * stride + p], zc = xt[r * stride + p] - t;
                    as = fmaf(gz, zc, as);
                    at += gz;
                    gt[r * stride + p] = gz * es;
                    xt[r * stride + p] = zc * eis;
                }
                emit(ea_off + 2 * d4 + c, as - (float)nvalid);   // d(-logdet)/ds = -1 per row
                emit(ea_off + 3 * d4 + c, at);
            }
        } else {
            // ---- first ElementwiseAffine inverted, transposed: x = (y - t) e^-s, -logdet_inverse contains +s
            for (int c = lane; c < d; c += kFitBlock) {
                const float s = f.ea0_log_scale[c], t = f.ea0_shift[c];
                const float es = fast_exp(s), eis = fast_exp(-s);
                float as = 0.f, at = 0.f;
                for (int r = 0; r < RPW; ++r) {
                    const float gx = gt[r * stride + c], xv = xt[r * stride + c];
                    const float gy = gx * eis;
                    as = fmaf(-gx, xv, as);
                    at -= gy;
                    gt[r * stride + c] = gy;
                    xt[r * stride + c] = fmaf(es, xv, t);
                }
                emit(ea_off + c, as + (float)nvalid);
                emit(ea_off + d4 + c, at);
            }
        }
        __syncthreads();
        for (int li = 0; li < g.n_coupling; ++li) {
            const int l = RKL ? li : g.n_coupling - 1 - li;
            const bool rev = (l & 1) == 0;
            const float* __restrict__ W = f.weights + l * g.layer_stride;
            const int64_t L0 = (int64_t)l * g.layer_stride;
            // ---- ROW phase (lane = row)
            if (rowlane) {
                float h1[HP], hl[HP];
                const float* b1 = W + o.b1;
#pragma unroll
                for (int k = 0; k < HP; ++k) h1[k] = b1[k];
                for (int j = 0; j < g.d_a; ++j) {
                    const float xj = xrow[phys(j, d, rev)];
                    const float* w = W + (int64_t)j * HP;
#pragma unroll
                    for (int k = 0; k < HP; ++k) h1[k] = fmaf(w[k], xj, h1[k]);
                }
#pragma unroll
                for (int k = 0; k < HP; ++k) h1[k] = fast_tanh(h1[k]);
                if (g.n_hl > 1) {
                    const float* Wh = W + o.wht;
                    const float* bh = W + o.bh;
#pragma unroll
                    for (int k = 0; k < HP; ++k) hl[k] = bh[k];
#pragma unroll
                    for (int i = 0; i < HP; ++i)
#pragma unroll
                        for (int k = 0; k < HP; ++k) hl[k] = fmaf(Wh[i * HP + k], h1[i], hl[k]);
#pragma unroll
                    for (int k = 0; k < HP; ++k) hl[k] = fast_tanh(hl[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < HP; ++k) hl[k] = h1[k];
                }
                float gh[HP];
#pragma unroll
                for (int k = 0; k < HP; ++k) gh[k] = 0.f;
                const float* W3 = W + o.w3;
                const float* b3 = W + o.b3;
                for (int t = 0; t < g.d_b; ++t) {
                    float ua = b3[t], ub = b3[g.d_b + t];
                    const float* wa = W3 + (int64_t)t * HP;
                    const float* wb = W3 + (int64_t)(g.d_b + t) * HP;
#pragma unroll
                    for (int k = 0; k < HP; ++k) {
                        ua = fmaf(wa[k], hl[k], ua);
                        ub = fmaf(wb[k], hl[k], ub);
                    }
                    const float alpha = fast_exp(fmaf(0.5f, ua, g.log1m)) + g.m;
                    const float ra = __builtin_amdgcn_rcpf(alpha);
                    const int p = phys(g.d_a + t, d, rev);
                    float da, db;
                    if constexpr (!RKL) {
                        const float xb = (xrow[p] - 0.5f * ub) * ra;
                        const float gz = grow[p];
                        xrow[p] = xb;                                    // the layer's input
                        grow[p] = gz * alpha;                            // dL/dx_b
                        const float ga = fmaf(gz, xb, valid ? -ra : 0.f);   // dL/dalpha: z_b = alpha x_b + beta, -log alpha
                        da = 0.5f * ga * (alpha - g.m);
                        db = 0.5f * gz;
                    } else {
                        // inverse layer v_b = (y_b - beta) / alpha with +log alpha in the loss; the tile holds v, dL/dv
                        const float vb = xrow[p], gv = grow[p];
                        xrow[p] = fmaf(alpha, vb, 0.5f * ub);            // the inverse layer's input y_b
                        const float gy = gv * ra;
                        grow[p] = gy;                                    // dL/dy_b
                        const float ga = ((valid ? 1.f : 0.f) - gv * vb) * ra;
                        da = 0.5f * ga * (alpha - g.m);
                        db = -0.5f * gy;
                    }
#pragma unroll
                    for (int k = 0; k < HP; ++k) gh[k] = fmaf(wa[k], da, fmaf(wb[k], db, gh[k]));
                }
                float dl[HP], df[HP];   // deltas (gradients of the pre-activations) of the last / first hidden layer
#pragma unroll
                for (int k = 0; k < HP; ++k) dl[k] = gh[k] * (1.f - hl[k] * hl[k]);
                if (g.n_hl > 1) {
                    const float* Wh = W + o.wht;
#pragma unroll
                    for (int i = 0; i < HP; ++i) {
                        float a = 0.f;
#pragma unroll
                        for (int k = 0; k < HP; ++k) a = fmaf(Wh[i * HP + k], dl[k], a);
                        df[i] = a * (1.f - h1[i] * h1[i]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < HP; ++k) df[k] = dl[k];
                }
                for (int j = 0; j < g.d_a; ++j) {
                    const float* w = W + (int64_t)j * HP;
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < HP; ++k) a = fmaf(w[k], df[k], a);
                    grow[phys(j, d, rev)] += a;
                }
                float* hrow = hb + lane * hs;
#pragma unroll
                for (int k = 0; k < HP; ++k) {
                    hrow[k] = hl[k];
                    hrow[HP + k] = dl[k];
                    hrow[2 * HP + k] = h1[k];
                    hrow[3 * HP + k] = df[k];
                }
            }
            __syncthreads();
            // ---- TRANSPOSED phases (lane = output coordinate; rows walked in order)
            // W3 (2 d_b, HP) and b3: the lane re-derives alpha of (row, t) from the row's stored activations
            for (int t = lane; t < g.d_b; t += kFitBlock) {
                float wa[HP], wb[RKL ? HP : 1], aa[HP], ab[HP];
                const float* war = W + o.w3 + (int64_t)t * HP;
                const float* wbr = W + o.w3 + (int64_t)(g.d_b + t) * HP;
#pragma unroll
                for (int k = 0; k < HP; ++k) {
                    wa[k] = war[k];
                    if constexpr (RKL) wb[k] = wbr[k];
                    aa[k] = 0.f;
                    ab[k] = 0.f;
                }
                const float ba = W[o.b3 + t], bb = W[o.b3 + g.d_b + t];
                float sa = 0.f, sb = 0.f;
                const int p = phys(g.d_a + t, d, rev);
                for (int r = 0; r < RPW; ++r) {
                    const float* hrow = hb + r * hs;
                    float h[HP];
                    float ua = ba, ub = bb;
#pragma unroll
                    for (int k = 0; k < HP; ++k) {
                        h[k] = hrow[k];
                        ua = fmaf(wa[k], h[k], ua);
                        if constexpr (RKL) ub = fmaf(wb[k], h[k], ub);
                    }
                    const float alpha = fast_exp(fmaf(0.5f, ua, g.log1m)) + g.m;
                    const float ra = __builtin_amdgcn_rcpf(alpha);
                    float da, db;
                    if constexpr (!RKL) {
                        const float gz = gt[r * stride + p] * ra, xb = xt[r * stride + p];
                        const float ga = fmaf(gz, xb, r < nvalid ? -ra : 0.f);
                        da = 0.5f * ga * (alpha - g.m);
                        db = 0.5f * gz;
                    } else {
                        const float gy = gt[r * stride + p];                         // = dL/dv / alpha
                        const float vb = (xt[r * stride + p] - 0.5f * ub) * ra;
                        const float ga = ((r < nvalid ? 1.f : 0.f) - gy * alpha * vb) * ra;
                        da = 0.5f * ga * (alpha - g.m);
                        db = -0.5f * gy;
                    }
                    sa += da;
                    sb += db;
#pragma unroll
                    for (int k = 0; k < HP; ++k) {
                        aa[k] = fmaf(da, h[k], aa[k]);
                        ab[k] = fmaf(db, h[k], ab[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < HP; ++k) {
                    emit(L0 + o.w3 + (int64_t)t * HP + k, aa[k]);
                    emit(L0 + o.w3 + (int64_t)(g.d_b + t) * HP + k, ab[k]);
                }
                emit(L0 + o.b3 + t, sa);
                emit(L0 + o.b3 + g.d_b + t, sb);
            }
            // Wh^T (HP_in, HP_out), bh, b1
            if (g.n_hl > 1) {
                for (int e = lane; e < HP * HP; e += kFitBlock) {
                    const int i = e / HP, k = e - i * HP;
                    float a = 0.f;
                    for (int r = 0; r < RPW; ++r) a = fmaf(hb[r * hs + 2 * HP + i], hb[r * hs + HP + k], a);
                    emit(L0 + o.wht + e, a);
                }
            }
            if (lane < HP) {
                float a1 = 0.f, a2 = 0.f;
                for (int r = 0; r < RPW; ++r) {
                    a1 += hb[r * hs + 3 * HP + lane];
                    a2 += hb[r * hs + HP + lane];
                }
                emit(L0 + o.b1 + lane, a1);
                if (g.n_hl > 1) emit(L0 + o.bh + lane, a2);
            }
            // W1^T (d_a, HP)
            for (int j = lane; j < g.d_a; j += kFitBlock) {
                float a[HP];
#pragma unroll
                for (int k = 0; k < HP; ++k) a[k] = 0.f;
                const int p = phys(j, d, rev);
                for (int r = 0; r < RPW; ++r) {
                    const float xj = xt[r * stride + p];
#pragma unroll
                    for (int k = 0; k < HP; ++k) a[k] = fmaf(xj, hb[r * hs + 3 * HP + k], a[k]);
                }
#pragma unroll
                for (int k = 0; k < HP; ++k) emit(L0 + (int64_t)j * HP + k, a[k]);
            }
            __syncthreads();
        }
        if constexpr (!RKL) {
            // ---- first ElementwiseAffine, transposed: the tile holds its OUTPUT y = e^s x + t and dL/dy
            for (int c = lane; c < d; c += kFitBlock) {
                const float t = f.ea0_shift[c];
                float as = 0.f, at = 0.f;
                for (int r = 0; r < RPW; ++r) {
                    const float gy = gt[r * stride + c];
                    as = fmaf(gy, xt[r * stride + c] - t, as);
                    at += gy;
                }
                emit(ea_off + c, as - (float)nvalid);
                emit(ea_off + d4 + c, at);
            }
        } else {
            // ---- last ElementwiseAffine inverted (logical coordinates), transposed: the tile holds v = (z - t) e^-s and dL/dv
            for (int c = lane; c < d; c += kFitBlock) {
                const int p = phys(c, d, rev_last);
                const float eis = fast_exp(-f.ea1_log_scale[c]);
                float as = 0.f, at = 0.f;
                for (int r = 0; r < RPW; ++r) {
                    const float gv = gt[r * stride + p];
                    as = fmaf(-gv, xt[r * stride + p], as);
                    at = fmaf(-gv, eis, at);
                }
                emit(ea_off + 2 * d4 + c, as + (float)nvalid);
                emit(ea_off + 3 * d4 + c, at);
            }
        }
        first = false;
    }
    // losses and row counts of this workgroup's rows: fixed-order sums over the lanes
    __syncthreads();
    lds[lane] = loss_acc;
    lds[64 + lane] = rows_acc;
    lds[128 + lane] = vloss_acc;
    lds[192 + lane] = vrows_acc;
    __syncthreads();
    if (lane < 4) {
        float a = 0.f;
        for (int r = 0; r < 64; ++r) a += lds[64 * lane + r];
        P[n_params + lane] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fold of the partial gradients + AdamW + the epoch loop's bookkeeping, one launch.
//   sums      per parameter: 16 threads add up to 16 slabs each (all loads in flight at once, added in slab order), one
//             thread adds the 16 sub-sums in order; the four tail values by a fixed butterfly -- every sum has a fixed
//             association, so a run repeats bit for bit.  Only the slabs of workgroups that produced gradients are summed
//             (a workgroup that only saw validation rows never writes its gradient entries).
//   AdamW     torch.optim.AdamW: p <- p (1 - lr wd);  m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g^2;
//             p <- p - lr (m / bc1) / (sqrt(v / bc2) + eps),  bc = 1 - beta^step.
//   run       (ctl given) flow_training._loop / torchflows' fit loop as nfmc drives it (jump.py:139-151: early stopping,
//             keep_best_weights, ValueError on divergence), decided here so that the host enqueues epochs without reading
//             anything back.  Call c works on the weights w_c:
//               validation rows:  their loss at w_c closes epoch c - 1 (best-so-far / early stopping; best weights <- w_c),
//                                 then -- unless the run stopped or this is the closing call -- the step w_c -> w_{c+1};
//               no validation:    the batch loss at w_c stands in, paired with w_{c+1} (the reference's `_loop`).
//             A non-finite batch loss ends the run (state[4] = 1: the host raises ValueError) or, for variational fits
//             with check_for_divergences = False, skips the epoch.  Once stopped / diverged every later call of the run is
//             a no-op (the gradient kernel returns at once, this kernel copies the state through).
struct FitFoldArgs {
    float* params;
    float* am;
    float* av;
    float* prev;
    float* best;
    const float* partial;
    int64_t pstride;
    int nparts_tail, nparts_grad;
    int64_t n_params;
    NfmcAdamW opt;
    float* status;
    int has_ctl, call, has_val;
    NfmcFitControl ctl;
    const float* state_in;
    float* state_out;
};

__global__ void __launch_bounds__(256) fit_fold_kernel(FitFoldArgs a) {
    __shared__ float red[4][4];
    __shared__ float sub[16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- run state
    float best = INFINITY, since = 0.f, applied = 0.f, stopped = 0.f, diverged = 0.f, booked = 0.f;
    if (a.has_ctl && a.call > 0) {
        best = a.state_in[NFMC_FIT_BEST_LOSS];
        since = a.state_in[NFMC_FIT_SINCE_BEST];
        applied = a.state_in[NFMC_FIT_APPLIED];
        stopped = a.state_in[NFMC_FIT_STOPPED];
        diverged = a.state_in[NFMC_FIT_DIVERGED];
        booked = a.state_in[NFMC_FIT_BOOKED];
    }
    const bool frozen = stopped != 0.f || diverged != 0.f;
    if (frozen) {   // uniform: the run ended in an earlier call
        if (blockIdx.x == 0 && tid < NFMC_FIT_STATE_FLOATS) a.state_out[tid] = a.state_in[tid];
        return;
    }
    // ---- loss sum, rows, validation loss sum, validation rows over all slabs (at most 256 of them)
    float t4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int w = tid; w < a.nparts_tail; w += 256) {
        const float* t = a.partial + (int64_t)w * a.pstride + a.n_params;
#pragma unroll
        for (int k = 0; k < 4; ++k) t4[k] += t[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) t4[k] = group_allreduce<64>(t4[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave][k] = t4[k];
    }
    __syncthreads();
    float sums[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sums[k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    const float rows = sums[1], loss = sums[0] / rows;
    const bool ok = fabsf(loss) <= 3.0e38f;
    const float val = sums[3] > 0.f ? sums[2] / sums[3] : loss;
    // ---- decisions (every thread of every workgroup takes the same ones)
    bool do_step = ok, prev_to_best = false, new_to_best = false;
    int step = a.opt.step;
    if (a.has_ctl) {
        const bool closing = a.has_val && a.call >= a.ctl.n_epochs;
        if (a.call == 0 && a.ctl.keep_best_weights) prev_to_best = true;   // the weights the run started from are the fallback
        if (a.has_val && a.call > 0) {   // the validation loss at w_c closes epoch c - 1
            if (!(fabsf(val) <= 3.0e38f)) {
                diverged = 1.f;
            } else if (val < best) {
                best = val;
                since = 0.f;
                prev_to_best = a.ctl.keep_best_weights != 0;
            } else {
                since += 1.f;
                if (a.ctl.early_stopping && since > (float)a.ctl.early_stopping_threshold) stopped = 1.f;
            }
            booked += 1.f;
        }
        do_step = false;
        if (!closing && stopped == 0.f && diverged == 0.f) {
            if (ok) {
                do_step = true;
                step = (int)applied + 1;
                applied += 1.f;
                if (!a.has_val) {   // the batch loss before the step stands in for the validation loss
                    if (loss < best) {
                        best = loss;
                        since = 0.f;
                        new_to_best = a.ctl.keep_best_weights != 0;
                    } else {
                        since += 1.f;
                        if (a.ctl.early_stopping && since > (float)a.ctl.early_stopping_threshold) stopped = 1.f;
                    }
                    booked += 1.f;
                }
            } else if (!a.ctl.skip_nonfinite) {
                diverged = 1.f;
            }
        }
        if (blockIdx.x == 0 && tid == 0) {
            a.state_out[NFMC_FIT_BEST_LOSS] = best;
            a.state_out[NFMC_FIT_SINCE_BEST] = since;
            a.state_out[NFMC_FIT_APPLIED] = applied;
            a.state_out[NFMC_FIT_STOPPED] = stopped;
            a.state_out[NFMC_FIT_DIVERGED] = diverged;
            a.state_out[NFMC_FIT_LAST_LOSS] = loss;
            a.state_out[NFMC_FIT_LAST_VAL] = val;
            a.state_out[NFMC_FIT_BOOKED] = booked;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        a.status[0] = loss;
        a.status[1] = ok ? 1.f : 0.f;
        a.status[2] = val;   // validation loss at the parameters BEFORE the step
    }
    // ---- gradient of parameter i = 16 blockIdx + c: 16 sub-sums of up to 16 slabs each
    const int c = tid & 15, part = tid >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + c;
    if (do_step) {
        const int chunk = (a.nparts_grad + 15) / 16;
        const int w0 = part * chunk, w1 = min(a.nparts_grad, w0 + chunk);
        float g = 0.f;
        if (i < a.n_params) {
            for (int w = w0; w < w1; w += 16) {
                float t[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) t[k] = w + k < w1 ? a.partial[(int64_t)(w + k) * a.pstride + i] : 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) g += t[k];
            }
        }
        sub[part][c] = g;
    }
    __syncthreads();
    if (part != 0 || i >= a.n_params) return;
    float p = a.params[i];
    if (a.prev) a.prev[i] = p;
    const bool reset = a.has_ctl && a.call == 0;
    float pn = p;
    if (do_step) {
        float gsum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) gsum += sub[k][c];
        const float gr = gsum / rows;
        const float bc1 = 1.f - powf(a.opt.beta1, (float)step), bc2 = 1.f - powf(a.opt.beta2, (float)step);
        pn = p * (1.f - a.opt.lr * a.opt.weight_decay);
        const float m = a.opt.beta1 * (reset ? 0.f : a.am[i]) + (1.f - a.opt.beta1) * gr;
        const float v = a.opt.beta2 * (reset ? 0.f : a.av[i]) + (1.f - a.opt.beta2) * gr * gr;
        a.am[i] = m;
        a.av[i] = v;
        pn = pn - a.opt.lr * (m / bc1) / (sqrtf(v / bc2) + a.opt.eps);
        a.params[i] = pn;
    } else if (reset) {
        a.am[i] = 0.f;
        a.av[i] = 0.f;
    }
    if (a.best) {
        if (new_to_best) a.best[i] = pn;
        else if (prev_to_best) a.best[i] = p;
    }
}

static bool fit_rows_shape(const NfmcRealNVP* f, int64_t n_params, int* hp_out, int* ch_out) {
    const int hp = nfmc_realnvp_padded_hidden(f->n_hidden);
    if (hp != 4 && hp != 8) return false;
    const int d_b = f->d - f->d / 2;
    const int ch = d_b <= 64 ? 1 : (d_b <= 128 ? 2 : 4);
    if (d_b > 256) return false;
    if (fit_rows_lds_bytes(n_params, hp, ch) > 160 * 1024) return false;
    *hp_out = hp;
    *ch_out = ch;
    return true;
}

// trainable floats of a flow in the fit's layout with the tightest packing (what nfmc_flow_fit_supported_f32 assumes)
static int64_t fit_min_params(const NfmcRealNVP* f) {
    const int64_t stride = (nfmc_coupling_layer_floats(f->d, f->n_hidden, f->n_hidden_layers, 0) + 3) / 4 * 4;
    const int64_t d4 = (f->d + 3) / 4 * 4;
    return (int64_t)(f->n_coupling > 0 ? f->n_coupling : 1) * stride + 4 * d4;
}

static bool fit_wide(const NfmcRealNVP* f);
static int64_t mfma_layer_floats_of(const NfmcRealNVP& f) { return nfmc_realnvp_layer_floats(f.d, f.n_hidden, f.n_hidden_layers); }
static bool fit_supported(const NfmcRealNVP* f) {
    if (fit_wide(f)) return true;
    if (!f || f->n_bins != 0 || f->d <= 0 || f->d > 512 || f->n_coupling < 0) return false;
    if (f->n_hidden <= 0 || f->n_hidden > 32 || f->n_hidden_layers < 1 || f->n_hidden_layers > 2) return false;
    const int hp = nfmc_realnvp_padded_hidden(f->n_hidden);
    if (hp <= 0) return false;
    if (hp <= 8) {
        int h, c;
        return fit_rows_shape(f, fit_min_params(f), &h, &c);
    }
    return f->d <= 256 && fit_lds_bytes(f->d, hp) <= 160 * 1024;   // widths 9..32: the row-per-lane kernel, d <= 256
}

// conditioners of width 33..128 at d = 64 / 128: the matrix-core kernel (fit_mfma.hip), blob in the matrix-core layout
static bool fit_wide(const NfmcRealNVP* f) {
    return f && f->n_bins == 0 && f->n_coupling > 0 && f->n_hidden > 32 && nfmc_mfma_supported(f->d, f->n_hidden, f->n_hidden_layers);
}

}  // namespace nfmc

using namespace nfmc;

extern "C" int nfmc_flow_fit_supported_f32(const NfmcRealNVP* flow) { return fit_supported(flow) ? 1 : 0; }

extern "C" int64_t nfmc_flow_fit_partial_floats(int64_t n, int64_t n_params) {
    if (n <= 0 || n_params <= 0) return 0;
    return (int64_t)256 * (n_params + kFitTail);   // one slab per workgroup; never more than 256 workgroups
}

extern "C" int64_t nfmc_flow_fit_workspace(const NfmcRealNVP* flow, int64_t n, int64_t n_val, int64_t n_params,
                                           int64_t* partial_floats) {
    if (partial_floats) *partial_floats = 0;
    if (!flow || n <= 0 || n_val < 0 || n_params <= 0 || !fit_supported(flow)) return 0;
    if (!fit_wide(flow)) {
        if (partial_floats) *partial_floats = nfmc_flow_fit_partial_floats(n, n_params);
        return 0;
    }
    const int grid = fit_mfma_grid(n, n_val);
    if (partial_floats) *partial_floats = (int64_t)grid * (n_params + kFitTail);
    return fit_mfma_ck_floats(flow->d, nfmc_realnvp_padded_hidden(flow->n_hidden), flow->n_hidden_layers, flow->n_coupling, grid) *
           (int64_t)sizeof(float);
}

// One call of a run (or one stand-alone step when ctl == NULL): gradient launch + fold launch.
static int fit_call(const NfmcFlowFit* fit, const NfmcPotential* pot, const float* x, int64_t n, const NfmcAdamW* opt,
                    const NfmcFitControl* ctl, int call, hipStream_t st) {
    const NfmcRealNVP& f = fit->flow;
    const int d4 = (f.d + 3) / 4 * 4;
    const bool has_val = !pot && fit->x_val && fit->n_val > 0;
    const int64_t nv = has_val ? fit->n_val : 0;
    const bool closing = ctl && has_val && call >= ctl->n_epochs;
    const int64_t n_train = closing ? 0 : n;   // the closing call only needs the validation loss
    const int64_t pstride = fit->n_params + kFitTail;
    const int hp = nfmc_realnvp_padded_hidden(f.n_hidden);
    const float* state_in = ctl && call > 0 ? fit->run_state + NFMC_FIT_STATE_FLOATS * (call & 1) : nullptr;
    float* state_out = ctl ? fit->run_state + NFMC_FIT_STATE_FLOATS * ((call + 1) & 1) : nullptr;
    NfmcPotential p0 = {};
    if (pot) p0 = *pot;
    int grid = 0, grad_slabs = 0;
    int rhp = 0, rch = 0;
    if (fit_wide(&f)) {
        const int64_t tiles = (n_train + kMfmaChainsFit - 1) / kMfmaChainsFit, vtiles = (nv + kMfmaChainsFit - 1) / kMfmaChainsFit;
        grid = fit_mfma_grid(n_train, nv);
        grad_slabs = (int)(tiles < grid ? tiles : grid);
        if (fit->partial_floats < (int64_t)grid * pstride) return NFMC_ESCRATCH;
        const int64_t ckf = fit_mfma_ck_floats(f.d, hp, f.n_hidden_layers, f.n_coupling, grid);
        if (!fit->scratch || fit->scratch_bytes < ckf * (int64_t)sizeof(float)) return NFMC_ESCRATCH;
        if ((reinterpret_cast<uintptr_t>(fit->params) & 15) != 0 || (reinterpret_cast<uintptr_t>(fit->scratch) & 15) != 0 ||
            (reinterpret_cast<uintptr_t>(x) & 15) != 0 || (nv > 0 && (reinterpret_cast<uintptr_t>(fit->x_val) & 15) != 0) ||
            (f.layer_stride & 3) != 0 || (fit->ea_off & 3) != 0)
            return NFMC_EALIGN;
        FitMfmaArgs a;
        a.f = f;
        a.pot = p0;
        a.x = x;
        a.n = n_train;
        a.xv = fit->x_val;
        a.nv = nv;
        a.partial = fit->partial;
        a.pstride = pstride;
        a.ea_off = fit->ea_off;
        a.d4 = d4;
        a.n_params = fit->n_params;
        a.tiles = tiles;
        a.vtiles = vtiles;
        a.ck = fit->scratch;
        a.run_state = state_in;
        const int rc = fit_mfma_launch(pot != nullptr, a, grid, st);
        if (rc != 0) return rc;
    } else if (fit_rows_shape(&f, fit->n_params, &rhp, &rch)) {
        if ((f.layer_stride & 3) != 0 || (fit->ea_off & 3) != 0 || (reinterpret_cast<uintptr_t>(fit->params) & 15) != 0)
            return NFMC_EALIGN;
        // rows per wave tile: 4 (2 at CH = 4: register budget) when that still gives every SIMD of the machine a tile, else 1
        const int S = (n_train + nv) >= 4096 ? (rch == 4 ? 2 : 4) : 1;
        const int64_t tiles = (n_train + S - 1) / S, tiles4 = (tiles + kFrWaves - 1) / kFrWaves * kFrWaves;
        const int64_t vtiles = (nv + S - 1) / S;
        const int64_t wgs = (tiles4 + vtiles + kFrWaves - 1) / kFrWaves;
        grid = (int)(wgs < 256 ? (wgs < 1 ? 1 : wgs) : 256);
        grad_slabs = (int)(tiles4 / kFrWaves < grid ? tiles4 / kFrWaves : grid);
        if (fit->partial_floats < (int64_t)grid * pstride) return NFMC_ESCRATCH;
        FitRowsArgs a;
        a.f = f;
        a.pot = p0;
        a.x = x;
        a.n = n_train;
        a.xv = fit->x_val;
        a.nv = nv;
        a.partial = fit->partial;
        a.pstride = pstride;
        a.ea_off = fit->ea_off;
        a.d4 = d4;
        a.n_params = fit->n_params;
        a.tiles4 = tiles4;
        a.vtiles = vtiles;
        a.params = fit->params;
        a.run_state = state_in;
        const size_t lds = fit_rows_lds_bytes(fit->n_params, rhp, rch);
        const int rc = rhp == 4 ? fit_rows_launch_h4(pot != nullptr, rch, S, a, grid, lds, st)
                                : fit_rows_launch_h8(pot != nullptr, rch, S, a, grid, lds, st);
        if (rc != 0) return rc;
    } else {
        const int rpw = fit_rows_per_wave(n_train);
        const int64_t tiles = (n_train + rpw - 1) / rpw, vtiles = nv > 0 ? (nv + rpw - 1) / rpw : 0;
        const int64_t all = tiles + vtiles;
        grid = (int)(all < 256 ? (all < 1 ? 1 : all) : 256);
        grad_slabs = (int)(tiles < grid ? tiles : grid);   // workgroups beyond the batch tiles never write gradient entries
        if (fit->partial_floats < (int64_t)grid * pstride) return NFMC_ESCRATCH;
        const size_t lds = fit_lds_bytes(f.d, hp, rpw);
#define NFMC_FIT_LAUNCH3(HPV, RKLV, RPWV)                                                                                 \
    {                                                                                                                     \
        auto kern = fit_grad_kernel<HPV, RKLV, RPWV>;                                                                     \
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        if (e != hipSuccess) return (int)e;                                                                               \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kFitBlock), lds, st, f, p0, x, n_train, fit->x_val, nv, fit->partial,   \
                           pstride, fit->ea_off, d4, fit->n_params, tiles, vtiles, state_in);                             \
    }
#define NFMC_FIT_LAUNCH2(HPV, RKLV)              \
    if (rpw == 16) NFMC_FIT_LAUNCH3(HPV, RKLV, 16) \
    else NFMC_FIT_LAUNCH3(HPV, RKLV, 64)
#define NFMC_FIT_LAUNCH(HPV)            \
    if (pot) NFMC_FIT_LAUNCH2(HPV, true) \
    else NFMC_FIT_LAUNCH2(HPV, false)
        switch (hp) {
            case 16: NFMC_FIT_LAUNCH(16) break;
            case 32: NFMC_FIT_LAUNCH(32) break;
            default: return NFMC_EUNSUPPORTED;
        }
#undef NFMC_FIT_LAUNCH
#undef NFMC_FIT_LAUNCH2
#undef NFMC_FIT_LAUNCH3
    }
    FitFoldArgs fa;
    fa.params = fit->params;
    fa.am = fit->adam_m;
    fa.av = fit->adam_v;
    fa.prev = fit->params_prev;
    fa.best = ctl ? fit->best : nullptr;
    fa.partial = fit->partial;
    fa.pstride = pstride;
    fa.nparts_tail = grid;
    fa.nparts_grad = grad_slabs;
    fa.n_params = fit->n_params;
    fa.opt = *opt;
    fa.status = fit->status;
    fa.has_ctl = ctl ? 1 : 0;
    fa.call = call;
    fa.has_val = has_val ? 1 : 0;
    fa.ctl = ctl ? *ctl : NfmcFitControl{};
    fa.state_in = state_in;
    fa.state_out = state_out;
    const int blocks = (int)((fit->n_params + 15) / 16);
    hipLaunchKernelGGL(fit_fold_kernel, dim3(blocks), dim3(256), 0, st, fa);
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

static int fit_check(const NfmcFlowFit* fit, const NfmcPotential* pot, const float* x, int64_t n, const NfmcAdamW* opt) {
    if (!fit || !x || !opt || n <= 0) return NFMC_EINVAL;
    const NfmcRealNVP& f = fit->flow;
    if (!fit->params || !fit->adam_m || !fit->adam_v || !fit->partial || !fit->status) return NFMC_EINVAL;
    if (!fit_supported(&f)) return NFMC_EUNSUPPORTED;
    if (fit_wide(&f) && f.layer_stride < mfma_layer_floats_of(f)) return NFMC_EINVAL;
    if (pot && pot->kind != NFMC_POT_QUADRATIC && pot->kind != NFMC_POT_FUNNEL) return NFMC_EUNSUPPORTED;
    const int d4 = (f.d + 3) / 4 * 4;
    if (fit->ea_off < (int64_t)f.n_coupling * f.layer_stride || fit->n_params < fit->ea_off + 4 * d4) return NFMC_EINVAL;
    // the flow's pointers must be views of the trainable vector in the gradient's layout
    if (f.weights != fit->params || f.ea0_log_scale != fit->params + fit->ea_off || f.ea0_shift != fit->params + fit->ea_off + d4 ||
        f.ea1_log_scale != fit->params + fit->ea_off + 2 * d4 || f.ea1_shift != fit->params + fit->ea_off + 3 * d4)
        return NFMC_EINVAL;
    return NFMC_OK;
}

extern "C" int nfmc_flow_fit_step_f32(const NfmcFlowFit* fit, const float* x, int64_t n, const NfmcAdamW* opt,
                                      nfmc_stream_t stream) {
    const int rc = fit_check(fit, nullptr, x, n, opt);
    if (rc != NFMC_OK) return rc;
    if (opt->step < 1) return NFMC_EINVAL;
    return fit_call(fit, nullptr, x, n, opt, nullptr, 0, (hipStream_t)stream);
}

extern "C" int nfmc_flow_variational_fit_step_f32(const NfmcFlowFit* fit, const NfmcPotential* pot, const float* z, int64_t n,
                                                  const NfmcAdamW* opt, nfmc_stream_t stream) {
    if (!pot) return NFMC_EINVAL;
    const int rc = fit_check(fit, pot, z, n, opt);
    if (rc != NFMC_OK) return rc;
    if (opt->step < 1) return NFMC_EINVAL;
    return fit_call(fit, pot, z, n, opt, nullptr, 0, (hipStream_t)stream);
}

extern "C" int nfmc_flow_fit_epochs_f32(const NfmcFlowFit* fit, const NfmcPotential* pot, const float* x, int64_t n,
                                        int64_t epoch_stride, const NfmcAdamW* opt, const NfmcFitControl* ctl,
                                        int32_t first_call, int32_t n_calls, nfmc_stream_t stream) {
    if (!ctl || first_call < 0 || n_calls < 0 || epoch_stride < 0) return NFMC_EINVAL;
    const int rc = fit_check(fit, pot, x, n, opt);
    if (rc != NFMC_OK) return rc;
    if (!fit->run_state || (ctl->keep_best_weights && !fit->best)) return NFMC_EINVAL;
    for (int c = first_call; c < first_call + n_calls; ++c) {
        const int r = fit_call(fit, pot, x + (int64_t)(c - first_call) * epoch_stride, n, opt, ctl, c, (hipStream_t)stream);
        if (r != NFMC_OK) return r;
    }
    return NFMC_OK;
}
