// RealNVP in the samplers' register layout (narrow conditioners, HP <= 8): the chain's coordinates stay in
// the VGPRs of its LPC lanes, so a flow pass can run inside the same kernel, on the same registers, as the
// MALA / HMC transitions (no HBM or LDS round trip of the state, and LPC x more lanes than the
// one-chain-per-lane kernels of flow_kernels.hip when the number of chains is small).
//
// Layout ("interleaved 4-blocks"): lane g of a chain group holds, in register i, coordinate
//     c = 4 * ((i / 4) * LPC + g) + (i % 4)
// i.e. 4-coordinate blocks dealt round-robin over the lanes: 16-byte coalesced IO, one Philox block per
// register quad, and -- when d == CPL * LPC and CPL == 8 -- register quad 0 of EVERY lane is the first
// half of the coordinates and quad 1 the second half, so the source/target roles of a coupling layer are
// compile-time per quad (no lane idles in either phase).
//
// Conditioner: each lane accumulates the first-layer pre-activations over its own source coordinates
// (weights from an LDS image, one ds_read_b128 per coordinate for HP = 4).  With LPC >= HP lanes per chain the
// hidden stack is DISTRIBUTED: a DPP reduce-scatter leaves lane g with the complete sum of hidden unit g % HP,
// the lane applies bias + tanh to that one unit, and a DPP all-gather hands every lane all HP activations (in a
// lane-dependent register order that the image's weight rows are pre-permuted for); further hidden layers are
// one HP-term dot product + one tanh per lane.  (Evaluating the stack redundantly on every lane, as the
// LPC < HP layouts still do, was half of a coupling layer's instructions at d = 64.)  Then every lane computes
// the affine parameters of its own target coordinates.  Coordinates that are not source (target) in a layer index an all-zero row,
// and carry an is-target flag of 0, which yields exactly alpha = 1, beta = 0, log alpha = 0 for them:
// no index arithmetic and no branches, also for ragged d.
#pragma once


#include "common.hpp"
#include "flow_device.hpp"

namespace nfmc {

// LDS image of one coupling layer, in floats, indexed by REGISTER SLOT s = i * LPC + g (the slot of the
// coordinate held by register i of lane g), so a lane's rows are `g` rows past a compile-time base and
// consecutive lanes read consecutive rows (bank-conflict free ds_read_b128):
//   W1 (DP x HP): row s = W1T[logical j] if the slot's coordinate is a SOURCE of this layer, else 0
//   b1 HP | [WhT HP x HP | bh HP] x (n_hl - 1)                                   (LPC < HP)
//   b1 HP | [HP rows of HP + 4: row u = weights INTO unit u, position r = from unit u ^ unit_xor(r) | bh[u] | 0 0 0] x (n_hl - 1)
//   W3 (DP x RS): row s = [(alpha weight, beta weight) x HP | b3_alpha | b3_beta | is_target | 0] (zeros if not target);
//       distributed: pair r belongs to hidden unit (g % HP) ^ unit_xor(r)
// followed, after all layers, by the ElementwiseAffine parameters as eight planes of DP floats (slot-indexed, so the
// lanes of a chain read consecutive words: conflict free):
//   e^ls0 | sh0 | e^-ls0 | ls0 | e^ls1 | sh1 | e^-ls1 | ls1   (ea1 already mapped through the final reversal)
// EXACT (d == DP, CPL >= 8): a layer's sources are one half of every lane's registers and its targets the other
// half, so W1 keeps only the source slots' rows and W3 only the target slots' rows (DP/2 each, indexed by
// (i - first register of the half) * LPC + g): half the image -- 31 instead of 62 KB at d = 256, HP = 8, which
// lifts the LDS limit from 2 to 5 workgroups per CU -- and half the staging work per workgroup.
// NB = 8: rational-quadratic spline couplings ('c-rqnsf', flow_device.hpp: rqs_coordinate).  A target slot's W3 row is then
//   [(3 NB - 1) x HP weights, output-major, hidden units in the lane class's order | (3 NB - 1) biases | is_target]
// = 116 floats at HP = 4, 208 at HP = 8 (the lane computes all 23 spline parameters of each of its target coordinates).
template <int CPL, int LPC, int HP, bool EXACT = false, int NB = 0>
struct FlowImage {
    static constexpr int NP = 3 * NB - 1;                    // spline parameters per target coordinate
    static constexpr int RS = NB ? NP * (HP + 1) + 1 : 2 * HP + 4;
    static_assert(RS % 4 == 0, "whole 16-byte groups per row");
    static constexpr int DP = CPL * LPC;
    static constexpr int ROWS = EXACT ? DP / 2 : DP;         // rows of W1 / of W3 per layer
    static constexpr bool DIST = LPC >= HP;                  // one lane class per hidden unit
    static constexpr int HROW = HP + 4;                      // distributed hidden-layer row: HP weights | bias | pad
    static constexpr int HL = DIST ? HP * HROW : HP * HP + HP;  // floats per hidden layer after the first
    __host__ __device__ static int mid_floats(int n_hl) { return HP + (n_hl - 1) * HL; }
    __host__ __device__ static int layer_floats(int n_hl) { return ROWS * HP + mid_floats(n_hl) + ROWS * RS; }
    static constexpr int EA = 8;   // ElementwiseAffine planes of DP floats: e^ls0, sh0, e^-ls0, ls0, e^ls1, sh1, e^-ls1, ls1
    __host__ __device__ static int total_floats(int n_hl, int n_coupling) { return n_coupling * layer_floats(n_hl) + EA * DP; }

    // One slot's rows of one coupling layer (registers; loads only, so that a caller can issue the loads of several
    // items before it stores any of them).
    struct SlotRows {
        float r1[HP], r3[RS];
        bool src, tgt;
    };
    // spline target rows are long (116 / 208 floats): copied piecewise, straight from the blob, by stage()
    __device__ static __forceinline__ void stage_slot_rqs(float* __restrict__ img, const NfmcRealNVP& f, int l, int s, int bmid,
                                                          int lf, int nmid) {
        const int d = f.d, d_a = d / 2, d_b = d - d_a;
        const bool rev = (l & 1) == 0;
        const float* __restrict__ B = f.weights;
        const int W = l * (int)f.layer_stride;
        const int W3 = W + d_a * HP + bmid;
        const int b3 = W3 + NP * d_b * HP;
        const int c = coord_of<CPL, LPC>(s % LPC, s / LPC);
        const int j = rev ? d - 1 - c : c;
        const bool src = c < d && j < d_a, tgt = c < d && j >= d_a;
        float* w = img + l * lf;
        float* w3 = w + ROWS * HP + nmid;
        const int sh = EXACT ? (s >= DP / 2 ? s - DP / 2 : s) : s;
        if (!EXACT || src) {
            const int w1 = W + (src ? j : 0) * HP;
#pragma unroll
            for (int k = 0; k < HP; ++k) w[sh * HP + k] = src ? B[w1 + k] : 0.f;
        }
        if (!EXACT || tgt) {
            const int tt = tgt ? j - d_a : 0;
            const int ub = (s % LPC) % HP;
            float* row = w3 + sh * RS;
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int k = 0; k < HP; ++k) {
                    const int kk = DIST ? (ub ^ unit_xor<HP>(k)) : k;
                    const float v = B[W3 + (tt * NP + q) * HP + kk];
                    row[q * HP + k] = tgt ? v : 0.f;
                }
                const float bq = B[b3 + tt * NP + q];
                row[NP * HP + q] = tgt ? bq : 0.f;
            }
            row[NP * HP + NP] = tgt ? 1.f : 0.f;
        }
    }
    __device__ static __forceinline__ SlotRows load_slot(const NfmcRealNVP& f, int l, int s, int bmid) {
        // 32-bit offsets from the (wave-uniform) blob base: scalar base + one VGPR offset per load instead of a 64-bit
        // address pair per load (the hoisted loads of a thread would otherwise cost ~2 VGPRs each)
        const int d = f.d, d_a = d / 2, d_b = d - d_a;
        const bool rev = (l & 1) == 0;
        const float* __restrict__ B = f.weights;
        const int W = l * (int)f.layer_stride;
        const int W3 = W + d_a * HP + bmid;
        const int b3 = W3 + 2 * d_b * HP;
        const int c = coord_of<CPL, LPC>(s % LPC, s / LPC);
        const int j = rev ? d - 1 - c : c;
        SlotRows o;
        o.src = c < d && j < d_a;
        o.tgt = c < d && j >= d_a;
        const int w1 = W + (o.src ? j : 0) * HP;
        const int tt = o.tgt ? j - d_a : 0;
        const int wa = W3 + tt * HP;
        const int wb = W3 + (d_b + tt) * HP;
        const int ub = (s % LPC) % HP;                       // the slot's lane class
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            const int kk = DIST ? (ub ^ unit_xor<HP>(k)) : k;
            const float a1 = B[w1 + k], aa = B[wa + kk], ab = B[wb + kk];   // always in bounds: predication below, not on the loads
            o.r1[k] = o.src ? a1 : 0.f;
            o.r3[2 * k] = o.tgt ? aa : 0.f;
            o.r3[2 * k + 1] = o.tgt ? ab : 0.f;
        }
        const float ba = B[b3 + tt], bb = B[b3 + d_b + tt];
        o.r3[2 * HP] = o.tgt ? ba : 0.f;
        o.r3[2 * HP + 1] = o.tgt ? bb : 0.f;
        o.r3[2 * HP + 2] = o.tgt ? 1.f : 0.f;
        o.r3[2 * HP + 3] = 0.f;
        return o;
    }
    __device__ static __forceinline__ void store_slot(float* __restrict__ img, const SlotRows& o, int l, int s, int lf, int nmid) {
        float* w = img + l * lf;
        float* w3 = w + ROWS * HP + nmid;
        // EXACT: registers [0, CPL/2) are the first half of the coordinates; row within the half's block
        const int sh = EXACT ? (s >= DP / 2 ? s - DP / 2 : s) : s;
        if (!EXACT || o.src) {
#pragma unroll
            for (int k = 0; k < HP; k += 4)
                *reinterpret_cast<float4*>(w + sh * HP + k) = make_float4(o.r1[k], o.r1[k + 1], o.r1[k + 2], o.r1[k + 3]);
        }
        if (!EXACT || o.tgt) {
#pragma unroll
            for (int k = 0; k < RS; k += 4)
                *reinterpret_cast<float4*>(w3 + sh * RS + k) = make_float4(o.r3[k], o.r3[k + 1], o.r3[k + 2], o.r3[k + 3]);
        }
    }
    __device__ static __forceinline__ void store_ea(float* __restrict__ ea, int s, float l0, float h0, float l1, float h1) {
        ea[s] = fast_exp(l0);
        ea[DP + s] = h0;
        ea[2 * DP + s] = fast_exp(-l0);
        ea[3 * DP + s] = l0;
        ea[4 * DP + s] = fast_exp(l1);
        ea[5 * DP + s] = h1;
        ea[6 * DP + s] = fast_exp(-l1);
        ea[7 * DP + s] = l1;
    }
    // element t of a layer's middle part (b1 and the hidden layers after the first): offset in the layer's blob, -1 = zero
    __device__ static __forceinline__ int mid_src(int d_a, int t) {
        const int m0 = d_a * HP;   // b1 | [WhT | bh] ...
        if (!DIST || t < HP) return m0 + t;
        const int hl = (t - HP) / HL, e = (t - HP) % HL;
        const int sw = m0 + HP + hl * (HP * HP + HP);   // WhT[in][out] | bh
        const int u = e / HROW, r = e % HROW;
        return r < HP ? sw + (u ^ unit_xor<HP>(r)) * HP + u : (r == HP ? sw + HP * HP + u : -1);
    }

    // All `nthreads` threads of the workgroup; blob layout: flow_device.hpp (W1T | b1 | [WhT | bh] | W3 | b3).
    // The image is a permutation of a few KB that sit in L2, so building it costs global-load LATENCY, not bandwidth:
    // the first version walked layers, middle parts and the elementwise-affine vectors in separate loops, i.e. 5-7
    // dependent global round trips (~2 us of an 11 us one-tile launch at d = 64, 6-12 us at d = 256).  Here every
    // thread first ISSUES the loads of everything it will store -- one slot row, one middle element, one slot of
    // elementwise-affine parameters; addresses clamped instead of branching, so they form one basic block with one
    // wait -- and then stores; whatever exceeds that (DP = 512, deep conditioners) follows in plain loops.
    __device__ static void stage(float* __restrict__ img, const NfmcRealNVP& f, int nthreads) {
        const int d = f.d, d_a = d / 2, n_hl = f.n_hidden_layers;
        const int lf = layer_floats(n_hl);
        const int nmid = mid_floats(n_hl);                        // image
        const int bmid = HP + (n_hl - 1) * (HP * HP + HP);        // blob
        const int t = threadIdx.x;
        const int nA = f.n_coupling * DP, nB = f.n_coupling * nmid;
        const bool revl = (f.n_coupling & 1) != 0;
        if constexpr (NB > 0) {   // spline couplings: plain loops (the rows are too long to hold one per thread in registers)
            for (int a = t; a < nA; a += nthreads) stage_slot_rqs(img, f, a / DP, a % DP, bmid, lf, nmid);
            for (int b = t; b < nB; b += nthreads) {
                const int q = mid_src(d_a, b % nmid);
                img[(b / nmid) * lf + ROWS * HP + b % nmid] = q < 0 ? 0.f : f.weights[(b / nmid) * (int)f.layer_stride + q];
            }
            float* ea = img + f.n_coupling * lf;
            for (int s = t; s < DP; s += nthreads) {
                const int p = coord_of<CPL, LPC>(s % LPC, s / LPC);
                const bool ok = p < d;
                const int c = revl ? d - 1 - p : p;
                store_ea(ea, s, ok ? f.ea0_log_scale[p] : 0.f, ok ? f.ea0_shift[p] : 0.f, ok ? f.ea1_log_scale[c] : 0.f,
                         ok ? f.ea1_shift[c] : 0.f);
            }
            return;
        }
        // ---- loads
        const int a0 = t < nA ? t : 0;
        const SlotRows ra = load_slot(f, a0 / DP, a0 % DP, bmid);
        const int b0 = t < nB ? t : 0;
        const int mo = mid_src(d_a, b0 % nmid);
        const float mv = f.weights[(b0 / nmid) * (int)f.layer_stride + (mo < 0 ? 0 : mo)];
        const int s0 = t < DP ? t : 0;
        const int p0 = coord_of<CPL, LPC>(s0 % LPC, s0 / LPC);
        const bool ok0 = p0 < d;
        const int pc = ok0 ? p0 : 0, cc = revl ? d - 1 - pc : pc;   // logical latent coordinate held at position p
        const float ls0 = f.ea0_log_scale[pc], sh0 = f.ea0_shift[pc], ls1 = f.ea1_log_scale[cc], sh1 = f.ea1_shift[cc];
        // ---- stores
        if (t < nA) store_slot(img, ra, a0 / DP, a0 % DP, lf, nmid);
        if (t < nB) img[(b0 / nmid) * lf + ROWS * HP + b0 % nmid] = mo < 0 ? 0.f : mv;
        float* ea = img + f.n_coupling * lf;
        if (t < DP) {
            store_ea(ea, s0, ok0 ? ls0 : 0.f, ok0 ? sh0 : 0.f, ok0 ? ls1 : 0.f, ok0 ? sh1 : 0.f);
        }
        // ---- the rest (more than one slot row / one middle element / one slot per thread: d > 128 or deep conditioners)
        for (int a = t + nthreads; a < nA; a += nthreads) store_slot(img, load_slot(f, a / DP, a % DP, bmid), a / DP, a % DP, lf, nmid);
        for (int b = t + nthreads; b < nB; b += nthreads) {
            const int q = mid_src(d_a, b % nmid);
            img[(b / nmid) * lf + ROWS * HP + b % nmid] = q < 0 ? 0.f : f.weights[(b / nmid) * (int)f.layer_stride + q];
        }
        for (int s = t + nthreads; s < DP; s += nthreads) {
            const int p = coord_of<CPL, LPC>(s % LPC, s / LPC);
            const bool ok = p < d;
            const int c = revl ? d - 1 - p : p;
            store_ea(ea, s, ok ? f.ea0_log_scale[p] : 0.f, ok ? f.ea0_shift[p] : 0.f, ok ? f.ea1_log_scale[c] : 0.f,
                     ok ? f.ea1_shift[c] : 0.f);
        }
    }
};

// Per-lane flow evaluator.  x[] always holds PHYSICAL positions (position p = coordinate p of x-space;
// on the latent side position p holds logical coordinate (odd #reversals ? d-1-p : p), as in flow_device.hpp).
// LEAN: scheduling fences between coordinates keep the weight loads from being hoisted en bloc, so the flow
// code adds few live registers to a kernel whose hot loop is something else (jump tail of the samplers).
// EXACT: d == CPL * LPC with CPL >= 8, so the first CPL/2 registers of every lane are the first half of the
// coordinates and the rest the second half: a layer's sources and targets are whole register ranges, known at
// compile time per layer parity, and only those rows of the image are read (half the LDS traffic, half the
// multiply-adds, and far fewer live registers than the generic path, which reads a zero row instead).
// N floats (N % 4 == 0) from a 16-byte aligned LDS row as ds_read_b128.  Left to itself the compiler reads the
// image rows with ds_read2_b32 (it cannot see the alignment): 4x the LDS instructions, and at a row stride of
// 12 or 20 floats every one of them is a 4-way bank conflict -- the register flow kernels were LDS-bound on it.
template <int N>
__device__ __forceinline__ void load_row16(float (&w)[N], const float* __restrict__ row) {
    static_assert(N % 4 == 0, "rows are whole 16-byte groups");
#pragma unroll
    for (int k = 0; k < N; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(__builtin_assume_aligned(row + k, 16));
        w[k] = v.x;
        w[k + 1] = v.y;
        w[k + 2] = v.z;
        w[k + 3] = v.w;
    }
}

// Two multiply-adds per instruction (v_pk_fma_f32, 1.25x the issue cost of one v_fma): (a.x, a.y) * b + (c.x, c.y).
// The operands are register PAIRS as the 128-bit LDS reads deliver them, so no moves are needed to form them: the
// image interleaves what is consumed together (hidden units k, k+1 of one source coordinate; the alpha and beta
// weights of one target coordinate).  Left to itself the compiler packs across two target coordinates and spends the
// gain on v_mov (20 % of the output-layer instructions).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, float b, f2 c) { return __builtin_elementwise_fma(a, (f2){b, b}, c); }

template <int CPL, int LPC, int HP, bool LEAN = false, bool EXACT = false, int NB = 0>
struct FlowB {
    using Img = FlowImage<CPL, LPC, HP, EXACT, NB>;
    static constexpr int DP = CPL * LPC;
    static constexpr bool kExact = EXACT;
    static_assert(!EXACT || CPL >= 8, "EXACT needs whole register quads per half");
    const float* img;  // LDS
    int n_hl, n_coupling, lf, g;
    float m, log1m, ea_ls, bound;

    __device__ __forceinline__ void init(const float* lds_img, const NfmcRealNVP& f, int g_) {
        img = lds_img;
        n_hl = f.n_hidden_layers;
        n_coupling = f.n_coupling;
        lf = Img::layer_floats(n_hl);
        m = f.min_scale;
        log1m = __logf(1.f - f.min_scale);
        bound = f.spline_bound;
        g = g_;
        // this lane's share of the (state-independent) log-determinant of the two ElementwiseAffine layers
        const float* ea = img + n_coupling * lf + g;
        ea_ls = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) ea_ls += ea[3 * DP + i * LPC] + ea[7 * DP + i * LPC];
    }

    // Make the image pointer opaque to the optimiser.  The weights are loop-invariant across the chain tiles of
    // a kernel, so LICM would otherwise hoist every weight load above the samplers' hot loop and keep ~100
    // values live in VGPRs through it (measured: 91 -> 208 VGPRs, occupancy 5 -> 2).
    __device__ __forceinline__ void launder() { asm volatile("" : "+v"(img)); }

    // one coupling layer in place; INVERSE: x_b = (z_b - beta)/alpha.  Returns this lane's share of the logdet.
    // REV: the layer sees logical coordinate j at physical position d-1-j (even layers); only used when EXACT.
    template <bool INVERSE, bool REV>
    __device__ __forceinline__ float coupling_impl(float (&x)[CPL], int l) const {
        constexpr int S0 = EXACT ? (REV ? CPL / 2 : 0) : 0, S1 = EXACT ? S0 + CPL / 2 : CPL;       // source registers
        constexpr int T0 = EXACT ? (REV ? 0 : CPL / 2) : 0, T1 = EXACT ? T0 + CPL / 2 : CPL;       // target registers
        const float* W1 = img + l * lf + g * HP;
        // hidden-stack weights come from the LDS image (one row per lane class when distributed, else wave-uniform
        // broadcast reads).  Through the scalar cache -- constant address space, s_load, SGPR operands -- the
        // redundant form cost 24 (HP = 4) / 80 (HP = 8) SGPRs per layer and measured 3-5 % slower at both widths.
        const float* b1 = img + l * lf + Img::ROWS * HP;
        float h[HP];
        {
            f2 hh[HP / 2];
#pragma unroll
            for (int k = 0; k < HP / 2; ++k) hh[k] = (f2){0.f, 0.f};
#pragma unroll
            for (int i = S0; i < S1; ++i) {  // generic path: zero rows for coordinates that are not sources of this layer
                float w[HP];
                load_row16<HP>(w, W1 + (i - S0) * LPC * HP);
#pragma unroll
                for (int k = 0; k < HP / 2; ++k) hh[k] = pk_fma((f2){w[2 * k], w[2 * k + 1]}, x[i], hh[k]);
                if constexpr (LEAN) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < HP / 2; ++k) {
                h[2 * k] = hh[k].x;
                h[2 * k + 1] = hh[k].y;
            }
        }
        if constexpr (Img::DIST) {
            const int ub = g & (HP - 1);
            float v = fast_tanh(group_reduce_scatter<HP, LPC>(h) + b1[ub]);
            group_all_gather<HP>(v, h);
            const float* Wh = b1 + HP + ub * Img::HROW;
            for (int hl = 1; hl < n_hl; ++hl) {
                float wr[Img::HROW];
                load_row16<Img::HROW>(wr, Wh);
                float t = wr[HP];
#pragma unroll
                for (int k = 0; k < HP; ++k) t = fmaf(wr[k], h[k], t);
                v = fast_tanh(t);
                group_all_gather<HP>(v, h);
                Wh += Img::HL;
            }
        } else {
            {
                float bb[HP];
                load_row16<HP>(bb, b1);
#pragma unroll
                for (int k = 0; k < HP; ++k) h[k] = fast_tanh(group_allreduce<LPC>(h[k]) + bb[k]);
            }
            const float* Wh = b1 + HP;
            for (int hl = 1; hl < n_hl; ++hl) {
                float t[HP];
                const float* bh = Wh + HP * HP;
                load_row16<HP>(t, bh);
#pragma unroll
                for (int i = 0; i < HP; ++i) {
                    float wr[HP];
                    load_row16<HP>(wr, Wh + i * HP);
#pragma unroll
                    for (int k = 0; k < HP; ++k) t[k] = fmaf(wr[k], h[i], t[k]);
                }
#pragma unroll
                for (int k = 0; k < HP; ++k) h[k] = fast_tanh(t[k]);
                Wh = bh + HP;
            }
        }
        const float* W3 = img + l * lf + Img::ROWS * HP + Img::mid_floats(n_hl) + g * Img::RS;
        float ld = 0.f;
        if constexpr (NB > 0) {
            // spline coupling: this lane evaluates the 3 NB - 1 conditioner outputs of each of its target coordinates
            // (weights in 16-byte LDS reads, one output at a time) and the monotone spline itself (rqs_coordinate)
            constexpr int NP = Img::NP;
            for (int i = T0; i < T1; ++i) {   // not unrolled: one coordinate's 23 outputs + the spline are ~350 instructions
                const float* row = W3 + (i - T0) * LPC * Img::RS;
                float raw[NP];
                {
                    float bb[NP + 1];
                    load_row16<NP + 1>(bb, row + NP * HP);
#pragma unroll
                    for (int q = 0; q < NP; ++q) raw[q] = bb[q];
                    if (!EXACT && bb[NP] == 0.f) continue;   // not a target of this layer (generic path): bitwise unchanged
                }
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    float w[HP];
                    load_row16<HP>(w, row + q * HP);
#pragma unroll
                    for (int k = 0; k < HP; ++k) raw[q] = fmaf(w[k], h[k], raw[q]);
                }
                float xi = x[0];
#pragma unroll
                for (int r = 1; r < CPL; ++r) xi = (r == i) ? x[r] : xi;   // register i without a dynamically indexed array
                const float yi = rqs_coordinate<INVERSE>(xi, raw, bound, ld);
#pragma unroll
                for (int r = 0; r < CPL; ++r) x[r] = (r == i) ? yi : x[r];
            }
            return INVERSE ? -ld : ld;
        }
#pragma unroll
        for (int i = T0; i < T1; ++i) {
            float w[Img::RS];
            load_row16<Img::RS>(w, W3 + (i - T0) * LPC * Img::RS);
            f2 uab = {w[2 * HP], w[2 * HP + 1]};   // (alpha, beta) pre-activations, advanced together
#pragma unroll
            for (int k = 0; k < HP; ++k) uab = pk_fma((f2){w[2 * k], w[2 * k + 1]}, h[k], uab);
            const float ua = uab.x, ub = uab.y;
            // generic path: pass-through and padding coordinates (flag 0) stay bitwise unchanged
            const float alpha = (EXACT || w[2 * HP + 2] != 0.f) ? fast_exp(fmaf(0.5f, ua, log1m)) + m : 1.f;
            const float beta = 0.5f * ub;
            ld += fast_ln(alpha);
            x[i] = INVERSE ? (x[i] - beta) * __builtin_amdgcn_rcpf(alpha) : fmaf(alpha, x[i], beta);
            if constexpr (LEAN || EXACT) __builtin_amdgcn_sched_barrier(0);
        }
        return INVERSE ? -ld : ld;
    }

    template <bool INVERSE>
    __device__ __forceinline__ float coupling(float (&x)[CPL], int l) const {
        if constexpr (EXACT) {
            return (l & 1) == 0 ? coupling_impl<INVERSE, true>(x, l) : coupling_impl<INVERSE, false>(x, l);
        } else {
            return coupling_impl<INVERSE, false>(x, l);
        }
    }

    // ---- TWO chains per lane group (exact-fit, distributed hidden stack only): the same lanes carry the same
    // coordinates of chains A and B, so every weight row read from LDS feeds both.  At d = 256 a coupling layer reads
    // 14 KB of image per chain and the CU's 128 B/clk of LDS bandwidth, not VALU issue, bounded the one-chain kernel
    // (C5 jump: ~30 us of LDS reads next to ~27 us of VALU issue); sharing the rows halves the reads at no extra
    // instruction.  Per chain the arithmetic is the one-chain code's, operation for operation (bitwise equal results).
    template <bool INVERSE, bool REV>
    __device__ __forceinline__ void coupling2_impl(float (&xa)[CPL], float (&xb)[CPL], int l, float& lda, float& ldb) const {
        static_assert(EXACT && Img::DIST, "two-chain evaluator: exact-fit layouts with a distributed hidden stack");
        constexpr int S0 = REV ? CPL / 2 : 0, S1 = S0 + CPL / 2;
        constexpr int T0 = REV ? 0 : CPL / 2, T1 = T0 + CPL / 2;
        const float* W1 = img + l * lf + g * HP;
        const float* b1 = img + l * lf + Img::ROWS * HP;
        float ha[HP], hb[HP];
        {
            f2 pa[HP / 2], pb[HP / 2];
#pragma unroll
            for (int k = 0; k < HP / 2; ++k) pa[k] = pb[k] = (f2){0.f, 0.f};
#pragma unroll
            for (int i = S0; i < S1; ++i) {
                float w[HP];
                load_row16<HP>(w, W1 + (i - S0) * LPC * HP);
#pragma unroll
                for (int k = 0; k < HP / 2; ++k) {
                    pa[k] = pk_fma((f2){w[2 * k], w[2 * k + 1]}, xa[i], pa[k]);
                    pb[k] = pk_fma((f2){w[2 * k], w[2 * k + 1]}, xb[i], pb[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < HP / 2; ++k) {
                ha[2 * k] = pa[k].x;
                ha[2 * k + 1] = pa[k].y;
                hb[2 * k] = pb[k].x;
                hb[2 * k + 1] = pb[k].y;
            }
        }
        {
            const int ub = g & (HP - 1);
            const float bias = b1[ub];
            float va = fast_tanh(group_reduce_scatter<HP, LPC>(ha) + bias);
            float vb = fast_tanh(group_reduce_scatter<HP, LPC>(hb) + bias);
            group_all_gather<HP>(va, ha);
            group_all_gather<HP>(vb, hb);
            const float* Wh = b1 + HP + ub * Img::HROW;
            for (int hl = 1; hl < n_hl; ++hl) {
                float wr[Img::HROW];
                load_row16<Img::HROW>(wr, Wh);
                float ta = wr[HP], tb = wr[HP];
#pragma unroll
                for (int k = 0; k < HP; ++k) {
                    ta = fmaf(wr[k], ha[k], ta);
                    tb = fmaf(wr[k], hb[k], tb);
                }
                va = fast_tanh(ta);
                vb = fast_tanh(tb);
                group_all_gather<HP>(va, ha);
                group_all_gather<HP>(vb, hb);
                Wh += Img::HL;
            }
        }
        const float* W3 = img + l * lf + Img::ROWS * HP + Img::mid_floats(n_hl) + g * Img::RS;
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int i = T0; i < T1; ++i) {
            float w[Img::RS];
            load_row16<Img::RS>(w, W3 + (i - T0) * LPC * Img::RS);
            f2 ua = {w[2 * HP], w[2 * HP + 1]}, ub = ua;
#pragma unroll
            for (int k = 0; k < HP; ++k) {
                ua = pk_fma((f2){w[2 * k], w[2 * k + 1]}, ha[k], ua);
                ub = pk_fma((f2){w[2 * k], w[2 * k + 1]}, hb[k], ub);
            }
            const float alpha_a = fast_exp(fmaf(0.5f, ua.x, log1m)) + m, alpha_b = fast_exp(fmaf(0.5f, ub.x, log1m)) + m;
            const float beta_a = 0.5f * ua.y, beta_b = 0.5f * ub.y;
            sa += fast_ln(alpha_a);
            sb += fast_ln(alpha_b);
            xa[i] = INVERSE ? (xa[i] - beta_a) * __builtin_amdgcn_rcpf(alpha_a) : fmaf(alpha_a, xa[i], beta_a);
            xb[i] = INVERSE ? (xb[i] - beta_b) * __builtin_amdgcn_rcpf(alpha_b) : fmaf(alpha_b, xb[i], beta_b);
            __builtin_amdgcn_sched_barrier(0);
        }
        lda += INVERSE ? -sa : sa;
        ldb += INVERSE ? -sb : sb;
    }
    template <bool INVERSE>
    __device__ __forceinline__ void coupling2(float (&xa)[CPL], float (&xb)[CPL], int l, float& lda, float& ldb) const {
        if ((l & 1) == 0) coupling2_impl<INVERSE, true>(xa, xb, l, lda, ldb);
        else coupling2_impl<INVERSE, false>(xa, xb, l, lda, ldb);
    }
    __device__ __forceinline__ void forward2(float (&xa)[CPL], float (&xb)[CPL], float& lda, float& ldb) const {
        const float* ea = img + n_coupling * lf + g;
        lda = ldb = ea_ls;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const float e = ea[i * LPC], sh = ea[DP + i * LPC];
            xa[i] = fmaf(e, xa[i], sh);
            xb[i] = fmaf(e, xb[i], sh);
        }
        for (int l = 0; l < n_coupling; ++l) coupling2<false>(xa, xb, l, lda, ldb);
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const float e = ea[4 * DP + i * LPC], sh = ea[5 * DP + i * LPC];
            xa[i] = fmaf(e, xa[i], sh);
            xb[i] = fmaf(e, xb[i], sh);
        }
    }
    __device__ __forceinline__ void inverse2(float (&xa)[CPL], float (&xb)[CPL], float& lda, float& ldb) const {
        const float* ea = img + n_coupling * lf + g;
        lda = ldb = -ea_ls;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const float sh = ea[5 * DP + i * LPC], ei = ea[6 * DP + i * LPC];
            xa[i] = (xa[i] - sh) * ei;
            xb[i] = (xb[i] - sh) * ei;
        }
        for (int l = n_coupling - 1; l >= 0; --l) coupling2<true>(xa, xb, l, lda, ldb);
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const float sh = ea[DP + i * LPC], ei = ea[2 * DP + i * LPC];
            xa[i] = (xa[i] - sh) * ei;
            xb[i] = (xb[i] - sh) * ei;
        }
    }

    // x -> z (z left in physical positions); returns this lane's share of logdet_forward.  The ElementwiseAffine
    // planes of the image carry e^ls and e^-ls (no transcendental per coordinate) and the lane's share of sum(ls) is a
    // constant of the flow (ea_ls, summed once in init()).
    __device__ __forceinline__ float forward(float (&x)[CPL]) const {
        const float* ea = img + n_coupling * lf + g;
        float ld = ea_ls;
#pragma unroll
        for (int i = 0; i < CPL; ++i) x[i] = fmaf(ea[i * LPC], x[i], ea[DP + i * LPC]);
        for (int l = 0; l < n_coupling; ++l) ld += coupling<false>(x, l);
#pragma unroll
        for (int i = 0; i < CPL; ++i) x[i] = fmaf(ea[4 * DP + i * LPC], x[i], ea[5 * DP + i * LPC]);
        return ld;
    }

    // z (physical positions) -> x; returns this lane's share of logdet_inverse
    __device__ __forceinline__ float inverse(float (&x)[CPL]) const {
        const float* ea = img + n_coupling * lf + g;
        float ld = -ea_ls;
#pragma unroll
        for (int i = 0; i < CPL; ++i) x[i] = (x[i] - ea[5 * DP + i * LPC]) * ea[6 * DP + i * LPC];
        for (int l = n_coupling - 1; l >= 0; --l) ld += coupling<true>(x, l);
#pragma unroll
        for (int i = 0; i < CPL; ++i) x[i] = (x[i] - ea[DP + i * LPC]) * ea[2 * DP + i * LPC];
        return ld;
    }
};

// latent z ~ N(0, I) for this lane's positions: position p holds logical latent coordinate (revl ? d-1-p : p).
// `replay`: NULL -> native stream 2, else the (n, d) latents of this transition.
// ALIGNED4: d % 4 == 0 is known at compile time (exact-fit layouts), so the per-coordinate branch is not emitted.
template <int CPL, int LPC, bool ALIGNED4 = false, int R = 10>
__device__ __forceinline__ void draw_latent(float (&z)[CPL], const float* __restrict__ replay, uint64_t seed,
                                            uint32_t step, uint32_t gchain, int64_t row, int64_t n, int d, int g,
                                            bool revl) {
    if (replay) {
        const float* src = replay + row * d;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int p = coord_of<CPL, LPC>(g, i);
            z[i] = (row < n && p < d) ? src[revl ? d - 1 - p : p] : 0.f;
        }
        return;
    }
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (ALIGNED4 || !revl || (d & 3) == 0) {
        // a register quad is one Philox block (reversed flows with d % 4 == 0: block d/4-1-b, elements reversed)
#pragma unroll
        for (int q = 0; q < CPL / 4; ++q) {
            const int b = q * LPC + g;
            float w[4];
            philox_normal4<R>(gchain, step, (uint32_t)(revl ? (d >> 2) - 1 - b : b), kTagLatent, k0, k1, w);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[4 * q + r] = (4 * b + r < d) ? (revl ? w[3 - r] : w[r]) : 0.f;
        }
    } else {
        // reversed AND ragged: the quad straddles two blocks; rare, so one (low-register) call per coordinate
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = d - 1 - coord_of<CPL, LPC>(g, i);
            float v = 0.f;
            if (c >= 0) {
                float w[4];
                philox_normal4<R>(gchain, step, (uint32_t)(c >> 2), kTagLatent, k0, k1, w);
                const int e = c & 3;
                v = e == 0 ? w[0] : (e == 1 ? w[1] : (e == 2 ? w[2] : w[3]));
            }
            z[i] = v;
        }
    }
}

}  // namespace nfmc
