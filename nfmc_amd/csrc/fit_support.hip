// Small device-side helpers of the flow (re)fit (f1) that replaced strings of torch launches on the refit path
// (round 3: ~0.45 ms of host work per refit at the C5 shape):
//   nfmc_flow_blob_copy_f32   the flow's nn.Parameters <-> the trainable vector (weight-blob layout), ONE launch instead
//                             of one slice copy per parameter tensor;
//   nfmc_rows_sample_f32      the refit buffer's shuffled split (tuning.py:58-65: rows shuffled, cut, capped): the first m
//                             rows of a keyed pseudo-random permutation of the N pooled rows, gathered in ONE launch --
//                             instead of torch.randperm(N) (a sort of N keys) + two index gathers.
#include "common.hpp"

namespace nfmc {

struct BlobPieces {
    NfmcBlobPiece p[NFMC_BLOB_MAX_PIECES];
    int64_t first[NFMC_BLOB_MAX_PIECES + 1];   // running element counts
    int n;
};

__global__ void __launch_bounds__(256) blob_copy_kernel(float* __restrict__ vec, BlobPieces b, int to_vector) {
    const int64_t total = b.first[b.n];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int k = 0;
        while (k + 1 < b.n && e >= b.first[k + 1]) ++k;
        const NfmcBlobPiece& p = b.p[k];
        const int64_t local = e - b.first[k];
        const int64_t r = local / p.cols, c = local - r * p.cols;
        float* v = vec + p.vec_off + r * p.vec_row_stride + c * p.vec_col_stride;
        if (to_vector) *v = p.param[local];
        else p.param[local] = *v;
    }
}

// ---- keyed pseudo-random permutation of [0, N): balanced Feistel network on 2 * half bits + cycle walking
__host__ __device__ inline uint32_t prp_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

struct PrpKeys {
    uint32_t k[NFMC_PRP_ROUNDS];
    int half;
};

__host__ __device__ inline uint64_t prp_index(uint64_t i, uint64_t n, const PrpKeys& key) {
    const uint32_t mask = (1u << key.half) - 1u;
    do {
        uint32_t L = (uint32_t)(i >> key.half), R = (uint32_t)(i & mask);
#pragma unroll
        for (int r = 0; r < NFMC_PRP_ROUNDS; ++r) {
            const uint32_t t = L ^ (prp_mix32(R + key.k[r]) & mask);
            L = R;
            R = t;
        }
        i = ((uint64_t)L << key.half) | R;
    } while (i >= n);
    return i;
}

static PrpKeys prp_keys(uint64_t n, uint64_t seed) {
    PrpKeys key;
    int bits = 2;
    while (bits < 62 && ((uint64_t)1 << bits) < n) ++bits;
    key.half = (bits + 1) / 2;
    uint64_t s = seed;
    for (int r = 0; r < NFMC_PRP_ROUNDS; ++r) {   // splitmix64
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        key.k[r] = (uint32_t)z;
    }
    return key;
}

// one wave per output row
__global__ void __launch_bounds__(256) rows_sample_kernel(const float* __restrict__ x, int64_t n, int d, PrpKeys key,
                                                          int64_t first, float* __restrict__ out, int64_t m,
                                                          int64_t* __restrict__ index_out) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < m; i += (int64_t)gridDim.x * 4) {
        const int64_t src = (int64_t)prp_index((uint64_t)(first + i), (uint64_t)n, key);
        const float* s = x + src * d;
        float* o = out + i * d;
        for (int c = lane; c < d; c += 64) o[c] = s[c];
        if (index_out && lane == 0) index_out[i] = src;
    }
}

}  // namespace nfmc

using namespace nfmc;

extern "C" int nfmc_flow_blob_copy_f32(float* vec, const NfmcBlobPiece* pieces, int32_t n_pieces, int32_t to_vector,
                                       nfmc_stream_t stream) {
    if (!vec || !pieces || n_pieces < 0) return NFMC_EINVAL;
    for (int base = 0; base < n_pieces; base += NFMC_BLOB_MAX_PIECES) {
        BlobPieces b;
        b.n = n_pieces - base < NFMC_BLOB_MAX_PIECES ? n_pieces - base : NFMC_BLOB_MAX_PIECES;
        b.first[0] = 0;
        for (int k = 0; k < b.n; ++k) {
            const NfmcBlobPiece& p = pieces[base + k];
            if (p.rows < 0 || p.cols < 0 || p.vec_off < 0 || (!p.param && (int64_t)p.rows * p.cols > 0)) return NFMC_EINVAL;
            b.p[k] = p;
            b.first[k + 1] = b.first[k] + (int64_t)p.rows * p.cols;
        }
        if (b.first[b.n] == 0) continue;
        const int64_t blocks = (b.first[b.n] + 255) / 256;
        hipLaunchKernelGGL(blob_copy_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream,
                           vec, b, to_vector);
    }
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

extern "C" int nfmc_rows_sample_f32(const float* x, int64_t n, int32_t d, uint64_t seed, int64_t first, float* out, int64_t m,
                                    int64_t* index_out, nfmc_stream_t stream) {
    if (!x || !out || n <= 0 || d <= 0 || m < 0 || first < 0 || first + m > n) return NFMC_EINVAL;
    if (m == 0) return NFMC_OK;
    const PrpKeys key = prp_keys((uint64_t)n, seed);
    const int64_t blocks = (m + 3) / 4;
    hipLaunchKernelGGL(rows_sample_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, x, n,
                       d, key, first, out, m, index_out);
    NFMC_HIP_CHECK_LAUNCH();
    return NFMC_OK;
}

extern "C" int64_t nfmc_rows_sample_index(int64_t n, uint64_t seed, int64_t i) {
    if (n <= 0 || i < 0 || i >= n) return -1;
    const PrpKeys key = prp_keys((uint64_t)n, seed);
    return (int64_t)prp_index((uint64_t)i, (uint64_t)n, key);
}
