// fp32 MFMA rate microbenchmark for gfx950: v_mfma_f32_16x16x4_f32 in a register-only loop, every CU busy.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma.hip -o tools/ubench_mfma.bin && tools/ubench_mfma.bin
// Variants: dependent chain (one accumulator) vs NACC independent accumulators; 1 or 2 waves per SIMD; with and without
// LDS reads of the A operand (one ds_read_b128 per 4 MFMAs, as in the NeuTra kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NACC, bool LDS>
__global__ void __launch_bounds__(512, 2) mfma_loop(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float img[128 * 132];
    for (int i = threadIdx.x; i < 128 * 132; i += blockDim.x) img[i] = seed * (float)(i % 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 b = {seed, seed * 2, seed * 3, seed * 4};
    f32x4 a = {1.f, 2.f, 3.f, 4.f};
    const float* arow = img + (lane & 15) * 132 + 4 * (lane >> 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (LDS) a = *reinterpret_cast<const f32x4*>(arow + 16 * g + (it & 7) * 132 * 16);
#pragma unroll
            for (int k = 0; k < NACC; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[r], acc[k], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int block, int grid) {
    float* out;
    hipMalloc(&out, (size_t)grid * block * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<NACC, LDS>), dim3(grid), dim3(block), 0, 0, out, iters, 1e-3f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = (double)grid * (block / 64) * iters * 8 * NACC * 4;
        if (rep == 2) printf("%-44s block %4d grid %5d: %7.3f ms  %6.1f TFLOP/s\n", name, block, grid, ms, mfmas * 2048 / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    run<1, false>("dependent chain, 2 waves/SIMD", 512, 256);
    run<1, false>("dependent chain, 1 wave/SIMD", 256, 256);
    run<2, false>("2 accumulators, 2 waves/SIMD", 512, 256);
    run<2, false>("2 accumulators, 1 wave/SIMD", 256, 256);
    run<4, false>("4 accumulators, 1 wave/SIMD", 256, 256);
    run<1, true>("dependent chain + LDS A reads, 2 waves/SIMD", 512, 256);
    run<2, true>("2 accumulators + LDS A reads, 2 waves/SIMD", 512, 256);
    run<1, false>("dependent chain, 2 waves/SIMD, 2 rounds", 512, 512);
    return 0;
}
