// Instruction-cost microbenchmark for gfx950: issue cost (cycles per wave-instruction per SIMD at full
// occupancy) of the integer / transcendental ops the RNG choices depend on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP 256

#define BENCH_KERNEL(NAME, BODY)                                                    \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters) {         \
        uint32_t a = threadIdx.x * 2654435761u + 1, b = blockIdx.x + 7, c = a ^ b, d = a + b; \
        uint32_t e = a * 3, f = b * 5, g = c * 7, h = d * 11;                        \
        uint64_t w = a, x2 = b, y2 = c, z2 = d;                                      \
        float fa = a * 1e-9f + 1.1f, fb = b * 1e-9f + 1.2f, fc = 1.3f, fd = 1.4f;    \
        typedef float f2_ __attribute__((ext_vector_type(2)));                       \
        f2_ pa = {fa, fb}, pb = {fb, fa}, pc = {fc, fd}, pd = {fd, fc};              \
        const uint64_t sm = __ballot(a & 1);                                         \
        for (int it = 0; it < iters; ++it) {                                         \
            _Pragma("unroll") for (int r = 0; r < REP / 4; ++r) { BODY }             \
        }                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)w ^ (uint32_t)x2 ^ (uint32_t)y2 ^ (uint32_t)z2 ^ \
            __float_as_uint(fa) ^ __float_as_uint(fb) ^ __float_as_uint(fc) ^ __float_as_uint(fd) ^ __float_as_uint(pa[0] + pa[1] + pb[0] + pb[1] + pc[0] + pc[1] + pd[0] + pd[1]); \
    }

// four independent chains per body -> REP instructions per iteration
BENCH_KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
BENCH_KERNEL(k_xor, asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_add, asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %0, 7\n v_alignbit_b32 %1, %1, %1, 9\n v_alignbit_b32 %2, %2, %2, 11\n v_alignbit_b32 %3, %3, %3, 13" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_mul_hi, asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_mad64, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %6, %1\n v_mad_u64_u32 %2, vcc, %6, %7, %2\n v_mad_u64_u32 %3, vcc, %7, %4, %3" : "+v"(w), "+v"(x2), "+v"(y2), "+v"(z2) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");)
BENCH_KERNEL(k_mul24, asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_mulhi24, asm volatile("v_mul_hi_u32_u24 %0, %0, %1\n v_mul_hi_u32_u24 %1, %1, %2\n v_mul_hi_u32_u24 %2, %2, %3\n v_mul_hi_u32_u24 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_xad, asm volatile("v_xad_u32 %0, %0, %1, %2\n v_xad_u32 %1, %1, %2, %3\n v_xad_u32 %2, %2, %3, %0\n v_xad_u32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %1, %1, 5, %2\n v_lshl_add_u32 %2, %2, 7, %3\n v_lshl_add_u32 %3, %3, 9, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_log, asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
BENCH_KERNEL(k_exp, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
BENCH_KERNEL(k_sin, asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
BENCH_KERNEL(k_sqrt, asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
BENCH_KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
BENCH_KERNEL(k_cvt_u2f, asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_dpp, asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)

BENCH_KERNEL(k_cndmask_s, asm volatile("v_cndmask_b32 %0, %0, %1, %4\n v_cndmask_b32 %1, %1, %2, %4\n v_cndmask_b32 %2, %2, %3, %4\n v_cndmask_b32 %3, %3, %0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(sm));)
BENCH_KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
BENCH_KERNEL(k_pk_fma, asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));)
BENCH_KERNEL(k_pk_mul, asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));)
BENCH_KERNEL(k_pk_add, asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));)
BENCH_KERNEL(k_and_or, asm volatile("v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %0\n v_and_or_b32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_mul_f32, asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
BENCH_KERNEL(k_add_f32, asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
BENCH_KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_fma4, asm volatile("v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %1, %2, %3, %1\n v_fma_f32 %2, %3, %0, %2\n v_fma_f32 %3, %0, %1, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
BENCH_KERNEL(k_cos, asm volatile("v_cos_f32 %0, %0\n v_cos_f32 %1, %1\n v_cos_f32 %2, %2\n v_cos_f32 %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)
BENCH_KERNEL(k_fmac_e32, asm volatile("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)

BENCH_KERNEL(k_cndmask_e64vcc, asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %3, %3, %0, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_cndmask_ind, asm volatile("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %5, vcc\n v_cndmask_b32_e32 %2, %2, %6, vcc\n v_cndmask_b32_e32 %3, %3, %7, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));)
BENCH_KERNEL(k_cndmask_setvcc, asm volatile("s_mov_b64 vcc, %4\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %1, %1, %2, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %3, %3, %0, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(sm) : "vcc");)

// gfx950's three-input bit op (truth table 0x96 = a ^ b ^ c) against the two VOP2 xors it replaces in a Philox round
BENCH_KERNEL(k_bitop3, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
BENCH_KERNEL(k_bitop3_s, asm volatile("v_bitop3_b32 %0, %0, %1, %4 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %4 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %4 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %4 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"((uint32_t)sm));)
BENCH_KERNEL(k_xor_pair, asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %4, %0\n v_xor_b32 %2, %2, %3\n v_xor_b32 %2, %4, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"((uint32_t)sm));)

struct Ent { const char* name; void (*fn)(uint32_t*, int); };

int main() {
    uint32_t* out;
    const int grid = 256 * 8, iters = 200;
    hipMalloc(&out, grid * 256 * 4);
    Ent ents[] = {{"v_fma_f32", k_fma}, {"v_xor_b32", k_xor}, {"v_add_u32", k_add}, {"v_alignbit_b32", k_alignbit},
                  {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mad_u64_u32", k_mad64},
                  {"v_mul_u32_u24", k_mul24}, {"v_mul_hi_u32_u24", k_mulhi24}, {"v_mad_u32_u24", k_mad24},
                  {"v_xad_u32", k_xad}, {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshl_add}, {"v_log_f32", k_log},
                  {"v_exp_f32", k_exp}, {"v_sin_f32", k_sin}, {"v_sqrt_f32", k_sqrt}, {"v_rcp_f32", k_rcp},
                  {"v_cvt_f32_u32", k_cvt_u2f}, {"v_cndmask_b32", k_cndmask}, {"v_add_f32_dpp", k_dpp},
                  {"v_cndmask(sgpr)", k_cndmask_s}, {"v_bfi_b32", k_bfi}, {"v_pk_fma_f32", k_pk_fma}, {"v_pk_mul_f32", k_pk_mul},
                  {"v_pk_add_f32", k_pk_add}, {"v_and_or_b32", k_and_or}, {"v_mul_f32", k_mul_f32}, {"v_cos_f32", k_cos},
                  {"v_fmac_f32(e32)", k_fmac_e32}, {"cndmask e64 vcc", k_cndmask_e64vcc}, {"cndmask e32 indep", k_cndmask_ind}, {"cndmask e32 vcc set", k_cndmask_setvcc},
                  {"v_bitop3_b32", k_bitop3}, {"v_bitop3_b32(sgpr)", k_bitop3_s}, {"v_xor x2 (dependent pair, sgpr)", k_xor_pair},
                  {"v_add_f32", k_add_f32}, {"v_mov_b32", k_mov}, {"v_fma_f32 (4 distinct regs)", k_fma4}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double base = 0;
    for (auto& e : ents) {
        hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, 10);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // wave-instructions per SIMD: grid*4 waves / 1024 SIMDs * iters * REP
        const double winstr = (double)grid * 4 / 1024.0 * iters * REP;
        const double ns_per = best * 1e6 / winstr;
        if (base == 0) base = ns_per;
        printf("%-18s %8.3f ms  %6.3f ns/wave-instr/SIMD  = %5.2f x v_fma\n", e.name, best, ns_per, ns_per / base);
    }
    return 0;
}
