// valu_rate4.hip -- when does v_xor_b32 keep its full rate on gfx950?  Variants of operand shape
// and neighbours.  Development tool (results in DESIGN.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define KERNEL(NAME, BODY)                                                                          \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, int iters, uint32_t seed)            \
    {                                                                                               \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,      \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, t0 = 1, t1 = 2, t2 = 3, t3 = 4;           \
        for (int i = 0; i < iters; i++) {                                                           \
            _Pragma("unroll") for (int k = 0; k < 8; k++)                                           \
            {                                                                                       \
                asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),     \
                             "+v"(a6), "+v"(a7), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));           \
            }                                                                                       \
        }                                                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ t0 ^ t1 ^ t2 ^ t3; \
    }
// 8 instructions per body in every variant
KERNEL(k_xor_inplace, "v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3\n v_xor_b32 %4, %4, %5\n v_xor_b32 %6, %6, %7\n v_xor_b32 %1, %1, %0\n v_xor_b32 %3, %3, %2\n v_xor_b32 %5, %5, %4\n v_xor_b32 %7, %7, %6")
KERNEL(k_xor_3op, "v_xor_b32 %8, %0, %1\n v_xor_b32 %9, %2, %3\n v_xor_b32 %10, %4, %5\n v_xor_b32 %11, %6, %7\n v_xor_b32 %0, %8, %9\n v_xor_b32 %2, %10, %11\n v_xor_b32 %4, %8, %11\n v_xor_b32 %6, %9, %10")
KERNEL(k_bcnt_only, "v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %2, %3, %2\n v_bcnt_u32_b32 %4, %5, %4\n v_bcnt_u32_b32 %6, %7, %6\n v_bcnt_u32_b32 %8, %1, %8\n v_bcnt_u32_b32 %9, %3, %9\n v_bcnt_u32_b32 %10, %5, %10\n v_bcnt_u32_b32 %11, %7, %11")
KERNEL(k_mix_alt, "v_xor_b32 %8, %0, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_xor_b32 %9, %0, %3\n v_bcnt_u32_b32 %4, %9, %4\n v_xor_b32 %10, %0, %5\n v_bcnt_u32_b32 %6, %10, %6\n v_xor_b32 %11, %0, %7\n v_bcnt_u32_b32 %2, %11, %2")
KERNEL(k_mix_indep, "v_xor_b32 %0, %0, %1\n v_bcnt_u32_b32 %8, %3, %8\n v_xor_b32 %2, %2, %1\n v_bcnt_u32_b32 %9, %5, %9\n v_xor_b32 %4, %4, %1\n v_bcnt_u32_b32 %10, %7, %10\n v_xor_b32 %6, %6, %1\n v_bcnt_u32_b32 %11, %3, %11")
KERNEL(k_mix_grouped, "v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %1\n v_xor_b32 %4, %4, %1\n v_xor_b32 %6, %6, %1\n v_bcnt_u32_b32 %8, %3, %8\n v_bcnt_u32_b32 %9, %5, %9\n v_bcnt_u32_b32 %10, %7, %10\n v_bcnt_u32_b32 %11, %3, %11")
KERNEL(k_and_bcnt, "v_and_b32 %0, %0, %1\n v_bcnt_u32_b32 %8, %3, %8\n v_and_b32 %2, %2, %1\n v_bcnt_u32_b32 %9, %5, %9\n v_and_b32 %4, %4, %1\n v_bcnt_u32_b32 %10, %7, %10\n v_and_b32 %6, %6, %1\n v_bcnt_u32_b32 %11, %3, %11")
KERNEL(k_add_fma, "v_add_f32 %0, %0, %1\n v_fma_f32 %8, %3, %5, %8\n v_add_f32 %2, %2, %1\n v_fma_f32 %9, %5, %7, %9\n v_add_f32 %4, %4, %1\n v_fma_f32 %10, %7, %3, %10\n v_add_f32 %6, %6, %1\n v_fma_f32 %11, %3, %5, %11")
typedef void (*kern_t)(uint32_t *, int, uint32_t);
static void run(const char *name, kern_t k, uint32_t *d)
{
    const int blocks = 256 * 8, iters = 5000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)blocks * 4 * iters * 64.0 / 1024.0; // wave-instructions per SIMD
    printf("%-16s %8.3f ms  %6.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / instr_per_simd);
}
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
#define R(k) run(#k, k, d)
    R(k_xor_inplace); R(k_xor_3op); R(k_bcnt_only); R(k_mix_alt); R(k_mix_indep); R(k_mix_grouped); R(k_and_bcnt); R(k_add_fma);
    return 0;
}
