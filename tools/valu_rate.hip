// valu_rate.hip -- instruction-rate microbenchmark (gfx950): sustained lane-ops/s of single
// VALU instructions at 8 waves per SIMD.  Development tool; results are recorded in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define KERNEL(NAME, ASM4)                                                                          \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, int iters, uint32_t seed)            \
    {                                                                                               \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,      \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;                                          \
        for (int i = 0; i < iters; i++) {                                                           \
            _Pragma("unroll") for (int k = 0; k < 8; k++)                                           \
            {                                                                                       \
                asm volatile(ASM4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),     \
                             "+v"(a6), "+v"(a7));                                                   \
            }                                                                                       \
        }                                                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;         \
    }

#define FOUR2(op) op " %0, %0, %1\n" op " %2, %2, %3\n" op " %4, %4, %5\n" op " %6, %6, %7"
#define FOUR3(op) op " %0, %0, %1, %2\n" op " %3, %3, %4, %5\n" op " %6, %6, %7, %0\n" op " %1, %1, %2, %3"

KERNEL(k_xor, FOUR2("v_xor_b32"))
KERNEL(k_xor64, FOUR2("v_xor_b32_e64"))
KERNEL(k_add, FOUR2("v_add_u32"))
KERNEL(k_sub, FOUR2("v_sub_u32"))
KERNEL(k_and, FOUR2("v_and_b32"))
KERNEL(k_lshl, FOUR2("v_lshlrev_b32"))
KERNEL(k_min, FOUR2("v_min_u32"))
KERNEL(k_max_i32, FOUR2("v_max_i32"))
KERNEL(k_min_u16, FOUR2("v_min_u16"))
KERNEL(k_sub_u16, FOUR2("v_sub_u16"))
KERNEL(k_addf, FOUR2("v_add_f32"))
KERNEL(k_mulf, FOUR2("v_mul_f32"))
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %2, %3, %2\n v_bcnt_u32_b32 %4, %5, %4\n v_bcnt_u32_b32 %6, %7, %6")
KERNEL(k_add3, FOUR3("v_add3_u32"))
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1\n v_lshl_or_b32 %2, %2, 3, %3\n v_lshl_or_b32 %4, %4, 3, %5\n v_lshl_or_b32 %6, %6, 3, %7")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %4, %4, 8, 8\n v_bfe_u32 %6, %6, 8, 8")
KERNEL(k_perm, FOUR3("v_perm_b32"))
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1\n v_alignbyte_b32 %2, %2, %3, 1\n v_alignbyte_b32 %4, %4, %5, 1\n v_alignbyte_b32 %6, %6, %7, 1")
KERNEL(k_mad_u24, FOUR3("v_mad_u32_u24"))
KERNEL(k_sad_u8, FOUR3("v_sad_u8"))
KERNEL(k_fma, FOUR3("v_fma_f32"))
KERNEL(k_med3, FOUR3("v_med3_i32"))
KERNEL(k_max3, FOUR3("v_max3_u32"))
KERNEL(k_pk_sub, "v_pk_sub_u16 %0, %0, %1 clamp\n v_pk_sub_u16 %2, %2, %3 clamp\n v_pk_sub_u16 %4, %4, %5 clamp\n v_pk_sub_u16 %6, %6, %7 clamp")
KERNEL(k_pk_min, FOUR2("v_pk_min_u16"))
KERNEL(k_pk_mad, FOUR3("v_pk_mad_u16"))
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc")
KERNEL(k_cmp_addc, "v_cmp_gt_u32 vcc, %0, %1\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_cmp_gt_u32 vcc, %4, %5\n v_addc_co_u32 %6, vcc, 0, %6, vcc")
KERNEL(k_xor_sdwa, "v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_xor_b32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_xor_b32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_xor_b32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL(k_dot4, FOUR3("v_dot4_u32_u8"))

typedef void (*kern_t)(uint32_t *, int, uint32_t);
static void run(const char *name, kern_t k, uint32_t *d)
{
    const int blocks = 256 * 8, iters = 10000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 32.0;
    printf("%-18s %8.3f ms  %7.2f T lane-ops/s\n", name, ms, ops / (ms * 1e-3) / 1e12);
}

int main()
{
    uint32_t *d;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
#define R(k) run(#k, k, d)
    R(k_xor); R(k_xor64); R(k_add); R(k_sub); R(k_and); R(k_lshl); R(k_min); R(k_max_i32); R(k_min_u16); R(k_sub_u16);
    R(k_addf); R(k_mulf); R(k_bcnt); R(k_add3); R(k_lshl_or); R(k_bfe); R(k_perm); R(k_alignbyte); R(k_mad_u24);
    R(k_sad_u8); R(k_fma); R(k_med3); R(k_max3); R(k_pk_sub); R(k_pk_min); R(k_pk_mad); R(k_cndmask); R(k_cmp_addc);
    R(k_xor_sdwa); R(k_dot4);
    return 0;
}
