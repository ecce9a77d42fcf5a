// valu_rate5.hip -- issue rate of the gfx950 vector instructions the round-3 detect / describe rewrites lean on
// (SWAR byte compares with v_bitop3_b32, shifts, v_sad_u8, v_rndne_f32, ...).  8 independent chains per lane,
// 8 waves per SIMD, all CUs.  Prints ns per wave-instruction per SIMD and T lane-ops/s chip-wide.
// Development tool (results in DESIGN.md section 4).
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
// two-source form: dst_k = op(dst_k, t_j)      three-source: dst_k = op(dst_k, t_j, t_j')
#define OP2(op) op " %0, %0, %8\n" op " %1, %1, %9\n" op " %2, %2, %10\n" op " %3, %3, %11\n" op " %4, %4, %8\n" op " %5, %5, %9\n" op " %6, %6, %10\n" op " %7, %7, %11"
#define OP2R(op) op " %0, %8, %0\n" op " %1, %9, %1\n" op " %2, %10, %2\n" op " %3, %11, %3\n" op " %4, %8, %4\n" op " %5, %9, %5\n" op " %6, %10, %6\n" op " %7, %11, %7"
#define OP3(op, sfx) op " %0, %0, %8, %9" sfx "\n" op " %1, %1, %9, %10" sfx "\n" op " %2, %2, %10, %11" sfx "\n" op " %3, %3, %11, %8" sfx "\n" op " %4, %4, %8, %10" sfx "\n" op " %5, %5, %9, %11" sfx "\n" op " %6, %6, %10, %8" sfx "\n" op " %7, %7, %11, %9" sfx
#define OP1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7"
KERNEL(k_add_u32, OP2("v_add_u32"))
KERNEL(k_sub_u32, OP2("v_sub_u32"))
KERNEL(k_and_b32, OP2("v_and_b32"))
KERNEL(k_or_b32, OP2("v_or_b32"))
KERNEL(k_lshrrev, OP2R("v_lshrrev_b32"))
KERNEL(k_lshlrev, OP2R("v_lshlrev_b32"))
KERNEL(k_bitop3, OP3("v_bitop3_b32", " bitop3:0xd8"))
KERNEL(k_and_or, OP3("v_and_or_b32", ""))
KERNEL(k_or3, OP3("v_or3_b32", ""))
KERNEL(k_bfi, OP3("v_bfi_b32", ""))
KERNEL(k_lshl_add, OP3("v_lshl_add_u32", ""))
KERNEL(k_add3, OP3("v_add3_u32", ""))
KERNEL(k_xad, OP3("v_xad_u32", ""))
KERNEL(k_sad_u8, OP3("v_sad_u8", ""))
KERNEL(k_msad_u8, OP3("v_msad_u8", ""))
KERNEL(k_perm, OP3("v_perm_b32", ""))
KERNEL(k_alignbit, OP3("v_alignbit_b32", ""))
KERNEL(k_mul_u24, OP2("v_mul_u32_u24"))
KERNEL(k_mad_u24, OP3("v_mad_u32_u24", ""))
KERNEL(k_mul_lo, OP2("v_mul_lo_u32"))
KERNEL(k_rndne, OP1("v_rndne_f32"))
KERNEL(k_cvt_ub0, OP1("v_cvt_f32_ubyte0"))
KERNEL(k_cvt_u32_f32, OP1("v_cvt_u32_f32"))
KERNEL(k_not, OP1("v_not_b32"))
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %10\n v_mov_b32 %3, %11\n v_mov_b32 %4, %8\n v_mov_b32 %5, %9\n v_mov_b32 %6, %10\n v_mov_b32 %7, %11")
KERNEL(k_mul_f32, OP2("v_mul_f32"))
KERNEL(k_fmac_f32, OP2("v_fmac_f32"))
KERNEL(k_pk_add_u16, OP2("v_pk_add_u16"))
KERNEL(k_max_u16, OP2("v_max_u16"))
KERNEL(k_min_u32, OP2("v_min_u32"))
KERNEL(k_cndmask, OP2("v_cndmask_b32"))
KERNEL(k_cmp_sgpr, "v_cmp_lt_u32 s[20:21], %0, %8\n v_cmp_lt_u32 s[22:23], %1, %9\n v_cmp_lt_u32 s[24:25], %2, %10\n v_cmp_lt_u32 s[26:27], %3, %11\n v_cmp_lt_u32 s[20:21], %4, %8\n v_cmp_lt_u32 s[22:23], %5, %9\n v_cmp_lt_u32 s[24:25], %6, %10\n v_cmp_lt_u32 s[26:27], %7, %11")
KERNEL(k_cmp_u16_sgpr, "v_cmp_lt_u16 s[20:21], %0, %8\n v_cmp_lt_u16 s[22:23], %1, %9\n v_cmp_lt_u16 s[24:25], %2, %10\n v_cmp_lt_u16 s[26:27], %3, %11\n v_cmp_lt_u16 s[20:21], %4, %8\n v_cmp_lt_u16 s[22:23], %5, %9\n v_cmp_lt_u16 s[24:25], %6, %10\n v_cmp_lt_u16 s[26:27], %7, %11")
KERNEL(k_readfirstlane, "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n v_readfirstlane_b32 s24, %4\n v_readfirstlane_b32 s25, %5\n v_readfirstlane_b32 s26, %6\n v_readfirstlane_b32 s27, %7")
KERNEL(k_dot4_u8, OP3("v_dot4_u32_u8", ""))
// packed fp32 (64-bit register pairs): 8 independent chains of 2 floats
typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL2(NAME, BODY)                                                                         \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, int iters, uint32_t seed)            \
    {                                                                                               \
        f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 * 3.f, a2 = a0 * 5.f, a3 = a0 * 7.f, t0 = {1.0001f, 0.9999f}, t1 = {0.5f, 0.25f}; \
        for (int i = 0; i < iters; i++) {                                                           \
            _Pragma("unroll") for (int k = 0; k < 16; k++)                                          \
            {                                                                                       \
                asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(t0), "v"(t1));     \
            }                                                                                       \
        }                                                                                           \
        const f2 r = a0 + a1 + a2 + a3;                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(r.x + r.y);                    \
    }
// 4 instructions per body, 16 bodies per iteration = 64 = 8 x the KERNEL macro's 8: same count per iteration as KERNEL
KERNEL2(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4")
KERNEL2(k_pk_add_f32, "v_pk_add_f32 %0, %0, %5\n v_pk_add_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %5\n v_pk_add_f32 %3, %3, %5")
KERNEL2(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5")
typedef void (*kern_t)(uint32_t *, int, uint32_t);
static void run(const char *name, kern_t k, uint32_t *d)
{
    const int blocks = 256 * 8, iters = 3000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * iters * 64.0;
    printf("%-16s %8.3f ms  %6.2f ns per wave-instruction per SIMD  %6.1f T lane-ops/s\n", name, ms, ms * 1e6 / (wave_instr / 1024.0),
           wave_instr * 64 / (ms * 1e-3) / 1e12);
}
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
#define R(k) run(#k, k, d)
    R(k_add_u32); R(k_sub_u32); R(k_and_b32); R(k_or_b32); R(k_lshrrev); R(k_lshlrev); R(k_bitop3); R(k_and_or); R(k_or3); R(k_bfi);
    R(k_lshl_add); R(k_add3); R(k_xad); R(k_sad_u8); R(k_msad_u8); R(k_perm); R(k_alignbit); R(k_mul_u24); R(k_mad_u24); R(k_mul_lo);
    R(k_rndne); R(k_cvt_ub0); R(k_cvt_u32_f32); R(k_not); R(k_mov); R(k_mul_f32); R(k_fmac_f32); R(k_pk_add_u16);
    R(k_max_u16); R(k_min_u32); R(k_cndmask); R(k_cmp_sgpr); R(k_cmp_u16_sgpr); R(k_readfirstlane); R(k_dot4_u8); R(k_pk_mul_f32); R(k_pk_add_f32); R(k_pk_fma_f32);
    return 0;
}
