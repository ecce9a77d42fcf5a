// valu_rate3.hip -- ceiling of the matcher's inner mix on gfx950: per candidate 8 x (v_xor_b32
// with an SGPR operand + v_bcnt_u32_b32 accumulate) + key + min, operands resident (no loads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t *out, const uint32_t *__restrict__ bsrc, int iters)
{
    uint32_t a[8];
    for (int j = 0; j < 8; j++) a[j] = (threadIdx.x + 1) * (2 * j + 3);
    uint32_t b[8];
    for (int j = 0; j < 8; j++) b[j] = bsrc[j]; // uniform -> SGPRs
    uint32_t best = 0xFFFFFFFFu;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            uint32_t dist = 0;
            if (MODE == 2) { // grouped: 8 independent xors back to back, then the bcnt chain
                uint32_t x0, x1, x2, x3, x4, x5, x6, x7;
                asm volatile("v_xor_b32 %0, %9, %17\n v_xor_b32 %1, %10, %18\n v_xor_b32 %2, %11, %19\n v_xor_b32 %3, %12, %20\n"
                             "v_xor_b32 %4, %13, %21\n v_xor_b32 %5, %14, %22\n v_xor_b32 %6, %15, %23\n v_xor_b32 %7, %16, %24\n"
                             "v_bcnt_u32_b32 %8, %0, %8\n v_bcnt_u32_b32 %8, %1, %8\n v_bcnt_u32_b32 %8, %2, %8\n v_bcnt_u32_b32 %8, %3, %8\n"
                             "v_bcnt_u32_b32 %8, %4, %8\n v_bcnt_u32_b32 %8, %5, %8\n v_bcnt_u32_b32 %8, %6, %8\n v_bcnt_u32_b32 %8, %7, %8"
                             : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7), "+v"(dist)
                             : "s"(b[0] + c), "s"(b[1] + c), "s"(b[2] + c), "s"(b[3] + c), "s"(b[4] + c), "s"(b[5] + c), "s"(b[6] + c), "s"(b[7] + c),
                               "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
            } else
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                uint32_t x;
                if (MODE == 0) { // SGPR operand
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(b[kk] + c), "v"(a[kk]));
                } else {         // VGPR operand
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(a[(kk + 1) & 7]), "v"(a[kk]));
                }
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(dist) : "v"(x));
            }
            uint32_t key = (dist << 16) | (uint32_t)(i * 8 + c);
            best = key < best ? key : best;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = best;
}
template <int MODE> void run(const char *name, uint32_t *d, uint32_t *b)
{
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, b, 20);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, b, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    double pairs = (double)blocks * 256 * iters * 8.0;
    printf("%-24s %8.3f ms  %7.3f T pairs/s\n", name, ms, pairs / (ms * 1e-3) / 1e12);
}
int main()
{
    uint32_t *d, *b; (void)hipMalloc(&d, 256 * 8 * 256 * 4); (void)hipMalloc(&b, 64); (void)hipMemset(b, 0x5a, 64);
    run<0>("xor(sgpr)+bcnt acc", d, b); run<1>("xor(vgpr)+bcnt acc", d, b); run<2>("8 xor then 8 bcnt", d, b);
    return 0;
}
