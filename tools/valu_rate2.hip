// valu_rate2.hip -- compiler-generated select / convert / float min-max rates (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a[8];
    float f[8];
    for (int j = 0; j < 8; j++) { a[j] = (threadIdx.x + seed) * (2 * j + 3); f[j] = (float)(a[j] & 255); }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                if (OP == 0) a[j] = (a[j] > a[j + 1]) ? a[j] + 7 : a[j + 1];            // cmp + cndmask (+add)
                if (OP == 1) f[j] = fminf(f[j] + 1.0f, f[j + 1]);                         // add + min f32
                if (OP == 2) f[j] = fmaxf(f[j] - f[j + 1], 0.0f) + f[j];                  // sub max add
                if (OP == 3) f[j] += (float)((a[j + 1] >> 8) & 255u);                     // cvt_f32_ubyte1 + add
                if (OP == 4) a[j] += (a[j + 1] < a[j]);                                   // cmp + addc
                if (OP == 5) a[j] = (a[j] & 0xffff0000u) | ((a[j] + a[j+1]) & 0xffffu);   // bfi
            }
        }
    }
    uint32_t r = 0;
    for (int j = 0; j < 8; j++) r ^= a[j] ^ __float_as_uint(f[j]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run(const char *name, uint32_t *d, double ops_per)
{
    const int blocks = 256 * 8, iters = 5000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 50, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    double stmts = (double)blocks * 256 * iters * 32.0;
    printf("%-28s %8.3f ms  %7.2f T statements/s (x%.0f instr)\n", name, ms, stmts / (ms * 1e-3) / 1e12, ops_per);
}
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("cmp+cndmask+add", d, 3); run<1>("addf+minf", d, 2); run<2>("subf+maxf+addf", d, 3);
    run<3>("cvt_f32_ubyte1+addf", d, 2); run<4>("cmp+addc", d, 2); run<5>("add+bfi", d, 2);
    return 0;
}
