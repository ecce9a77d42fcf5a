// How fast are byte-misaligned LDS reads on gfx950?  (decides whether detect's ring test may fetch its
// 16 ring pixels with 7 unaligned b32/b64 reads instead of 17 byte reads)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_unaligned_rate tools/lds_unaligned_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32u __attribute__((aligned(1)));
typedef uint64_t u64u __attribute__((aligned(1)));

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, int mis)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[16384];
    for (int i = threadIdx.x; i < 16384 / 4; i += 256) reinterpret_cast<uint32_t *>(s)[i] = i * 2654435761u;
    __syncthreads();
    // pseudo-random 72-byte-pitch positions like the detect tile
    uint32_t a = ((threadIdx.x * 37u) % 60u) * 72u + ((threadIdx.x * 13u) & 63u) + 8u;
    a = (a & ~7u) + mis; // mis = 0: 8-byte aligned; 1..7: misaligned
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint8_t *p = s + ((a + u * 648u) & 8191u);
            if (MODE == 0) acc += *p;
            if (MODE == 1) acc += *reinterpret_cast<const u32u *>(p);
            if (MODE == 2) { const uint64_t v = *reinterpret_cast<const u64u *>(p); acc += (uint32_t)v ^ (uint32_t)(v >> 32); }
        }
        a += acc & 8u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    uint32_t *d;
    hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000, blocks = 1024;
    const char *names[3] = {"ds_read_u8 ", "ds_read_b32", "ds_read_b64"};
    for (int mode = 0; mode < 3; mode++)
        for (int mis = 0; mis < 8; mis += (mode == 0 ? 8 : 1)) {
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, mis);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, mis);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, mis);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            // wave-instructions = blocks * 4 waves * iters * 8; per CU: / 256 CUs
            const double per_cu = (double)blocks * 4 * iters * 8 / 256;
            printf("%s misalign %d: %.3f ms  -> %.2f ns per wave-instruction per CU\n", names[mode], mis, best, best * 1e6 / per_cu);
        }
    return 0;
}
