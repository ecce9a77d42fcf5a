// calib_copy.hip -- known-byte-count streaming kernels to calibrate rocprofv3 FETCH_SIZE /
// WRITE_SIZE on gfx950 for the access widths the orbfe kernels use (4 B and 16 B per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void copy_dword(const uint32_t *__restrict__ s, uint32_t *__restrict__ d, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i] + 1u;
}
__global__ void copy_dwordx4(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = s[i];
        v.x += 1u;
        d[i] = v;
    }
}
int main()
{
    const size_t bytes = 512ull << 20; // 512 MiB each way: larger than the 256 MiB Infinity Cache
    void *s, *d;
    (void)hipMalloc(&s, bytes);
    (void)hipMalloc(&d, bytes);
    (void)hipMemset(s, 1, bytes);
    (void)hipMemset(d, 0, bytes);
    for (int it = 0; it < 3; it++) {
        hipLaunchKernelGGL(copy_dword, dim3(2048), dim3(256), 0, 0, (const uint32_t *)s, (uint32_t *)d, bytes / 4);
        hipLaunchKernelGGL(copy_dwordx4, dim3(2048), dim3(256), 0, 0, (const uint4 *)s, (uint4 *)d, bytes / 16);
    }
    (void)hipDeviceSynchronize();
    printf("each launch reads %zu bytes and writes %zu bytes\n", bytes, bytes);
    return 0;
}
