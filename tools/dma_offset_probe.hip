// Does the instruction offset of global_load_lds_dwordx4 move the global address, the LDS address, or both?
// (match_mfma.hip fills a ring slot with two 1 KiB DMAs 1024 bytes apart on BOTH sides.)  Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint32_t *src, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint32_t s[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) s[i] = 0xDEADBEEFu;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * lane),
                                     (__attribute__((address_space(3))) void *)s, 16, 1024, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = s[i];
}
int main()
{
    std::vector<uint32_t> h(2048);
    for (int i = 0; i < 2048; i++) h[i] = i;
    uint32_t *d, *o;
    (void)hipMalloc(&d, 8192); (void)hipMalloc(&o, 4096);
    (void)hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    uint32_t r[1024];
    (void)hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    int first = -1;
    for (int i = 0; i < 1024; i++) if (r[i] != 0xDEADBEEFu) { first = i; break; }
    printf("first written LDS dword: %d (LDS byte %d), holds global dword %u (global byte %u)\n", first, first * 4, first >= 0 ? r[first] : 0, first >= 0 ? r[first] * 4 : 0);
    return 0;
}
