// Do unaligned dword global loads, unaligned global_load_lds_dword and misaligned ds_read_b128
// return the right bytes on gfx950?  hipcc --offload-arch=gfx950 -O2 -o tools/unaligned_probe tools/unaligned_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint8_t* src, uint32_t* out_plain, uint32_t* out_dma, uint4* out_lds, int shift) {
    __shared__ __attribute__((aligned(16))) uint32_t s[64 * 5];
    const int lane = threadIdx.x;
    const uint8_t* p = src + shift + 4 * lane;
    out_plain[lane] = *reinterpret_cast<const uint32_t*>(p);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                     (__attribute__((address_space(3))) void*)s, 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out_dma[lane] = s[lane];
    __syncthreads();
    for (int i = lane; i < 320; i += 64) s[i] = reinterpret_cast<const uint32_t*>(src)[i];
    __syncthreads();
    // lane reads 16 bytes at byte offset 16 * lane + ((lane + shift) & 3): per-lane different misalignment
    const uint32_t a = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)s + 16 * lane + ((lane + shift) & 3);
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out_lds[lane] = v;
}
int main() {
    std::vector<uint8_t> h(2048);
    for (int i = 0; i < 2048; i++) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t* d; uint32_t *o1, *o2; uint4* o3;
    hipMalloc(&d, 2048); hipMalloc(&o1, 256); hipMalloc(&o2, 256); hipMalloc(&o3, 1024);
    hipMemcpy(d, h.data(), 2048, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; shift++) {
        k<<<1, 64>>>(d, o1, o2, o3, shift);
        uint32_t a[64], b[64]; uint8_t c[1024];
        hipMemcpy(a, o1, 256, hipMemcpyDeviceToHost); hipMemcpy(b, o2, 256, hipMemcpyDeviceToHost);
        hipMemcpy(c, o3, 1024, hipMemcpyDeviceToHost);
        int bad1 = 0, bad2 = 0, bad3 = 0;
        for (int l = 0; l < 64; l++) {
            uint32_t want = 0;
            for (int j = 0; j < 4; j++) want |= (uint32_t)h[shift + 4 * l + j] << (8 * j);
            bad1 += a[l] != want; bad2 += b[l] != want;
            for (int j = 0; j < 16; j++) bad3 += c[16 * l + j] != h[16 * l + ((l + shift) & 3) + j];
        }
        printf("shift %d: plain dword %d/64 wrong, LDS-DMA dword %d/64 wrong, misaligned ds_read_b128 %d/1024 bytes wrong (%s)\n",
               shift, bad1, bad2, bad3, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
