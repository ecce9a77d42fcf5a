// Where does global_load_lds_dwordx3 put lane L's 12 bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint32_t* src, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t s[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) s[i] = 0xDEADBEEFu;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 3 * lane),
                                     (__attribute__((address_space(3))) void*)s, 12, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = s[i];
}
int main() {
    std::vector<uint32_t> h(256);
    for (int i = 0; i < 256; i++) h[i] = i;  // lane L loads dwords 3L, 3L+1, 3L+2
    uint32_t *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o);
    uint32_t r[512];
    hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
    for (int i = 0; i < 40; i++) printf("%s%d", i ? " " : "lds dwords: ", r[i] == 0xDEADBEEFu ? -1 : (int)r[i]);
    printf("\n");
    int contiguous = 1, stride16 = 1;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 3; j++) { contiguous &= r[3 * l + j] == (uint32_t)(3 * l + j); stride16 &= r[4 * l + j] == (uint32_t)(3 * l + j); }
    printf("lane*12 contiguous: %d   lane*16 stride: %d\n", contiguous, stride16);
    return 0;
}
