// sstore_probe.hip -- can a gfx950 wave write SGPRs straight to global memory (s_store_dwordx4 + s_dcache_wb), into
// 52-byte records whose other dwords are written by vector stores from OTHER workgroups?  (The describe kernel holds
// a keypoint's 256 descriptor bits as four 64-bit ballots in SGPRs; moving them to VGPRs for a vector store costs 8
// v_mov per keypoint.)  Checks every byte on the host and times both forms.  Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline uint32_t mix(uint32_t a) { a ^= a >> 15; a *= 0x2C1B3C6Du; a ^= a >> 12; a *= 0x297A2D39u; a ^= a >> 15; return a; }
template <bool SCALAR>
__global__ void __launch_bounds__(256) k_store(uint32_t *rec, int n, int per_wave)
{
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (int j = 0; j < per_wave; j++) {
        // records dealt so that neighbours in memory come from different workgroups
        const int r = __builtin_amdgcn_readfirstlane(j * (gridDim.x * 4) + wave);
        if (r >= n) break;
        u32x4 a, b;
        for (int k = 0; k < 4; k++) {
            a[k] = __builtin_amdgcn_readfirstlane(mix(13u * r + 5 + k));
            b[k] = __builtin_amdgcn_readfirstlane(mix(13u * r + 9 + k));
        }
        uint32_t *p = rec + (size_t)r * 13;
        if (lane < 5) p[lane] = mix(13u * r + lane); // header by vector stores
        if (SCALAR) {
            const uint32_t *q = p + 5;
            asm volatile("s_store_dwordx4 %0, %2, 0x0\n s_store_dwordx4 %1, %2, 0x10" ::"s"(a), "s"(b), "s"(q) : "memory");
        } else if (lane == 0) {
            for (int k = 0; k < 4; k++) { p[5 + k] = a[k]; p[9 + k] = b[k]; }
        }
    }
    if (SCALAR) asm volatile("s_dcache_wb" ::: "memory");
}
int main()
{
    const int n = 1 << 20, blocks = 2048, per_wave = (n + blocks * 4 - 1) / (blocks * 4);
    uint32_t *d; (void)hipMalloc(&d, (size_t)n * 52);
    std::vector<uint32_t> h((size_t)n * 13);
    for (int mode = 0; mode < 2; mode++) {
        (void)hipMemset(d, 0xEE, (size_t)n * 52);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        for (int rep = 0; rep < 10; rep++) {
            if (mode) hipLaunchKernelGGL(k_store<true>, dim3(blocks), dim3(256), 0, 0, d, n, per_wave);
            else hipLaunchKernelGGL(k_store<false>, dim3(blocks), dim3(256), 0, 0, d, n, per_wave);
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h.data(), d, (size_t)n * 52, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int r = 0; r < n; r++)
            for (int k = 0; k < 13; k++) {
                const uint32_t want = k < 5 ? mix(13u * r + k) : (k < 9 ? mix(13u * r + 5 + (k - 5)) : mix(13u * r + 9 + (k - 9)));
                bad += h[(size_t)r * 13 + k] != want;
            }
        printf("%s stores: %.3f ms per launch (%d records), %ld wrong dwords\n", mode ? "scalar" : "vector", ms / 10, n, bad);
    }
    return 0;
}
