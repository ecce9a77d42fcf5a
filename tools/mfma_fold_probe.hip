// Probe (round 5): what does ONE wave per SIMD pay for vector instructions placed between v_mfma_f32_16x16x128_f8f6f4 (e2m1)?
// The matcher needs one v_max3_f32 per MFMA.  MI355X_MICROARCH.md prices a 16-cycle bf16 MFMA at 8 cycles of vector issue and
// a v_max3_f32 at 4, so one fold per MFMA should hide; round 1 measured that it does not for THIS instruction.  This probe
// times hand-written streams with s_memtime (cycles, exact) on every CU, 1 and 2 waves per SIMD:
//   M     32 MFMAs (8 independent accumulators x 4)                     MX   MFMA, v_max3 alternating (independent registers)
//   MXX   MFMA + 2 v_max3                                               MA   MFMA + v_add_u32          MN   MFMA + s_nop 0
//   B     the same with v_mfma_f32_16x16x32_bf16 (the guide's instruction): B, BX
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_fold_probe tools/mfma_fold_probe.hip && tools/mfma_fold_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

#define MF(acc) "v_mfma_f32_16x16x128_f8f6f4 %" #acc ", %8, %9, %" #acc " cbsz:4 blgp:4\n\t"
#define BF(acc) "v_mfma_f32_16x16x32_bf16 %" #acc ", %8, %9, %" #acc "\n\t"
#define MX(r) "v_max3_f32 %" #r ", %" #r ", %14, %15\n\t"
#define AD(r) "v_add_u32 %" #r ", %" #r ", %14\n\t"
#define NP "s_nop 0\n\t"

template <int MODE>
__global__ void __launch_bounds__(512) probe(int iters, unsigned long long *out, float *sink)
{
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    u4 a = {0x22222222u, 0x22222222u, 0x22222222u, 0x22222222u}, b = {0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u};
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y = 0.5f, z = 0.25f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define STREAM(A0, F0, A1, F1, A2, F2, A3, F3, A4, F4, A5, F5, A6, F6, A7, F7)                                                  \
    asm volatile(A0 F0 A1 F1 A2 F2 A3 F3 A4 F4 A5 F5 A6 F6 A7 F7                                                                 \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)                               \
                 : "v"(a), "v"(b), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y), "v"(z))
        // x0..x3 are read-modify-write inside; declared as inputs only to stay below the operand limit (they are dead outside)
        if (MODE == 0) STREAM(MF(0), "", MF(1), "", MF(2), "", MF(3), "", MF(4), "", MF(5), "", MF(6), "", MF(7), "");
        if (MODE == 1) STREAM(MF(0), MX(10), MF(1), MX(11), MF(2), MX(12), MF(3), MX(13), MF(4), MX(10), MF(5), MX(11), MF(6), MX(12), MF(7), MX(13));
        if (MODE == 2) STREAM(MF(0), MX(10) MX(11), MF(1), MX(12) MX(13), MF(2), MX(10) MX(11), MF(3), MX(12) MX(13), MF(4), MX(10) MX(11), MF(5), MX(12) MX(13), MF(6), MX(10) MX(11), MF(7), MX(12) MX(13));
        if (MODE == 3) STREAM(MF(0), AD(10), MF(1), AD(11), MF(2), AD(12), MF(3), AD(13), MF(4), AD(10), MF(5), AD(11), MF(6), AD(12), MF(7), AD(13));
        if (MODE == 4) STREAM(MF(0), NP, MF(1), NP, MF(2), NP, MF(3), NP, MF(4), NP, MF(5), NP, MF(6), NP, MF(7), NP);
        if (MODE == 5) STREAM(BF(0), "", BF(1), "", BF(2), "", BF(3), "", BF(4), "", BF(5), "", BF(6), "", BF(7), "");
        if (MODE == 6) STREAM(BF(0), MX(10), BF(1), MX(11), BF(2), MX(12), BF(3), MX(13), BF(4), MX(10), BF(5), MX(11), BF(6), MX(12), BF(7), MX(13));
        if (MODE == 7) STREAM("", MX(10), "", MX(11), "", MX(12), "", MX(13), "", MX(10), "", MX(11), "", MX(12), "", MX(13));
        // folds that READ the accumulators of MFMAs issued four earlier (the matcher's real dependency): physical registers
        // (an inline-asm operand cannot name one register of a tuple): accumulators v[100:131], running maxima v132..v135
#define MFP(lo, hi) "v_mfma_f32_16x16x128_f8f6f4 v[" #lo ":" #hi "], %0, %1, v[" #lo ":" #hi "] cbsz:4 blgp:4\n\t"
        if (MODE == 8)
            asm volatile(MFP(100, 103) "v_max3_f32 v132, v132, v116, v120\n\t" MFP(104, 107) "v_max3_f32 v133, v133, v117, v121\n\t"
                         MFP(108, 111) "v_max3_f32 v134, v134, v118, v122\n\t" MFP(112, 115) "v_max3_f32 v135, v135, v119, v123\n\t"
                         MFP(116, 119) "v_max3_f32 v132, v132, v100, v104\n\t" MFP(120, 123) "v_max3_f32 v133, v133, v101, v105\n\t"
                         MFP(124, 127) "v_max3_f32 v134, v134, v102, v106\n\t" MFP(128, 131) "v_max3_f32 v135, v135, v103, v107\n\t"
                         :
                         : "v"(a), "v"(b)
                         : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114",
                           "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129",
                           "v130", "v131", "v132", "v133", "v134", "v135");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    float s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3] + x0 + x1 + x2 + x3;
    if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
static double run(int threads, int iters)
{
    unsigned long long *d;
    float *sink;
    const int waves = 256 * threads / 64;
    hipMalloc(&d, waves * 8);
    hipMalloc(&sink, 64);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, iters, d, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    hipFree(d);
    hipFree(sink);
    return (double)h[waves / 2] / ((double)iters * 8); // median cycles per MFMA slot (8 slots per iteration)
}

int main()
{
    const int iters = 20000;
    const char *names[] = {"M    32 MFMA f8f6f4 e2m1", "MX   MFMA + v_max3", "MXX  MFMA + 2 v_max3", "MA   MFMA + v_add_u32", "MN   MFMA + s_nop 0",
                           "B    MFMA 16x16x32 bf16", "BX   bf16 MFMA + v_max3", "X    v_max3 alone", "MXd  MFMA + v_max3 reading older accumulators"};
    for (int threads : {256, 512, 768}) {
        double r[9] = {run<0>(threads, iters), run<1>(threads, iters), run<2>(threads, iters), run<3>(threads, iters), run<4>(threads, iters),
                       run<5>(threads, iters), run<6>(threads, iters), run<7>(threads, iters), run<8>(threads, iters)};
        for (int k = 0; k < 9; k++)
            printf("%d wave(s) per SIMD  %-48s %.2f cycles per slot per wave (%.2f per SIMD)\n", threads / 256, names[k], r[k], r[k] / (threads / 256));
    }
    return 0;
}
