// Probe: do the matrix pipe and the vector ALU of a CDNA4 SIMD run concurrently ACROSS WAVES for the
// instructions this library uses -- v_mfma_f32_16x16x128_f8f6f4 (e2m1 operands; the matcher) against full-rate VALU
// (v_add_u32 / v_bitop3-class; detect) and against v_max3_f32 (the matcher's own epilogue)?
// One workgroup of 512 threads per CU (2 waves per SIMD); wave w of a SIMD pair gets a role:
//   role M: `iters` x 8 independent MFMAs          role V: `iters` x 64 v_add_u32 (4 independent chains)
//   role X: `iters` x 64 v_max3_f32
// Timed three ways on all 256 CUs: every wave M (+ idle partner), every wave V, and M beside V on the same SIMD.
// If the pipes co-execute, t(M beside V) ~ max(t(M), t(V)); if issue is exclusive, ~ t(M) + t(V).
//   hipcc --offload-arch=gfx950 -O3 -o tools/coexec_probe tools/coexec_probe.hip && tools/coexec_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// roles per wave parity inside a SIMD: waves 0-3 = first wave of SIMD 0-3, waves 4-7 = second wave (CDNA deals the
// waves of a workgroup round-robin over the SIMDs)
enum { IDLE = 0, MFMA = 1, VADD = 2, VMAX3 = 3 };

// 512 threads: one wave of role A and one of role B per SIMD; 1024 threads: one wave of role A and THREE of role B per SIMD
// (three dependent-chain VALU waves keep a SIMD's vector issue saturated: the 2-wave form leaves issue slots idle)
__global__ void __launch_bounds__(1024) coexec_kernel(int role_a, int role_b, int iters, float *out)
{
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? role_a : role_b;
    float sink = 0.f;
    if (role == MFMA) {
        v8i fa, fb;
        for (int i = 0; i < 8; ++i) {
            fa[i] = 0x22222222 * (i < 4);
            fb[i] = 0x20202020 * (i < 4);
        }
        v4f acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, acc[i], 4, 4, 0, 0, 0, 0);
        for (int i = 0; i < 8; ++i) sink += acc[i][0] + acc[i][3];
    } else if (role == VADD) {
        uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(c));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(a));
            }
        sink = (float)(a + b + c + d);
    } else if (role == VMAX3) {
        float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(b) : "v"(c), "v"(d));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(a));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
            }
        sink = a + b + c + d;
    }
    if (sink == 12345.678f) out[0] = sink;
}

static int g_threads = 512;
static float run(int ra, int rb, int iters, float *d_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(coexec_kernel, dim3(256), dim3(g_threads), 0, 0, ra, rb, iters, d_out);
    hipEventRecord(e0);
    for (int w = 0; w < 5; w++) hipLaunchKernelGGL(coexec_kernel, dim3(256), dim3(g_threads), 0, 0, ra, rb, iters, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main()
{
    float *d_out;
    hipMalloc(&d_out, 64);
    const int iters = 4000;
    const char *name[] = {"idle", "mfma", "v_add", "v_max3"};
    const int cases[][2] = {{MFMA, IDLE}, {VADD, IDLE}, {VMAX3, IDLE}, {MFMA, MFMA}, {VADD, VADD}, {MFMA, VADD}, {MFMA, VMAX3}, {VADD, VMAX3}};
    float t[8];
    for (int c = 0; c < 8; c++) {
        t[c] = run(cases[c][0], cases[c][1], iters, d_out);
        printf("wave A %-6s beside wave B %-6s : %.4f ms\n", name[cases[c][0]], name[cases[c][1]], t[c]);
    }
    printf("mfma beside v_add : %.4f ms; alone %.4f and %.4f; sum %.4f, max %.4f -> overlap fraction %.2f\n", t[5], t[0], t[1],
           t[0] + t[1], t[0] > t[1] ? t[0] : t[1], (t[0] + t[1] - t[5]) / (t[0] < t[1] ? t[0] : t[1]));
    printf("mfma beside v_max3: %.4f ms; alone %.4f and %.4f; sum %.4f, max %.4f -> overlap fraction %.2f\n", t[6], t[0], t[2],
           t[0] + t[2], t[0] > t[2] ? t[0] : t[2], (t[0] + t[2] - t[6]) / (t[0] < t[2] ? t[0] : t[2]));
    printf("(overlap fraction 1 = the shorter stream is hidden entirely, 0 = issue is exclusive)\n");
    // saturated form: ONE wave of role A beside THREE of role B on every SIMD
    g_threads = 1024;
    const int sat[][2] = {{MFMA, IDLE}, {IDLE, VADD}, {IDLE, VMAX3}, {MFMA, VADD}, {MFMA, VMAX3}, {VADD, VADD}};
    float u[6];
    for (int c = 0; c < 6; c++) {
        u[c] = run(sat[c][0], sat[c][1], iters, d_out);
        printf("1 wave %-6s beside 3 waves %-6s per SIMD: %.4f ms\n", name[sat[c][0]], name[sat[c][1]], u[c]);
    }
    printf("saturated: mfma + 3 x v_add  %.4f ms; alone %.4f and %.4f; sum %.4f -> overlap fraction %.2f\n", u[3], u[0], u[1], u[0] + u[1],
           (u[0] + u[1] - u[3]) / (u[0] < u[1] ? u[0] : u[1]));
    printf("saturated: mfma + 3 x v_max3 %.4f ms; alone %.4f and %.4f; sum %.4f -> overlap fraction %.2f\n", u[4], u[0], u[2], u[0] + u[2],
           (u[0] + u[2] - u[4]) / (u[0] < u[2] ? u[0] : u[2]));
    printf("vector issue rate at saturation: %.2f cycles per v_add wave-instruction per SIMD (4 waves), %.2f per v_max3 (3 waves)\n",
           u[5] * 1e-3 * 2.4e9 / (4.0 * iters * 64), u[2] * 1e-3 * 2.4e9 / (3.0 * iters * 64));
    hipFree(d_out);
    return 0;
}
