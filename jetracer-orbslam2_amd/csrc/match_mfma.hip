// The matrix-core 256-bit matcher.  Its own translation unit because it is built with
// -fno-honor-nans (no v_max_f32 x, x, x canonicalisation in front of every maximum; every value
// here is a finite integer) and -mllvm -amdgpu-mfma-vgpr-form (MFMA results land in VGPRs, where
// v_max_f32 can read them, instead of AGPRs + v_accvgpr_read).  Nothing in this file depends on
// include/orbfe_math.h's rounding contract.
#include "device_common.hpp"
#include "orbfe_internal.hpp"

namespace orbfe {

// 256-bit brute force on the matrix cores (window < 0, at most kMmaS keypoints per frame).
//
// Hamming distance of binary vectors is dist(a, b) = |a| + |b| - 2 a.b, and the all-pairs a.b
// of two frames is a {0,1} matrix product with K = 256: exact in any format that holds 0 and 1
// and accumulates in f32 (all sums are integers < 2^24).  gfx950's v_mfma_scale_f32_16x16x128
// _f8f6f4 takes 4-bit e2m1 operands, so one instruction covers 16 x 16 pairs x 128 bits.
//
// (1) match_expand_kernel rewrites every descriptor as 256 e2m1 nibbles (1.0 = 0x2) in MFMA
//     fragment order -- per 16 keypoints: [k-step 2][lane 64][16 B], lane = 16 * (word & 3) +
//     (keypoint & 15) -- so that one wave-wide 16-byte load IS the A or B operand (the order
//     of the 256 bits inside K does not matter as long as both frames use the same one).  It
//     also writes colkey[j] = -(|b_j| * S + j), S = 16384 (-1e30 for padding).
// (2) match_mfma_kernel: block = 128 queries of frame p (8 A fragments x 2 k-steps, resident
//     in VGPRs), its 4 waves take every 4th block of 16 candidates of frame p + 1.  B is
//     block-scaled by 2^15 = 2S (E8M0 142) and the accumulator starts at colkey[j], so the
//     MFMA pair itself yields key' = 2S a.b - S|b_j| - j = -(S (dist - |a|) + j) exactly, and
//     the whole epilogue is one v_max_f32 per pair: the maximum key' is the lexicographic
//     minimum (dist, j), the same winner as the packed-key v_min_u32 of the VALU kernel.
//     The 64 partial maxima per query (4 waves x 16 column classes) are reduced through LDS.
constexpr int kMmaS = 16384;
constexpr int kMmaRows = 128;
constexpr int kMmaLds = 68; // floats per query row in LDS: 64 + 4 keeps writes and b128 reads conflict-free
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// 8 bits -> 8 e2m1 nibbles (bit b -> nibble b): 0x2 (= 1.0) or 0
__device__ __forceinline__ uint32_t spread_bits_e2m1(uint32_t x)
{
    x &= 0xFFu;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x << 1;
}

__global__ void __launch_bounds__(256)
match_expand_kernel(const orbfe_keypoint *__restrict__ records, const int32_t *__restrict__ counts, int cap, int capP,
                    uint4 *__restrict__ mexp, float *__restrict__ mkey)
{
    const int f = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 3, w = t & 7; // 8 consecutive lanes = the 8 words of keypoint i
    if (i >= capP) return;
    const bool live = i < counts[f];
    uint32_t word = 0;
    if (live) word = reinterpret_cast<const uint32_t *>(records + (size_t)f * cap + i)[5 + w];
    int pop = __popc(word);
    pop += __shfl_xor(pop, 1);
    pop += __shfl_xor(pop, 2);
    pop += __shfl_xor(pop, 4);
    const uint4 e = make_uint4(spread_bits_e2m1(word), spread_bits_e2m1(word >> 8), spread_bits_e2m1(word >> 16),
                               spread_bits_e2m1(word >> 24));
    mexp[(size_t)f * capP * 8 + (size_t)(i >> 4) * 128 + (w >> 2) * 64 + (w & 3) * 16 + (i & 15)] = e;
    if (w == 0) mkey[(size_t)f * capP + i] = live ? -(float)(pop * kMmaS + i) : -1e30f;
}

__global__ void __launch_bounds__(256)
match_mfma_kernel(const uint4 *__restrict__ mexp, const float *__restrict__ mkey, const int32_t *__restrict__ counts,
                  int cap, int capP, int max_dist, int32_t *__restrict__ out_idx, int32_t *__restrict__ out_dist)
{
    __shared__ float s_best[kMmaRows * kMmaLds];
    int p, blk;
    xcd_remap(gridDim.x, gridDim.y, &p, &blk); // all query blocks of a pair share one L2
    const int nA = counts[p], nB = counts[p + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row0 = blk * kMmaRows;
    const uint4 *__restrict__ Ea = mexp + (size_t)p * capP * 8;
    const uint4 *__restrict__ Eb = mexp + (size_t)(p + 1) * capP * 8;
    const float *__restrict__ Kb = mkey + (size_t)(p + 1) * capP;
    const bool active = row0 < nA && nB > 0; // block-uniform

    if (active) {
        v8i a[8][2];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int ab = (row0 >> 4) + m;
            const bool in = ab * 16 < capP;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const uint4 q = in ? Ea[(size_t)ab * 128 + ks * 64 + lane] : make_uint4(0, 0, 0, 0);
                a[m][ks] = (v8i){(int)q.x, (int)q.y, (int)q.z, (int)q.w, 0, 0, 0, 0};
            }
        }
        v4f best[8];
#pragma unroll
        for (int m = 0; m < 8; m++) best[m] = (v4f){-3e38f, -3e38f, -3e38f, -3e38f};
        const int nBb = (nB + 15) >> 4;
        int bb = wv;
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
        float c = -1e30f;
        if (bb < nBb) {
            q0 = Eb[(size_t)bb * 128 + lane];
            q1 = Eb[(size_t)bb * 128 + 64 + lane];
            c = Kb[bb * 16 + (lane & 15)];
        }
        while (bb < nBb) {
            const int nb = bb + 4;
            uint4 n0 = q0, n1 = q1;
            float nc = c;
            if (nb < nBb) { // prefetch the next candidate block under this one's MFMAs
                n0 = Eb[(size_t)nb * 128 + lane];
                n1 = Eb[(size_t)nb * 128 + 64 + lane];
                nc = Kb[nb * 16 + (lane & 15)];
            }
            const v8i b0 = (v8i){(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w, 0, 0, 0, 0};
            const v8i b1 = (v8i){(int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w, 0, 0, 0, 0};
            const v4f cv = (v4f){c, c, c, c};
#pragma unroll
            for (int m = 0; m < 8; m++) {
                // cbsz = blgp = 4: e2m1 operands; scales are E8M0 bytes: A x 2^0 (127), B x 2^15 (142)
                v4f acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][0], b0, cv, 4, 4, 0, 127, 0, 142);
                acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][1], b1, acc, 4, 4, 0, 127, 0, 142);
#pragma unroll
                for (int r = 0; r < 4; r++) best[m][r] = __builtin_fmaxf(best[m][r], acc[r]);
            }
            q0 = n0;
            q1 = n1;
            c = nc;
            bb = nb;
        }
        // C/D layout: lane holds rows 4 * (lane >> 4) + r of each 16-row fragment, column lane & 15
#pragma unroll
        for (int m = 0; m < 8; m++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                s_best[(m * 16 + 4 * (lane >> 4) + r) * kMmaLds + wv * 16 + (lane & 15)] = best[m][r];
    }
    __syncthreads();
    const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
    float v = -3e38f;
    if (active) {
        const float4 *src = reinterpret_cast<const float4 *>(s_best + row * kMmaLds + half * 32);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 x = src[k];
            v = fmaxf(fmaxf(v, fmaxf(x.x, x.y)), fmaxf(x.z, x.w));
        }
    }
    v = fmaxf(v, __shfl_xor(v, 1));
    const int i = row0 + row;
    if (half == 0 && i < cap) {
        bool ok = active && i < nA && v > -1e29f;
        int bj = -1, bd = -1;
        if (ok) {
            const int nk = -(int)v;                                       // S * (dist - |a_i|) + j
            const int pop_a = (-(int)mkey[(size_t)p * capP + i]) >> 14; // |a_i|
            bj = nk & (kMmaS - 1);
            bd = pop_a + (nk >> 14);
            ok = bd <= max_dist;
        }
        out_idx[(size_t)p * cap + i] = ok ? bj : -1;
        if (out_dist) out_dist[(size_t)p * cap + i] = ok ? bd : -1;
    }
}


void launch_match_mfma(const orbfe_keypoint *d_records, const int32_t *d_counts, int n_frames, int cap, int capP,
                       int max_dist, uint4 *mexp, float *mkey, int32_t *d_idx, int32_t *d_dist, hipStream_t stream)
{
    static_assert(kMmaS == kMmaMaxKeypoints, "key packing");
    hipLaunchKernelGGL(match_expand_kernel, dim3((capP * 8 + 255) / 256, n_frames), dim3(256), 0, stream, d_records,
                       d_counts, cap, capP, mexp, mkey);
    hipLaunchKernelGGL(match_mfma_kernel, dim3((capP + kMmaRows - 1) / kMmaRows, n_frames - 1), dim3(256), 0, stream,
                       mexp, mkey, d_counts, cap, capP, max_dist, d_idx, d_dist);
}

} // namespace orbfe
