// The matrix-core 256-bit matcher.  Its own translation unit because it is built with
// -fno-honor-nans (no v_max_f32 x, x, x canonicalisation in front of every maximum; every value
// here is a finite integer) and -mllvm -amdgpu-mfma-vgpr-form (MFMA results land in VGPRs, where
// v_max3_f32 can read them, instead of AGPRs + v_accvgpr_read).  Nothing in this file depends on
// include/orbfe_math.h's rounding contract.
#include "device_common.hpp"
#include "orbfe_internal.hpp"

#include <type_traits>

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
//     also writes colkey[j] = -(|b_j| * S + j) / 2S, S = 16384 (-1e30 for padding).
// (2) match_mfma_kernel: block = 128 queries of frame p (8 A fragments x 2 k-steps, resident
//     in VGPRs), its 4 waves take every 4th block of 16 candidates of frame p + 1, streamed
//     through a per-wave LDS ring by LDS-DMA.  The accumulator starts at colkey[j], so the
//     MFMA pair itself yields key' = a.b - |b_j| / 2 - j / 2S = -(S (dist - |a|) + j) / 2S
//     exactly (8 integer + 15 fraction bits: every partial sum fits the f32 significand), and
//     the whole epilogue is a running maximum (one v_max3_f32 per two pairs): the maximum key'
//     is the lexicographic minimum (dist, j), the same winner as the packed-key v_min_u32 of
//     the VALU kernel.  The 64 partial maxima per query (4 waves x 16 column classes) are
//     reduced through LDS.
//     Round 1 carried the keys as integers and let the instruction's block scale multiply B by
//     2S; with both scales 0 hipcc emits the plain v_mfma_f32_16x16x128_f8f6f4 (a 64-bit
//     encoding, no scale operands to read) and the same loop runs 5 % faster.
// Measured (tools/mfma_probe.hip, MI355X): the MFMA alone issues at 19 T pairs/s; with the
// epilogue as v_max_f32 the step structure tops out at 12.5 T (VALU issue does not overlap these
// MFMAs), with v_max3_f32 at 14.7 T; the kernel reaches ~12 T.
constexpr int kMmaS = 16384;
constexpr float kKeyUnit = 1.0f / (2 * kMmaS); // keys are carried as (integer key) / 2S: a power of two, exact
constexpr int kRing = 4;                 // candidate blocks in flight per wave
constexpr int kSlotBytes = 2048 + 256;   // one block: 2 x 1 KB fragments + 16 column keys x 4 copies (a b128 read = the C tuple)
// NW waves per workgroup (4; 2 = the small-footprint form, 18 KB of LDS instead of 37, for co-residency experiments):
// floats per query row in LDS: 16 NW + 4 keeps writes and b128 reads conflict-free
constexpr int mma_lds_row(int nw) { return 16 * nw + 4; }
constexpr int mma_lds_bytes(int rb, int nw)
{
    return nw * kRing * kSlotBytes > 16 * rb * mma_lds_row(nw) * 4 ? nw * kRing * kSlotBytes : 16 * rb * mma_lds_row(nw) * 4;
}
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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
                    uint4 *__restrict__ mexp, float *__restrict__ mkey, float4 *__restrict__ mkey4)
{
    const int f = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 3, w = t & 7; // 8 consecutive lanes = the 8 words of keypoint i
    if (i >= capP) return;
    const bool live = i < clamp_count(counts[f], cap);
    uint32_t word = 0;
    if (live) word = reinterpret_cast<const uint32_t *>(records + (size_t)f * cap + i)[5 + w];
    int pop = __popc(word);
    pop += __shfl_xor(pop, 1);
    pop += __shfl_xor(pop, 2);
    pop += __shfl_xor(pop, 4);
    const uint4 e = make_uint4(spread_bits_e2m1(word), spread_bits_e2m1(word >> 8), spread_bits_e2m1(word >> 16),
                               spread_bits_e2m1(word >> 24));
    mexp[(size_t)f * capP * 8 + (size_t)(i >> 4) * 128 + (w >> 2) * 64 + (w & 3) * 16 + (i & 15)] = e;
    if (w == 0) {
        const float k = live ? -(float)(pop * kMmaS + i) * kKeyUnit : -1e30f;
        mkey[(size_t)f * capP + i] = k;
        // the register-streamed kernel (match_mfma2_kernel) loads the accumulator-init tuple {k, k, k, k} of its column with
        // one 16-byte load
        mkey4[(size_t)f * capP + i] = make_float4(k, k, k, k);
    }
}

// RB = query row blocks of 16 per workgroup: 8 (128 queries), or 4 when the call is so small that 128-query
// workgroups would not even put one on every CU (one to ~15 pairs of 2000: 13 -> 10 us for a single pair,
// 23 -> 17 us at 4800 keypoints; from 32 pairs on the larger block wins again, measured)
template <int RB, int NW>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(3)))
match_mfma_kernel(const uint4 *__restrict__ mexp, const float *__restrict__ mkey, const int32_t *__restrict__ counts,
                  int cap, int capP, int first, int stride, int max_dist, int32_t *__restrict__ out_idx,
                  int32_t *__restrict__ out_dist)
{
    // the candidate ring while the MFMA loop runs, then (after a barrier) the partial maxima
    constexpr int kMmaLds = mma_lds_row(NW);
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[mma_lds_bytes(RB, NW)];
    float *s_best = reinterpret_cast<float *>(s_mem);
    int pk, blk;
    xcd_remap(gridDim.x, gridDim.y, &pk, &blk); // all query blocks of a pair share one L2
    const int p = first + pk * stride;
    const int nA = clamp_count(counts[p], cap), nB = clamp_count(counts[p + 1], cap);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform: block indices stay in SGPRs
    constexpr int kMmaRows = 16 * RB;
    const int row0 = blk * kMmaRows;
    const uint4 *__restrict__ Ea = mexp + (size_t)p * capP * 8;
    const uint4 *__restrict__ Eb = mexp + (size_t)(p + 1) * capP * 8;
    const float *__restrict__ Kb = mkey + (size_t)(p + 1) * capP;
    const bool active = row0 < nA && nB > 0; // block-uniform

    if (active) {
        v8i a[RB][2];
#pragma unroll
        for (int m = 0; m < RB; m++) {
            const int ab = (row0 >> 4) + m;
            const bool in = ab * 16 < capP;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const uint4 q = in ? Ea[(size_t)ab * 128 + ks * 64 + lane] : make_uint4(0, 0, 0, 0);
                a[m][ks] = (v8i){(int)q.x, (int)q.y, (int)q.z, (int)q.w, 0, 0, 0, 0};
            }
        }
        v4f best[RB];
#pragma unroll
        for (int m = 0; m < RB; m++) best[m] = (v4f){-3e38f, -3e38f, -3e38f, -3e38f};
        const int nBb = (nB + 15) >> 4;
        // This wave's candidate blocks wv, wv + 4, ... stream through a ring of kRing LDS slots
        // filled by LDS-DMA (global_load_lds: no VGPRs, nothing the register allocator could
        // copy while in flight).  The fragment layout written by match_expand_kernel is exactly
        // lane-linear, so one 16-byte DMA per k-step lands as the operand image.  Each slot is
        // filled by exactly 3 DMAs and VMEM retires in order, so slot s is complete once at
        // most 3 * (kRing - 1) younger DMAs are outstanding.  Past the wave's last block the
        // index is clamped: every step issues its 3 DMAs and the count stays exact.
        unsigned char *ring = s_mem + wv * (kRing * kSlotBytes);
        const uint32_t ring_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)ring;
        const uint32_t lds16 = ring_lds + lane * 16, ldsk = ring_lds + (lane & 15) * 16; // this lane's read addresses
        const int T = wv < nBb ? (nBb - wv + NW - 1) / NW : 0; // blocks of this wave
        const uint32_t lane_off16 = (uint32_t)lane * 16u, lane_off4 = (uint32_t)(lane >> 2) * 4u;
        auto issue = [&](int s, int t) {
            int lb = wv + NW * t;
            lb = lb < nBb ? lb : nBb - 1;
            // uniform base + 32-bit lane offset: the SGPR-base form of the instruction, no 64-bit vector add
            const char *src = reinterpret_cast<const char *>(Eb + (size_t)lb * 128) + lane_off16;
            unsigned char *dst = ring + s * kSlotBytes; // wave-uniform; lane L lands at dst + size * L
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            // the instruction offset moves the global AND the LDS address (tools/dma_offset_probe.hip): the second
            // kilobyte needs no second address pair and no second M0 write
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 1024, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(Kb + lb * 16) + lane_off4),
                                             (__attribute__((address_space(3))) void *)(dst + 2048), 4, 0, 0);
        };
        // One step = TWO candidate blocks (slots s, s + 1; s even).  On this chip VALU issue does
        // not overlap these MFMAs (a step of 16 MFMAs + 32 v_max_f32 measures 381 cycles, not
        // 256), so the epilogue is cut to ONE v_max3_f32 per two results: max3(best, blockA,
        // blockB).  The column keys sit in LDS as 16 keys x 4 copies, so one ds_read_b128 IS the
        // 4-register accumulator-init tuple (no v_mov broadcast).
        // The slots are read with hand-written ds_reads: hipcc would put its own vmcnt(0) in front
        // of a compiler-visible LDS read that may alias a pending DMA.  Reads and their wait are
        // ONE asm block: no register is in flight outside it.
        auto step2 = [&](int s, int t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (kRing - 2)) : "memory");
            u32x4 q0, q1, p0, p1;
            v4f cv, dv;
            asm volatile("ds_read_b128 %0, %6 offset:%8\n\tds_read_b128 %1, %6 offset:%9\n\t"
                         "ds_read_b128 %2, %7 offset:%10\n\t"
                         "ds_read_b128 %3, %6 offset:%11\n\tds_read_b128 %4, %6 offset:%12\n\t"
                         "ds_read_b128 %5, %7 offset:%13\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(cv), "=&v"(p0), "=&v"(p1), "=&v"(dv)
                         : "v"(lds16), "v"(ldsk), "n"(s * kSlotBytes), "n"(s * kSlotBytes + 1024),
                           "n"(s * kSlotBytes + 2048), "n"((s + 1) * kSlotBytes), "n"((s + 1) * kSlotBytes + 1024),
                           "n"((s + 1) * kSlotBytes + 2048)
                         : "memory");
            const v8i b0 = (v8i){(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w, 0, 0, 0, 0};
            const v8i b1 = (v8i){(int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w, 0, 0, 0, 0};
            const v8i d0 = (v8i){(int)p0.x, (int)p0.y, (int)p0.z, (int)p0.w, 0, 0, 0, 0};
            const v8i d1 = (v8i){(int)p1.x, (int)p1.y, (int)p1.z, (int)p1.w, 0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < RB; m++) {
                // cbsz = blgp = 4: e2m1 operands; scale arguments 0, 0 select the unscaled instruction
                v4f acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][0], b0, cv, 4, 4, 0, 0, 0, 0);
                v4f acd = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][0], d0, dv, 4, 4, 0, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][1], b1, acc, 4, 4, 0, 0, 0, 0);
                acd = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][1], d1, acd, 4, 4, 0, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++)
                    best[m][r] = __builtin_fmaxf(__builtin_fmaxf(best[m][r], acc[r]), acd[r]); // v_max3_f32
            }
            // the slots are refilled only after their reads: the MFMAs above consumed them
            issue(s, t + kRing);
            issue(s + 1, t + 1 + kRing);
        };
        static_assert(kRing == 4, "the loop below is written for 4 slots = 2 double steps");
#pragma unroll
        for (int s = 0; s < kRing; s++) issue(s, s);
        // a block past the wave's last one is the clamped last block again: maxima are idempotent
        int t = 0;
        for (; t + 4 <= T; t += 4) {
            step2(0, t);
            step2(2, t + 2);
        }
        if (t < T) step2(0, t);
        if (t + 2 < T) step2(2, t + 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // drain the clamped tail DMAs
        __syncthreads();                                  // every wave is done with its ring: s_best reuses it
        // C/D layout: lane holds rows 4 * (lane >> 4) + r of each 16-row fragment, column lane & 15
#pragma unroll
        for (int m = 0; m < RB; m++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                s_best[(m * 16 + 4 * (lane >> 4) + r) * kMmaLds + wv * 16 + (lane & 15)] = best[m][r];
    }
    __syncthreads();
    // the 16 NW partial maxima of a query row (NW waves x 16 column classes): two threads per row with 4 waves (32 floats
    // each, met by one shuffle), one thread per row with 2 waves, one thread for two rows with 1 wave
    constexpr int kTPR = NW >= 4 ? 2 : 1, kFPT = 16 * NW / kTPR, kRowStep = 64 * NW / kTPR;
    for (int row = (int)threadIdx.x / kTPR; row < kMmaRows; row += kRowStep) {
        const int part = (int)threadIdx.x % kTPR;
        float v = -3e38f;
        if (active) {
            const float4 *src = reinterpret_cast<const float4 *>(s_best + row * kMmaLds + part * kFPT);
#pragma unroll
            for (int k = 0; k < kFPT / 4; k++) {
                const float4 x = src[k];
                v = fmaxf(fmaxf(v, fmaxf(x.x, x.y)), fmaxf(x.z, x.w));
            }
        }
        if (kTPR == 2) v = fmaxf(v, __shfl_xor(v, 1));
        const int i = row0 + row;
        if (part == 0 && i < cap) {
            bool ok = active && i < nA && v > -1e29f;
            int bj = -1, bd = -1;
            if (ok) {
                const int nk = -(int)(v * (2 * kMmaS));                                       // S * (dist - |a_i|) + j
                const int pop_a = (-(int)(mkey[(size_t)p * capP + i] * (2 * kMmaS))) >> 14; // |a_i|
                bj = nk & (kMmaS - 1);
                bd = pop_a + (nk >> 14);
                ok = bd <= max_dist;
            }
            out_idx[(size_t)pk * cap + i] = ok ? bj : -1;
            if (out_dist) out_dist[(size_t)pk * cap + i] = ok ? bd : -1;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same product with the candidate blocks streamed through REGISTERS and the epilogue UNDER the MFMAs.
//
// What round 4's counters said about match_mfma_kernel: 512 MFMAs x 16 cycles + 856 other vector instructions x ~4.6 =
// the 12 200 cycles a wave takes -- the two kinds of work add, they do not overlap (the 0.156 "co-execution" the counter
// shows is the MFMAs' own issue cycles).  A 16-cycle MFMA holds the SIMD's vector issue for 8 of its 16 cycles
// (MI355X_MICROARCH.md, row 'vector-instruction ISSUE cost'); a v_max3_f32 costs 4: one fold per MFMA FITS in the gap -- if it
// is an independent instruction that sits right behind the MFMA in the SAME wave's stream.  In the round-4 loop the four
// folds of a row block came after its four MFMAs and depended on them, so every wave alternated "wait for the pipe" with
// "fold", and three such waves per SIMD queued on the matrix pipe instead of filling each other's gaps.
// Here the stream is software-pipelined by one row block: MFMA (block m), fold (block m - 1), MFMA, fold, ... pinned with
// sched_group_barrier, the last block's folds carried into the next step.  And the candidate fragments no longer pass
// through LDS: the fragment image written by match_expand_kernel is lane-linear, so a plain 16-byte global load per lane
// IS the B operand -- no LDS-DMA, no M0 writes, no ds_read, no counted vmcnt by hand (hipcc counts plain loads itself);
// the ring is three register slots of two blocks, filled two steps ahead.
// Same arithmetic, same keys, same reduction through LDS at the end: bit-identical results (the 111 matcher tests).
template <int RB, int NW, int WPE>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
match_mfma2_kernel(const uint4 *__restrict__ mexp, const float *__restrict__ mkey, const float4 *__restrict__ mkey4,
                   const int32_t *__restrict__ counts, int cap, int capP, int first, int stride, int max_dist,
                   int32_t *__restrict__ out_idx, int32_t *__restrict__ out_dist)
{
    constexpr int kMmaLds = mma_lds_row(NW);
    constexpr int kMmaRows = 16 * RB;
    __shared__ __attribute__((aligned(16))) float s_best[kMmaRows * kMmaLds];
    int pk, blk;
    xcd_remap(gridDim.x, gridDim.y, &pk, &blk); // all query blocks of a pair share one L2
    const int p = first + pk * stride;
    const int nA = clamp_count(counts[p], cap), nB = clamp_count(counts[p + 1], cap);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int row0 = blk * kMmaRows;
    const uint4 *__restrict__ Ea = mexp + (size_t)p * capP * 8;
    const uint4 *__restrict__ Eb = mexp + (size_t)(p + 1) * capP * 8;
    const float4 *__restrict__ Kb = mkey4 + (size_t)(p + 1) * capP;
    const bool active = row0 < nA && nB > 0; // block-uniform

    if (active) {
        v8i a[RB][2];
#pragma unroll
        for (int m = 0; m < RB; m++) {
            const int ab = (row0 >> 4) + m;
            const bool in = ab * 16 < capP;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const uint4 q = in ? Ea[(size_t)ab * 128 + ks * 64 + lane] : make_uint4(0, 0, 0, 0);
                a[m][ks] = (v8i){(int)q.x, (int)q.y, (int)q.z, (int)q.w, 0, 0, 0, 0};
            }
        }
        v4f best[RB];
#pragma unroll
        for (int m = 0; m < RB; m++) best[m] = (v4f){-3e38f, -3e38f, -3e38f, -3e38f};
        const int nBb = (nB + 15) >> 4;
        const int T = wv < nBb ? (nBb - wv + NW - 1) / NW : 0; // candidate blocks of this wave: wv, wv + NW, ...
        const int S = (T + 1) >> 1;                               // steps of two blocks
        struct Blk {
            uint4 b0, b1; // the two k-steps of the fragment
            float4 k;     // the column key, four times: the accumulator-init tuple
        };
        // a block past the wave's last one is the clamped last block again: maxima are idempotent
        // buffer loads: the frame's fragment image and its key tuples behind two 128-bit descriptors, the block as the
        // instruction's SCALAR offset, the lane as a constant 32-bit vector offset -- no address arithmetic on the vector ALU
        const __amdgpu_buffer_rsrc_t rE = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(Eb), 0, capP * 128, 0x00020000);
        const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(Kb), 0, capP * 16, 0x00020000);
        const int lane16 = lane * 16, col16 = (lane & 15) * 16;
        const int diag_nb = max_dist < -1000 ? 1 : nBb; // TEMPORARY diagnostic
        auto load = [&](int t) {
            int lb = wv + NW * t;
            lb = lb < diag_nb ? lb : diag_nb - 1;
            Blk r;
            r.b0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rE, lane16, lb * 2048, 0));
            r.b1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rE, lane16 + 1024, lb * 2048, 0));
            r.k = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rK, col16, lb * 256, 0));
            return r;
        };
        // the two row blocks whose folds are still owed: the fold of block m - 2 sits under the MFMAs of block m (one block
        // of lag is not enough: its last MFMA would be the instruction right in front of the fold that reads it)
        const v4f kLow = (v4f){-3e38f, -3e38f, -3e38f, -3e38f};
        v4f p1c = kLow, p1d = kLow, p2c = kLow, p2d = kLow;
        // SPREAD: the six loads that refill slot (N0, N1) go out one per row block instead of six in a row
        auto step = [&](const Blk &X, const Blk &Y, Blk *N0, Blk *N1, int tn) {
            int lbn0 = wv + NW * tn, lbn1 = wv + NW * (tn + 1);
            lbn0 = lbn0 < nBb ? lbn0 : nBb - 1;
            lbn1 = lbn1 < nBb ? lbn1 : nBb - 1;
            const v8i b0 = (v8i){(int)X.b0.x, (int)X.b0.y, (int)X.b0.z, (int)X.b0.w, 0, 0, 0, 0};
            const v8i b1 = (v8i){(int)X.b1.x, (int)X.b1.y, (int)X.b1.z, (int)X.b1.w, 0, 0, 0, 0};
            const v8i d0 = (v8i){(int)Y.b0.x, (int)Y.b0.y, (int)Y.b0.z, (int)Y.b0.w, 0, 0, 0, 0};
            const v8i d1 = (v8i){(int)Y.b1.x, (int)Y.b1.y, (int)Y.b1.z, (int)Y.b1.w, 0, 0, 0, 0};
            const v4f cv = (v4f){X.k.x, X.k.y, X.k.z, X.k.w}, dv = (v4f){Y.k.x, Y.k.y, Y.k.z, Y.k.w};
#pragma unroll
            for (int m = 0; m < RB; m++) {
                v4f &fold = best[(m + RB - 2) % RB]; // m = 0, 1: the previous step's last two row blocks
                // cbsz = blgp = 4: e2m1 operands; scale arguments 0, 0 select the unscaled instruction
                v4f acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][0], b0, cv, 4, 4, 0, 0, 0, 0);
                fold[0] = __builtin_fmaxf(__builtin_fmaxf(fold[0], p2c[0]), p2d[0]); // v_max3_f32
                v4f acd = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][0], d0, dv, 4, 4, 0, 0, 0, 0);
                fold[1] = __builtin_fmaxf(__builtin_fmaxf(fold[1], p2c[1]), p2d[1]);
                acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][1], b1, acc, 4, 4, 0, 0, 0, 0);
                fold[2] = __builtin_fmaxf(__builtin_fmaxf(fold[2], p2c[2]), p2d[2]);
                acd = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][1], d1, acd, 4, 4, 0, 0, 0, 0);
                fold[3] = __builtin_fmaxf(__builtin_fmaxf(fold[3], p2c[3]), p2d[3]);
                p2c = p1c;
                p2d = p1d;
                p1c = acc;
                p1d = acd;
                // the emitted order: one matrix instruction, one vector instruction, four times
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); // VALU
                }
                if (N0) {
                    if (m == 0) N0->b0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rE, lane16, lbn0 * 2048, 0));
                    if (m == 1) N0->b1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rE, lane16 + 1024, lbn0 * 2048, 0));
                    if (m == 2) N0->k = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rK, col16, lbn0 * 256, 0));
                    if (m == 3) N1->b0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rE, lane16, lbn1 * 2048, 0));
                    if (m == 4) N1->b1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rE, lane16 + 1024, lbn1 * 2048, 0));
                    if (m == 5) N1->k = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rK, col16, lbn1 * 256, 0));
                }
                // ... and nothing crosses from one row block to the next: each is scheduled on its own (left free, the
                // scheduler lets the pattern decay over a three-step loop body)
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // three register slots of two blocks each, filled two steps ahead; whole groups of three steps first (no exit
        // inside the loop: the compiler then counts the outstanding loads itself, vmcnt(n) at the first use of a slot)
        Blk s0a = load(0), s0b = load(1), s1a = load(2), s1b = load(3), s2a, s2b;
        int i = 0;
        if (max_dist < -3000) { // TEMPORARY diagnostic: no loads inside the loop at all
            s2a = load(4);
            s2b = load(5);
            for (; i + 3 <= S; i += 3) {
                step(s0a, s0b, nullptr, nullptr, 0);
                step(s1a, s1b, nullptr, nullptr, 0);
                step(s2a, s2b, nullptr, nullptr, 0);
            }
        }
        for (; i + 3 <= S; i += 3) {
            step(s0a, s0b, &s2a, &s2b, 2 * i + 4);
            step(s1a, s1b, &s0a, &s0b, 2 * i + 6);
            step(s2a, s2b, &s1a, &s1b, 2 * i + 8);
        }
        if (i < S) step(s0a, s0b, nullptr, nullptr, 0);     // the remaining one or two steps: their blocks are already in flight
        if (i + 1 < S) step(s1a, s1b, nullptr, nullptr, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            best[RB - 2][r] = __builtin_fmaxf(__builtin_fmaxf(best[RB - 2][r], p2c[r]), p2d[r]);
            best[RB - 1][r] = __builtin_fmaxf(__builtin_fmaxf(best[RB - 1][r], p1c[r]), p1d[r]);
        }
        // C/D layout: lane holds rows 4 * (lane >> 4) + r of each 16-row fragment, column lane & 15
#pragma unroll
        for (int m = 0; m < RB; m++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                s_best[(m * 16 + 4 * (lane >> 4) + r) * kMmaLds + wv * 16 + (lane & 15)] = best[m][r];
    }
    __syncthreads();
    constexpr int kTPR = NW >= 4 ? 2 : 1, kFPT = 16 * NW / kTPR, kRowStep = 64 * NW / kTPR;
    for (int row = (int)threadIdx.x / kTPR; row < kMmaRows; row += kRowStep) {
        const int part = (int)threadIdx.x % kTPR;
        float v = -3e38f;
        if (active) {
            const float4 *src = reinterpret_cast<const float4 *>(s_best + row * kMmaLds + part * kFPT);
#pragma unroll
            for (int k = 0; k < kFPT / 4; k++) {
                const float4 x = src[k];
                v = fmaxf(fmaxf(v, fmaxf(x.x, x.y)), fmaxf(x.z, x.w));
            }
        }
        if (kTPR == 2) v = fmaxf(v, __shfl_xor(v, 1));
        const int i = row0 + row;
        if (part == 0 && i < cap) {
            bool ok = active && i < nA && v > -1e29f;
            int bj = -1, bd = -1;
            if (ok) {
                const int nk = -(int)(v * (2 * kMmaS));                                       // S * (dist - |a_i|) + j
                const int pop_a = (-(int)(mkey[(size_t)p * capP + i] * (2 * kMmaS))) >> 14; // |a_i|
                bj = nk & (kMmaS - 1);
                bd = pop_a + (nk >> 14);
                ok = bd <= max_dist;
            }
            out_idx[(size_t)pk * cap + i] = ok ? bj : -1;
            if (out_dist) out_dist[(size_t)pk * cap + i] = ok ? bd : -1;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 5, second form: the block tile.  match_mfma2_kernel showed (tools/r5_match_diag.py) that with the folds under the
// MFMAs the loop reaches the matrix pipe's pace when the candidate operands are free (0.85 ms per 4096 frames), and that
// what it pays on top -- 0.35 ms -- is the ISSUE of its six 1-KB vector loads per step, the same whether they hit L1 or L2.
// Every wave of a workgroup streamed the whole candidate frame for itself.  Here the four waves of a workgroup hold
// DIFFERENT queries (4 x 128 = 512 per workgroup) and consume the SAME candidate blocks: one ring of three slots in LDS,
// filled by LDS-DMA -- each wave brings a quarter of a step's fragments -- and read by all four with ds_read_b128.  Vector
// memory instructions per MFMA fall to a quarter (and L2 -> CU traffic with them); a wave's partial maxima never leave it
// (16 column classes met by four lane shuffles at the end), so there is no cross-wave reduction through LDS either.
// One s_barrier per step orders both directions: a wave passes it only after ITS pieces of this step have landed
// (vmcnt) and after it has consumed the previous step's slot, so behind the barrier the slot of this step is complete and
// the slot of the step before may be refilled.  Same keys, same folds: bit-identical results.
constexpr int kTileRing = 3;                     // slots; a slot = one step = four candidate blocks
constexpr int kTileBlk = 2048 + 256;             // fragment image + 16 keys x 4 copies
constexpr int kTileSlot = 4 * kTileBlk;
template <int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
match_mfma3_kernel(const uint4 *__restrict__ mexp, const float *__restrict__ mkey, const int32_t *__restrict__ counts,
                   int cap, int capP, int first, int stride, int max_dist, int32_t *__restrict__ out_idx,
                   int32_t *__restrict__ out_dist)
{
    constexpr int RB = 8;
    __shared__ __attribute__((aligned(16))) unsigned char s_ring[kTileRing * kTileSlot];
    int pk, blk;
    xcd_remap(gridDim.x, gridDim.y, &pk, &blk); // all query tiles of a pair share one L2
    const int p = first + pk * stride;
    const int nA = clamp_count(counts[p], cap), nB = clamp_count(counts[p + 1], cap);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int row0 = blk * 512 + wv * 128; // this wave's 128 queries
    const uint4 *__restrict__ Ea = mexp + (size_t)p * capP * 8;
    const uint4 *__restrict__ Eb = mexp + (size_t)(p + 1) * capP * 8;
    const float *__restrict__ Kb = mkey + (size_t)(p + 1) * capP;
    if (!(blk * 512 < nA && nB > 0)) { // workgroup-uniform: nothing to match
        for (int i = blk * 512 + (int)threadIdx.x; i < blk * 512 + 512 && i < cap; i += 256) {
            out_idx[(size_t)pk * cap + i] = -1;
            if (out_dist) out_dist[(size_t)pk * cap + i] = -1;
        }
        return;
    }
    v8i a[RB][2];
#pragma unroll
    for (int m = 0; m < RB; m++) {
        const int ab = (row0 >> 4) + m;
        const bool in = ab * 16 < capP;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const uint4 q = in ? Ea[(size_t)ab * 128 + ks * 64 + lane] : make_uint4(0, 0, 0, 0);
            a[m][ks] = (v8i){(int)q.x, (int)q.y, (int)q.z, (int)q.w, 0, 0, 0, 0};
        }
    }
    v4f best[RB];
#pragma unroll
    for (int m = 0; m < RB; m++) best[m] = (v4f){-3e38f, -3e38f, -3e38f, -3e38f};
    const int nBb = (nB + 15) >> 4;
    const int S = (nBb + 3) >> 2; // steps of four blocks, the same for every wave
    const uint32_t ring_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)s_ring;
    const uint32_t lds16 = ring_lds + lane * 16, ldsk = ring_lds + (lane & 15) * 16; // this lane's read addresses
    const uint32_t lane_off16 = (uint32_t)lane * 16u, lane_off4 = (uint32_t)(lane >> 2) * 4u;
    // this wave's three pieces of step t: block wv of the step -- its two k-steps and its keys.  Every wave has exactly
    // three DMAs per step in flight, so one counted wait serves all.
    auto issue = [&](int slot, int t) {
        int lb = 4 * t + wv;
        lb = lb < nBb ? lb : nBb - 1; // past the end: the last block again (maxima are idempotent)
        unsigned char *dst = s_ring + slot * kTileSlot + wv * kTileBlk;
        const char *src = reinterpret_cast<const char *>(Eb + (size_t)lb * 128) + lane_off16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 1024, 0); // (the offset moves both sides)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(Kb + lb * 16) + lane_off4),
                                         (__attribute__((address_space(3))) void *)(dst + 2048), 4, 0, 0);
    };
    const v4f kLow = (v4f){-3e38f, -3e38f, -3e38f, -3e38f};
    v4f p1c = kLow, p1d = kLow, p2c = kLow, p2d = kLow; // the two row blocks whose folds are still owed (match_mfma2_kernel)
    // 32 MFMAs over two candidate blocks with the folds of two row blocks ago between them
    auto mma2 = [&](const u32x4 &q0, const u32x4 &q1, const v4f &cv, const u32x4 &r0, const u32x4 &r1, const v4f &dv) {
        const v8i b0 = (v8i){(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w, 0, 0, 0, 0};
        const v8i b1 = (v8i){(int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w, 0, 0, 0, 0};
        const v8i d0 = (v8i){(int)r0.x, (int)r0.y, (int)r0.z, (int)r0.w, 0, 0, 0, 0};
        const v8i d1 = (v8i){(int)r1.x, (int)r1.y, (int)r1.z, (int)r1.w, 0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < RB; m++) {
            v4f &fold = best[(m + RB - 2) % RB]; // m = 0, 1: the previous pass's last two row blocks
            v4f acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][0], b0, cv, 4, 4, 0, 0, 0, 0);
            fold[0] = __builtin_fmaxf(__builtin_fmaxf(fold[0], p2c[0]), p2d[0]); // v_max3_f32
            v4f acd = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][0], d0, dv, 4, 4, 0, 0, 0, 0);
            fold[1] = __builtin_fmaxf(__builtin_fmaxf(fold[1], p2c[1]), p2d[1]);
            acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][1], b1, acc, 4, 4, 0, 0, 0, 0);
            fold[2] = __builtin_fmaxf(__builtin_fmaxf(fold[2], p2c[2]), p2d[2]);
            acd = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][1], d1, acd, 4, 4, 0, 0, 0, 0);
            fold[3] = __builtin_fmaxf(__builtin_fmaxf(fold[3], p2c[3]), p2d[3]);
            p2c = p1c;
            p2d = p1d;
            p1c = acc;
            p1d = acd;
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) { // the emitted order: one matrix instruction, one vector instruction, four times
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0); // nothing crosses from one row block to the next
        }
    };
    auto step = [&](auto slot_c, int t) {
        constexpr int s = decltype(slot_c)::value;
        // my pieces of this step have landed (three younger DMAs may still be in flight: the next step's); then everybody's
        asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
        issue((s + 2) % kTileRing, t + 2); // refills the slot the previous step read: every wave is past those reads
        u32x4 q0, q1, r0, r1, u0, u1, w0, w1;
        v4f cv, dv, ev, fv;
        // hand-written reads: hipcc would put its own vmcnt(0) in front of a compiler-visible LDS read that may alias a DMA.
        // Blocks 0, 1 are waited for; the reads of blocks 2, 3 stay in flight under the first 32 MFMAs.
        asm volatile("ds_read_b128 %0, %6 offset:%8\n\tds_read_b128 %1, %6 offset:%9\n\t"
                     "ds_read_b128 %2, %7 offset:%10\n\t"
                     "ds_read_b128 %3, %6 offset:%11\n\tds_read_b128 %4, %6 offset:%12\n\t"
                     "ds_read_b128 %5, %7 offset:%13\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(cv), "=&v"(r0), "=&v"(r1), "=&v"(dv)
                     : "v"(lds16), "v"(ldsk), "n"(s * kTileSlot), "n"(s * kTileSlot + 1024), "n"(s * kTileSlot + 2048),
                       "n"(s * kTileSlot + kTileBlk), "n"(s * kTileSlot + kTileBlk + 1024), "n"(s * kTileSlot + kTileBlk + 2048)
                     : "memory");
        asm volatile("ds_read_b128 %0, %6 offset:%8\n\tds_read_b128 %1, %6 offset:%9\n\t"
                     "ds_read_b128 %2, %7 offset:%10\n\t"
                     "ds_read_b128 %3, %6 offset:%11\n\tds_read_b128 %4, %6 offset:%12\n\t"
                     "ds_read_b128 %5, %7 offset:%13"
                     : "=&v"(u0), "=&v"(u1), "=&v"(ev), "=&v"(w0), "=&v"(w1), "=&v"(fv)
                     : "v"(lds16), "v"(ldsk), "n"(s * kTileSlot + 2 * kTileBlk), "n"(s * kTileSlot + 2 * kTileBlk + 1024),
                       "n"(s * kTileSlot + 2 * kTileBlk + 2048), "n"(s * kTileSlot + 3 * kTileBlk),
                       "n"(s * kTileSlot + 3 * kTileBlk + 1024), "n"(s * kTileSlot + 3 * kTileBlk + 2048)
                     : "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma2(q0, q1, cv, r0, r1, dv);
        // the second half's operands: the wait is tied to the registers, so nothing reads (or copies) them before it
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(u0), "+v"(u1), "+v"(ev), "+v"(w0), "+v"(w1), "+v"(fv)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        mma2(u0, u1, ev, w0, w1, fv);
    };
    issue(0, 0);
    issue(1, 1);
    int i = 0;
    for (; i + 3 <= S; i += 3) {
        step(std::integral_constant<int, 0>{}, i);
        step(std::integral_constant<int, 1>{}, i + 1);
        step(std::integral_constant<int, 2>{}, i + 2);
    }
    if (i < S) step(std::integral_constant<int, 0>{}, i); // S is the same for the four waves: the barriers stay matched
    if (i + 1 < S) step(std::integral_constant<int, 1>{}, i + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the clamped tail DMAs land before the workgroup's LDS is released
#pragma unroll
    for (int r = 0; r < 4; r++) {
        best[RB - 2][r] = __builtin_fmaxf(__builtin_fmaxf(best[RB - 2][r], p2c[r]), p2d[r]);
        best[RB - 1][r] = __builtin_fmaxf(__builtin_fmaxf(best[RB - 1][r], p1c[r]), p1d[r]);
    }
    // C/D layout: lane holds rows 4 * (lane >> 4) + r of each 16-row fragment, column class lane & 15: the row's maximum is
    // the maximum over its 16 lanes
#pragma unroll
    for (int m = 0; m < RB; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float v = best[m][r];
            v = fmaxf(v, __shfl_xor(v, 1));
            v = fmaxf(v, __shfl_xor(v, 2));
            v = fmaxf(v, __shfl_xor(v, 4));
            v = fmaxf(v, __shfl_xor(v, 8));
            const int i_q = row0 + m * 16 + 4 * (lane >> 4) + r;
            if ((lane & 15) == 0 && i_q < cap) {
                bool ok = i_q < nA && v > -1e29f;
                int bj = -1, bd = -1;
                if (ok) {
                    const int nk = -(int)(v * (2 * kMmaS));                                         // S * (dist - |a_i|) + j
                    const int pop_a = (-(int)(mkey[(size_t)p * capP + i_q] * (2 * kMmaS))) >> 14; // |a_i|
                    bj = nk & (kMmaS - 1);
                    bd = pop_a + (nk >> 14);
                    ok = bd <= max_dist;
                }
                out_idx[(size_t)pk * cap + i_q] = ok ? bj : -1;
                if (out_dist) out_dist[(size_t)pk * cap + i_q] = ok ? bd : -1;
            }
        }
}

void launch_match_mfma(const orbfe_keypoint *d_records, const int32_t *d_counts, int n_frames, int n_pairs, int first,
                       int stride, int cap, int capP, int max_dist, uint4 *mexp, float *mkey, float4 *mkey4, int32_t *d_idx,
                       int32_t *d_dist, hipStream_t stream)
{
    static_assert(kMmaS == kMmaMaxKeypoints, "key packing");
    hipLaunchKernelGGL(match_expand_kernel, dim3((capP * 8 + 255) / 256, n_frames), dim3(256), 0, stream, d_records,
                       d_counts, cap, capP, mexp, mkey, mkey4);
    if (const char *v2 = getenv("ORBFE_MATCH_V2")) { // round-5 A/B: "<waves per workgroup><waves per SIMD>", e.g. 41, 42, 82
        const dim3 grid((capP + 127) / 128, n_pairs);
#define ORBFE_V2(NWV, WPEV)                                                                                              \
    hipLaunchKernelGGL((match_mfma2_kernel<8, NWV, WPEV>), grid, dim3(64 * NWV), 0, stream, mexp, mkey, mkey4, d_counts, cap, \
                       capP, first, stride, max_dist, d_idx, d_dist)
        if (!strcmp(v2, "41")) { ORBFE_V2(4, 1); return; }
        if (!strcmp(v2, "42")) { ORBFE_V2(4, 2); return; }
        if (!strcmp(v2, "43")) { ORBFE_V2(4, 3); return; }
        if (!strcmp(v2, "82")) { ORBFE_V2(8, 2); return; }
        if (!strcmp(v2, "21")) { ORBFE_V2(2, 1); return; }
        if (!strcmp(v2, "22")) { ORBFE_V2(2, 2); return; }
#undef ORBFE_V2
        if (v2[0] == 't') { // the block tile: t2 / t3 = waves per SIMD
            const dim3 grid((capP + 511) / 512, n_pairs);
            if (v2[1] == '3')
                hipLaunchKernelGGL((match_mfma3_kernel<3>), grid, dim3(256), 0, stream, mexp, mkey, d_counts, cap, capP, first, stride,
                                   max_dist, d_idx, d_dist);
            else
                hipLaunchKernelGGL((match_mfma3_kernel<2>), grid, dim3(256), 0, stream, mexp, mkey, d_counts, cap, capP, first, stride,
                                   max_dist, d_idx, d_dist);
            return;
        }
        // 192 queries per workgroup (12 row blocks): a third fewer candidate loads per MFMA, 2 waves per SIMD still fit
        if (!strcmp(v2, "c22")) {
            hipLaunchKernelGGL((match_mfma2_kernel<12, 2, 2>), dim3((capP + 191) / 192, n_pairs), dim3(128), 0, stream, mexp, mkey, mkey4,
                               d_counts, cap, capP, first, stride, max_dist, d_idx, d_dist);
            return;
        }
        if (!strcmp(v2, "c42")) {
            hipLaunchKernelGGL((match_mfma2_kernel<12, 4, 2>), dim3((capP + 191) / 192, n_pairs), dim3(256), 0, stream, mexp, mkey, mkey4,
                               d_counts, cap, capP, first, stride, max_dist, d_idx, d_dist);
            return;
        }
    }
    const char *nw = getenv("ORBFE_MATCH_WAVES"); // "2": the 18 KB / 2-wave form (A/B and co-residency probes)
    if (nw && nw[0] == '2')
        hipLaunchKernelGGL((match_mfma_kernel<8, 2>), dim3((capP + 127) / 128, n_pairs), dim3(128), 0, stream, mexp, mkey,
                           d_counts, cap, capP, first, stride, max_dist, d_idx, d_dist);
    else if (nw && nw[0] == '1')
        hipLaunchKernelGGL((match_mfma_kernel<8, 1>), dim3((capP + 127) / 128, n_pairs), dim3(64), 0, stream, mexp, mkey,
                           d_counts, cap, capP, first, stride, max_dist, d_idx, d_dist);
    else if ((long long)n_pairs * ((capP + 127) / 128) >= 256) // at least one 128-query workgroup per CU
        hipLaunchKernelGGL((match_mfma_kernel<8, 4>), dim3((capP + 127) / 128, n_pairs), dim3(256), 0, stream, mexp, mkey,
                           d_counts, cap, capP, first, stride, max_dist, d_idx, d_dist);
    else
        hipLaunchKernelGGL((match_mfma_kernel<4, 4>), dim3((capP + 63) / 64, n_pairs), dim3(256), 0, stream, mexp, mkey,
                           d_counts, cap, capP, first, stride, max_dist, d_idx, d_dist);
}

} // namespace orbfe
