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
// epilogue as v_max_f32 the step structure tops out at 12.5 T, with v_max3_f32 at 14.7 T; this
// kernel reaches ~12 T.  It serves small and medium calls; large calls take (3) match_tile_kernel
// below, which needs neither (1) nor its scratch.
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
                    uint4 *__restrict__ mexp, float *__restrict__ mkey)
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
    if (w == 0) mkey[(size_t)f * capP + i] = live ? -(float)(pop * kMmaS + i) * kKeyUnit : -1e30f;
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
// (3) match_tile_kernel (round 5): the block tile that expands its own operands -- what large calls run.
//
// What round 4's counters said about match_mfma_kernel: 512 MFMAs x 16 cycles + 856 other vector instructions x ~4.6 are
// the 12 200 cycles a wave takes; the two kinds of work ADD (the 0.156 "co-execution" its counter shows is the MFMAs' own
// issue cycles).  tools/mfma_fold_probe.hip measures what the chip allows: in one wave's stream an MFMA followed by an
// INDEPENDENT v_max3_f32 costs 19 cycles against 18 for the bare MFMA (two folds: 19.5) -- a 16-cycle MFMA holds the
// SIMD's vector issue for only part of its time (MI355X_MICROARCH.md, row 'vector-instruction ISSUE cost').  In the
// round-4 loop the four folds of a row block came right behind its four MFMAs and depended on them, so a wave alternated
// "wait for the pipe" with "fold" and three such waves per SIMD queued on the matrix pipe.  Three changes, each measured
// (tools/experiments/README.md, matcher_forms.patch keeps the two intermediate kernels):
//  a. the folds are software-pipelined by TWO row blocks (one is not enough: the fold would read the MFMA issued right in
//     front of it) and the order MFMA, fold, MFMA, fold is pinned -- sched_group_barrier per instruction pair,
//     sched_barrier per row block; left free, hipcc's scheduler lets the pattern decay over an unrolled loop body;
//  b. with a. the loop runs at the matrix pipe's pace when its operands are free (0.85 ms per 4096 frames) and pays
//     0.35 ms for the ISSUE of the six 1-KB vector loads a wave needs per 32 MFMAs, L1 hit or not (every wave streamed the
//     whole candidate frame for itself: 36 B per cycle and CU against the L1's 64).  So the four waves of a workgroup hold
//     DIFFERENT queries (4 x 128 = 512 per workgroup) and consume the SAME candidate blocks from a ring in LDS: vector
//     memory instructions and L2 -> CU bytes per MFMA fall to a quarter; a wave's partial maxima never leave it (16 column
//     classes met by four lane shuffles at the end), so there is no cross-wave reduction either;
//  c. every candidate block now enters a workgroup ONCE, so the workgroup can afford to build it: match_expand_kernel's
//     e2m1 scratch (128 B + a key per keypoint, written once and read as the A and as the B operand: the stage moved 5.2 x
//     its algorithmic bytes, and the kernel was 0.29 ms of every step) is not used.  Wave w takes block w of a step --
//     two dwords of descriptor per lane (lane L: words L >> 4 and (L >> 4) + 4 of keypoint L & 15, exactly the fragment
//     entries [k-step 0 / 1][lane L]), eight dwords of e2m1 nibbles made by ten ANDs / shifts in the MFMA gaps next to the folds, the
//     popcount for the column key met by two lane shuffles -- and writes fragment image and key tuples into the ring with
//     ds_write_b128.  The queries are expanded the same way in the prologue.  No LDS-DMA, no scratch: 32 bytes of HBM
//     traffic per keypoint and workgroup instead of 128 + 128 per keypoint and call plus the scratch's own round trip.
// The ring has two slots of four blocks.  A slot is written one step after its last read and read one step after that;
// ONE s_barrier per step, in the middle, orders both (see `step`).  All LDS traffic of the loop is hand-written asm: the
// compiler must neither wait for the prefetched global loads at a barrier nor reorder the ring's reads and writes.  (And
// no divergent branch in the loop: an `if (lane < 16)` around the key store split the body into blocks and cost 335
// spilled registers; the four lanes of a keypoint hold the same key and all write it.)
// Same keys, same folds, same winner: bit-identical to the other forms and to the oracle (the 111 matcher tests run on
// every form).  Measured: match stage 1.654 -> 1.522 ms per 4096 frames, step 5.99 -> 5.79 ms.
constexpr int kTileBlk = 2048 + 256; // one candidate block in the ring: fragment image + 16 keys x 4 copies
constexpr int kTileSlot = 4 * kTileBlk;
// The tile kernel's own order of the 256 bits inside K (any order serves as long as queries and candidates share it): dword d of
// a lane's fragment entry holds bits d, d + 4, ..., d + 28 of its descriptor word, one per nibble, so the CANDIDATE side of the
// expansion -- the one inside the loop -- is a plain AND per dword.  The e2m1 code a set bit carries is whatever the AND leaves
// (0x1 = 0.5, 0x2 = 1.0, 0x4 = 2.0; bit 3 of a nibble is the sign, so dword 3 is shifted down by one); the QUERY side, expanded
// once per kernel, carries the reciprocal, and every product of two set bits is exactly 1:
//     dword          0                    1                    2                    3
//     candidate      w & 0x11111111 (0.5) w & 0x22222222 (1.0) w & 0x44444444 (2.0) (w >> 1) & 0x44444444 (2.0)
//     query          2.0                  1.0                  0.5                  0.5
// Ten full-rate instructions per two descriptor words, where the bytewise spread above (bit b -> nibble b, the order the scratch
// of the other forms is defined in) compiles to 2 v_mul_u32_u24_sdwa + v_bitop3 + a quarter-rate v_mul_lo_u32 + v_and for each
// of the eight bytes (0.19 ms of the kernel's 1.59 per 4096-frame step, profiles/r05_tile_ablation.txt).
template <int D> __device__ __forceinline__ uint32_t tile_cand_dword(uint32_t w)
{
    return D == 0 ? w & 0x11111111u : D == 1 ? w & 0x22222222u : D == 2 ? w & 0x44444444u : (w >> 1) & 0x44444444u;
}
__device__ __forceinline__ u32x4 tile_cand_word(uint32_t w)
{
    return (u32x4){tile_cand_dword<0>(w), tile_cand_dword<1>(w), tile_cand_dword<2>(w), tile_cand_dword<3>(w)};
}
// x summed over lanes L and L ^ 16 (L ^ 32): gfx950's row swaps are vector instructions -- v0.row1 <-> v1.row0, v0.row3 <-> v1.row2
// (v0.rows 2, 3 <-> v1.rows 0, 1), so with both operands = x the two results are the two rows' (halves') values in every lane
__device__ __forceinline__ int sum_lanes_xor16(int x)
{
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    return (int)(r[0] + r[1]);
}
__device__ __forceinline__ int sum_lanes_xor32(int x)
{
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    return (int)(r[0] + r[1]);
}
// The maximum over the 16 lanes of a DPP row, in every lane of it, by four v_max_f32 with DPP operands: lanes ^ 1 and ^ 2 inside a
// quad, then the mirrored half row (the other quad: all four of its lanes hold its maximum by now) and the mirrored row.  (The
// __shfl_xor this replaces is a ds_bpermute_b32 round trip per exchange: 128 of them in a row were the kernel's epilogue.)
template <int CTRL> __device__ __forceinline__ float max_with_dpp(float v)
{
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return __builtin_fmaxf(v, __builtin_bit_cast(float, o));
}
__device__ __forceinline__ float max_over_row16(float v)
{
    v = max_with_dpp<0xB1>(v);  // quad_perm [1, 0, 3, 2]
    v = max_with_dpp<0x4E>(v);  // quad_perm [2, 3, 0, 1]
    v = max_with_dpp<0x141>(v); // row_half_mirror
    return max_with_dpp<0x140>(v); // row_mirror
}
// colkey of candidate kp with |b| = pc (see (2) above); past the count (kp < 0): never wins
__device__ __forceinline__ float column_key(int pc, int kp) { return kp >= 0 ? -(float)(pc * kMmaS + kp) * kKeyUnit : -1e30f; }
__device__ __forceinline__ u32x4 tile_query_word(uint32_t w)
{
    return (u32x4){(w & 0x11111111u) << 2, w & 0x22222222u, (w & 0x44444444u) >> 2, (w >> 3) & 0x11111111u};
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
match_tile_kernel(const orbfe_keypoint *__restrict__ records, const int32_t *__restrict__ counts, int cap, int first,
                   int stride, int max_dist, int32_t *__restrict__ out_idx, int32_t *__restrict__ out_dist)
{
    constexpr int RB = 8;
    constexpr int kRing2 = 2;
    __shared__ __attribute__((aligned(16))) unsigned char s_ring[kRing2 * kTileSlot];
    __shared__ uint16_t s_popA[512]; // |a| of the workgroup's queries (0..256)
    int pk, blk;
    xcd_remap(gridDim.x, gridDim.y, &pk, &blk); // all query tiles of a pair share one L2
    const int p = first + pk * stride;
    const int nA = clamp_count(counts[p], cap), nB = clamp_count(counts[p + 1], cap);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int row0 = blk * 512 + wv * 128; // this wave's 128 queries
    if (!(blk * 512 < nA && nB > 0)) {     // workgroup-uniform: nothing to match
        for (int i = blk * 512 + (int)threadIdx.x; i < blk * 512 + 512 && i < cap; i += 256) {
            out_idx[(size_t)pk * cap + i] = -1;
            if (out_dist) out_dist[(size_t)pk * cap + i] = -1;
        }
        return;
    }
    // descriptor word `w` of record `kp` of frame f (the record is 13 dwords, the descriptor its dwords 5..12)
    const uint32_t *__restrict__ recA = reinterpret_cast<const uint32_t *>(records + (size_t)p * cap);
    const uint32_t *__restrict__ recB = reinterpret_cast<const uint32_t *>(records + (size_t)(p + 1) * cap);
    const int kq = lane & 15, wq = lane >> 4;
    v8i a[RB][2];
    v4f best[RB];
    const int nBb = (nB + 15) >> 4;
    const int S = (nBb + 3) >> 2; // steps of four blocks, the same for every wave
    const uint32_t ring_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)s_ring;
    const uint32_t lds16 = ring_lds + lane * 16, ldsk = ring_lds + (lane & 15) * 16; // this lane's read addresses
    // this wave's source of step t: block 4 t + w of the candidate frame, two descriptor words per lane
    struct Src {
        uint32_t w0, w1;
        int kp; // candidate index, or -1 past the count (key -1e30: never wins)
    };
    // (blocks past the count clamp to the last record and carry dead keys: maxima are idempotent, a dead key never wins)
    const unsigned char *__restrict__ recB8 = reinterpret_cast<const unsigned char *>(recB);
    const uint32_t lane_off = 20u + 4u * (uint32_t)wq; // descriptor word wq of a record; word wq + 4 is 16 bytes on
    int kp_next = wv * 16 + kq;                        // the candidate this lane fetches next: 64 on per step
    auto fetch = [&]() {
        const int kp = kp_next;
        kp_next += 64;
        const uint32_t kc = min((uint32_t)kp, (uint32_t)(nB - 1)); // (nB >= 1 here) a legal record for the load
        const uint32_t off = kc * 52u + lane_off;                  // 32 bits: one frame's records are < 4 GB by far
        Src r;
        r.w0 = *reinterpret_cast<const uint32_t *>(recB8 + off);
        r.w1 = *reinterpret_cast<const uint32_t *>(recB8 + off + 16);
        r.kp = kp < nB ? kp : -1;
        return r;
    };
    // the expansion of a source, in pieces a row block's scheduling region can take one at a time
    struct Exp {
        u32x4 f0, f1;
        int pc; // |b| on its way: this lane's two words, then the keypoint's eight
        float key;
    };
    // fragment image + key tuples of this wave's block into slot `s` (all LDS traffic of the loop is hand-written: the
    // compiler must neither wait for the prefetched global loads at a barrier nor reorder reads and writes of the ring)
    auto store = [&](auto slot_c, const Exp &e) {
        constexpr int sl = decltype(slot_c)::value;
        const uint32_t base = lds16 + (uint32_t)(sl * kTileSlot) + (uint32_t)wv * kTileBlk;
        const uint32_t kbase = ldsk + (uint32_t)(sl * kTileSlot) + (uint32_t)wv * kTileBlk;
        const v4f k4 = (v4f){e.key, e.key, e.key, e.key};
        // the key tuple: the four lanes of a keypoint hold the same key and write it to the same 16 bytes -- no branch (a
        // divergent `if (lane < 16)` here split the loop body into blocks and cost the kernel 335 spilled registers)
        asm volatile("ds_write_b128 %0, %2\n\tds_write_b128 %0, %3 offset:1024\n\tds_write_b128 %1, %4 offset:2048" ::"v"(base), "v"(kbase),
                     "v"(e.f0), "v"(e.f1), "v"(k4)
                     : "memory");
    };
    struct Ops {
        u32x4 q0, q1, r0, r1;
        v4f cv, dv;
    };
    auto read2 = [](auto off_c, Ops &o, uint32_t lds16, uint32_t ldsk) { // (the prologue's; the loop spreads its reads, see mma2)
        constexpr int off = decltype(off_c)::value;
        asm volatile("ds_read_b128 %0, %6 offset:%8\n\tds_read_b128 %1, %6 offset:%9\n\t"
                     "ds_read_b128 %2, %7 offset:%10\n\t"
                     "ds_read_b128 %3, %6 offset:%11\n\tds_read_b128 %4, %6 offset:%12\n\t"
                     "ds_read_b128 %5, %7 offset:%13"
                     : "=&v"(o.q0), "=&v"(o.q1), "=&v"(o.cv), "=&v"(o.r0), "=&v"(o.r1), "=&v"(o.dv)
                     : "v"(lds16), "v"(ldsk), "n"(off), "n"(off + 1024), "n"(off + 2048), "n"(off + kTileBlk),
                       "n"(off + kTileBlk + 1024), "n"(off + kTileBlk + 2048)
                     : "memory");
    };
    auto landed = [](Ops &o) { // the wait is tied to the registers, so nothing reads (or copies) them before it
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.q0), "+v"(o.q1), "+v"(o.cv), "+v"(o.r0), "+v"(o.r1), "+v"(o.dv)::"memory");
    };
    const uint32_t wr16 = lds16 + (uint32_t)wv * kTileBlk, wrk = ldsk + (uint32_t)wv * kTileBlk; // this wave's block of a slot
    const v4f kLow = (v4f){-3e38f, -3e38f, -3e38f, -3e38f};
    v4f p1c = kLow, p1d = kLow, p2c = kLow, p2d = kLow; // the two row blocks whose folds are still owed
    // Half a step: 32 MFMAs over the two candidate blocks in `o`, the folds of two row blocks ago between them, and behind
    // row blocks 0..2 the six LDS reads of the NEXT two blocks into `n` (offset RD), two per row block: a burst of six
    // ds_read_b128 in front of the MFMAs held the wave's issue for ~50 cycles per half, two in a gap cost ~3
    // (MI355X_MICROARCH.md, LDS: 'issued between MFMAs').  The FIRST half of a step also carries the expansion of the
    // source block `sr` into `ex` -- one dword per row block, the popcount met across the four lanes of a keypoint by
    // v_permlane16_swap / v_permlane32_swap (vector instructions; the __shfl_xor of the prologue is two ds_bpermute round
    // trips, which stood exposed in the middle of every step), the key in row block 4; the SECOND half writes `ex` into
    // the ring at offset WR behind row blocks 3..5.  Everything that touches LDS is hand-written asm between two
    // sched_barriers, so it sits exactly where it is written.
    auto mma2 = [&](auto first_c, auto rd_c, auto wr_c, const Ops &o, Ops &n, Src &sr, Exp &ex) {
        constexpr bool kFirst = decltype(first_c)::value;
        constexpr int RD = decltype(rd_c)::value, WR = decltype(wr_c)::value;
        const v8i b0 = (v8i){(int)o.q0.x, (int)o.q0.y, (int)o.q0.z, (int)o.q0.w, 0, 0, 0, 0};
        const v8i b1 = (v8i){(int)o.q1.x, (int)o.q1.y, (int)o.q1.z, (int)o.q1.w, 0, 0, 0, 0};
        const v8i d0 = (v8i){(int)o.r0.x, (int)o.r0.y, (int)o.r0.z, (int)o.r0.w, 0, 0, 0, 0};
        const v8i d1 = (v8i){(int)o.r1.x, (int)o.r1.y, (int)o.r1.z, (int)o.r1.w, 0, 0, 0, 0};
        auto row_block = [&](auto m_c) { // (a constant m: sched_group_barrier takes literals only)
            constexpr int m = decltype(m_c)::value;
            v4f &fold = best[(m + RB - 2) % RB]; // m = 0, 1: the previous pass's last two row blocks
            v4f acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][0], b0, o.cv, 4, 4, 0, 0, 0, 0);
            fold[0] = __builtin_fmaxf(__builtin_fmaxf(fold[0], p2c[0]), p2d[0]); // v_max3_f32
            v4f acd = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][0], d0, o.dv, 4, 4, 0, 0, 0, 0);
            fold[1] = __builtin_fmaxf(__builtin_fmaxf(fold[1], p2c[1]), p2d[1]);
            acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][1], b1, acc, 4, 4, 0, 0, 0, 0);
            fold[2] = __builtin_fmaxf(__builtin_fmaxf(fold[2], p2c[2]), p2d[2]);
            acd = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m][1], d1, acd, 4, 4, 0, 0, 0, 0);
            fold[3] = __builtin_fmaxf(__builtin_fmaxf(fold[3], p2c[3]), p2d[3]);
            p2c = p1c;
            p2d = p1d;
            p1c = acc;
            p1d = acd;
            if (kFirst) { // piece m of the expansion
                if (m == 0) ex.f0.x = tile_cand_dword<0>(sr.w0), ex.pc = __popc(sr.w0) + __popc(sr.w1);
                if (m == 1) ex.f0.y = tile_cand_dword<1>(sr.w0), ex.pc = sum_lanes_xor16(ex.pc);
                if (m == 2) ex.f0.z = tile_cand_dword<2>(sr.w0), ex.pc = sum_lanes_xor32(ex.pc);
                if (m == 3) ex.f0.w = tile_cand_dword<3>(sr.w0);
                if (m == 4) ex.f1.x = tile_cand_dword<0>(sr.w1), ex.key = -(float)(ex.pc * kMmaS + sr.kp) * kKeyUnit;
                if (m == 5) ex.f1.y = tile_cand_dword<1>(sr.w1), ex.key = sr.kp >= 0 ? ex.key : -1e30f; // = column_key()
                if (m == 6) ex.f1.z = tile_cand_dword<2>(sr.w1);
                if (m == 7) ex.f1.w = tile_cand_dword<3>(sr.w1);
            }
            if (!kFirst && m == 6) sr = fetch(); // the buffer's next source: five vector instructions and the two loads
            // the emitted order: one matrix instruction, then the vector instructions due -- a fold, and the piece's
            // instructions (kPiece[m] of them) dealt out over the four gaps
            constexpr int kPiece[8] = {3, 4, 4, 2, 4, 3, 1, 2};
            constexpr int extra = kFirst ? kPiece[m] : m == 6 ? 5 : 0;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1 + (extra + 3) / 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1 + (extra + 2) / 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1 + (extra + 1) / 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1 + extra / 4, 0);
            if (!kFirst && m == 6) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            __builtin_amdgcn_sched_barrier(0); // nothing crosses from one row block to the next
            if (m == 0)
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                             : "=&v"(n.q0), "=&v"(n.q1)
                             : "v"(lds16), "n"(RD), "n"(RD + 1024)
                             : "memory");
            if (m == 1)
                asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%5"
                             : "=&v"(n.cv), "=&v"(n.r0)
                             : "v"(ldsk), "v"(lds16), "n"(RD + 2048), "n"(RD + kTileBlk)
                             : "memory");
            if (m == 2)
                asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%5"
                             : "=&v"(n.r1), "=&v"(n.dv)
                             : "v"(lds16), "v"(ldsk), "n"(RD + kTileBlk + 1024), "n"(RD + kTileBlk + 2048)
                             : "memory");
            if (!kFirst && m == 3) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wr16), "v"(ex.f0), "n"(WR) : "memory");
            if (!kFirst && m == 4) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wr16), "v"(ex.f1), "n"(WR + 1024) : "memory");
            if (!kFirst && m == 5) {
                // the key tuple: the four lanes of a keypoint hold the same key and write it to the same 16 bytes -- no branch
                // (a divergent `if (lane < 16)` here split the loop body into blocks and cost the kernel 335 spilled registers)
                const v4f k4 = (v4f){ex.key, ex.key, ex.key, ex.key};
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wrk), "v"(k4), "n"(WR + 2048) : "memory");
            }
            if (m <= 2 || (!kFirst && m <= 5)) __builtin_amdgcn_sched_barrier(0);
        };
        row_block(std::integral_constant<int, 0>{});
        row_block(std::integral_constant<int, 1>{});
        row_block(std::integral_constant<int, 2>{});
        row_block(std::integral_constant<int, 3>{});
        row_block(std::integral_constant<int, 4>{});
        row_block(std::integral_constant<int, 5>{});
        row_block(std::integral_constant<int, 6>{});
        row_block(std::integral_constant<int, 7>{});
    };
    // the expansion of a source block in one piece (the prologue's)
    auto expand = [&](const Src &sr) {
        Exp e;
        e.f0 = tile_cand_word(sr.w0);
        e.f1 = tile_cand_word(sr.w1);
        e.pc = sum_lanes_xor32(sum_lanes_xor16(__popc(sr.w0) + __popc(sr.w1)));
        e.key = column_key(e.pc, sr.kp);
        return e;
    };
    Ops P, Q;
    Exp E;
    // prologue.  Every global load of it is issued up front -- the queries' raw words (16 registers; their 64-register
    // fragments are made after the ring's prologue) and the sources of steps 0..3 -- so the workgroup pays one memory
    // latency, not three in a row (in-kernel stamps, profiles/r05_tile_stamps.txt: the prologue was 13 000 cycles of a
    // workgroup's 85 000, the epilogue 18 600).
    uint32_t qw0[RB], qw1[RB];
#pragma unroll
    for (int m = 0; m < RB; m++) {
        const int kp = row0 + m * 16 + kq;
        const int kc = kp < nA ? kp : nA - 1; // (nA >= 1 here; rows past the count: zero fragments, results discarded below)
        qw0[m] = recA[(size_t)kc * 13 + 5 + wq];
        qw1[m] = recA[(size_t)kc * 13 + 9 + wq];
    }
    const Src s0 = fetch(), s1 = fetch();
    Src nx0 = fetch();
    Src nx1 = fetch();
    // steps 0 and 1 expanded into the two slots
    store(std::integral_constant<int, 0>{}, expand(s0));
    store(std::integral_constant<int, 1>{}, expand(s1));
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    read2(std::integral_constant<int, 0>{}, P, lds16, ldsk);
#pragma unroll
    for (int m = 0; m < RB; m++) {
        const bool in = row0 + m * 16 + kq < nA;
        const uint32_t w0 = in ? qw0[m] : 0u, w1 = in ? qw1[m] : 0u;
        const u32x4 f0 = tile_query_word(w0), f1 = tile_query_word(w1);
        a[m][0] = (v8i){(int)f0.x, (int)f0.y, (int)f0.z, (int)f0.w, 0, 0, 0, 0};
        a[m][1] = (v8i){(int)f1.x, (int)f1.y, (int)f1.z, (int)f1.w, 0, 0, 0, 0};
        const int pc = sum_lanes_xor32(sum_lanes_xor16(__popc(w0) + __popc(w1)));
        if (wq == 0) s_popA[wv * 128 + m * 16 + kq] = (uint16_t)pc;
    }
#pragma unroll
    for (int m = 0; m < RB; m++) best[m] = (v4f){-3e38f, -3e38f, -3e38f, -3e38f};
    // One step = slot s = four candidate blocks and ONE barrier, in the middle: by then this wave holds every operand of the
    // slot in registers (and its own ring writes of the step before have completed: the same lgkmcnt(0)), so behind the
    // barrier slot s may take step t + 2, and the other slot, written a step ago, is complete for everybody.  Both halves'
    // LDS reads are requested 32 MFMAs before they are needed; the expansion rides in the first half's MFMA gaps.
    // The source block a step expands was fetched two steps earlier into one of TWO register buffers, refilled right after
    // their use (step t: buffer t & 1 holds block 4 (t + 2) + w and takes block 4 (t + 4) + w in row block 6 of the second half).  Two buffers and a loop
    // unrolled by two: no buffer ever changes registers, so the loads stay in flight across the loop's back-edge (three
    // rotating values made hipcc copy them there, behind an s_waitcnt vmcnt(0) that cut the prefetch distance to one step).
    auto step = [&](auto slot_c, Src &buf) {
        constexpr int s = decltype(slot_c)::value;
        typedef std::integral_constant<int, s * kTileSlot> this_slot;
        landed(P); // blocks 0, 1: requested in the previous step
        __builtin_amdgcn_sched_barrier(0);
        mma2(std::true_type{}, std::integral_constant<int, s * kTileSlot + 2 * kTileBlk>{}, this_slot{}, P, Q, buf, E);
        landed(Q);
        asm volatile("s_barrier" ::: "memory");
        // blocks 0, 1 of the other slot into P, step t + 2 into the slot everybody has just finished with
        mma2(std::false_type{}, std::integral_constant<int, ((s + 1) % kRing2) * kTileSlot>{}, this_slot{}, Q, P, buf, E);
    };
    int i = 0;
    for (; i + 2 <= S; i += 2) {
        step(std::integral_constant<int, 0>{}, nx0);
        step(std::integral_constant<int, 1>{}, nx1);
    }
    if (i < S) step(std::integral_constant<int, 0>{}, nx0); // S is the same for the four waves: the barriers stay matched
    landed(P); // (the read of the slot after the last one: harmless, but it must have returned before the registers die)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        best[RB - 2][r] = __builtin_fmaxf(__builtin_fmaxf(best[RB - 2][r], p2c[r]), p2d[r]);
        best[RB - 1][r] = __builtin_fmaxf(__builtin_fmaxf(best[RB - 1][r], p1c[r]), p1d[r]);
    }
    // C/D layout: lane holds rows 4 * (lane >> 4) + r of each 16-row fragment, column class lane & 15: the row's maximum is
    // the maximum over its 16 lanes, left in all of them.  Then every lane takes ONE row of each half of the row blocks --
    // lane L: r = L & 3, m = ((L & 15) >> 2) + 4 h -- so the 128 results leave in two passes of 64 full lanes (one pass per
    // (m, r) with one lane in sixteen writing was 32 branches, 32 LDS reads and 32 waits).
#pragma unroll
    for (int m = 0; m < RB; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) best[m][r] = max_over_row16(best[m][r]);
    const int r_sel = lane & 3, m_sel = (lane & 15) >> 2;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        float x[4];
#pragma unroll
        for (int mm = 0; mm < 4; mm++) {
            const v4f &b4 = best[4 * h + mm];
            x[mm] = r_sel == 0 ? b4[0] : r_sel == 1 ? b4[1] : r_sel == 2 ? b4[2] : b4[3];
        }
        const float v = m_sel == 0 ? x[0] : m_sel == 1 ? x[1] : m_sel == 2 ? x[2] : x[3];
        const int q = (4 * h + m_sel) * 16 + 4 * (lane >> 4) + r_sel; // this lane's query of the wave's 128
        const int i_q = row0 + q;
        if (i_q < cap) {
            bool ok = i_q < nA && v > -1e29f;
            const int nk = -(int)(ok ? v * (2 * kMmaS) : 0.0f); // S * (dist - |a_i|) + j
            const int bj = nk & (kMmaS - 1);
            const int bd = (int)s_popA[wv * 128 + q] + (nk >> 14); // |a_i| from the prologue
            ok = ok && bd <= max_dist;
            out_idx[(size_t)pk * cap + i_q] = ok ? bj : -1;
            if (out_dist) out_dist[(size_t)pk * cap + i_q] = ok ? bd : -1;
        }
    }
}

// Which form a call takes.  The tile kernel's workgroup holds 512 queries and is the faster form whenever the chip is filled
// (it needs no expansion pass and no scratch: 255 pairs of 2000 keypoints 0.075 ms against 0.104 streamed, 255 pairs of 405:
// 0.009 against 0.022, 15 pairs of 8192: 0.077 against 0.092 -- bench.py --mode match, ORBFE_MATCH=stream | tile).  Small
// calls are the exception (tools/r5_match_forms_sweep.py, profiles/r05_match_forms_sweep.txt): a tile workgroup runs a whole
// candidate frame past its queries, so with fewer (pair, tile) items than half the CUs the 128-query workgroups of the
// streamed forms spread the same work over more of the chip (1 pair of 2000: 0.010 ms against 0.022; the forms meet at ~128
// items); and between 256 and 448 items the CUs that get a second tile workgroup run both at half pace (320 items of 4800
// keypoints: 0.088 against 0.080).  form: 0 = by size, 1 = the expand + stream forms, 2 = the tile form (ORBFE_MATCH=stream|
// tile, read when the context is created: A/B timing, and the tests run both).
bool match_mfma_uses_tile(int n_pairs, int capP, int form)
{
    if (form == 1) return false;
    if (form == 2) return true;
    const long long items = (long long)n_pairs * ((capP + 511) / 512);
    return items >= 128 && !(items > 256 && items < 448);
}

void launch_match_mfma(const orbfe_keypoint *d_records, const int32_t *d_counts, int n_frames, int n_pairs, int first,
                       int stride, int cap, int capP, int max_dist, uint4 *mexp, float *mkey, int form, int32_t *d_idx,
                       int32_t *d_dist, hipStream_t stream)
{
    static_assert(kMmaS == kMmaMaxKeypoints, "key packing");
    if (match_mfma_uses_tile(n_pairs, capP, form)) {
        hipLaunchKernelGGL(match_tile_kernel, dim3((capP + 511) / 512, n_pairs), dim3(256), 0, stream, d_records, d_counts, cap,
                           first, stride, max_dist, d_idx, d_dist);
        return;
    }
    hipLaunchKernelGGL(match_expand_kernel, dim3((capP * 8 + 255) / 256, n_frames), dim3(256), 0, stream, d_records,
                       d_counts, cap, capP, mexp, mkey);
    if ((long long)n_pairs * ((capP + 127) / 128) >= 256) // at least one 128-query workgroup per CU
        hipLaunchKernelGGL((match_mfma_kernel<8, 4>), dim3((capP + 127) / 128, n_pairs), dim3(256), 0, stream, mexp, mkey,
                           d_counts, cap, capP, first, stride, max_dist, d_idx, d_dist);
    else
        hipLaunchKernelGGL((match_mfma_kernel<4, 4>), dim3((capP + 63) / 64, n_pairs), dim3(256), 0, stream, mexp, mkey,
                           d_counts, cap, capP, first, stride, max_dist, d_idx, d_dist);
}

} // namespace orbfe
