// device_common.hpp -- device helpers shared by stage_kernels.hip and batch_kernels.hip.
#pragma once
#include "orbfe_internal.hpp"

namespace orbfe {

// rBRIEF pattern, 256 rows of (Px, Py, Qx, Qy) (include/orbfe_pattern.h)
static __constant__ __attribute__((aligned(16))) int8_t c_pattern[ORBFE_PATTERN_TESTS * 4] = {
    ORBFE_PATTERN_VALUES};

// the same table as floats (one 16-byte load per test instead of four byte->float converts)
static __constant__ __attribute__((aligned(16))) float c_pattern_f[ORBFE_PATTERN_TESTS * 4] = {
    ORBFE_PATTERN_VALUES};

// Sum of v over the 64 lanes of the wave, returned in every lane.  DPP adds inside the VALU
// (row_shr 1,2,4,8, row_bcast 15, row_bcast 31) instead of six LDS-crossbar ds_bpermute hops.
__device__ inline int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true); // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true); // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}

// Inclusive prefix sum over the 64 lanes (same DPP ladder); lane 63 holds the total.
__device__ inline int wave_incl_scan_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);
    return v;
}

// Exclusive prefix count of `flag` over a block of NW waves (all threads must call it); *total =
// number of set flags.  s_wave: NW ints of LDS.
template <int NW = 4>
__device__ inline int block_excl_scan(bool flag, int *s_wave, int *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t m = __ballot(flag);
    const int pre = (int)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave[wv] = (int)__popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int u = 0; u < NW; u++) {
        const int v = s_wave[u];
        off += (u < wv) ? v : 0;
        tot += v;
    }
    *total = tot;
    return off + pre;
}

// Orientation patch half-widths, floor(sqrtf(225 - dy*dy) + 0.5) for dy = 0..15
// (src/cuda/orb.cu:106; dy = 15 gives 0).
static __constant__ int8_t c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 7, 5, 0};

// XCD-aware work placement.  A grid of (items_per_frame, n_frames) blocks is dispatched x
// fastest and dealt round-robin over the 8 XCDs, so blocks b and b + 8 share an L2.  With the
// plain mapping the tiles / keypoints of ONE frame are spread over all 8 L2s and every halo
// row or patch line crosses the fabric up to 8 times (measured: 4.1x the algorithmic bytes
// in detect).  This remap gives every frame to one XCD (frame % 8), walking its items in
// order, so re-reads hit that XCD's 4 MiB L2.  Placement is a speed matter only: any
// dispatch order gives the same results.
__device__ inline void xcd_remap(int items, int n_frames, int *frame, int *item)
{
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    const int f8 = n_frames & ~7;
    const int lim = items * f8;
    if (L < lim) {
        const int xcd = L & 7, k = L >> 3;
        const int q = k / items;
        *frame = q * 8 + xcd;
        *item = k - q * items;
    } else { // the last n_frames % 8 frames: plain order
        const int k = L - lim;
        const int q = k / items;
        *frame = f8 + q;
        *item = k - q * items;
    }
}

// The same placement without the division, for the kernels that take a DeviceGeom: the host launches
// frame_grid(items, n) = (8 * items, ceil(n / 8)) blocks, so that blockIdx.x & 7 is the XCD the block lands on and
// also the frame's index inside its group of 8; frames past n (last row only) return false.  Fewer than 8 frames:
// the plain (items, n) grid.  xcd_remap's runtime division was ~40 scalar instructions per wave, and the scalar unit
// issues once per SIMD turn like the vector ALU.
__device__ inline bool frame_item(const DeviceGeom &g, int *frame, int *item)
{
    if (g.grid8) {
        *frame = blockIdx.y * 8 + (blockIdx.x & 7);
        *item = blockIdx.x >> 3;
        return *frame < g.n_frames;
    }
    *frame = blockIdx.y;
    *item = blockIdx.x;
    return true;
}

// tile index -> tile row without a runtime division: the host passes magic_of(tiles_x) (batch_kernels.hip)
__device__ inline int div_by_magic(int n, uint32_t magic) { return magic ? (int)__umulhi((uint32_t)n, magic) : n; }

// f1  RGB8 -> gray, src/cuda/cuda_RGB_to_Grayscale.cu:10-23: floor((B*0.07 + G*0.72 + R*0.21) + 0.5)
// evaluated in double, left to right, without contraction (see oracle_rgb_to_grayscale).
__device__ inline uint32_t rgb_to_gray1(uint32_t r, uint32_t g, uint32_t b)
{
    ORBFE_NO_CONTRACT
    double t = (double)(float)b * 0.07;
    t = t + (double)(float)g * 0.72;
    t = t + (double)(float)r * 0.21;
    return (uint32_t)(int)floor(t + 0.5);
}
// four pixels = 12 bytes R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3 -> one dword of gray
__device__ inline uint32_t rgb4_to_gray4(uint32_t w0, uint32_t w1, uint32_t w2)
{
    const uint32_t g0 = rgb_to_gray1(w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u);
    const uint32_t g1 = rgb_to_gray1(w0 >> 24, w1 & 255u, (w1 >> 8) & 255u);
    const uint32_t g2 = rgb_to_gray1((w1 >> 16) & 255u, w1 >> 24, w2 & 255u);
    const uint32_t g3 = rgb_to_gray1((w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24);
    return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
}

// Pixel accessors: px(row, col) -> value.  GlobalPx reads the image; LdsPatch reads a patch
// that one wave staged in LDS (origin = image position of patch byte (0, 0)).
struct GlobalPx {
    const uint8_t *img;
    int pitch;
    __device__ inline int operator()(int row, int col) const { return img[(size_t)row * pitch + col]; }
};
struct LdsPatch {
    const uint8_t *base; // LDS
    int pitch_bytes, oy, ox;
    __device__ inline int operator()(int row, int col) const
    {
        return base[(row - oy) * pitch_bytes + (col - ox)];
    }
};

// Intensity-centroid moments of the radius-15 disc around (kx, ky) (orb.cu:77-134).  One
// wave: lane = patch column (0..30) + 32 * half; half 0 sums the centre row and the rows
// above, half 1 the rows below.  Integer sums (exact), xor-butterfly add reduction.
template <typename Px>
__device__ inline void patch_moments(const Px &px, int w, int h, int kx, int ky, int lane,
                                     int *m10_out, int *m01_out)
{
    const int col = lane & 31, half = lane >> 5;
    int m10 = 0, m01 = 0;
    const int dx = col - 15;
    const int adx = dx < 0 ? -dx : dx;
    const int tdx = kx + dx;
    if (col < 31 && tdx > 0 && tdx < w) {
        if (half == 0) {
            m10 += dx * px(ky, tdx); // centre row: no row test (:94-102)
#pragma unroll
            for (int dy = 1; dy < 16; dy++)
                if (ky - dy > 0 && adx <= c_umax[dy]) {
                    const int v = px(ky - dy, tdx);
                    m01 -= dy * v;
                    m10 += dx * v;
                }
        } else {
#pragma unroll
            for (int dy = 1; dy < 16; dy++)
                if (ky + dy < h && adx <= c_umax[dy]) {
                    const int v = px(ky + dy, tdx);
                    m01 += dy * v;
                    m10 += dx * v;
                }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m10 += __shfl_xor(m10, off);
        m01 += __shfl_xor(m01, off);
    }
    *m10_out = m10;
    *m01_out = m01;
}

// ------------------------------------------------------------------------------------
// rBRIEF (orb.cu:17-75) + 32-bit "compression" (orb.cu:145-169).  One wave per keypoint.
// In round r (0..3) lane t evaluates pattern test 64 r + t, and the 64-bit __ballot of the
// outcomes IS descriptor bytes 8r .. 8r+7 (bit j of byte b is test 8b + j).  All 64 lanes of
// the wave must be active when orb_describe is called.
// ------------------------------------------------------------------------------------
__device__ inline bool orb_border_zero(int lx, int ly, int w, int h, int radians)
{
    return radians ? (lx < 19 || lx > w - 20 || ly < 19 || ly > h - 20)
                   : (lx < 17 || lx > w - 17 || ly < 17 || ly > h - 17);
}

// Round-half-even of v (|v| < 2^22) plus an integer, in two full-rate instructions: adding
// 1.5 * 2^23 makes the float's low mantissa bits the rounded integer (default rounding mode is
// nearest-even, exactly what CUDA __float2int_rn / rintf do), so
//     as_int(v + 12582912.0f) - 0x4B400000 == (int)rintf(v)
// and the integer offset folds into the subtraction.  Bit-identical to orbfe_rn_int for this range
// (the oracle keeps the rintf form; the parity tests compare the two on every descriptor).
__device__ inline int rn_plus(float v, int add)
{
    ORBFE_NO_CONTRACT
    const float r = v + 12582912.0f;
    return (int)__float_as_uint(r) + (add - 0x4B400000);
}

// rBRIEF from a patch staged in LDS (`bytes` = the LDS array, row pitch PITCH bytes); c0 = byte
// offset of the keypoint inside it.  Same arithmetic as orb_describe below; only the address of a
// sample is formed differently, entirely in full-rate f32 ops (integer multiplies by 36 / 44 are
// quarter rate on this chip): with M = 1.5 * 2^23, row + M and col + M round the coordinates
// (ties to even, as rn_plus), (row + M) - M is the rounded row as an exact float, and
// fmaf(row, PITCH, col + M) = M + col + PITCH * row exactly (all integers < 2^22), whose bit
// pattern is 0x4B400000 + (col + PITCH * row): one integer add gives the LDS byte address.
// pat[r] = pattern row 64 r + lane (the caller loads the four rows once per wave: they are the same
// for every keypoint).
template <int PITCH>
__device__ inline void orb_describe_lds(const uint8_t *bytes, int c0, float a, float b, const float4 (&pat)[4],
                                        uint64_t d[4])
{
    ORBFE_NO_CONTRACT
    const float M = 12582912.0f;
    const uint32_t cbias = (uint32_t)c0 - 0x4B400000u;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float4 pt = pat[r];
        const float fpx = pt.x, fpy = pt.y, fqx = pt.z, fqy = pt.w;
        const float p1 = fpx * b, p2 = fpy * a, p3 = fpx * a, p4 = fpy * b;
        const float q1 = fqx * b, q2 = fqy * a, q3 = fqx * a, q4 = fqy * b;
#ifdef ORBFE_DESCRIBE_MAGIC_ROUND
        const float prow = ((p1 + p2) + M) - M, qrow = ((q1 + q2) + M) - M;
#else
        // the rounded row as an exact float in ONE instruction (v_rndne_f32 = round half to even, as the
        // reference's __float2int_rn) instead of the add / subtract pair of the magic-constant form
        const float prow = __builtin_rintf(p1 + p2), qrow = __builtin_rintf(q1 + q2);
#endif
        const float pcol = (p3 - p4) + M, qcol = (q3 - q4) + M;
        const float pa = __builtin_fmaf(prow, (float)PITCH, pcol);
        const float qa = __builtin_fmaf(qrow, (float)PITCH, qcol);
        const int t0 = bytes[__float_as_uint(pa) + cbias];
        const int t1 = bytes[__float_as_uint(qa) + cbias];
        d[r] = __ballot(t0 < t1);
    }
}

// (a, b) = (cos, sin) of the steering angle (orb_steer below)
template <typename Px>
__device__ inline void orb_describe(const Px &px, int lx, int ly, float a, float b, int lane, uint64_t d[4])
{
    ORBFE_NO_CONTRACT
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float4 pt = reinterpret_cast<const float4 *>(c_pattern_f)[64 * r + lane];
        const float fpx = pt.x, fpy = pt.y, fqx = pt.z, fqy = pt.w;
        const float p1 = fpx * b, p2 = fpy * a, p3 = fpx * a, p4 = fpy * b;
        const float q1 = fqx * b, q2 = fqy * a, q3 = fqx * a, q4 = fqy * b;
        const int prow = rn_plus(p1 + p2, ly), pcol = rn_plus(p3 - p4, lx);
        const int qrow = rn_plus(q1 + q2, ly), qcol = rn_plus(q3 - q4, lx);
        const int t0 = px(prow, pcol);
        const int t1 = px(qrow, qcol);
        d[r] = __ballot(t0 < t1);
    }
}

// cos / sin of the descriptor steering angle: the stored angle (radians) times pi/180 in the
// reference (Q7), or the angle itself with angle_in_radians.
__device__ inline void orb_steer(float angle, int radians, float *a, float *b)
{
    ORBFE_NO_CONTRACT
    const float ang = radians ? angle : angle * ORBFE_DEG2RAD_F;
    orbfe_sincosf(ang, b, a);
}

__device__ inline uint32_t orb_compress(const uint64_t d[4])
{
    uint32_t out = 0;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int k = 0; k < 8; k++) out |= (uint32_t)(((d[r] >> (8 * k)) & 0xFFull) == 1ull) << (8 * r + k);
    return out;
}

} // namespace orbfe
