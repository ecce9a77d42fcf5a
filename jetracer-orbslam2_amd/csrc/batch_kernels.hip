// batch_kernels.hip -- the throughput path: many frames per call, context-owned pyramid.
// gfx950 only.  Pipeline per orbfe_extract call (all on the caller's stream):
//   memset cell keys -> blur (level 0) -> halving (levels 1..L-1) -> fused FAST + grid NMS
//   -> per-frame selection / compaction -> orientation + rBRIEF -> 52-byte records.
// The f32 response maps of the reference (src/cuda/fast.cu, nms.cu) never exist in HBM:
// scores live in LDS as u16 and the per-cell winner is an atomicMax of nms_key().
#include "orbfe_internal.hpp"
#include "device_common.hpp"

#include <vector>

namespace orbfe {

// ------------------------------------------------------------------------------------
// a2 batch  blur: one thread = 4 adjacent outputs (one dword store).  Seams every 32 columns
// (Q2) fall on dword-group boundaries, so only the first/last pixel of a group can see one.
// VEC = source rows can be read as aligned dwords.
// ------------------------------------------------------------------------------------
template <bool VEC>
__global__ void __launch_bounds__(256)
blur_batch_kernel(uint8_t *__restrict__ dst, int dst_pitch, size_t dst_fstride,
                  const uint8_t *__restrict__ src, int src_pitch, size_t src_fstride, int w, int h)
{
    int f, item; // item = (row block, column block), x fastest
    xcd_remap(gridDim.x, gridDim.y, &f, &item);
    const int xblocks = (w + 255) / 256;
    const int by = item / xblocks, bx = item - by * xblocks;
    const int x0 = (bx * 64 + (threadIdx.x & 63)) * 4;
    const int y = by * 4 + (threadIdx.x >> 6);
    if (x0 >= w || y >= h) return;
    dst += (size_t)f * dst_fstride;
    src += (size_t)f * src_fstride;
    uint32_t *o = reinterpret_cast<uint32_t *>(dst + (size_t)y * dst_pitch + x0);
    if (y == 0 || y >= h - 2) {
        *o = 0u;
        return;
    }
    int col[3][6]; // rows y-1, y, y+1; columns x0-1 .. x0+4
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const uint8_t *row = src + (size_t)(y - 1 + r) * src_pitch;
        if (VEC && x0 + 3 < w) {
            const uint32_t v = *reinterpret_cast<const uint32_t *>(row + x0);
            col[r][1] = v & 255u;
            col[r][2] = (v >> 8) & 255u;
            col[r][3] = (v >> 16) & 255u;
            col[r][4] = v >> 24;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) col[r][1 + i] = (x0 + i < w) ? row[x0 + i] : 0;
        }
        col[r][0] = ((x0 & 31) == 0) ? col[r][1] : (int)row[x0 - 1];
        // right neighbour of pixel x0+3; a seam or the image edge substitutes the pixel itself
        col[r][5] = (((x0 + 3) & 31) == 31 || x0 + 3 >= w - 1) ? col[r][4] : (int)row[x0 + 4];
    }
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // x == w-1 inside the group: its right neighbour is itself
        const bool last = (x0 + i == w - 1);
        const int a_r = last ? col[0][1 + i] : col[0][2 + i];
        const int b_r = last ? col[1][1 + i] : col[1][2 + i];
        const int c_r = last ? col[2][1 + i] : col[2][2 + i];
        const int s = 2 * col[0][1 + i] + 4 * col[1][1 + i] + 2 * col[2][1 + i] + col[0][i] + a_r +
                      2 * col[1][i] + 2 * b_r + col[2][i] + c_r;
        out |= (uint32_t)((s + 8) >> 4) << (8 * i);
    }
    *o = out; // columns >= w of the last group land in the row padding (pitch is 64-aligned)
}

// a3 batch  halving: one thread = 4 outputs from two aligned 8-byte reads.
__global__ void __launch_bounds__(256)
halfsample_batch_kernel(uint8_t *__restrict__ pyr, size_t fstride, size_t src_off, int src_pitch,
                        size_t dst_off, int dst_pitch, int dw, int dh)
{
    int f, item;
    xcd_remap(gridDim.x, gridDim.y, &f, &item);
    const int xblocks = (dw + 255) / 256;
    const int by = item / xblocks, bx = item - by * xblocks;
    const int x0 = (bx * 64 + (threadIdx.x & 63)) * 4;
    const int y = by * 4 + (threadIdx.x >> 6);
    if (x0 >= dw || y >= dh) return;
    uint8_t *base = pyr + (size_t)f * fstride;
    const uint8_t *t = base + src_off + (size_t)(2 * y) * src_pitch + 2 * x0;
    const uint2 a = *reinterpret_cast<const uint2 *>(t);
    const uint2 b = *reinterpret_cast<const uint2 *>(t + src_pitch);
    // horizontal pair sums of bytes, two 16-bit lanes per dword
    const uint32_t m = 0x00FF00FFu;
    const uint32_t s0 = (a.x & m) + ((a.x >> 8) & m) + (b.x & m) + ((b.x >> 8) & m);
    const uint32_t s1 = (a.y & m) + ((a.y >> 8) & m) + (b.y & m) + ((b.y >> 8) & m);
    const uint32_t out = ((s0 >> 2) & 255u) | (((s0 >> 18) & 255u) << 8) |
                         (((s1 >> 2) & 255u) << 16) | (((s1 >> 18) & 255u) << 24);
    *reinterpret_cast<uint32_t *>(base + dst_off + (size_t)y * dst_pitch + x0) = out;
}

// a3 batch  the small levels (8 and up: at most 1/256 of the frame's width and height) of one frame
// in ONE workgroup, level after level with a barrier in between, instead of one launch per level:
// a 12-level 4K pyramid is launch-bound (4 launches of ~100 pixels each).
__global__ void __launch_bounds__(256)
pyramid_tail_kernel(DeviceGeom g, uint8_t *__restrict__ pyr, int first_level)
{
    uint8_t *fb = pyr + (size_t)blockIdx.x * g.frame_stride;
    for (int l = first_level; l < g.L; l++) {
        const int dw = g.lv[l].w, dh = g.lv[l].h;
        if (dw == 0 || dh == 0) break;
        const uint8_t *src = fb + g.lv[l - 1].offset;
        uint8_t *dst = fb + g.lv[l].offset;
        const int sp = g.lv[l - 1].pitch, dp = g.lv[l].pitch;
        for (int i = threadIdx.x; i < dw * dh; i += 256) {
            const int y = i / dw, x = i - y * dw;
            const uint8_t *q = src + (size_t)(2 * y) * sp + 2 * x;
            dst[(size_t)y * dp + x] = (uint8_t)(((unsigned)q[0] + q[1] + q[sp] + q[sp + 1]) >> 2);
        }
        __syncthreads(); // level l is complete (and visible to this workgroup) before level l + 1 reads it
    }
}

// ------------------------------------------------------------------------------------
// a2 + a3 fused: blur and ALL halvings of one 128x128 level-0 tile in one workgroup (levels
// 0..7; 2^7 = 128).  The reference's blur has seams every 32 columns (Q2), so a tile needs no
// horizontal halo at all, only one input row above and below.  Thread = one dword column
// (4 pixels) x 16 rows with a 3-row sliding window of the horizontal 1-2-1 sums, kept as two
// 16-bit lanes per dword (even / odd pixels: plain 32-bit adds never carry across lanes
// because every partial sum is < 2^16).  Left / right neighbour dwords come from the adjacent
// lanes by DPP (row_shr:1 / row_shl:1; a DPP row is 16 lanes = 64 pixels = two seam segments).
// Level 1 (2x2 of level 0) and level 2 (2x2 of level 1) are formed in registers as the rows
// stream by; level 2 goes to LDS and levels 3..7 are a five-step LDS cascade.
// Requires W % 4 == 0 and a dword-aligned source; otherwise the unfused kernels above run.
// ------------------------------------------------------------------------------------
// RGB = the source is interleaved RGB8 (f1): the gray conversion happens in the row load, so
// the gray frame never exists in HBM.
template <bool RGB>
__global__ void __launch_bounds__(256)
pyramid_fused_kernel(DeviceGeom g, uint8_t *__restrict__ pyr, const uint8_t *__restrict__ src, int src_pitch,
                     size_t src_fstride, int tiles_x, uint32_t tiles_x_magic, int keys_per_tile, uint32_t *__restrict__ cellkey)
{
    __shared__ uint8_t s_l2[32 * 32], s_l3[16 * 16], s_l4[8 * 8], s_l5[4 * 4], s_l6[2 * 2];
    int f, tile;
    if (!frame_item(g, &f, &tile)) return;
    const int ty = div_by_magic(tile, tiles_x_magic), tx = tile - ty * tiles_x;
    const int tid = threadIdx.x, cx = tid & 31, rg = tid >> 5;
    // by-product: the frame's cell keys are cleared for the detection that follows (saves the
    // memset launch of detect_batch); tile t clears the t-th slice
    {
        const int per = keys_per_tile; // ceil(K / tiles per frame), from the host (a runtime division here was ~35 instructions)
        const int hi = (tile + 1) * per < g.K ? (tile + 1) * per : g.K;
        for (int i = tile * per + tid; i < hi; i += 256) cellkey[(size_t)f * g.K + i] = 0u;
    }
    const int W = g.W, H = g.H;
    const int x0 = tx * 128 + 4 * cx;
    const int ybase = ty * 128 + rg * 16;
    src += (size_t)f * src_fstride;
    uint8_t *fb = pyr + (size_t)f * g.frame_stride;
    const bool col_ok = x0 < W;
    const uint32_t M = 0x00FF00FFu;
    const bool seam_l = (cx & 7) == 0, seam_r = (cx & 7) == 7 || x0 + 4 >= W;

    // The 18 input rows of this thread (y - 1 .. y + 16; RGB: three dwords each) are loaded up front, all in
    // flight together.  The loads are unconditional (addresses clamped into the frame, values masked
    // afterwards): a conditional load cannot be hoisted, and with one load + s_waitcnt vmcnt(0) per
    // row (which also waits for the previous row's stores) a wave had one request in flight.
    uint32_t rowv[18][RGB ? 3 : 1];
    {
        const int xc = col_ok ? x0 : 0;
#pragma unroll
        for (int k = 0; k < 18; k++) {
            int yy = ybase - 1 + k;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            const uint32_t *p = reinterpret_cast<const uint32_t *>(src + (size_t)yy * src_pitch + (RGB ? 3 : 1) * xc);
#pragma unroll
            // (streaming loads: a source frame is read exactly once)
            for (int j = 0; j < (RGB ? 3 : 1); j++) rowv[k][j] = __builtin_nontemporal_load(p + j);
        }
    }
    // horizontal 1-2-1 sums of input row yy, as (even pixels, odd pixels) 16-bit lane pairs
    auto hrow = [&](int yy, int k, uint32_t &hE, uint32_t &hO) {
        uint32_t v = 0;
        if (col_ok && yy >= 0 && yy < H) {
            if constexpr (RGB) v = rgb4_to_gray4(rowv[k][0], rowv[k][1], rowv[k][2]);
            else v = rowv[k][0];
        }
        const uint32_t Ld = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true); // lane - 1
        const uint32_t Rd = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, true); // lane + 1
        const uint32_t Lb = seam_l ? (v & 0xFFu) : (Ld >> 24);
        const uint32_t Rb = seam_r ? (v >> 24) : (Rd & 0xFFu);
        const uint32_t E = v & M, O = (v >> 8) & M;
        hE = ((O << 16) | Lb) + 2u * E + O;
        hO = E + 2u * O + ((E >> 16) | (Rb << 16));
    };

    uint32_t aE, aO, bE, bO, cE, cO; // h of rows y-1, y, y+1
    hrow(ybase - 1, 0, aE, aO);
    hrow(ybase, 1, bE, bO);
    uint32_t prev_s = 0;  // level-0 pair sums of the previous (even) row
    uint32_t prev_l1 = 0; // level-1 row of the previous row pair
    const bool want1 = g.L > 1, want2 = g.L > 2;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int y = ybase + r;
        hrow(y + 1, r + 2, cE, cO);
        uint32_t oE = ((aE + 2u * bE + cE + 0x00080008u) >> 4) & M;
        uint32_t oO = ((aO + 2u * bO + cO + 0x00080008u) >> 4) & M;
        if (y == 0 || y >= H - 2) { // rows the reference never writes (Q1)
            oE = 0;
            oO = 0;
        }
        if (col_ok && y < H)
            *reinterpret_cast<uint32_t *>(fb + g.lv[0].offset + (size_t)y * g.lv[0].pitch + x0) = oE | (oO << 8);
        aE = bE; aO = bO; bE = cE; bO = cO;
        const uint32_t sum = oE + oO; // (px0 + px1, px2 + px3)
        if ((r & 1) == 0) {
            prev_s = sum;
        } else if (want1) {
            const uint32_t l1 = ((prev_s + sum) >> 2) & M; // two level-1 pixels
            if (col_ok && y < H) // both source rows exist <=> level-1 row y/2 exists (y is odd here)
                *reinterpret_cast<uint16_t *>(fb + g.lv[1].offset + (size_t)(y >> 1) * g.lv[1].pitch + (x0 >> 1)) =
                    (uint16_t)((l1 & 0xFFu) | ((l1 >> 8) & 0xFF00u));
            if ((r & 3) == 1) {
                prev_l1 = l1;
            } else if (want2) {
                const uint32_t u = prev_l1 + l1;
                s_l2[((rg * 16 + r) >> 2) * 32 + cx] = (uint8_t)(((u & 0xFFFFu) + (u >> 16)) >> 2);
            }
        }
    }
    if (g.L <= 2) return;
    __syncthreads();
    // levels 3..7 in LDS; a pixel of level l exists iff x < W >> l and y < H >> l, and every
    // existing pixel has four existing parents (floor sizes), so garbage never propagates
    auto quad = [](const uint8_t *p, int pitch, int x, int y) {
        const uint8_t *q = p + (2 * y) * pitch + 2 * x;
        return (uint8_t)(((unsigned)q[0] + q[1] + q[pitch] + q[pitch + 1]) >> 2);
    };
    if (g.L > 3) s_l3[tid] = quad(s_l2, 32, tid & 15, tid >> 4);
    __syncthreads();
    if (g.L > 4 && tid < 64) s_l4[tid] = quad(s_l3, 16, tid & 7, tid >> 3);
    __syncthreads();
    if (g.L > 5 && tid < 16) s_l5[tid] = quad(s_l4, 8, tid & 3, tid >> 2);
    __syncthreads();
    if (g.L > 6 && tid < 4) s_l6[tid] = quad(s_l5, 4, tid & 1, tid >> 1);
    __syncthreads();
    // stores: level l tile is (128 >> l)^2 at (tx, ty) * (128 >> l)
    auto put = [&](const uint8_t *p, int l, int n) { // n = tile edge at this level
        for (int i = tid; i < n * n; i += 256) {
            const int yy = i / n, xx = i - yy * n;
            const int gx = tx * n + xx, gy = ty * n + yy;
            if (gx < g.lv[l].w && gy < g.lv[l].h) fb[g.lv[l].offset + (size_t)gy * g.lv[l].pitch + gx] = p[i];
        }
    };
    put(s_l2, 2, 32);
    if (g.L > 3) put(s_l3, 3, 16);
    if (g.L > 4) put(s_l4, 4, 8);
    if (g.L > 5) put(s_l5, 5, 4);
    if (g.L > 6) put(s_l6, 6, 2);
    if (g.L > 7 && tid == 0 && tx < g.lv[7].w && ty < g.lv[7].h)
        fb[g.lv[7].offset + (size_t)ty * g.lv[7].pitch + tx] = quad(s_l6, 2, 0, 0);
}

// ------------------------------------------------------------------------------------
// a5 + a6 fused  FAST score + 3x3 NMS + per-cell maximum for one 64x64 tile of one level of
// one frame.  Most pixels are not corners, so the work is staged to keep the lanes busy:
//   A  load the pixel tile (4-px halo: ring radius 3 + NMS radius 1) into LDS as dwords
//   B  compass pre-test on the 4 pixels of a dword at once, bytes in place (compass4: byte-wise compares as
//      v_sub + v_bitop3): a cyclic arc of >= 9 ring pixels holds one pixel of each opposite pair (N,S) and
//      (E,W), one of >= 12 holds 3 of the 4 compass pixels.  Survivors are compacted, branch-free, into ONE LDS
//      queue per workgroup.  (Pure work-skipping, like the reference's opposite-pair prechecks,
//      fast.cu:98-124: it rejects no corner.)
//   C  full 16-pixel ring test on queued candidates only, batches of 64 dealt round-robin to the waves, two ring
//      pixels per packed-u16 op (saturating subtracts give the score terms and the label flags at once); scores
//      go to a u16 LDS tile; positive scores inside the tile are queued again, in place
//   D  strict 3x3 maximum test on the positives only; winners atomicMax their nms_key()
//      into the cell (LDS for cells >= 4 px, global for smaller cells)
// ------------------------------------------------------------------------------------
// (The stop-after-a-phase and ring-read profiling builds of detect_tile_kernel / describe_tile_kernel -- wrong results by
// design -- are not in this file: tools/experiments/profiling_probes.patch adds them to a scratch copy of the sources,
// tools/build_variant.sh -p applies it, tools/phase_counters.sh uses those builds.)

// entry i of the context's tile list (8 bytes, 8-byte aligned) through the constant address space: s_load_dwordx2
static_assert(sizeof(TileDesc) == 8, "TileDesc is read as one 64-bit scalar load");
__device__ inline TileDesc load_tile_desc(const TileDesc *tiles, int i)
{
    const uint64_t raw = reinterpret_cast<const __attribute__((address_space(4))) uint64_t *>(reinterpret_cast<uintptr_t>(tiles))[i];
    return TileDesc{(int16_t)(raw & 0xFFFFu), (int16_t)((raw >> 16) & 0xFFFFu), (int16_t)((raw >> 32) & 0xFFFFu), 0};
}

constexpr int kSteerMaxBreaks = 224; // LDS copy of the steering table's break points in the tile describe kernel (222 today)

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ inline us2 U2(uint32_t v) { return __builtin_bit_cast(us2, v); }
__device__ inline uint32_t U1(us2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ inline us2 ssub(us2 a, us2 b) { return __builtin_elementwise_sub_sat(a, b); }

constexpr int kPxW = kTileW + 8, kPxH = kTileH + 8; // 72 x 72 pixel tile
constexpr int kPxDw = kPxW / 4;                      // 18 dwords per row
constexpr int kScW = kTileW + 2, kScH = kTileH + 2; // 66 x 66 score tile (1-px halo)
constexpr int kScPitch = kScW + 2;                   // 68
constexpr int kMaxLdsCells = (kTileW / 4) * (kTileH / 4);
#ifndef ORBFE_DETECT_TILES_PER_WG
#define ORBFE_DETECT_TILES_PER_WG 4
#endif
constexpr int kDetectTilesPerWg = ORBFE_DETECT_TILES_PER_WG; // consecutive tile-list entries a batch-path workgroup takes (detect_tile_kernel)

// ---- compass pre-test on the four pixels of a dword AT ONCE, bytes in place (round 3).  Rounds 1-2 unpacked the
// bytes into pairs of 16-bit lanes for v_pk_min/max/sub_u16: 50 instructions per dword, 27 of them at the packed
// (4-cycle) rate.  Here every step is a full-rate 32-bit instruction (v_add / v_sub / v_and / v_or and the 3-input
// v_bitop3_b32, 2.76 cycles: tools/valu_rate5.hip), 43 per dword.
//   byte-wise a > b, result in bit 7 of each byte:  t = (b | H) - (a & L)  never borrows across bytes and its
//   bit 7 says a_low7 <= b_low7;  a > b  =  (a7 & ~b7) | (~(a7 ^ b7) & ~t7)  -- one v_bitop3 (table 0x71).
//   thresholds: s = (c + t) mod 256 with ovf = c > 255 - t (then no pixel can be brighter), d = (c - t) mod 256 with
//   unf = c < t (then none can be darker); any t in 1..254.
// Result: bit 8 i + 7 set <=> pixel i of the dword may be a corner; other bits clear.
constexpr uint32_t kH8 = 0x80808080u, kL7 = 0x7F7F7F7Fu;
#define ORBFE_BITOP3(a, b, c, tbl) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tbl))
constexpr int kGT = 0x71;    // (a & ~b) | (~(a ^ b) & ~c)
constexpr int kXor3 = 0x96;  // a ^ b ^ c
constexpr int kOrAnd = 0xA8; // (a | b) & c
constexpr int kAndNotAnd = 0x20; // a & ~b & c
constexpr int kAndOr = 0xEA; // (a & b) | c
struct CompassConst { // wave-uniform, from the threshold
    uint32_t T, TL, T7, nT7, K, KH;
};
__device__ inline CompassConst compass_const(int threshold)
{
    const uint32_t T = (uint32_t)threshold * 0x01010101u, K = (uint32_t)(255 - threshold) * 0x01010101u;
    return CompassConst{T, T & kL7, T & kH8, ~T & kH8, K, K | kH8};
}
// MODE 0: arcs 9..11, MODE 1: arc >= 12 (closed forms, below); MODE 2: the reference's own opposite-pair
// prechecks (fast.cu:98-124), literally -- the ring test runs iff (W or E is not similar to the centre) and
// (N or S is not similar): the only form that is valid for an ARBITRARY corner table (stage API).
template <int MODE>
__device__ inline uint32_t compass4(uint32_t C, uint32_t Wd, uint32_t Ed, uint32_t Nd, uint32_t Sd, const CompassConst &k)
{
    const uint32_t cL = C & kL7, cH = C | kH8, u = C & kH8;
    // brighter side
    const uint32_t ovf = ORBFE_BITOP3(C, k.K, k.KH - cL, kGT);          // c > 255 - t
    const uint32_t s = ORBFE_BITOP3(cL + k.TL, u, k.T7, kXor3);         // (c + t) mod 256
    const uint32_t sH = s | kH8;
    // darker side
    const uint32_t z = cH - k.TL;
    const uint32_t unf = ORBFE_BITOP3(k.T, C, z, kGT);                   // t > c
    const uint32_t d = ORBFE_BITOP3(z, u, k.nT7, kXor3);                // (c - t) mod 256
    const uint32_t dL = d & kL7;
    auto brighter = [&](uint32_t v) { return ORBFE_BITOP3(v, s, sH - (v & kL7), kGT); }; // v > s
    auto darker = [&](uint32_t v) { return ORBFE_BITOP3(d, v, (v | kH8) - dL, kGT); };   // d > v
    const uint32_t bN = brighter(Nd), bS = brighter(Sd), bE = brighter(Ed), bW = brighter(Wd);
    const uint32_t dN = darker(Nd), dS = darker(Sd), dE = darker(Ed), dW = darker(Wd);
    uint32_t br, dk;
    if (MODE == 0) {
        // an arc of >= 9 leaves at most 7 contiguous ring pixels out, so it holds one pixel of EVERY opposite pair:
        // (N or S) and (E or W) brighter, or the same darker
        br = ORBFE_BITOP3(bE, bW, bN | bS, kOrAnd);
        dk = ORBFE_BITOP3(dE, dW, dN | dS, kOrAnd);
    } else if (MODE == 1) {
        // arc >= 12: three of the four compass pixels lie in the arc
        br = ORBFE_BITOP3(bN & bS, bE | bW, bE & bW & (bN | bS), kAndOr);
        dk = ORBFE_BITOP3(dN & dS, dE | dW, dE & dW & (dN | dS), kAndOr);
    } else {
        // not similar = brighter or darker; (W or E not similar) and (N or S not similar).  The overflow / underflow
        // masks apply per polarity, so they are folded in before the polarities mix.
        const uint32_t nsv = ORBFE_BITOP3(bN | bS, ovf, ORBFE_BITOP3(dN | dS, unf, kH8, kAndNotAnd), 0xBA /* (a & ~b) | c */);
        const uint32_t nsh = ORBFE_BITOP3(bE | bW, ovf, ORBFE_BITOP3(dE | dW, unf, kH8, kAndNotAnd), 0xBA);
        return nsv & nsh & kH8;
    }
    return ORBFE_BITOP3(br, ovf, kH8, kAndNotAnd) | ORBFE_BITOP3(dk, unf, kH8, kAndNotAnd);
}

// FAST score (0 = not a corner) of the pixel at LDS byte pointer p, two ring pixels per op.
// arc 9..12: the arc test in closed form; arc 0: the caller's corner table decides (lut[dark] | lut[bright],
// fast.cu:230-231), whatever it holds.
__device__ inline int fast_score_packed(const uint8_t *p, uint32_t t2, int arc, const uint8_t *__restrict__ lut)
{
    const uint32_t c = p[0];
    const us2 c2 = U2(c | (c << 16));
    const us2 hi = c2 + U2(t2), lo = ssub(c2, U2(t2));
    const us2 one = U2(0x00010001u);
    us2 sb = U2(0), sd = U2(0);
    uint32_t bright = 0, dark = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { // pair (ring i, ring i + 8)
        const uint32_t v0 = p[ring_dy(i) * kPxW + ring_dx(i)];
        const uint32_t v1 = p[ring_dy(i + 8) * kPxW + ring_dx(i + 8)];
        const us2 v = U2(v0 | (v1 << 16));
        const us2 ab = ssub(v, hi), ad = ssub(lo, v); // score terms v - (c+t), (c-t) - v, or 0
        sb = U2(U1(sb) + U1(ab)); // 8 terms <= 255 per 16-bit lane: no carry between the lanes, so a full-rate v_add_u32
        sd = U2(U1(sd) + U1(ad));
        // mask |= (term != 0) << ring position, both lanes at once: v_pk_min_u16 makes the flags 0 / 1, v_dot2_u32_u16
        // adds flag.lo << i + flag.hi << (i + 8) to ONE 16-bit mask (rounds 1-2 shifted two 8-bit masks with v_pk_mad
        // and merged them afterwards: 4 instructions more per candidate).  Inline asm because hipcc rewrites min(x, 1)
        // into per-half compare + select chains (3x the instructions).
        uint32_t fb, fd;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(fb) : "v"(U1(ab)), "v"(U1(one)));
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(fd) : "v"(U1(ad)), "v"(U1(one)));
        const uint32_t wgt = (1u << i) | (1u << (i + 24)); // lo lane: ring i, hi lane: ring i + 8
        asm("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(bright) : "v"(fb), "s"(wgt));
        asm("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(dark) : "v"(fd), "s"(wgt));
    }
    if (arc == 0) {
        if (!(lut[bright] | lut[dark])) return 0;
    } else {
        // a pixel is never both brighter and darker, so the two masks are disjoint and only one of them can hold
        // >= arc (>= 9 of 16) ones at all: ONE arc test, on the mask that has the population for it
        const uint32_t m = __popc(bright) >= arc ? bright : dark;
        if (!orbfe_has_arc(m, arc)) return 0;
    }
    const uint32_t xb = U1(sb), xd = U1(sd);
    const int rb = (int)((xb & 0xFFFFu) + (xb >> 16)), rd = (int)((xd & 0xFFFFu) + (xd >> 16));
    return rb > rd ? rb : rd;
}

// STAGE = the stage API's orbfe_detect (one frame, caller-owned pitched levels): the tile is found from the
// per-level tile counts instead of the context's tile list, and the tile's scores are also written to
// the caller's f32 response maps (the reference's detect leaves them filled, fast.cu:292-407).
struct StageTiles {
    int first[8 + 1];  // first tile of level l; first[n_levels] = number of tiles
    int tiles_x[8];
    float *resp[8];    // response map of level l (may be null) ...
    int resp_pitch[8]; // ... and its pitch in floats
    const uint8_t *lut; // the caller's 65536-byte corner table (ARC = 0)
};

// ARC = the minimum arc length as a compile-time constant (9..12): the compass test's variant and the
// closed-form arc test then carry no run-time selects (8 fewer instructions per compass trip, ~10 per
// ring-test batch).  ARC = 0 (the stage API): no arc is assumed at all -- the pre-test is the reference's own
// (fast.cu:98-124) and every candidate looks its two masks up in the caller's table, so the result is the
// reference's for ANY table contents (fast.cuh:25-26, :42-48 treat it as opaque).
template <bool STAGE, int ARC, bool MULTI>
__global__ void __launch_bounds__(256)
detect_tile_kernel(DeviceGeom g, const uint8_t *__restrict__ pyr, const TileDesc *__restrict__ tiles,
                   uint32_t *__restrict__ cellkey, int tile_first, int tile_step, int n_items, StageTiles st)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_px32[kPxH * kPxDw];
    __shared__ __attribute__((aligned(16))) uint16_t s_sc[kScH * kScPitch];
    // per-wave queues (wave w owns the tasks tid = 64 w + lane of every trip): the fill level
    // is a wave-uniform register, so compaction needs no atomics and no block barrier
    constexpr int kMainTrips = kTileH * 16 / 256;       // trips over the tile proper (16 groups per row)
    constexpr int kTrips = kMainTrips + 1;              // + one trip for the halo ring
    constexpr int kQ1 = kTrips * 64 * 4;                // candidates a wave can produce
    static_assert(32 + 2 * kScH <= 256, "halo trip must fit one block");
    // (r << 7) | px.  The positives queue q2 lives IN PLACE in q1: its write index never passes
    // the read index (positives so far <= candidates consumed), same wave, in-order LDS.
    // ONE candidate queue per workgroup (round 3; rounds 1-2 had one per wave): 256 dump halfwords (one per thread,
    // phase B's branch-free compaction), then up to 4 * kQ1 entries.  A wave reserves its run with one LDS atomic, and
    // the ring test walks the queue in batches of 64 dealt round-robin to the waves: the last, partly filled batch
    // exists once per tile instead of once per wave (3.36 -> 3.02 batches per wave on the bench scenes).
    __shared__ uint16_t s_q[256 + 4 * kQ1];
    __shared__ int s_qcount;
    __shared__ uint32_t s_key[kMaxLdsCells];

    int f, item;
    if (!frame_item(g, &f, &item)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint8_t *s_px = reinterpret_cast<const uint8_t *>(s_px32);
    // MULTI: a workgroup takes kT consecutive entries of its shard of the tile list, one after the other, and loads tile
    // u + 1's pixels into registers while it works on tile u (r5: in-kernel stamps, profiles/r05_phase_stamps.txt, showed a
    // workgroup waiting 26 % of its 7.7 us for the tile's round trip to memory at its head -- six workgroups per CU hide most
    // of that, not all).  Only launches of many rounds of workgroups take this form (launch_detect_tiles): a quarter of the
    // workgroups, each four times as long, fill the chip worse when there are few (one 4K frame: 0.031 -> 0.050 ms).
    constexpr int kT = MULTI ? kDetectTilesPerWg : 1;
    static_assert(!(STAGE && MULTI), "the stage API keeps one tile per workgroup");
    const int tile0 = item * kT;
    const int n_here = n_items - tile0 < kT ? n_items - tile0 : kT; // (>= 1: the grid is ceil(n_items / kT) wide)
    auto tile_desc = [&](int u) {
        if (STAGE) {
            const int tile_id = tile0 + u;
            int lvl = 0;
#pragma unroll
            for (int i = 1; i < 8; i++) lvl += (i < g.Ld && tile_id >= st.first[i]) ? 1 : 0;
            const int t = tile_id - st.first[lvl], tyy = t / st.tiles_x[lvl];
            return TileDesc{(int16_t)lvl, (int16_t)(t - tyy * st.tiles_x[lvl]), (int16_t)tyy, 0};
        }
        // shard: every tile_step-th tile.  The index is wave-uniform and the list is never written by a kernel: read it
        // through the constant address space, i.e. with ONE s_load_dwordx2 (as a global load it was two vector loads
        // and two v_readfirstlane at the head of every wave)
        return load_tile_desc(tiles, tile_first + (tile0 + u) * tile_step);
    };
    // ---- A: pixel tile as dwords (x0 - 4 is dword aligned; rows are 64-byte aligned).  Dword i of
    //         the tile = (row i / 18, group i % 18); i advances by 256 per trip = 14 rows + 4 groups
    //         with one carry, so there is no division.  Batch path: the context's pyramid has guard bands, every tile loads
    //         without range tests, every load is unconditional, so all kLoadTrips requests of a thread are in
    //         flight together (a conditional load is followed by its own s_waitcnt vmcnt(0): six memory
    //         latencies in a row).  Lanes past the tile's last dword re-load their first one.
    constexpr int kLoadTrips = (kPxH * kPxDw + 255) / 256;
    uint32_t pxv[kLoadTrips];
    auto load_tile = [&](const TileDesc &t) {
        const int P = g.lv[t.level].pitch;
        const uint8_t *img = pyr + (size_t)f * g.frame_stride + g.lv[t.level].offset;
        const int x0 = t.tx * kTileW, y0 = t.ty * kTileH;
        int r = tid / kPxDw, q = tid - r * kPxDw;
        uint32_t off = (uint32_t)(__mul24(y0 - 4 + r, P) + x0 - 4 + 4 * q);
        const uint32_t dstep = (uint32_t)((256 / kPxDw) * P + 4 * (256 % kPxDw));
        // rows above the level's first one give a NEGATIVE offset (guard band / previous level): the base moves
        // up by 4 rows + 4 bytes (scalar arithmetic) so that every lane offset is a non-negative 32-bit number --
        // the SGPR-base form of the load, no sign extension and no 64-bit vector add per load
        const uint8_t *img0 = img - (ptrdiff_t)(4 * P + 4);
        off += (uint32_t)(4 * P + 4);
        const uint32_t off0 = off;
#pragma unroll
        for (int t2 = 0; t2 < kLoadTrips; t2++) {
            const bool in = 256 * (t2 + 1) <= kPxH * kPxDw || 256 * t2 + tid < kPxH * kPxDw;
            pxv[t2] = *reinterpret_cast<const uint32_t *>(img0 + (in ? off : off0));
            off += dstep;
            q += 256 % kPxDw;
            if (q >= kPxDw) { // carry into the next row
                q -= kPxDw;
                off += (uint32_t)(P - 4 * kPxDw);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int t2 = 0; t2 < kLoadTrips; t2++)
            if (256 * (t2 + 1) <= kPxH * kPxDw || 256 * t2 + tid < kPxH * kPxDw) s_px32[256 * t2 + tid] = pxv[t2];
    };
    auto zero_scores = [&]() {
        static_assert((kScH * kScPitch * 2) % 16 == 0, "score tile is zeroed with 16-byte stores");
        constexpr int kScQuads = kScH * kScPitch / 8; // 561 stores: two full rounds and a short one, no loop
        static_assert(kScQuads > 512 && kScQuads <= 768, "score tile zeroing is unrolled for 3 rounds");
        reinterpret_cast<uint4 *>(s_sc)[tid] = make_uint4(0u, 0u, 0u, 0u);
        reinterpret_cast<uint4 *>(s_sc)[tid + 256] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < kScQuads - 512) reinterpret_cast<uint4 *>(s_sc)[tid + 512] = make_uint4(0u, 0u, 0u, 0u);
    };
    TileDesc td = tile_desc(0);
    // stage API (caller-owned levels): only tiles that lie inside the image with their halo load without range tests
    const bool all_in = !STAGE || (td.tx * kTileW >= 4 && td.tx * kTileW + kTileW + 4 <= g.lv[td.level].pitch && td.ty * kTileH >= 4 &&
                                   td.ty * kTileH + kTileH + 4 <= g.lv[td.level].h); // uniform
    if (all_in) {
        load_tile(td);
        store_tile();
    } else {
        const int P = g.lv[td.level].pitch, H = g.lv[td.level].h;
        const uint8_t *img = pyr + (size_t)f * g.frame_stride + g.lv[td.level].offset;
        const int x0 = td.tx * kTileW, y0 = td.ty * kTileH;
        int r = tid / kPxDw, q = tid - r * kPxDw;
        uint32_t off = (uint32_t)(__mul24(y0 - 4 + r, P) + x0 - 4 + 4 * q); // wraps harmlessly when unused
        const uint32_t dstep = (uint32_t)((256 / kPxDw) * P + 4 * (256 % kPxDw));
#pragma unroll
        for (int i0 = 0; i0 < kPxH * kPxDw; i0 += 256) {
            const int i = i0 + tid;
            if (i0 + 256 <= kPxH * kPxDw || i < kPxH * kPxDw) {
                uint32_t v = 0;
                const int gy = y0 - 4 + r, gx = x0 - 4 + 4 * q;
                if (gy >= 0 && gy < H && gx >= 0 && gx < P) v = *reinterpret_cast<const uint32_t *>(img + off);
                s_px32[i] = v;
            }
            off += dstep;
            r += 256 / kPxDw;
            q += 256 % kPxDw;
            if (q >= kPxDw) { // carry into the next row
                q -= kPxDw;
                r += 1;
                off += (uint32_t)(P - 4 * kPxDw);
            }
        }
    }
    zero_scores();
    if (tid < kMaxLdsCells) s_key[tid] = 0u; // (kMaxLdsCells = 256: the whole array, whatever the tile's cell count)
    if (tid == 0) s_qcount = 0;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint16_t *q1 = s_q + 256; // the queue proper
    // slot of this wave's p-th positive: in place, inside the batches the wave itself has consumed (64 (wv + 4 j) ..)
    auto q2slot = [wv](int p) { return ((p >> 6) << 8) + (wv << 6) + (p & 63); };
    const uint32_t t2 = (uint32_t)g.threshold * 0x00010001u;
    const CompassConst ck = compass_const(g.threshold);
    constexpr int kCompass = ARC == 0 ? 2 : (ARC >= 12 ? 1 : 0);
    static_assert(kMaxLdsCells == 256, "s_key is cleared by one store per thread");
    for (int u = 0; u < n_here; u++) { // uniform
    const int l = td.level;
    const int W = g.lv[l].w, H = g.lv[l].h;
    const int x0 = td.tx * kTileW, y0 = td.ty * kTileH;
    const int c = g.cell >> l, lc = ilog2(c);
    const bool lds_cells = c >= 4;
    const int lnc = 6 - lc, ncx = 1 << lnc, ncy = 1 << lnc; // cells per tile edge: 64 / c, c = 1 .. 64 a power of two
    static_assert(kTileW == 64 && kTileH == 64, "cells per tile edge as a shift");
    __syncthreads(); // the tile's pixels, the cleared score tile, cell keys and queue counter are in LDS
    // the next tile's pixels: requested now, stored behind the ring test (phase C is the last reader of this tile's)
    TileDesc td_next = td;
    if (kT > 1 && u + 1 < n_here) {
        td_next = tile_desc(u + 1);
        load_tile(td_next);
    }

    // ---- B: compass pre-test, 4 pixels per lane; score-tile row r <-> image y0 - 1 + r,
    //         pixel-tile column px <-> image x0 - 4 + px
    int n1 = 0; // wave-uniform fill level of q1
    // Task = one dword group q (pixels px = 4q .. 4q+3) of score row r.  The first kMainTrips
    // trips cover the tile proper (rows 1..kTileH, groups 1..16, 16 rows per trip: the lane keeps its group, so
    // the column masks are formed once); the last trip the 1-pixel halo ring (rows 0 and kScH-1, and the single
    // pixels px = 3 / px = 68 of every row).  A trip's four flags sit in bits 7, 15, 23, 31; trip t's word is
    // shifted right by t, so that ONE word holds all 20 flags of the lane: bit 8 i + 7 - t = (trip t, pixel i).
    // `inner` tiles (no image border inside the score tile) skip the range tests.
    const bool inner = x0 >= 4 && x0 + kTileW < W - 3 && y0 >= 4 && y0 + kTileH < H - 3;
    static_assert(kTrips <= 8, "one flag word per lane");
    // pixels i of a group starting at image column xb that are >= 3 from the left / right border, as flag bits
    auto colmask = [&](int xb) {
        int lo_i = 3 - xb, hi_i = W - 3 - xb; // pixel i valid iff lo_i <= i < hi_i
        lo_i = lo_i < 0 ? 0 : (lo_i > 4 ? 4 : lo_i);
        hi_i = hi_i < 0 ? 0 : (hi_i > 4 ? 4 : hi_i);
        const uint32_t m4 = hi_i > lo_i ? (((1u << hi_i) - 1u) & ~((1u << lo_i) - 1u)) : 0u;
        return ((m4 * 0x00204081u) & 0x01010101u) << 7; // bit i -> bit 8 i + 7
    };
    uint32_t allflags = 0;
    {
        // a lane's four trips are four ADJACENT rows (a 4 x 4 pixel block; the wave covers a band of 16 rows), so
        // that the 64 consecutive queue entries of a ring-test batch are a compact patch of the tile: with rows 16
        // apart per lane (rounds 1-2) a batch held the same columns of rows r, r + 16, r + 32, r + 48, and 16 rows
        // of 18 dwords are a multiple of the 32 LDS banks -- systematic 4-way conflicts on all 17 ring reads
        const int q = (tid & 15) + 1, r0 = 4 * (tid >> 4) + 1;
        const uint32_t cm = inner ? kH8 : colmask(x0 - 4 + 4 * q);
        const uint32_t *row = s_px32 + (r0 + 3) * kPxDw + q;
        static_assert(kMainTrips == 4, "a lane's trips are the rows of its 4 x 4 block");
        // a wave's band of 16 rows starts at image row y0 + 16 wv: in the last row of tiles of a level the bands that lie
        // entirely at or below H - 3 hold no testable pixel (640x480: half of each of the 10 bottom tiles of level 0) and
        // skip their four compass trips -- wave-uniform, so a scalar branch (r4)
        const int n_trips = (y0 + 16 * wv < H - 3) ? kMainTrips : 0;
#pragma unroll
        for (int trip = 0; trip < kMainTrips; trip++, row += kPxDw) {
            if (trip >= n_trips) break;
            const uint32_t C = row[0], L = row[-1], Rr = row[1];
            const uint32_t Nd = row[-3 * kPxDw], Sd = row[3 * kPxDw];
            const uint32_t Wd = __builtin_amdgcn_alignbyte(C, L, 1);  // bytes x-3 of the 4 pixels
            const uint32_t Ed = __builtin_amdgcn_alignbyte(Rr, C, 3); // bytes x+3
            uint32_t f = compass4<kCompass>(C, Wd, Ed, Nd, Sd, ck) & cm;
            if (!inner) {
                const int y = y0 - 1 + r0 + trip;
                f = (y >= 3 && y < H - 3) ? f : 0u;
            }
            allflags |= f >> trip;
        }
    }
    uint32_t ehalo = 0; // (r << 7) | 4 q of the halo trip's group
    if (tid < 32 + 2 * kScH) { // waves 0..2
        int r, q;
        uint32_t mask;
        if (tid < 32) {
            q = (tid & 15) + 1;
            r = tid < 16 ? 0 : kScH - 1;
            mask = kH8;
        } else if (tid < 32 + kScH) {
            q = 0;
            r = tid - 32;
            mask = 0x80000000u; // px = 3
        } else {
            q = kPxDw - 1;
            r = tid - 32 - kScH;
            mask = 0x00000080u; // px = 68
        }
        if (!inner) {
            const int y = y0 - 1 + r;
            mask &= (y >= 3 && y < H - 3) ? colmask(x0 - 4 + 4 * q) : 0u;
        }
        const uint32_t *row = s_px32 + (r + 3) * kPxDw;
        const uint32_t C = row[q];
        // (q = 0 / 17: the neighbour dword is the adjacent row's -- a valid LDS word whose bytes only reach pixels the
        // mask drops: px = 3 takes its W from C's byte 0, px = 68 its E from C's byte 3 -- so the reads need no test)
        const uint32_t L = row[q - 1], Rr = row[q + 1];
        const uint32_t Nd = row[q - 3 * kPxDw], Sd = row[q + 3 * kPxDw];
        const uint32_t Wd = __builtin_amdgcn_alignbyte(C, L, 1);
        const uint32_t Ed = __builtin_amdgcn_alignbyte(Rr, C, 3);
        allflags |= (compass4<kCompass>(C, Wd, Ed, Nd, Sd, ck) & mask) >> kMainTrips;
        ehalo = (uint32_t)((r << 7) | (4 * q));
    }
    // wave-level compaction, ONCE for all trips: a lane's candidates take consecutive slots starting at
    // the exclusive prefix of the per-lane counts (one DPP scan per tile and wave).  The 20 stores carry no
    // branch and no EXEC change (each was v_and + v_cmp + s_and_saveexec + ... + s_or: 40 scalar instructions per
    // wave): a clear flag sends the store to the lane's own dump halfword behind the queue.
    {
        const int cnt = __popc(allflags);
        const int incl = wave_incl_scan_i32(cnt);
        n1 = __builtin_amdgcn_readlane(incl, 63);
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_qcount, n1); // the wave's run in the workgroup's queue
        base = __builtin_amdgcn_readfirstlane(base);
        // byte offsets inside s_q: the dump halfwords lie BELOW the queue, so that next slot - dump is a small
        // positive number and flag * delta is one v_mad_u32_u24
        const uint32_t dump = (uint32_t)tid * 2u;
        uint32_t delta = 512u + (uint32_t)(base + incl - cnt) * 2u - dump;
        uint8_t *qb = reinterpret_cast<uint8_t *>(s_q);
        const uint32_t emain = (uint32_t)(((4 * (tid >> 4) + 1) << 7) | (4 * ((tid & 15) + 1)));
#pragma unroll
        for (int trip = 0; trip < kTrips; trip++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t bit = (allflags >> (8 * i + 7 - trip)) & 1u;
                const uint32_t e = trip < kMainTrips ? emain + (uint32_t)(trip << 7) + (uint32_t)i : ehalo + (uint32_t)i;
                *reinterpret_cast<uint16_t *>(qb + dump + __umul24(bit, delta)) = (uint16_t)e;
                // delta += 2 * bit as ONE instruction (left to itself hipcc keeps a count and rebuilds delta from it: two)
                asm("v_lshl_add_u32 %0, %1, 1, %0" : "+v"(delta) : "v"(bit));
            }
    }
    __syncthreads(); // the queue is complete
    const int nq = s_qcount;

    // ---- C: full ring test, batches of 64 candidates dealt round-robin to the waves
    int n2 = 0;
    for (int i0 = 64 * wv; i0 < nq; i0 += 256) {
        const int i = i0 + lane;
        bool pos = false;
        int e = 0;
        if (i < nq) {
            e = q1[i];
            const int r = e >> 7, px = e & 127;
            const int sc = fast_score_packed(s_px + (r + 3) * kPxW + px, t2, ARC, st.lut);
            if (sc) {
                s_sc[r * kScPitch + px - 3] = (uint16_t)sc;
                pos = r >= 1 && r <= kTileH && px >= 4 && px < 4 + kTileW;
            }
        }
        const uint64_t m = __ballot(pos);
        if (pos) q1[q2slot(n2 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)))] = (uint16_t)e;
        n2 += (int)__popcll(m);
    }
    __syncthreads(); // every wave's scores are in s_sc
    if (kT > 1 && u + 1 < n_here) store_tile(); // (uniform) nobody reads this tile's pixels any more

    if (STAGE && st.resp[l]) { // the tile's part of the caller's response map, zeros included
        float *rp = st.resp[l];
        const int rpitch = st.resp_pitch[l];
        for (int i = tid; i < kTileW * kTileH; i += 256) {
            const int ry = i / kTileW, rx = i - ry * kTileW;
            if (x0 + rx < W && y0 + ry < H)
                rp[(size_t)(y0 + ry) * rpitch + x0 + rx] = (float)s_sc[(ry + 1) * kScPitch + rx + 1];
        }
    }

    // ---- D: strict 3x3 maximum on the positives, then the cell's maximum key
    for (int i = lane; i < n2; i += 64) {
        const int e = q1[q2slot(i)];
        const int r = e >> 7, px = e & 127;
        const uint16_t *q = &s_sc[r * kScPitch + px - 3];
        const int v = q[0];
        // strictly greater than all 8 neighbours <=> greater than their maximum (three v_max3 and one v_max instead
        // of eight compares chained by EXEC updates: the short-circuit form cost 16 scalar instructions per trip)
        const int m0 = max(max((int)q[-kScPitch - 1], (int)q[-kScPitch]), (int)q[-kScPitch + 1]);
        const int m1 = max(max((int)q[-1], (int)q[1]), (int)q[kScPitch - 1]);
        const int m2 = max((int)q[kScPitch], (int)q[kScPitch + 1]);
        if (v <= max(max(m0, m1), m2)) continue;
        const int rx = px - 4, ry = r - 1;
        const uint32_t key = nms_key(v, l, x0 + rx, y0 + ry, g.cell);
        if (lds_cells) {
            atomicMax(&s_key[(ry >> lc) * ncx + (rx >> lc)], key);
        } else {
            const int cx = (x0 + rx) >> lc, cy = (y0 + ry) >> lc;
            if (cx < g.cells_x && cy < g.cells_y)
                atomicMax(&cellkey[(size_t)f * g.K + cy * g.cells_x + cx], key);
        }
    }
    const bool more = kT > 1 && u + 1 < n_here; // uniform
    if (!lds_cells && !more) return;
    __syncthreads(); // the cell keys are final; every wave is done with the score tile and the queue
    if (lds_cells) {
        for (int i = tid; i < ncx * ncy; i += 256) {
            const uint32_t key = s_key[i];
            if (key == 0u) continue;
            if (more) s_key[i] = 0u; // (the same thread that has just read it)
            const int cx = (x0 >> lc) + (i & (ncx - 1)), cy = (y0 >> lc) + (i >> lnc);
            if (cx < g.cells_x && cy < g.cells_y)
                atomicMax(&cellkey[(size_t)f * g.K + cy * g.cells_x + cx], key);
        }
    }
    if (!more) return;
    zero_scores();
    if (tid == 0) s_qcount = 0;
    td = td_next;
    } // tiles of this workgroup
}

// ------------------------------------------------------------------------------------
// Selection + compaction, one workgroup per frame.  Reference mode keeps every non-empty
// cell; max_features = N keeps the N best by (score desc, cell asc) (EXT ii).  Output is in
// cell order either way, so it is deterministic.  Also fills the optional SoA view.
// ------------------------------------------------------------------------------------
constexpr int kSelThreads = 1024; // one workgroup per frame: wide, because the kernel is latency-bound
constexpr int kSelWaves = kSelThreads / 64;

template <bool SOA> // SOA: the optional cell-indexed view is wanted (a separate instantiation: its six
                     // pointers and per-cell stores otherwise cost the common case ~1000 SGPR spill moves)
__global__ void __launch_bounds__(kSelThreads)
select_kernel(DeviceGeom g, const uint32_t *__restrict__ cellkey, uint4 *__restrict__ sel, uint16_t *__restrict__ cellslot,
              int32_t *__restrict__ selcount, int32_t *__restrict__ counts_out, orbfe_soa soa)
{
    constexpr int kBinsPer = 4096 / kSelThreads; // score bins owned by one thread
    __shared__ uint32_t s_hist[4096];
    __shared__ int s_wave[kSelWaves];
    __shared__ int s_thr, s_quota;
    const int f = blockIdx.x, tid = threadIdx.x;
    const uint32_t *keys = cellkey + (size_t)f * g.K;

    if (tid == 0) {
        s_thr = 0;          // keep score > s_thr ...
        s_quota = 0;        // ... plus the first s_quota cells with score == s_thr
    }
    // A thread takes kPer consecutive cells of every chunk of kSelThreads * kPer cells.  The frame's cell
    // keys are read twice (histogram, compaction).  Everything the histogram needs is requested up
    // front, unconditionally (index clamped, value masked), so that all of a thread's loads are in
    // flight together instead of one load + wait per trip: kRegChunks chunks stay in registers for the
    // compaction, kHistChunks more (a 4K frame with 16-px cells has 32 400 cells = 8 chunks) only feed
    // the histogram and are read again, one chunk ahead, during the compaction.
    constexpr int kPer = 4, kChunkCells = kSelThreads * kPer;
    constexpr int kRegChunks = 4, kHistChunks = 4;
    const int n_chunks = (g.K + kChunkCells - 1) / kChunkCells;
    // (one 16-byte load per thread and chunk when the frame's keys are 16-byte aligned and the four cells exist)
    const bool vec = g.K >= 4 && (((size_t)f * g.K) & 3u) == 0 && (reinterpret_cast<uintptr_t>(cellkey) & 15u) == 0; // block-uniform
    auto load_chunk = [&](int c, uint32_t (&v)[kPer]) { // keys of chunk c, 0 past the end
        static_assert(kPer == 4, "uint4");
        const int k0 = c * kChunkCells + kPer * tid;
        if (vec) {
            const int kc = k0 + kPer <= g.K ? k0 : 0; // a group that crosses the end falls back to the masked tail below
            const uint4 q = *reinterpret_cast<const uint4 *>(keys + kc);
            v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
            if (k0 + kPer > g.K) {
#pragma unroll
                for (int j = 0; j < kPer; j++) v[j] = k0 + j < g.K ? keys[k0 + j] : 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < kPer; j++) {
                const int k = k0 + j;
                v[j] = keys[k < g.K ? k : g.K - 1];
            }
#pragma unroll
            for (int j = 0; j < kPer; j++) v[j] = k0 + j < g.K ? v[j] : 0u;
        }
    };
    uint32_t kreg[kRegChunks][kPer], khist[kHistChunks][kPer];
#pragma unroll
    for (int c = 0; c < kRegChunks; c++) load_chunk(c, kreg[c]);
    const bool more = g.max_features > 0 && n_chunks > kRegChunks; // block-uniform
    if (more) {
#pragma unroll
        for (int c = 0; c < kHistChunks; c++) load_chunk(kRegChunks + c, khist[c]);
    }
    if (g.max_features > 0) {
        for (int i = tid; i < 4096; i += kSelThreads) s_hist[i] = 0u;
        __syncthreads();
        auto count = [&](const uint32_t (&v)[kPer]) {
#pragma unroll
            for (int j = 0; j < kPer; j++) {
                const uint32_t sc = v[j] >> 15;
                if (sc) atomicAdd(&s_hist[sc], 1u);
            }
        };
#pragma unroll
        for (int c = 0; c < kRegChunks; c++) count(kreg[c]);
        if (more) {
#pragma unroll
            for (int c = 0; c < kHistChunks; c++) count(khist[c]);
            for (int c = kRegChunks + kHistChunks; c < n_chunks; c++) { // still larger grids
                uint32_t v[kPer];
                load_chunk(c, v);
                count(v);
            }
        }
        __syncthreads();
        // thread t owns bins kBinsPer * t ...; `above` = how many scores lie in higher bins:
        // total - inclusive prefix over threads (DPP wave scan + one cross-wave step)
        int mine = 0;
        for (int b = 0; b < kBinsPer; b++) mine += (int)s_hist[kBinsPer * tid + b];
        const int incl = wave_incl_scan_i32(mine);
        if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
        __syncthreads();
        int wave_off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kSelWaves; w++) {
            const int v = s_wave[w];
            wave_off += w < (tid >> 6) ? v : 0;
            total += v;
        }
        int above = total - (wave_off + incl); // scores in bins owned by higher threads
        for (int b = kBinsPer - 1; b >= 0; b--) {
            const int n = (int)s_hist[kBinsPer * tid + b];
            if (above < g.max_features && above + n >= g.max_features && n > 0) {
                s_thr = kBinsPer * tid + b;
                s_quota = g.max_features - above;
            }
            above += n;
        }
    }
    __syncthreads();
    const int thr = s_thr, quota = s_quota;

    // Compaction in cell order.  Kept cells = every score above the threshold plus the first `quota`
    // cells AT the threshold, so a cell's slot is (# above before it) + min(# ties before it, quota):
    // one block-wide prefix of a packed (above, tie) count per chunk -- DPP scan inside the wave, one
    // barrier per chunk for the cross-wave table, which alternates between two buffers.
    __shared__ uint32_t s_cnt[2][kSelWaves];
    const int lane = tid & 63, wv = tid >> 6;
    constexpr bool want_soa = SOA;
    int n_gt = 0, n_tie = 0;
    int ecy = (kPer * tid) / g.cells_x, ecx = kPer * tid - ecy * g.cells_x; // this thread's first cell of chunk 0
    const int chunk_dy = kChunkCells / g.cells_x, chunk_dx = kChunkCells - chunk_dy * g.cells_x;
    auto emit = [&](int c, const uint32_t (&key)[kPer], int par) {
        const int k0 = c * kChunkCells + kPer * tid;
        int ccx = ecx, ccy = ecy; // cell coordinates of k0 (no division per key: emit runs for c = 0, 1, 2, ...)
        ecx += chunk_dx;
        ecy += chunk_dy;
        if (ecx >= g.cells_x) ecx -= g.cells_x, ecy++;
        uint32_t mine = 0; // (above, ties) among this thread's cells, packed 16 : 16
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const int score = (int)(key[j] >> 15); // 0 past the end of the grid
            mine += (score > thr ? 1u : 0u) + ((thr > 0 && score == thr) ? 0x10000u : 0u); // thr = 0: every non-empty cell
        }
        const uint32_t incl = (uint32_t)wave_incl_scan_i32((int)mine);
        if (lane == 63) s_cnt[par][wv] = incl;
        __syncthreads();
        uint32_t off = 0, tot = 0;
#pragma unroll
        for (int u = 0; u < kSelWaves; u++) {
            const uint32_t v = s_cnt[par][u];
            off += u < wv ? v : 0u;
            tot += v;
        }
        const uint32_t before = off + incl - mine;
        int gt_before = n_gt + (int)(before & 0xFFFFu), tie_before = n_tie + (int)(before >> 16);
        uint32_t slots[kPer];
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const int k = k0 + j;
            const bool in = k < g.K;
            const int score = (int)(key[j] >> 15);
            const bool gt = score > thr, tie = thr > 0 && score == thr;
            const bool keep = gt || (tie && tie_before < quota);
            const int slot = gt_before + (tie_before < quota ? tie_before : quota);
            gt_before += gt;
            tie_before += tie;
            int sc = 0, l = 0, x = 0, y = 0;
            // without the SoA view only the kept cells need their position: it is filled in below, from the
            // compacted list (a quarter of the cells at C5), not here for every cell
            if (want_soa && in) nms_decode(key[j], ccx, ccy, g.cell, &sc, &l, &x, &y);
            if (++ccx == g.cells_x) ccx = 0, ccy++; // cell k + 1
            if (keep && sel)
                sel[(size_t)f * g.cap + slot] = make_uint4((uint32_t)k, (uint32_t)x | ((uint32_t)y << 16), key[j], 0u);
            slots[j] = keep ? (uint32_t)slot : 0xFFFFu; // cap <= 65535
            if (want_soa && in) {
                const size_t o = (size_t)f * g.K + k;
                if (soa.d_pos) {
                    soa.d_pos[2 * o] = (float)x;
                    soa.d_pos[2 * o + 1] = (float)y;
                }
                if (soa.d_score) soa.d_score[o] = (float)sc;
                if (soa.d_level) soa.d_level[o] = l;
                if (!keep) { // unselected / empty cells: zero angle and descriptor (Q5)
                    if (soa.d_angle) soa.d_angle[o] = 0.0f;
                    if (soa.d_desc32) soa.d_desc32[o] = 0u;
                    if (soa.d_desc) {
                        uint32_t *d = reinterpret_cast<uint32_t *>(soa.d_desc + 32 * o);
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) d[jj] = 0u;
                    }
                }
            }
        }
        uint16_t *cs = cellslot + (size_t)f * g.K + k0;
        if (vec && k0 + kPer <= g.K) { // the four u16 slots as one 8-byte store
            *reinterpret_cast<uint2 *>(cs) = make_uint2(slots[0] | (slots[1] << 16), slots[2] | (slots[3] << 16));
        } else {
#pragma unroll
            for (int j = 0; j < kPer; j++)
                if (k0 + j < g.K) cs[j] = (uint16_t)slots[j];
        }
        n_gt += (int)(tot & 0xFFFFu);
        n_tie += (int)(tot >> 16);
    };
#pragma unroll
    for (int c = 0; c < kRegChunks; c++)
        if (c < n_chunks) emit(c, kreg[c], c & 1); // block-uniform condition
    if (n_chunks > kRegChunks) {
        // the rest: the next chunk's keys are requested before this chunk's prefix, so their latency is hidden
        uint32_t cur[kPer], nxt[kPer];
        load_chunk(kRegChunks, cur);
        for (int c = kRegChunks; c < n_chunks; c++) {
            load_chunk(c + 1, nxt);
            emit(c, cur, c & 1);
#pragma unroll
            for (int j = 0; j < kPer; j++) cur[j] = nxt[j];
        }
    }
    const int n_sel = n_gt + (n_tie < quota ? n_tie : quota);
    if (!want_soa && sel) { // positions of the kept cells
        __syncthreads();    // the list entries were written by other threads of this workgroup
        for (int i = tid; i < n_sel; i += kSelThreads) {
            uint4 *e = sel + (size_t)f * g.cap + i;
            const uint32_t k = e->x, key = e->z;
            const int cy = (int)k / g.cells_x;
            int sc, l, x, y;
            nms_decode(key, (int)k - cy * g.cells_x, cy, g.cell, &sc, &l, &x, &y);
            e->y = (uint32_t)x | ((uint32_t)y << 16);
        }
    }
    if (tid == 0) {
        selcount[f] = n_sel;
        if (counts_out) counts_out[f] = n_sel;
    }
}

// ------------------------------------------------------------------------------------
// a8 + a9 + a10 fused: one wave per selected keypoint -> one 52-byte record (+ SoA view).
// Orientation and descriptor sample the level-0 image only (Q10).
// ------------------------------------------------------------------------------------
// The patch a keypoint needs (radius R: 15 for the moments disc and the reference-mode
// descriptor, 19 once the descriptor really rotates) is staged in LDS by the wave with
// independent LDS-DMA loads (one memory latency instead of ~20 dependent byte gathers); moments
// and the 512 descriptor samples then read LDS.
// Staging is the largest part of this kernel and is bound by the texture path, which wants every
// quad of lanes to fetch 16 contiguous bytes: with 4-byte DMAs over 36-byte rows (9 dwords, quads
// straddling rows) the stage took 0.226 ms, with 32-byte rows 0.174 ms, unaligned origins were
// slower still (ablations in DESIGN.md 4.3).  So the patch origin is ((x - R) & ~15, y - R), rows are
// 48 (R = 15) / 64 (R = 19) bytes, and one lane moves one aligned 16-byte chunk
// (global_load_lds_dwordx4): 2 / 3 DMA instructions per patch.  The keypoint sits at byte
// (R, R + phase) of its patch, phase = (x - R) & 15.
constexpr int kKpw = 4; // keypoints per wave (consecutive slots = neighbouring cells)

template <int R>
struct PatchGeom {
    static constexpr int kRows = 2 * R + 1;
    // bytes one lane stages.  (The 12-byte LDS-DMA would fit 36-byte rows, but it writes lane L at
    // LDS base + 16 L, leaving a 4-byte hole after every 12 bytes: tools/dma12_probe.hip.)
    static constexpr int kChunk = 16;
    static constexpr int kAlign = 16;                              // alignment of the patch origin
    static constexpr int kPitch = (2 * R + 1 + kAlign - 1 + kChunk - 1) / kChunk * kChunk; // 48 / 64
    static constexpr int kChunksRow = kPitch / kChunk;
    static constexpr int kChunks = kRows * kChunksRow;             // chunks per patch
    static constexpr int kTrips = (kChunks + 63) / 64;             // DMA instructions per patch
    static constexpr int kPatchBytes = (kChunks * kChunk + 3) / 4 * 4;
    // Intensity-centroid moments on the matrix cores: m10 = sum dx * I, m01 = sum dy * I over
    // the radius-15 disc are dot products of the patch bytes with fixed weights.  One
    // v_mfma_i32_16x16x64_i8 takes 16 rows x 64 bytes: row = (keypoint, K slice s), 4 keypoints x
    // 4 slices of 64 consecutive patch bytes; column = (slice s', x or y weights) -- the entries
    // with s = s' are the partial sums, the rest is ignored.  kMom such steps cover the 31 disc
    // rows (which start kStart bytes into the patch).  The alignment phase is absorbed by reading
    // the A fragment at byte address patch + phase (a misaligned ds_read_b128, which gfx950
    // serves): the linear byte stream shifted by phase has the keypoint at column R of every
    // row, and the bytes that wrap into the next row fall on columns >= kPitch - 15 > 30, whose
    // weight is 0.  Pixels enter as I ^ 0x80 (signed I - 128); exact, because the disc's weights
    // sum to zero.
    static constexpr int kStart = (R - 15) * kPitch;          // multiple of 16
    static constexpr int kMom = (31 * kPitch + 255) / 256;    // 6 (R = 15) / 8 (R = 19)
    static_assert(kPitch - (kAlign - 1) > 30, "wrapped bytes must fall outside the disc columns");
};

// Weight operand of the moment MFMAs for patch radius R, in B-fragment order [kMom][lane 64][16]:
// lane = 16 * chunk + column, column = 2 * slice + (0: dx weights, 1: dy weights), columns 8..15 zero.
// Disc chords: orb.cu:79-80 (u_max), the same table as c_umax in device_common.hpp.
static std::vector<int8_t> make_moment_weights(int R)
{
    static const int u[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 7, 5, 0};
    const int pitch = R == 15 ? PatchGeom<15>::kPitch : PatchGeom<19>::kPitch, start = (R - 15) * pitch,
              n = (31 * pitch + 255) / 256;
    std::vector<int8_t> w((size_t)n * 64 * 16, 0);
    for (int ks = 0; ks < n; ks++)
        for (int lane = 0; lane < 64; lane++) {
            const int col = lane & 15, c = lane >> 4;
            if (col >= 8) continue;
            const int sl = col >> 1, xy = col & 1;
            for (int j = 0; j < 16; j++) {
                const int pb = start + ks * 256 + sl * 64 + c * 16 + j; // patch byte
                const int dy = pb / pitch - R, dx = pb % pitch - R;
                const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
                if (ady <= 15 && adx <= u[ady]) w[((size_t)ks * 64 + lane) * 16 + j] = (int8_t)(xy ? dy : dx);
            }
        }
    return w;
}

// One wave = kKpw keypoints, in three passes so that the transcendental math is not repeated
// by all 64 lanes for every keypoint:
//   1  stage the kKpw patches in LDS; their moments = kMom int8 MFMAs for all four at once
//   2  lanes 0..kKpw-1 each take one keypoint: atan2f, steering cos/sin -- ONCE per wave
//   3  per keypoint: 256 rotated tests from its LDS patch, 4 ballots = the descriptor
typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// DL = EXT iv, descriptor_level: a keypoint's patch comes from the pyramid level that won its cell, at
// position >> level, with that level's size in every bound (its own instantiation: level 0 stays as it was).
template <int R, bool SOA, bool DL>
__global__ void __launch_bounds__(256)
describe_kernel(DeviceGeom g, const uint8_t *__restrict__ pyr, const uint4 *__restrict__ sel,
                const int32_t *__restrict__ selcount, const uint4 *__restrict__ momw,
                orbfe_keypoint *__restrict__ records, orbfe_soa soa, SteerArgs steer)
{
    using G = PatchGeom<R>;
    // TAB: sample offsets from the orientation table (as in describe_tile_kernel; R = 15 <=> !angle_in_radians)
    constexpr bool TAB = R == 15;
    constexpr int kRows = G::kRows, kPitch = G::kPitch, kPatchBytes = G::kPatchBytes;
    // + 256 bytes: the last moment step of the last patch reads (zero-weighted) bytes past it
    __shared__ __attribute__((aligned(16))) uint8_t s_patch[4 * kKpw * kPatchBytes + 256];
    static_assert(G::kStart + G::kMom * 256 + 15 + 4 <= kPatchBytes + 256, "moment reads stay inside s_patch");

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // scalar: wave-uniform
    int f, blk;
    if (!frame_item(g, &f, &blk)) return;
    const int n = selcount[f];
    const int slot0 = (blk * 4 + wv) * kKpw; // wave-uniform
    if (slot0 >= n) return;
    const int nk = n - slot0 < kKpw ? n - slot0 : kKpw;
    const uint8_t *fbase = pyr + (size_t)f * g.frame_stride;
    const uint4 *fsel = sel + (size_t)f * g.cap + slot0;
    uint8_t *wpatch = s_patch + wv * (kKpw * kPatchBytes); // this wave's kKpw patches
    // weight fragments of the moment MFMAs and this lane's four rBRIEF pattern rows (the same for
    // every keypoint): issued first, they arrive under the patch staging
    uint4 bw[G::kMom];
#pragma unroll
    for (int ks = 0; ks < G::kMom; ks++) bw[ks] = momw[ks * 64 + lane];
    float4 pat[4];
    // TAB: lane i holds break points i, i + 64, i + 128, i + 192 of the table (+inf past the end): a keypoint's
    // interval = the number of break points <= its orientation = four ballots + four s_bcnt1, no search
    float brk[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (!TAB) pat[r] = reinterpret_cast<const float4 *>(c_pattern_f)[64 * r + lane];
        else brk[r] = 64 * r + lane < steer.n_breaks ? steer.breaks[64 * r + lane] : __builtin_inff();
    }
    static_assert(kSteerMaxBreaks <= 256, "four break points per lane");

    int kx[kKpw], ky[kKpw], kax[kKpw], m10[kKpw], m01[kKpw], klv[kKpw];
    uint32_t kcell[kKpw], kkey[kKpw];
    // chunk c of a patch = (row c / kChunksRow, 16-byte column c % kChunksRow); trip t stages chunks
    // 64 t + lane.  Their (row, byte column) relative to the patch origin, once per wave; the image offset
    // is row * pitch + column with the pitch of the keypoint's level.
    int vrow[G::kTrips], vcol[G::kTrips];
#pragma unroll
    for (int t = 0; t < G::kTrips; t++) {
        const int c = 64 * t + lane;
        vrow[t] = c / G::kChunksRow;
        vcol[t] = G::kChunk * (c - vrow[t] * G::kChunksRow);
    }
    // the wave's selection entries: uniform addresses = scalar loads, all four issued before the first
    // use (behind the uniform `it >= nk` test each would be followed by its own wait)
    uint4 srs[kKpw];
#pragma unroll
    for (int it = 0; it < kKpw; it++) srs[it] = fsel[it < nk ? it : nk - 1];
    // ---- pass 1: stage + moments
#pragma unroll
    for (int it = 0; it < kKpw; it++) {
        kx[it] = ky[it] = kax[it] = m10[it] = m01[it] = klv[it] = 0;
        kcell[it] = kkey[it] = 0;
        if (it >= nk) continue; // uniform
        const uint4 sr = srs[it];
        kcell[it] = sr.x;
        kkey[it] = sr.z;
        const int l = DL ? 7 - (int)((sr.z >> 12) & 7u) : 0; // uniform: the level that won the cell
        klv[it] = l;
        const int Wl = g.lv[l].w, Hl = g.lv[l].h, P = g.lv[l].pitch;
        const uint8_t *img = fbase + g.lv[l].offset;
        const int x = (int)(sr.y & 0xFFFFu) >> l, y = (int)(sr.y >> 16) >> l; // level coordinates (exact: pos = coord << l)
        kx[it] = x;
        ky[it] = y;
        const int oy = y - R;
        const int ax = (x - R) & ~(G::kAlign - 1); // may be negative, still a multiple of kAlign
        kax[it] = ax;
        uint8_t *sp = wpatch + it * kPatchBytes;
        // Plain copies when (a) every chunk lies inside the image's rows and padded pitch and (b) the
        // disc meets no excluded pixel (row <= 0, row >= H, column <= 0, column >= W: orb.cu:98,112,
        // 119).  Bytes outside the disc or the image width then carry zero weight and are never
        // sampled by the descriptor (17 / 19-px guard band), so their values do not matter.
        const bool inner = oy >= 0 && oy + kRows <= Hl && ax >= 0 && ax + kPitch <= P && y > 15 && y + 15 < Hl &&
                           x > 15 && x + 15 < Wl; // uniform
        if (inner) {
            // LDS-DMA: global_load_lds_dwordx4 writes lane L's 16 bytes to (uniform LDS base) + 16 L
            // with no VGPR round trip and no ds_write.  Address = scalar patch origin + the lane's
            // patch-relative offset of this trip: no VALU per DMA; every lane address is 16-byte
            // aligned (P is a multiple of 64).
            const uint8_t *pb = img + ((ptrdiff_t)oy * P + ax);
#pragma unroll
            for (int t = 0; t < G::kTrips; t++) {
                // (the size must be a literal: a template-dependent constant there keeps the host pass
                // from emitting the kernel's stub)
                static_assert(G::kChunk == 16, "DMA size literal below");
                if (64 * (t + 1) <= G::kChunks || 64 * t + lane < G::kChunks)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pb + (uint32_t)(__mul24(vrow[t], P) + vcol[t])),
                                                     (__attribute__((address_space(3))) void *)(sp + 1024 * t), 16, 0, 0);
            }
        } else {
            // border keypoints: dword by dword, excluded pixels staged as 0
            for (int i = lane; i < kPatchBytes / 4; i += 64) {
                const int r = i / (kPitch / 4), q = i - r * (kPitch / 4);
                const int gy = oy + r, gx = ax + 4 * q;
                uint32_t v = 0;
                if (gy > 0 && gy < Hl && gx >= 0 && gx < Wl) {
                    v = *reinterpret_cast<const uint32_t *>(img + (uint32_t)(__mul24(gy, P) + gx));
                    if (gx == 0) v &= 0xFFFFFF00u;
                    const int nv = Wl - gx; // valid bytes in this dword
                    if (nv < 4) v &= (1u << (8 * nv)) - 1u;
                }
                reinterpret_cast<uint32_t *>(sp)[i] = v;
            }
        }
    }
    // LDS-DMA data is only ordered behind this wave's vmcnt: wait for every staged dword before
    // the first LDS read (the compiler's own wait insertion does not cover it reliably)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // LDS ops of one wave are in order
    __builtin_amdgcn_wave_barrier();
    {
        // A fragment: lane = 16 * chunk + row, row = keypoint + 4 * slice: 16 consecutive bytes of the
        // phase-shifted patch stream.  Hand-written misaligned ds_read_b128 (a C++ uint4 load would
        // promise 16-byte alignment); reads and their wait are one asm block.
        const int row = lane & 15;
        int phase = kx[0] - R - kax[0];
#pragma unroll
        for (int it = 1; it < kKpw; it++) phase = (row & 3) == it ? kx[it] - R - kax[it] : phase;
        // 16 bytes from an arbitrary byte address = 5 aligned dwords realigned by v_alignbyte (byte-misaligned
        // wide LDS reads are served one lane at a time on gfx950: tools/lds_unaligned_rate.hip)
        const int abyte = (row & 3) * kPatchBytes + G::kStart + (row >> 2) * 64 + (lane >> 4) * 16 + phase;
        const uint32_t *a32 = reinterpret_cast<const uint32_t *>(wpatch + (abyte & ~3));
        const uint32_t ash = (uint32_t)abyte & 3u;
        u32x4 av[G::kMom];
#pragma unroll
        for (int ks = 0; ks < G::kMom; ks++) {
            const uint32_t *q = a32 + ks * 64;
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
            av[ks].x = __builtin_amdgcn_alignbyte(d1, d0, ash);
            av[ks].y = __builtin_amdgcn_alignbyte(d2, d1, ash);
            av[ks].z = __builtin_amdgcn_alignbyte(d3, d2, ash);
            av[ks].w = __builtin_amdgcn_alignbyte(d4, d3, ash);
        }
        v4i acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < G::kMom; ks++) {
            const uint4 bv = bw[ks];
            const v4i a = {(int)(av[ks].x ^ 0x80808080u), (int)(av[ks].y ^ 0x80808080u), (int)(av[ks].z ^ 0x80808080u),
                           (int)(av[ks].w ^ 0x80808080u)};
            const v4i b = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w};
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
        }
        // D[row][col] sits in lane 16 * (row / 4) + col, register row % 4: keypoint `it`, slice s,
        // weights xy -> register it of lane 16 s + 2 s + xy
#pragma unroll
        for (int it = 0; it < kKpw; it++) {
#pragma unroll
            for (int sl = 0; sl < 4; sl++) {
                m10[it] += __builtin_amdgcn_readlane(acc[it], 18 * sl);
                m01[it] += __builtin_amdgcn_readlane(acc[it], 18 * sl + 1);
            }
        }
    }
    // ---- pass 2: lane `it` owns keypoint `it`
    int lm10 = m10[0], lm01 = m01[0];
#pragma unroll
    for (int it = 1; it < kKpw; it++) {
        lm10 = lane == it ? m10[it] : lm10;
        lm01 = lane == it ? m01[it] : lm01;
    }
    const float langle = orbfe_atan2f((float)lm01, (float)lm10);
    float la = 0.f, lb = 0.f;
    if (!TAB) orb_steer(langle, g.angle_in_radians, &la, &lb);
    // TAB: the four keypoints' table rows, requested together before the first is used
    uint4 rows[kKpw];
    if (TAB) {
#pragma unroll
        for (int it = 0; it < kKpw; it++) {
            const float ang = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(langle), it));
            int iv = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) iv += (int)__popcll(__ballot(brk[r] <= ang));
            rows[it] = steer.table[iv * 64 + lane];
        }
    }
    // ---- pass 3: descriptors + records
#pragma unroll
    for (int it = 0; it < kKpw; it++) {
        if (it >= nk) continue;
        const float angle = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(langle), it));
        const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(la), it));
        const float b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lb), it));
        const int x = kx[it], y = ky[it], l = klv[it];
        // samples are addressed relative to s_patch: the keypoint's byte offset in it is one constant
        uint64_t d[4] = {0, 0, 0, 0};
        if (!orb_border_zero(x, y, g.lv[l].w, g.lv[l].h, g.angle_in_radians)) {
            const int c0 = (wv * kKpw + it) * kPatchBytes + R * G::kPitch + x - kax[it];
            if (TAB) {
                const uint32_t w[4] = {rows[it].x, rows[it].y, rows[it].z, rows[it].w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int t0 = s_patch[c0 + (int)(int16_t)(w[r] & 0xFFFFu)];
                    const int t1 = s_patch[c0 + ((int)w[r] >> 16)];
                    d[r] = __ballot(t0 < t1);
                }
                uint64_t t; // undo the round schedule (steer_table.cpp), scalar unit
                t = (d[1] ^ d[3]) & steer.sched_mask[3]; d[1] ^= t; d[3] ^= t;
                t = (d[0] ^ d[2]) & steer.sched_mask[2]; d[0] ^= t; d[2] ^= t;
                t = (d[2] ^ d[3]) & steer.sched_mask[1]; d[2] ^= t; d[3] ^= t;
                t = (d[0] ^ d[1]) & steer.sched_mask[0]; d[0] ^= t; d[1] ^= t;
            } else {
                orb_describe_lds<G::kPitch>(s_patch, c0, a, b, pat, d);
            }
        }

        // every value is wave-uniform: lane 0 stores the 13 dwords of the record
        if (lane == 0) {
            const int score = (int)(kkey[it] >> 15), level = 7 - (int)((kkey[it] >> 12) & 7u);
            uint32_t *rec = reinterpret_cast<uint32_t *>(records + (size_t)f * g.cap + slot0 + it);
            rec[0] = __float_as_uint((float)(x << l)); // records keep level-0 coordinates
            rec[1] = __float_as_uint((float)(y << l));
            rec[2] = __float_as_uint((float)score);
            rec[3] = (uint32_t)level;
            rec[4] = __float_as_uint(angle);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                rec[5 + 2 * k] = (uint32_t)d[k];
                rec[6 + 2 * k] = (uint32_t)(d[k] >> 32);
            }
            if (SOA && (soa.d_angle || soa.d_desc32 || soa.d_desc)) { // (SOA: its own instantiation, as in select_kernel)
                const size_t o = (size_t)f * g.K + kcell[it];
                if (soa.d_angle) soa.d_angle[o] = angle;
                if (soa.d_desc32) soa.d_desc32[o] = orb_compress(d);
                if (soa.d_desc) {
                    uint32_t *sd = reinterpret_cast<uint32_t *>(soa.d_desc + 32 * o);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        sd[2 * k] = (uint32_t)d[k];
                        sd[2 * k + 1] = (uint32_t)(d[k] >> 32);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// a8 + a9 + a10 fused, tile form: one workgroup = one 64x64 level-0 tile of one frame and every
// selected keypoint whose cell lies in it (at most one per cell, so <= (64 / cell)^2 <= 64).
// The tile plus its R-pixel halo is staged in LDS ONCE (LDS-DMA, 16-byte chunks) and shared by
// all its keypoints: ~330 staged bytes per keypoint on the bench scenes instead of a private
// 1488-byte patch each, no per-keypoint staging latency, and 9 KB of LDS per workgroup instead
// of 24, so eight workgroups fit a CU.  The keypoint list is read straight from the cell keys
// and the cell -> slot map select_kernel wrote; records go to their slots, so the output is the
// same cell-ordered array as before.
// Pixels the reference's moment loops exclude (row <= 0, row >= H, column <= 0, column >= W:
// orb.cu:98,112,119) are staged as 0; the descriptor never samples them (17 / 19-px guard band).
// ------------------------------------------------------------------------------------
constexpr int kDTile = 64;

template <int R>
struct TileGeom {
    static constexpr int kHaloX = (R + 15) / 16 * 16;                 // 16 (R = 15) / 32 (R = 19): chunk-aligned
    // + one spare chunk per row: 96 / 128 bytes put rows 4 (or all!) apart on the same LDS banks; with 112 /
    // 144 the row stride is 28 / 36 dwords and the descriptor's byte gathers spread over twice the banks
    static constexpr int kPitch = kDTile + 2 * kHaloX + 16;           // 112 / 144 bytes
    static constexpr int kRows = kDTile + 2 * R;                      // 94 / 102
    static constexpr int kChunksRow = kPitch / 16;
    static constexpr int kChunks = kRows * kChunksRow;                // 564 / 816
    static constexpr int kBytes = kChunks * 16;
    static constexpr int kPad = 256;                                  // zero-weighted moment reads past the last row
    static constexpr int kTrips = (kChunks + 255) / 256;              // DMA instructions per thread
    // Moments on the matrix cores: one v_mfma_i32_16x16x64_i8 row = (keypoint, slice s), its 64
    // K-bytes = columns 0..31 of TWO consecutive rows of the keypoint's 31-row disc box (lane chunk
    // c: row 8 ks + 2 s + (c >> 1), bytes 16 (c & 1) ..); 4 slices x 2 rows = 8 rows per step, so
    // kMom = 4 steps cover rows 0..31 (row 31 and column 31 carry zero weight).
    static constexpr int kMom = 4;
    static_assert((63 + R - 15 + 31) * kPitch + 63 + kHaloX - 15 + 32 <= kBytes + kPad, "moment reads stay inside the tile array");
};

// B operand of the tile kernel's moment MFMAs, [kMom][lane 64][16]: lane = 16 * chunk + column,
// column = 2 * slice + (0: dx weights, 1: dy weights), columns 8..15 zero.  Disc chords: orb.cu:79-80.
static std::vector<int8_t> make_tile_moment_weights()
{
    static const int u[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 7, 5, 0};
    std::vector<int8_t> w((size_t)4 * 64 * 16, 0);
    for (int ks = 0; ks < 4; ks++)
        for (int lane = 0; lane < 64; lane++) {
            const int col = lane & 15, c = lane >> 4;
            if (col >= 8) continue;
            const int sl = col >> 1, xy = col & 1;
            for (int j = 0; j < 16; j++) {
                const int row = 8 * ks + 2 * sl + (c >> 1), bx = 16 * (c & 1) + j;
                const int dy = row - 15, dx = bx - 15;
                const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
                if (ady <= 15 && adx <= 15 && adx <= u[ady]) w[((size_t)ks * 64 + lane) * 16 + j] = (int8_t)(xy ? dy : dx);
            }
        }
    return w;
}

// DL = EXT iv, descriptor_level: one workgroup = one 64x64 tile of ONE PYRAMID LEVEL (the detection tile list)
// and the keypoints that level won inside it, sampled at position >> level.  A level-l tile spans
// (64 / (cell >> l))^2 cells, up to 4096, so the keypoint list is gathered in passes of at most 64 (wave 0
// appends whole 64-cell groups while they fit); with DL = false there is exactly one group and one pass.
// threads of a tile-describe workgroup.  128 since late r3: the kernel waits for its tile's round trip to memory at the
// head of every workgroup (4.3), and with two waves per tile a CU holds 11 tiles in flight (LDS-bound) instead of 8
// (wave-slot-bound): 1.390 -> 1.326 ms per 4096 frames.  -DORBFE_DESCRIBE_THREADS=256 is the A/B build.
#ifndef ORBFE_DESCRIBE_THREADS
#define ORBFE_DESCRIBE_THREADS 128
#endif
constexpr int kDescThreads = ORBFE_DESCRIBE_THREADS;
static_assert(kDescThreads == 128 || kDescThreads == 256, "whole waves; the moment sums are zeroed by threads 0..127");

template <int R, bool SOA, bool DL>
__global__ void __launch_bounds__(kDescThreads)
describe_tile_kernel(DeviceGeom g, const uint8_t *__restrict__ pyr, const uint32_t *__restrict__ cellkey,
                     const uint16_t *__restrict__ cellslot, const uint4 *__restrict__ momw, int tiles_x, uint32_t tiles_x_magic,
                     const TileDesc *__restrict__ tiles, orbfe_keypoint *__restrict__ records, orbfe_soa soa, SteerArgs steer)
{
    using G = TileGeom<R>;
    constexpr int P = G::kPitch;
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[G::kBytes + G::kPad];
    // the pass's keypoints: x | y << 16 (level coordinates); LDS byte offset of the keypoint inside the tile, bit 31 =
    // all-zero descriptor (guard band, orb.cu:34); byte offset of its record in the frame's block.  Everything the
    // descriptor loop needs per keypoint is precomputed here by one lane per keypoint: in that loop it was ~25 scalar
    // instructions per keypoint, and the scalar unit issues one instruction per SIMD turn like the vector ALU
    __shared__ uint32_t s_kxy[64], s_kc0[64], s_kslot[64];
    __shared__ int s_mom[128];                               // their moments: m10, m01
    __shared__ float s_ang[64], s_cos[R == 15 ? 1 : 64], s_sin[R == 15 ? 1 : 64]; // angle and steering (cos, sin: only without the table)
    __shared__ uint8_t s_kiv[64];                            // interval of the orientation in the steering table (TAB; < 256)
    __shared__ float s_breaks[R == 15 ? kSteerMaxBreaks : 1]; // the table's break points: the bisection of the angle lanes reads LDS
                                                              // (from global memory its 10 dependent loads were 15 % of the kernel)
    __shared__ int s_nkp, s_cursor;
    // TAB: the rotated pattern comes from the orientation table (steer_table.cpp) instead of being computed: in the
    // degrees-as-radians regime (R = 15 <=> !angle_in_radians) the 512 rotated sample positions are a piecewise
    // constant function of the orientation with ~220 pieces, so a keypoint's 8 samples per lane are one 16-byte load
    // (none at all in the piece around 0, which holds 70 % of the orientations: its row stays in registers) + 8 adds
    // instead of 80 float instructions.  The table is generated with this very arithmetic and checked against it for
    // every float in [-pi, pi] (orbfe_selfcheck_steer_table).
    constexpr bool TAB = R == 15; // (R = 15 is launched only with !angle_in_radians, and then the context holds the table)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int f, tile;
    if (!frame_item(g, &f, &tile)) return;
    int l = 0, tx, ty;
    if (DL) {
        const TileDesc td = load_tile_desc(tiles, tile); // (scalar load, as in detect_tile_kernel)
        l = td.level;
        tx = td.tx;
        ty = td.ty;
    } else {
        ty = div_by_magic(tile, tiles_x_magic);
        tx = tile - ty * tiles_x;
    }
    const int W = g.lv[l].w, H = g.lv[l].h; // the sampled level (l = 0 unless DL)
    const int x0 = tx * kDTile, y0 = ty * kDTile;
    const int ox = x0 - G::kHaloX, oy = y0 - R; // image position of tile byte (0, 0)
    const uint8_t *img = pyr + (size_t)f * g.frame_stride + g.lv[l].offset;
    const int IP = g.lv[l].pitch;

    // ---- stage the tile: chunk q = (row q / kChunksRow, 16-byte column q % kChunksRow) lands at LDS byte 16 q.
    //      Addresses are clamped into the image (rows) and the padded pitch (columns): every load is
    //      legal, and what lies outside the image is zeroed afterwards (border tiles only).
#pragma unroll
    for (int t = 0; t < (G::kChunks + kDescThreads - 1) / kDescThreads; t++) {
        const int q = kDescThreads * t + tid;
        if (kDescThreads * (t + 1) <= G::kChunks || q < G::kChunks) {
            const int r = q / G::kChunksRow, cc = q - r * G::kChunksRow;
            int gy = oy + r, gx = ox + 16 * cc;
            gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
            gx = gx < 0 ? 0 : (gx > IP - 16 ? IP - 16 : gx);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(img + (uint32_t)(__mul24(gy, IP) + gx)),
                                             (__attribute__((address_space(3))) void *)(s_tile + 16 * (kDescThreads * t + 64 * wv)), 16, 0, 0);
        }
    }
    if (TAB)
        for (int i = tid; i < steer.n_breaks; i += kDescThreads) s_breaks[i] = steer.breaks[i];
    // cells of this tile: edge cl = cell >> l pixels of the level, n per tile edge (1 .. 64), in groups of 64
    const int cl = g.cell >> l;
    const int n = kDTile / cl, ln = ilog2(n);
    const int ngroups = DL ? (n * n + 63) >> 6 : 1;
    // weight fragments of the moment MFMAs and this lane's four rBRIEF pattern rows (loaded in the first pass)
    uint4 bw[G::kMom];
    float4 pat[4];
    uint4 central_off = make_uint4(0u, 0u, 0u, 0u);
    int cursor = 0; // next cell group (uniform)
    bool first = true;
    for (;;) {
        if (tid < 128) s_mom[tid] = 0;
        // ---- the pass's keypoints (wave 0; in the first pass while the DMAs fly): one lane per cell of a group
        if (wv == 0) {
            int cnt = 0;
            do {
                const int idx = 64 * cursor + lane;
                const int cxl = idx & (n - 1), cyl = idx >> ln;
                const int cx = tx * n + cxl, cy = ty * n + cyl;
                const bool valid = idx < n * n && cx < g.cells_x && cy < g.cells_y;
                const int k = cy * g.cells_x + cx;
                uint32_t slot = 0xFFFFu, key = 0;
                if (valid) {
                    slot = cellslot[(size_t)f * g.K + k];
                    key = cellkey[(size_t)f * g.K + k];
                }
                const bool keep = valid && slot != 0xFFFFu && (!DL || 7 - (int)((key >> 12) & 7u) == l);
                const uint64_t m = __ballot(keep);
                const int c = (int)__popcll(m);
                if (DL && cnt + c > 64) break; // this group waits for the next pass (an empty list takes any group)
                if (keep) {
                    int sc, lv, x, y;
                    nms_decode(key, cx, cy, g.cell, &sc, &lv, &x, &y);
                    const int pos = cnt + (int)__popcll(m & ((1ull << lane) - 1ull));
                    const int xl = x >> l, yl = y >> l;
                    s_kxy[pos] = (uint32_t)xl | ((uint32_t)yl << 16);
                    s_kc0[pos] = (uint32_t)((yl - oy) * P + xl - ox) | (orb_border_zero(xl, yl, W, H, g.angle_in_radians) ? 0x80000000u : 0u);
                    s_kslot[pos] = slot * (uint32_t)sizeof(orbfe_keypoint); // (the SoA view recomputes the cell index from (x, y))
                    // the record's head goes out from here, one lane per keypoint (in phase C it cost every keypoint's
                    // wave 3 converts + 5 moves + 2 stores for one active lane); records keep level-0 coordinates
                    uint32_t *rec = reinterpret_cast<uint32_t *>(records + (size_t)f * g.cap + slot);
                    rec[0] = __float_as_uint((float)x);
                    rec[1] = __float_as_uint((float)y);
                    rec[2] = __float_as_uint((float)sc);
                    rec[3] = (uint32_t)lv;
                }
                cnt += c;
                cursor++;
            } while (DL && cursor < ngroups);
            if (lane == 0) {
                s_nkp = cnt;
                s_cursor = cursor;
            }
        }
        // LDS-DMA data is ordered only behind the issuing wave's vmcnt; then the barrier publishes it
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nkp = s_nkp;
        if (DL) cursor = s_cursor;
        // uniform: no (further) keypoint in this tile.  WRITE-BACK INVARIANT of the scalar descriptor stores below: this
        // is the only exit a wave can take AFTER it has issued s_store_dwordx4 (passes 2.. of the DL form), and it is safe
        // because every pass ends with s_waitcnt lgkmcnt(0) + s_dcache_wb before it loops back here -- a wave never
        // reaches this return with an unwritten scalar store.  tests/test_gpu_round4.py::test_scalar_descriptor_stores_equal_the_vector_store_build A/Bs the path.
        if (nkp == 0) return;
        if (first) {
            // ---- border tiles: zero what the moments exclude / what lies outside the image: rows gy <= 0 and
            //      gy >= H, columns gx <= 0 and gx >= W (only the strips concerned are touched)
            if (ox <= 0 || ox + P > W || oy <= 0 || oy + G::kRows > H) { // uniform
                uint32_t *t32 = reinterpret_cast<uint32_t *>(s_tile);
                constexpr int PD = P / 4;
                const int rtop = oy <= 0 ? 1 - oy : 0;                       // rows [0, rtop) lie at gy <= 0
                const int rbot = H - oy < G::kRows ? H - oy : G::kRows;      // rows [rbot, kRows) at gy >= H
                for (int i = tid; i < rtop * PD; i += kDescThreads) t32[i] = 0u;
                for (int i = rbot * PD + tid; i < G::kRows * PD; i += kDescThreads) t32[i] = 0u;
                // column strips of the rows in between: `lanes` threads per row, one dword each per pass
                auto strip = [&](int d0, int nd, int first_keep, int last_keep) { // dwords [d0, d0 + nd); bytes outside [first_keep, last_keep) go
                    const int sh = nd <= 8 ? 3 : 5, d = d0 + (tid & ((1 << sh) - 1));
                    if (d >= d0 + nd) return;
                    uint32_t keep = 0xFFFFFFFFu;
                    const int lo = first_keep - 4 * d, hi = last_keep - 4 * d; // byte j of this dword kept iff lo <= j < hi
                    if (lo >= 4 || hi <= 0) keep = 0u;
                    else {
                        if (lo > 0) keep &= 0xFFFFFFFFu << (8 * lo);
                        if (hi < 4) keep &= (1u << (8 * hi)) - 1u;
                    }
                    for (int r = rtop + (tid >> sh); r < rbot; r += kDescThreads >> sh) t32[r * PD + d] &= keep;
                };
                if (ox <= 0) strip(0, (1 - ox + 3) / 4, 1 - ox, P);          // gx <= 0  <=> byte < 1 - ox
                if (ox + P > W) strip((W - ox) / 4, PD - (W - ox) / 4, 0, W - ox); // gx >= W <=> byte >= W - ox
                __syncthreads();
            }
#pragma unroll
            for (int ks = 0; ks < G::kMom; ks++) bw[ks] = momw[ks * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (!TAB) pat[r] = reinterpret_cast<const float4 *>(c_pattern_f)[64 * r + lane];
            if (TAB) central_off = steer.table[steer.central * 64 + lane];
            first = false;
        }

        // ---- phase A: moments of groups of kKpw keypoints on the matrix cores -> s_m10 / s_m01
        for (int grp = wv; grp * kKpw < nkp; grp += kDescThreads / 64) { // uniform
            const int k0 = grp * kKpw;
            const int nk = nkp - k0 < kKpw ? nkp - k0 : kKpw;
            // A fragment of lane (chunk c, row = keypoint + 4 * slice) = 16 bytes of disc-box row
            // 8 ks + 2 slice + (c >> 1), columns 16 (c & 1) .. + 15, at the keypoint's own byte position
            const int row = lane & 15, c = lane >> 4;
            const int j = k0 + ((row & 3) < nk ? (row & 3) : nk - 1);
            const uint32_t xy = s_kxy[j];
            const int kb = ((int)(xy >> 16) - 15 - oy) * P + (int)(xy & 0xFFFFu) - 15 - ox;
            // 16 bytes from an arbitrary byte address = 5 aligned dwords realigned by v_alignbyte: byte-misaligned
            // wide LDS reads are served one lane at a time on gfx950 (26.8 ns per wave-instruction against < 3 ns
            // aligned, tools/lds_unaligned_rate.hip)
            const int abyte = kb + (2 * (row >> 2) + (c >> 1)) * P + 16 * (c & 1);
            const uint32_t *a32 = reinterpret_cast<const uint32_t *>(s_tile + (abyte & ~3));
            const uint32_t ash = (uint32_t)abyte & 3u;
            u32x4 av[G::kMom];
#pragma unroll
            for (int ks = 0; ks < G::kMom; ks++) {
                const uint32_t *q = a32 + ks * 2 * P; // 8 rows further
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
                av[ks].x = __builtin_amdgcn_alignbyte(d1, d0, ash);
                av[ks].y = __builtin_amdgcn_alignbyte(d2, d1, ash);
                av[ks].z = __builtin_amdgcn_alignbyte(d3, d2, ash);
                av[ks].w = __builtin_amdgcn_alignbyte(d4, d3, ash);
            }
            v4i acc = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < G::kMom; ks++) {
                const uint4 bv = bw[ks];
                const v4i a = {(int)(av[ks].x ^ 0x80808080u), (int)(av[ks].y ^ 0x80808080u), (int)(av[ks].z ^ 0x80808080u),
                               (int)(av[ks].w ^ 0x80808080u)};
                const v4i b = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
            }
            // D[row][col] sits in lane 16 * (row / 4) + col, register row % 4: keypoint `it`, slice s, weights xy
            // -> register `it` of lane 18 s + xy.  The four slices meet in LDS: lanes 18 s + xy add register `it`
            // into s_mom[2 (k0 + it) + xy] (4 adders per word, LDS atomics: integer, order-free)
            const int sl = lane >> 4, cxy = (lane & 15) - 2 * sl; // this lane's slice and column offset inside it
            if (cxy == 0 || cxy == 1) {
#pragma unroll
                for (int it = 0; it < kKpw; it++)
                    if (it < nk) atomicAdd(&s_mom[2 * (k0 + it) + cxy], acc[it]);
            }
        }
        __syncthreads();
        // ---- phase B: one lane per keypoint: atan2f and the steering cos / sin, ONCE per pass
        if (wv == 0 && lane < nkp) {
            const float ang = orbfe_atan2f((float)s_mom[2 * lane + 1], (float)s_mom[2 * lane]);
            s_ang[lane] = ang;
            if (!TAB) { // (with the table the steering cos / sin are never formed: ~60 instructions of this serial phase)
                float ca, sb;
                orb_steer(ang, g.angle_in_radians, &ca, &sb);
                s_cos[lane] = ca;
                s_sin[lane] = sb;
            }
            if (TAB) { // interval = number of break points <= the orientation: branch-free bisection
                int pos = 0;
#pragma unroll
                for (int step = 128; step > 0; step >>= 1) { // (kSteerMaxBreaks <= 255)
                    const int t = pos + step;
                    if (t <= steer.n_breaks && s_breaks[t - 1] <= ang) pos = t;
                }
                s_kiv[lane] = (uint8_t)pos;
            }
            *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(records + (size_t)f * g.cap) + s_kslot[lane] + 16) = __float_as_uint(ang);
        }
        __syncthreads();
        // ---- phase C: descriptors, keypoints dealt round-robin to the waves.  A keypoint's 256 bits are four
        //      wave-uniform ballots, i.e. SGPR pairs: they go to the record by SCALAR stores (s_store_dwordx4: gfx950
        //      still has them; tools/sstore_probe.hip checks them against vector stores into the same cache lines),
        //      so the loop carries no vector instruction for the output at all
        // (uniform, but the compiler does not see it through xcd_remap's division: pin it to SGPRs)
        const uint64_t frv = reinterpret_cast<uint64_t>(records + (size_t)f * g.cap);
        const char *frame_rec = reinterpret_cast<const char *>(
            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)frv) |
            ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(frv >> 32)) << 32));
        // lane i holds keypoint i's four loop inputs; a keypoint's turn reads them with v_readlane (no LDS round
        // trip in the loop).  The empty asm makes the pattern rows' loads complete HERE: the compiler otherwise
        // waits for them inside the loop, four s_waitcnt per keypoint.
        const uint32_t my_c0 = s_kc0[lane], my_roff = s_kslot[lane];
        const uint32_t my_cos = TAB ? 0u : __float_as_uint(s_cos[lane]), my_sin = TAB ? 0u : __float_as_uint(s_sin[lane]);
        const int my_iv = TAB ? (int)s_kiv[lane] : 0;
        if (!TAB) {
#pragma unroll
            for (int r = 0; r < 4; r++) asm volatile("" ::"v"(pat[r].x), "v"(pat[r].y), "v"(pat[r].z), "v"(pat[r].w));
        } else {
            asm volatile("" ::"v"(central_off.x), "v"(central_off.y), "v"(central_off.z), "v"(central_off.w));
        }
        for (int j = wv; j < nkp; j += kDescThreads / 64) { // uniform
            const uint32_t c0f = (uint32_t)__builtin_amdgcn_readlane((int)my_c0, j);
            const uint32_t roff = (uint32_t)__builtin_amdgcn_readlane((int)my_roff, j);
            const float a = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)my_cos, j));
            const float b = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)my_sin, j));
            const char *rec = frame_rec + roff;
            uint64_t d[4] = {0, 0, 0, 0};
            if (!(c0f >> 31)) {
                if (TAB) {
                    const int iv = __builtin_amdgcn_readlane(my_iv, j);
                    uint4 off = central_off;
                    if (iv != steer.central) off = steer.table[iv * 64 + lane]; // uniform branch
                    const uint32_t w[4] = {off.x, off.y, off.z, off.w};
#pragma unroll
                    for (int r = 0; r < 4; r++) { // (P, Q) of round r as two signed 16-bit LDS offsets
                        const int t0 = s_tile[(int)c0f + (int)(int16_t)(w[r] & 0xFFFFu)];
                        const int t1 = s_tile[(int)c0f + ((int)w[r] >> 16)];
                        d[r] = __ballot(t0 < t1);
                    }
                    // round r holds bit t of word p_t[r] (the bank-aware schedule of steer_table.cpp): four masked
                    // swaps of whole ballots put the words back -- wave-uniform operands, scalar unit only
                    uint64_t t;
                    t = (d[1] ^ d[3]) & steer.sched_mask[3]; d[1] ^= t; d[3] ^= t;
                    t = (d[0] ^ d[2]) & steer.sched_mask[2]; d[0] ^= t; d[2] ^= t;
                    t = (d[2] ^ d[3]) & steer.sched_mask[1]; d[2] ^= t; d[3] ^= t;
                    t = (d[0] ^ d[1]) & steer.sched_mask[0]; d[0] ^= t; d[1] ^= t;
                } else {
                    orb_describe_lds<P>(s_tile, (int)c0f, a, b, pat, d);
                }
            }
            const u32x4 lo = {(uint32_t)d[0], (uint32_t)(d[0] >> 32), (uint32_t)d[1], (uint32_t)(d[1] >> 32)};
            const u32x4 hi = {(uint32_t)d[2], (uint32_t)(d[2] >> 32), (uint32_t)d[3], (uint32_t)(d[3] >> 32)};
#ifdef ORBFE_DESC_VECTOR_STORE // (A/B build: the stores as vector instructions of lane 0)
            if (lane == 0) {
                uint32_t *rw = reinterpret_cast<uint32_t *>(const_cast<char *>(rec));
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    rw[5 + k] = lo[k];
                    rw[9 + k] = hi[k];
                }
            }
#else
            asm volatile("s_store_dwordx4 %0, %2, 0x14\n\ts_store_dwordx4 %1, %2, 0x24" ::"s"(lo), "s"(hi), "s"(rec) : "memory");
#endif
            if (SOA && (soa.d_angle || soa.d_desc32 || soa.d_desc)) { // (SOA: its own instantiation, as in select_kernel)
                if (lane == 0) {
                    const float angle = s_ang[j];
                    const uint32_t xy = s_kxy[j];
                    const int X = (int)(xy & 0xFFFFu) << l, Y = (int)(xy >> 16) << l;
                    const int cell = (Y / g.cell) * g.cells_x + X / g.cell;
                    const size_t o = (size_t)f * g.K + cell;
                    if (soa.d_angle) soa.d_angle[o] = angle;
                    if (soa.d_desc32) soa.d_desc32[o] = orb_compress(d);
                    if (soa.d_desc) {
                        uint32_t *sd = reinterpret_cast<uint32_t *>(soa.d_desc + 32 * o);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            sd[2 * k] = (uint32_t)d[k];
                            sd[2 * k + 1] = (uint32_t)(d[k] >> 32);
                        }
                    }
                }
            }
        }
#if !defined(ORBFE_DESC_VECTOR_STORE)
        // scalar stores sit in the scalar data cache until written back; the write-back covers only stores that
        // have reached the cache, hence the wait
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
#endif
        if (!DL || cursor >= ngroups) break;
        __syncthreads(); // every wave is done with this pass's lists before the next gather overwrites them
    }
}

// ------------------------------------------------------------------------------------
// a11 over a batch: records of frame p (prev) against frame p + 1 (curr).
// ------------------------------------------------------------------------------------
__device__ inline uint32_t compress_words(const uint32_t w[8])
{
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int k = 0; k < 4; k++) out |= (uint32_t)(((w[j] >> (8 * k)) & 255u) == 1u) << (4 * j + k);
    return out;
}

__global__ void __launch_bounds__(256)
match_batch_ref_kernel(const orbfe_keypoint *__restrict__ records, const int32_t *__restrict__ counts,
                       int cap, int first, int stride, float win, int max_ham, int32_t *__restrict__ out_idx,
                       int32_t *__restrict__ out_dist)
{
    __shared__ float s_x[32], s_y[32];
    __shared__ uint32_t s_d[32];
    const int pk = blockIdx.y, p = first + pk * stride; // pair ordinal, prev frame
    const int nA = clamp_count(counts[p], cap), nB = clamp_count(counts[p + 1], cap);
    const uint32_t *A = reinterpret_cast<const uint32_t *>(records + (size_t)p * cap);
    const uint32_t *B = reinterpret_cast<const uint32_t *>(records + (size_t)(p + 1) * cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int tid = i & 31;
    const bool live = i < nA;
    float px = 0.f, py = 0.f;
    uint32_t d = 0;
    if (live) {
        const uint32_t *r = A + 13 * (size_t)i;
        px = __uint_as_float(r[0]);
        py = __uint_as_float(r[1]);
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = r[5 + j];
        d = compress_words(w);
    }
    int best = 9999999, pair = -1;
    for (int base = 0; base < nB; base += 32) {
        __syncthreads();
        if (threadIdx.x < 32 && base + (int)threadIdx.x < nB) {
            const uint32_t *r = B + 13 * (size_t)(base + threadIdx.x);
            s_x[threadIdx.x] = __uint_as_float(r[0]);
            s_y[threadIdx.x] = __uint_as_float(r[1]);
            uint32_t w[8];
#pragma unroll
            for (int j = 0; j < 8; j++) w[j] = r[5 + j];
            s_d[threadIdx.x] = compress_words(w);
        }
        __syncthreads();
        const int m = (base + 32 >= nB) ? nB - base : 32;
        if (live && tid < m) {
            int j = tid;
            for (int s = 0; s < m; s++) {
                if (fabsf(px - s_x[j]) <= win && fabsf(py - s_y[j]) <= win) {
                    const int hd = __popc(d ^ s_d[j]);
                    if (hd < max_ham && hd < best) {
                        best = hd;
                        pair = base + j;
                    }
                }
                j = (j + 1 == m) ? 0 : j + 1;
            }
        }
    }
    if (i < cap) {
        out_idx[(size_t)pk * cap + i] = live ? pair : -1;
        if (out_dist) out_dist[(size_t)pk * cap + i] = (live && pair >= 0) ? best : -1;
    }
}

// 256-bit brute force, two kernels.
// (1) match_gather_kernel repacks the 52-byte records into dense, 32-byte aligned descriptor
//     rows and float2 positions in context scratch (one coalesced pass, ~90 B per keypoint).
// (2) match_batch_256_kernel: thread = one query of frame p (8 dwords in VGPRs).  The
//     candidates of frame p + 1 are wave-uniform, so they are read with scalar loads
//     (s_load_dwordx8 through the scalar cache) and enter v_xor as SGPR operands: no LDS,
//     no per-lane loads in the loop.  Popcounts chain through v_bcnt_u32_b32's accumulate
//     operand.  The running minimum is a v_min_u32 on the packed key (dist << 16 | j): its
//     minimum is the lexicographic (dist, j) one.
struct alignas(32) Desc8 {
    uint32_t w[8];
};
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256)
match_gather_kernel(const orbfe_keypoint *__restrict__ records, const int32_t *__restrict__ counts, int cap,
                    Desc8 *__restrict__ mdesc, float2 *__restrict__ mpos)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    const size_t o = (size_t)f * cap + i;
    Desc8 d = {{0, 0, 0, 0, 0, 0, 0, 0}};
    float2 p = make_float2(0.f, 0.f);
    if (i < clamp_count(counts[f], cap)) {
        const uint32_t *r = reinterpret_cast<const uint32_t *>(records + o);
        p = make_float2(__uint_as_float(r[0]), __uint_as_float(r[1]));
#pragma unroll
        for (int k = 0; k < 8; k++) d.w[k] = r[5 + k];
    }
    mdesc[o] = d;
    mpos[o] = p;
}

template <bool WINDOW>
__global__ void __launch_bounds__(256)
match_batch_256_kernel(const Desc8 *__restrict__ mdesc, const float2 *__restrict__ mpos,
                       const int32_t *__restrict__ counts, int cap, int first, int stride, int window, int max_dist,
                       int32_t *__restrict__ out_idx, int32_t *__restrict__ out_dist)
{
    int pk, blk;
    xcd_remap(gridDim.x, gridDim.y, &pk, &blk); // all query blocks of a pair share one L2
    const int p = first + pk * stride;
    const int nA = clamp_count(counts[p], cap), nB = clamp_count(counts[p + 1], cap);
    const int i = blk * 256 + threadIdx.x;
    const Desc8 *__restrict__ Bd = mdesc + (size_t)(p + 1) * cap;
    const float2 *__restrict__ Bp = mpos + (size_t)(p + 1) * cap;
    const bool live = i < nA;
    Desc8 a = {{0, 0, 0, 0, 0, 0, 0, 0}};
    float2 pa = make_float2(0.f, 0.f);
    if (live) {
        a = mdesc[(size_t)p * cap + i];
        if (WINDOW) pa = mpos[(size_t)p * cap + i];
    }
    const float win = (float)window;
    uint32_t best = 0xFFFFFFFFu;
    // distance key of one candidate whose descriptor is in SGPRs
    auto key_of = [&](const u32x8 &b, int j) {
        uint32_t dist, key;
        {
            const uint32_t x = a.w[0] ^ b[0];
            asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(dist) : "v"(x)); // first word: accumulate into literal 0
        }
#pragma unroll
        for (int k = 1; k < 8; k++) {
            const uint32_t x = a.w[k] ^ b[k];
            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(dist) : "v"(x));
        }
        asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(key) : "v"(dist), "s"(j)); // dist << 16 | j, j wave-uniform
        if (WINDOW) {
            const float2 pb = Bp[j];
            if (fabsf(pa.x - pb.x) > win || fabsf(pa.y - pb.y) > win) key = 0xFFFFFFFFu;
        }
        return key;
    };
    auto umin = [](uint32_t x, uint32_t y) { return x < y ? x : y; };
    auto group_min = [&](const u32x8 (&g)[4], int j) {
        const uint32_t k0 = key_of(g[0], j), k1 = key_of(g[1], j + 1), k2 = key_of(g[2], j + 2), k3 = key_of(g[3], j + 3);
        best = umin(umin(best, k0), k1); // -> v_min3_u32
        best = umin(umin(best, k2), k3);
    };
    // Hand-made scalar-load pipeline.  SMEM returns out of order, so the only usable wait is
    // lgkmcnt(0); hipcc places a group's s_loads right before its wait, which exposes the full
    // scalar-cache latency once per group.  Here the loads of the NEXT group are issued, the
    // CURRENT group (~100 VALU instructions) is computed, and only then comes the wait.  The
    // asm operands pin that order: `issue` is ordered before the compute through a.w[0], the
    // wait after it through `best`, and the loaded tuples only become usable through the wait.
    // No load is in flight across the loop back edge.
    auto issue = [&](u32x8 (&g)[4], const Desc8 *src) {
        asm volatile("s_load_dwordx8 %0, %5, 0x0\n\ts_load_dwordx8 %1, %5, 0x20\n\t"
                     "s_load_dwordx8 %2, %5, 0x40\n\ts_load_dwordx8 %3, %5, 0x60"
                     : "=&s"(g[0]), "=&s"(g[1]), "=&s"(g[2]), "=&s"(g[3]), "+v"(a.w[0])
                     : "s"(src)
                     : "memory");
    };
    auto wait = [&](u32x8 (&g)[4]) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(g[0]), "+s"(g[1]), "+s"(g[2]), "+s"(g[3]), "+v"(best), "+v"(a.w[0]));
    };
    // whole waves beyond nA have nothing to do (the loop has no barrier)
    const bool wave_live = __builtin_amdgcn_readfirstlane((int)(__ballot(live) != 0ull)) != 0;
    if (wave_live) {
        int j = 0;
        if (nB >= 4) {
            u32x8 g0[4], g1[4];
            issue(g0, Bd);
            wait(g0);
            for (; j + 8 <= nB; j += 8) { // invariant: g0 = candidates j .. j+3, landed
                issue(g1, Bd + j + 4);
                group_min(g0, j);
                wait(g1);
                if (j + 12 <= nB) issue(g0, Bd + j + 8);
                group_min(g1, j + 4);
                wait(g0);
            }
            if (j + 4 <= nB) {
                group_min(g0, j);
                j += 4;
            }
        }
        for (; j < nB; j++) {
            const Desc8 b = Bd[j];
            u32x8 bv;
#pragma unroll
            for (int k = 0; k < 8; k++) bv[k] = b.w[k];
            const uint32_t k0 = key_of(bv, j);
            best = k0 < best ? k0 : best;
        }
    }
    if (i < cap) {
        const int bd = (int)(best >> 16), bj = (int)(best & 0xFFFFu);
        const bool ok = live && best != 0xFFFFFFFFu && bd <= max_dist;
        out_idx[(size_t)pk * cap + i] = ok ? bj : -1;
        if (out_dist) out_dist[(size_t)pk * cap + i] = ok ? bd : -1;
    }
}

// groups: 0 = by launch size, 1 = one tile per workgroup, 2 = tile groups (orbfe_ctx::detect_groups, ORBFE_DETECT_GROUPS)
static bool detect_uses_groups(int n_items, int n_frames, int groups)
{
    if (groups) return groups == 2;
    return (long long)n_items * n_frames >= 4ll * kDetectTilesPerWg * 12288;
}
static void launch_detect_tiles(const DeviceGeom &g, int n_items, hipStream_t stream, const uint8_t *pyr, const TileDesc *tiles,
                                uint32_t *cellkey, int tile_first, int tile_step, int groups)
{
    const StageTiles st{};
    // tile groups (MULTI) when the launch is at least ~32 rounds of workgroups even then (1536 resident workgroups: 256 CUs x
    // 6): C2 in steps of 4096 frames gains 2 % (640 k tiles), 256 frames (40 k tiles) LOSE 10 % to the coarser tail
    const bool multi = detect_uses_groups(n_items, g.n_frames, groups);
    const unsigned wgs = (unsigned)(multi ? (n_items + kDetectTilesPerWg - 1) / kDetectTilesPerWg : n_items); // per frame
    const dim3 grid = g.grid8 ? dim3(8u * wgs, (unsigned)(g.n_frames + 7) / 8u) : dim3(wgs, (unsigned)g.n_frames); // = frame_grid()
#define ORBFE_DETECT_LAUNCH(ARC)                                                                                                   \
    do {                                                                                                                           \
        if (multi)                                                                                                                 \
            hipLaunchKernelGGL((detect_tile_kernel<false, ARC, true>), grid, dim3(256), 0, stream, g, pyr, tiles, cellkey,          \
                               tile_first, tile_step, n_items, st);                                                                \
        else                                                                                                                       \
            hipLaunchKernelGGL((detect_tile_kernel<false, ARC, false>), grid, dim3(256), 0, stream, g, pyr, tiles, cellkey,         \
                               tile_first, tile_step, n_items, st);                                                                \
    } while (0)
    switch (g.arc) { // validated to 9..12 where the geometry is built
    case 9: ORBFE_DETECT_LAUNCH(9); break;
    case 10: ORBFE_DETECT_LAUNCH(10); break;
    case 11: ORBFE_DETECT_LAUNCH(11); break;
    default: ORBFE_DETECT_LAUNCH(12); break;
    }
#undef ORBFE_DETECT_LAUNCH
}

// Stage API: cell keys (left in the caller's d_score buffer, 4 bytes per cell) -> the reference's feature
// grid: score as float in place, position, level (nms.cu:246-252; empty cells: 0, (0,0), 0 -- Q5)
__global__ void stage_decode_kernel(int K, int cells_x, int cell0, float *__restrict__ score_and_key,
                                    float *__restrict__ d_pos, int *__restrict__ d_level)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const uint32_t key = __float_as_uint(score_and_key[k]);
    int s, l, x, y;
    nms_decode(key, k % cells_x, k / cells_x, cell0, &s, &l, &x, &y);
    score_and_key[k] = (float)s;
    d_pos[2 * k] = (float)x;
    d_pos[2 * k + 1] = (float)y;
    d_level[k] = l;
}

int launch_detect_stage(const orbfe_pyramid_level *lv, int n_levels, int threshold, const uint8_t *d_lut, float *d_pos,
                        float *d_score, int *d_level, hipStream_t stream)
{
    DeviceGeom g;
    memset(&g, 0, sizeof(g));
    g.W = (int)lv[0].image_width;
    g.H = (int)lv[0].image_height;
    g.L = g.Ld = n_levels;
    g.cell = 32; // CELL_SIZE_WIDTH / HEIGHT, src/SlamGpuPipeline/defines.h:17-18
    g.cells_x = (g.W + 31) / 32;
    g.cells_y = (g.H + 31) / 32;
    g.K = g.cells_x * g.cells_y;
    g.cap = g.K;
    g.threshold = threshold;
    g.arc = 0; // unknown and not needed: the table decides
    StageTiles st;
    memset(&st, 0, sizeof(st));
    st.lut = d_lut;
    int n_tiles = 0;
    for (int l = 0; l < n_levels; l++) {
        g.lv[l].w = (int)lv[l].image_width;
        g.lv[l].h = (int)lv[l].image_height;
        g.lv[l].pitch = (int)lv[l].image_pitch;
        g.lv[l].offset = (size_t)(reinterpret_cast<uintptr_t>(lv[l].image) - reinterpret_cast<uintptr_t>(lv[0].image)); // modulo 2^64
        st.first[l] = n_tiles;
        st.tiles_x[l] = (g.lv[l].w + kTileW - 1) / kTileW;
        st.resp[l] = lv[l].response;
        st.resp_pitch[l] = (int)(lv[l].response_pitch / sizeof(float));
        if (g.lv[l].w >= 7 && g.lv[l].h >= 7) // else no pixel is >= 3 from every border: the map is all zero
            n_tiles += st.tiles_x[l] * ((g.lv[l].h + kTileH - 1) / kTileH);
        else if (lv[l].response && g.lv[l].w > 0 && g.lv[l].h > 0 &&
                 hipMemset2DAsync(lv[l].response, lv[l].response_pitch, 0, (size_t)g.lv[l].w * sizeof(float), g.lv[l].h,
                                  stream) != hipSuccess)
            return ORBFE_ERR_HIP;
    }
    for (int l = n_levels; l <= 8; l++) st.first[l] = n_tiles;
    uint32_t *keys = reinterpret_cast<uint32_t *>(d_score); // 4 bytes per cell: the key, then (decode) the score
    if (hipMemsetAsync(keys, 0, (size_t)g.K * sizeof(uint32_t), stream) != hipSuccess) return ORBFE_ERR_HIP;
    if (n_tiles > 0)
        hipLaunchKernelGGL((detect_tile_kernel<true, 0, false>), dim3(n_tiles, 1), dim3(256), 0, stream, g, lv[0].image,
                           (const TileDesc *)nullptr, keys, 0, 1, n_tiles, st);
    hipLaunchKernelGGL(stage_decode_kernel, dim3((g.K + 255) / 256), dim3(256), 0, stream, g.K, g.cells_x, g.cell, d_score,
                       d_pos, d_level);
    return hipGetLastError() == hipSuccess ? ORBFE_OK : ORBFE_ERR_HIP;
}

// ------------------------------------------------------------------------------------
// 256-bit matching inside a position window: an index instead of brute force.  A query only has to look
// at the candidates whose position lies within +-window pixels, i.e. in the few grid cells around it, so
// the curr frame's records are bucketed by cell (counting sort, one workgroup per frame) and a query walks
// one contiguous range of the sorted list per cell row of its window: ~10 candidates instead of all 2000
// at the bench's C3 shape (stereo pairs, +-16 px).  The result is the same lexicographic minimum
// (distance, index) over the candidates inside the window as the brute-force form (oracle_match256).
// ------------------------------------------------------------------------------------
__device__ inline int bucket_coord(float v, float inv_cell, int n)
{
    // floor(v / cell) clamped to [0, n - 1]; monotone in v, so |a - b| <= w implies
    // bucket(a - w) <= bucket(b) <= bucket(a + w).  fmaxf / fminf also absorb a NaN.
    return (int)fminf(fmaxf(floorf(v * inv_cell), 0.0f), (float)(n - 1));
}

// bend[b] = END of bucket b in `sorted` (= start of bucket b + 1); one 1024-thread workgroup per curr frame.
// Thread t owns the `per` consecutive buckets [t * per, (t + 1) * per): one block-wide scan of the per-thread
// sums instead of a pass per 256 buckets (26 -> 10 us per 128 frames of 2000 records at K = 6360).
constexpr int kBucketThreads = 1024;
// LDS = the K bucket counters live in LDS (K * 4 bytes of dynamic shared memory): counting, prefix and the
// scatter cursor never leave the CU, and the bucket ends reach global memory as one coalesced pass.  With
// the counters in global memory the prefix alone was 2 x ceil(K / 1024) dependent L2 round trips per
// thread (17 us per 848x480 frame with 8-px cells, 5 us now).  Grids too large for LDS keep the global form.
template <bool LDS>
__global__ void __launch_bounds__(kBucketThreads)
match_bucket_kernel(const orbfe_keypoint *__restrict__ records, const int32_t *__restrict__ counts, int cap, int first,
                    int stride, int K, int cells_x, int cells_y, float inv_cell, int32_t *__restrict__ bend_all,
                    uint16_t *__restrict__ sorted_all, uint32_t *__restrict__ d32_all)
{
    extern __shared__ int s_bucket[];
    __shared__ int s_wave[kBucketThreads / 64];
    const int f = first + blockIdx.x * stride + 1; // the pair's curr frame
    const int n = clamp_count(counts[f], cap);
    const uint32_t *R = reinterpret_cast<const uint32_t *>(records + (size_t)f * cap);
    int32_t *bend = bend_all + (size_t)f * K;
    int *cnt = LDS ? s_bucket : bend;
    uint16_t *sorted = sorted_all + (size_t)f * cap;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // counters written by atomics are read past this CU's L1 when they live in global memory
    auto peek = [&](int k) { return LDS ? cnt[k] : __hip_atomic_load(&cnt[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (int k = tid; k < K; k += kBucketThreads) cnt[k] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kBucketThreads) {
        const int b = bucket_coord(__uint_as_float(R[13 * (size_t)i + 1]), inv_cell, cells_y) * cells_x +
                      bucket_coord(__uint_as_float(R[13 * (size_t)i]), inv_cell, cells_x);
        atomicAdd(&cnt[b], 1);
        if (d32_all) { // reference-mode matching: the 32-bit "compressed" descriptor (orb.cu:145-169) of every record
            uint32_t w[8];
#pragma unroll
            for (int j = 0; j < 8; j++) w[j] = R[13 * (size_t)i + 5 + j];
            d32_all[(size_t)f * cap + i] = compress_words(w);
        }
    }
    __syncthreads();
    // exclusive prefix sum over the K counts
    const int per = (K + kBucketThreads - 1) / kBucketThreads;
    const int k0 = tid * per, k1 = k0 + per < K ? k0 + per : K;
    int mine = 0;
    for (int k = k0; k < k1; k++) mine += peek(k);
    const int incl = wave_incl_scan_i32(mine);
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int run = incl - mine; // start of this thread's first bucket
    for (int u = 0; u < wv; u++) run += s_wave[u];
    for (int k = k0; k < k1; k++) {
        const int c = peek(k);
        cnt[k] = run; // start of bucket k
        run += c;
    }
    __syncthreads();
    // scatter: the atomic cursor turns every start into the bucket's end
    for (int i = tid; i < n; i += kBucketThreads) {
        const int b = bucket_coord(__uint_as_float(R[13 * (size_t)i + 1]), inv_cell, cells_y) * cells_x +
                      bucket_coord(__uint_as_float(R[13 * (size_t)i]), inv_cell, cells_x);
        sorted[atomicAdd(&cnt[b], 1)] = (uint16_t)i;
    }
    if (LDS) {
        __syncthreads();
        for (int k = tid; k < K; k += kBucketThreads) bend[k] = cnt[k];
    }
}

static void launch_match_bucket(const orbfe_keypoint *d_records, const int32_t *d_counts, int n_pairs, int cap, int first,
                                int stride, const DeviceGeom &g, float inv_cell, int32_t *bend, uint16_t *sorted,
                                uint32_t *d32, hipStream_t stream)
{
    // dynamic K * 4 bytes + the kernel's static s_wave must fit the 64 KiB a workgroup may ask for by default
    if ((size_t)g.K * sizeof(int) + (kBucketThreads / 64) * sizeof(int) <= 64 * 1024)
        hipLaunchKernelGGL(match_bucket_kernel<true>, dim3(n_pairs), dim3(kBucketThreads), (size_t)g.K * sizeof(int), stream,
                           d_records, d_counts, cap, first, stride, g.K, g.cells_x, g.cells_y, inv_cell, bend, sorted, d32);
    else
        hipLaunchKernelGGL(match_bucket_kernel<false>, dim3(n_pairs), dim3(kBucketThreads), 0, stream, d_records, d_counts,
                           cap, first, stride, g.K, g.cells_x, g.cells_y, inv_cell, bend, sorted, d32);
}

// The walk gathers: every thread reads its own ~10 candidate records, 64 different cache lines per load
// instruction, and neighbouring queries (records are in cell order) read nearly the same candidates one
// after the other -- the texture path's address rate, not latency or bytes, bounded it (22 us per 128
// pairs at C3 whether the loads were issued one by one or 24 at a time).  So a workgroup of 256
// consecutive queries first finds the cell rows its windows span, stages that contiguous piece of the
// sorted list -- bucket ends, indices and the 40 bytes of position + descriptor per candidate -- in LDS
// once (~400 records for 256 queries at C3 instead of 256 x 8 gathers), and the walk then reads LDS.
// A piece that does not fit (arbitrary records can put a whole frame into a few rows) takes the
// row-by-row kernel's path through global memory.
constexpr int kWinStage = 512;  // candidate records a workgroup can stage (20 KB): 2 per thread
constexpr int kWinEnds = 2048;  // bucket ends it can stage (8 KB): 8 per thread
__global__ void __launch_bounds__(256)
match_window_kernel(const orbfe_keypoint *__restrict__ records, const int32_t *__restrict__ counts, int cap, int first,
                    int stride, int K, int cells_x, int cells_y, float inv_cell, const int32_t *__restrict__ bend_all,
                    const uint16_t *__restrict__ sorted_all, int window, int max_dist, int32_t *__restrict__ out_idx,
                    int32_t *__restrict__ out_dist)
{
    __shared__ uint32_t s_rec[kWinStage * 10];
    __shared__ int s_end[kWinEnds];
    __shared__ uint16_t s_j[kWinStage];
    __shared__ int s_rmin, s_rmax;
    int pk, blk;
    xcd_remap(gridDim.x, gridDim.y, &pk, &blk); // all query blocks of a pair share one L2
    const int p = first + pk * stride;
    const int nA = clamp_count(counts[p], cap), nB = clamp_count(counts[p + 1], cap);
    const int i = blk * 256 + threadIdx.x;
    const bool live = i < nA && nB > 0;
    const uint32_t *B = reinterpret_cast<const uint32_t *>(records + (size_t)(p + 1) * cap);
    const int32_t *bend = bend_all + (size_t)(p + 1) * K;
    const uint16_t *sorted = sorted_all + (size_t)(p + 1) * cap;
    if (threadIdx.x == 0) s_rmin = cells_y, s_rmax = -1;
    __syncthreads();
    float ax = 0.0f, ay = 0.0f;
    const float win = (float)window;
    uint32_t a[8];
    int bx0 = 0, bx1 = 0, by0 = 0, by1 = -1;
    if (live) {
        const uint32_t *A = reinterpret_cast<const uint32_t *>(records + (size_t)p * cap) + 13 * (size_t)i;
        ax = __uint_as_float(A[0]), ay = __uint_as_float(A[1]);
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = A[5 + k];
        bx0 = bucket_coord(ax - win, inv_cell, cells_x), bx1 = bucket_coord(ax + win, inv_cell, cells_x);
        by0 = bucket_coord(ay - win, inv_cell, cells_y), by1 = bucket_coord(ay + win, inv_cell, cells_y);
    }
    { // rows spanned by the workgroup's windows: wave minimum / maximum first, one LDS atomic per wave
        int lo = live ? by0 : cells_y, hi = live ? by1 : -1;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo = min(lo, __shfl_xor(lo, d));
            hi = max(hi, __shfl_xor(hi, d));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&s_rmin, lo);
            atomicMax(&s_rmax, hi);
        }
    }
    __syncthreads();
    const int rmin = s_rmin, rmax = s_rmax; // block-uniform from here on
    uint32_t best = 0xFFFFFFFFu;
    if (rmax >= rmin) {
        const int c_lo = rmin * cells_x, c_hi = (rmax + 1) * cells_x; // cells [c_lo, c_hi)
        const int T0 = c_lo > 0 ? bend[c_lo - 1] : 0, T1 = bend[c_hi - 1];
        const int n = T1 - T0, n_ends = c_hi - c_lo + 1; // ends of cells c_lo - 1 .. c_hi - 1
        if (n <= 0) {
            // no candidate in any of these rows
        } else if (n <= kWinStage && n_ends <= kWinEnds) {
            // all requests first (clamped addresses), then the LDS stores: one round trip for the ends and the
            // indices together, one more for the records
            int ev[kWinEnds / 256];
            uint32_t jv[kWinStage / 256];
#pragma unroll
            for (int u = 0; u < kWinEnds / 256; u++) {
                const int c = c_lo - 1 + (int)threadIdx.x + 256 * u; // cell whose end this is
                ev[u] = bend[c < 0 ? 0 : (c < K ? c : K - 1)];
            }
#pragma unroll
            for (int u = 0; u < kWinStage / 256; u++) {
                const int t = (int)threadIdx.x + 256 * u;
                jv[u] = sorted[T0 + (t < n ? t : 0)];
            }
            uint32_t rv[kWinStage / 256][10];
#pragma unroll
            for (int u = 0; u < kWinStage / 256; u++) {
                const uint32_t *r = B + 13 * (size_t)jv[u];
                rv[u][0] = r[0];
                rv[u][1] = r[1];
#pragma unroll
                for (int k = 0; k < 8; k++) rv[u][2 + k] = r[5 + k];
            }
#pragma unroll
            for (int u = 0; u < kWinEnds / 256; u++) {
                const int e = (int)threadIdx.x + 256 * u;
                if (e < n_ends) s_end[e] = c_lo - 1 + e >= 0 ? ev[u] : 0;
            }
#pragma unroll
            for (int u = 0; u < kWinStage / 256; u++) {
                const int t = (int)threadIdx.x + 256 * u;
                if (t < n) {
                    s_j[t] = (uint16_t)jv[u];
#pragma unroll
                    for (int k = 0; k < 10; k++) s_rec[10 * t + k] = rv[u][k];
                }
            }
            __syncthreads();
            if (live) {
                for (int by = by0; by <= by1; by++) {
                    const int g0 = by * cells_x + bx0, g1 = by * cells_x + bx1;
                    int t = s_end[g0 - c_lo] - T0; // end of cell g0 - 1
                    const int e = s_end[g1 - c_lo + 1] - T0;
                    for (; t < e; t++) {
                        const uint32_t *r = s_rec + 10 * t;
                        if (fabsf(ax - __uint_as_float(r[0])) > win || fabsf(ay - __uint_as_float(r[1])) > win) continue;
                        uint32_t dist = 0;
#pragma unroll
                        for (int k = 0; k < 8; k++) dist += __popc(a[k] ^ r[2 + k]);
                        const uint32_t key = (dist << 16) | s_j[t];
                        best = key < best ? key : best;
                    }
                }
            }
        } else if (live) {
            for (int by = by0; by <= by1; by++) {
                const int g0 = by * cells_x + bx0, g1 = by * cells_x + bx1;
                int t = g0 > 0 ? bend[g0 - 1] : 0;
                const int e = bend[g1];
                for (; t < e; t++) {
                    const uint32_t j = sorted[t];
                    const uint32_t *r = B + 13 * (size_t)j;
                    if (fabsf(ax - __uint_as_float(r[0])) > win || fabsf(ay - __uint_as_float(r[1])) > win) continue;
                    uint32_t dist = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) dist += __popc(a[k] ^ r[5 + k]);
                    const uint32_t key = (dist << 16) | j;
                    best = key < best ? key : best;
                }
            }
        }
    }
    if (i >= cap) return;
    const int bd = (int)(best >> 16), bj = (int)(best & 0xFFFFu);
    const bool ok = best != 0xFFFFFFFFu && bd <= max_dist;
    out_idx[(size_t)pk * cap + i] = ok ? bj : -1;
    if (out_dist) out_dist[(size_t)pk * cap + i] = ok ? bd : -1;
}

// Reference-mode matching (post_processing.cu:92-200) through the same cell index.  The reference's thread
// visits the curr keypoints tile by tile (32 per tile), inside a tile of m entries in the order
// j = (s + tid) % m, skips a tile when tid >= m (Q8), and keeps the FIRST strictly smaller distance: the
// winner is the lexicographic minimum of (distance, visiting rank) with rank = 32 * tile + (j - tid) mod m
// -- an order-free reduction, so only the few candidates inside the +-window cells have to be looked at
// instead of all of them tile by tile.  Same result as match_batch_ref_kernel, bit for bit.
__global__ void __launch_bounds__(256)
match_window_ref_kernel(const orbfe_keypoint *__restrict__ records, const int32_t *__restrict__ counts, int cap, int first,
                        int stride, int K, int cells_x, int cells_y, float inv_cell, const int32_t *__restrict__ bend_all,
                        const uint16_t *__restrict__ sorted_all, const uint32_t *__restrict__ d32_all, int window,
                        int max_ham, int32_t *__restrict__ out_idx, int32_t *__restrict__ out_dist)
{
    int pk, blk;
    xcd_remap(gridDim.x, gridDim.y, &pk, &blk);
    const int p = first + pk * stride;
    const int nA = clamp_count(counts[p], cap), nB = clamp_count(counts[p + 1], cap);
    const int i = blk * 256 + threadIdx.x;
    if (i >= cap) return;
    uint32_t best = 0xFFFFFFFFu;
    const int tid = i & 31;
    if (i < nA && nB > 0) {
        const uint32_t *A = reinterpret_cast<const uint32_t *>(records + (size_t)p * cap) + 13 * (size_t)i;
        const uint32_t *B = reinterpret_cast<const uint32_t *>(records + (size_t)(p + 1) * cap);
        const int32_t *bend = bend_all + (size_t)(p + 1) * K;
        const uint16_t *sorted = sorted_all + (size_t)(p + 1) * cap;
        const uint32_t *d32 = d32_all + (size_t)(p + 1) * cap;
        const float ax = __uint_as_float(A[0]), ay = __uint_as_float(A[1]), win = (float)window;
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = A[5 + k];
        const uint32_t da = compress_words(w);
        const int bx0 = bucket_coord(ax - win, inv_cell, cells_x), bx1 = bucket_coord(ax + win, inv_cell, cells_x);
        const int by0 = bucket_coord(ay - win, inv_cell, cells_y), by1 = bucket_coord(ay + win, inv_cell, cells_y);
        for (int by = by0; by <= by1; by++) {
            const int g0 = by * cells_x + bx0, g1 = by * cells_x + bx1;
            int t = g0 > 0 ? bend[g0 - 1] : 0;
            const int e = bend[g1];
            for (; t < e; t++) {
                const int j = sorted[t];
                const int base = j & ~31, m = base + 32 >= nB ? nB - base : 32;
                if (tid >= m) continue; // the reference's thread skips this (partial) tile
                const uint32_t *r = B + 13 * (size_t)j;
                if (fabsf(ax - __uint_as_float(r[0])) > win || fabsf(ay - __uint_as_float(r[1])) > win) continue;
                const int hd = __popc(da ^ d32[j]);
                if (hd >= max_ham) continue;
                int rot = (j - base) - tid;
                rot = rot < 0 ? rot + m : rot;
                const uint32_t key = ((uint32_t)hd << 16) | (uint32_t)(base + rot);
                best = key < best ? key : best;
            }
        }
    }
    int bj = -1, bd = -1;
    if (best != 0xFFFFFFFFu) {
        const int rank = (int)(best & 0xFFFFu), base = rank & ~31, m = base + 32 >= nB ? nB - base : 32;
        int jl = (rank - base) + tid;
        jl = jl >= m ? jl - m : jl;
        bj = base + jl;
        bd = (int)(best >> 16);
    }
    out_idx[(size_t)pk * cap + i] = bj;
    if (out_dist) out_dist[(size_t)pk * cap + i] = bd;
}

} // namespace orbfe

// ======================================================================================
// C ABI
// ======================================================================================
using namespace orbfe;

namespace orbfe {
// Exhaustive check of the orientation table against the arithmetic it replaces: one thread per float `angle` in
// [first, first + count) of the ORDERED bit patterns (positive floats ascending), both signs, all 512 rotated points.
// round in which lane `lane` evaluates descriptor word `word` under the schedule masks (inverse of sched_perm)
__device__ inline int sched_round(const SteerArgs &st, int lane, int word)
{
    int p[4] = {0, 1, 2, 3}, t;
    if (st.sched_mask[0] >> lane & 1) { t = p[0]; p[0] = p[1]; p[1] = t; }
    if (st.sched_mask[1] >> lane & 1) { t = p[2]; p[2] = p[3]; p[3] = t; }
    if (st.sched_mask[2] >> lane & 1) { t = p[0]; p[0] = p[2]; p[2] = t; }
    if (st.sched_mask[3] >> lane & 1) { t = p[1]; p[1] = p[3]; p[3] = t; }
    return p[0] == word ? 0 : (p[1] == word ? 1 : (p[2] == word ? 2 : 3));
}
__global__ void __launch_bounds__(256) steer_check_kernel(SteerArgs st, int pitch, uint32_t first, uint32_t count,
                                                          unsigned long long *bad)
{
    ORBFE_NO_CONTRACT
    const int16_t *tab = reinterpret_cast<const int16_t *>(st.table);
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < 2ull * count; i += (uint64_t)gridDim.x * 256) {
        const uint32_t bits = first + (uint32_t)(i >> 1) | ((uint32_t)(i & 1) << 31);
        const float angle = __uint_as_float(bits);
        int pos = 0;
#pragma unroll
        for (int step = 512; step > 0; step >>= 1) {
            const int t = pos + step;
            if (t <= st.n_breaks && st.breaks[t - 1] <= angle) pos = t;
        }
        float a, b;
        orb_steer(angle, 0, &a, &b);
        const int16_t *row_tab = tab + (size_t)pos * 512;
        for (int t = 0; t < ORBFE_PATTERN_TESTS; t++)
#pragma unroll
            for (int which = 0; which < 2; which++) {
                const float px = (float)c_pattern[4 * t + 2 * which], py = (float)c_pattern[4 * t + 2 * which + 1];
                const float p1 = px * b, p2 = py * a, p3 = px * a, p4 = py * b;
                const int off = (int)__builtin_rintf(p1 + p2) * pitch + (int)__builtin_rintf(p3 - p4);
                mine += off != (int)row_tab[(t & 63) * 8 + 2 * sched_round(st, t & 63, t >> 6) + which];
            }
    }
    if (mine) atomicAdd(bad, mine);
}

} // namespace orbfe

static inline hipStream_t S(orbfe_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
// grid and geometry of a launch over (items per frame) x (n frames) for kernels that place themselves with frame_item()
// n / d for 0 <= n < 65536 and 1 <= d <= 256 as one s_mul_hi_u32 (device_common.hpp div_by_magic): magic = ceil(2^32 / d),
// 0 for d == 1; exact because n * (magic * d - 2^32) < 65536 * 256 < 2^32
static inline uint32_t magic_of(int d) { return d <= 1 ? 0u : (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); }
static inline dim3 frame_grid(int items, int n) { return n >= 8 ? dim3(8u * (unsigned)items, (unsigned)(n + 7) / 8u) : dim3(items, n); }
static inline DeviceGeom with_frames(const DeviceGeom &g, int n)
{
    DeviceGeom r = g;
    r.n_frames = n;
    r.grid8 = n >= 8;
    return r;
}
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Identity of the graph capture `stream` is in (0 = not capturing).  The "cell keys are already
// clear" shortcut between build_pyramid and detect_batch is a host-side flag, so it is only valid
// when both calls are issued on the same stream in the same mode: both eager, or both inside the
// same capture (then the graph contains the clearing kernel too).
static unsigned long long capture_id_of(hipStream_t stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(stream, &st, &id) != hipSuccess) {
        (void)hipGetLastError();
        return ~0ull; // unknown: never matches
    }
    return st == hipStreamCaptureStatusActive ? (id ? id : ~0ull - 1) : 0ull;
}

// ---- which kernels a call runs: ONE definition, used by the dispatch below and by orbfe_dispatch_info ----
static bool describe_uses_patch(const orbfe_ctx *ctx, int n_frames)
{
    // (the tile kernel finds a cell's record through the 16-bit cell -> slot map, 0xFFFF = none: frames that can
    // hold more than 65534 records take the patch kernel, which walks the 32-bit selection list)
    return ctx->describe_patch == 2 || (ctx->describe_patch == 1 && (long long)n_frames * ctx->g.cap >= 32768) ||
           ctx->g.cap > 65534;
}
enum MatchPath { kMatchRefWindow, kMatchRefLiteral, kMatchMfma, kMatchWindow256, kMatchValu256 };
static MatchPath match_path(const orbfe_ctx *ctx, int n_frames, int mode, int window)
{
    const DeviceGeom &g = ctx->g;
    const int wc = window >= 0 ? 2 * ((window + g.cell - 1) / g.cell) + 1 : 0; // cells per window edge, at most
    const bool few_cells = window >= 0 && (long long)wc * wc * 4 <= (long long)g.K;
    if (mode == 0)
        return (ctx->d_bend && ctx->d_bd32 && n_frames <= ctx->cfg.max_batch && few_cells) ? kMatchRefWindow : kMatchRefLiteral;
    if (window < 0 && ctx->d_mexp) return kMatchMfma;        // all candidates, <= 16384 per frame: matrix cores
    if (ctx->d_bend && few_cells) return kMatchWindow256;    // a window that spans few cells: walk the cell index
    return kMatchValu256;
}

// ---- the context's pyramid layout and detection tile list, from the configuration alone (no device) --------------
// guard bands: a detection tile reads rows y0 - 4 .. y0 + 67 and columns x0 - 4 .. x0 + 67 of its level without range
// tests (what lies outside the image is never used: the validity masks of phase B); inside the buffer that is a
// neighbouring level or frame, at its two ends it is these bands.
static_assert(kPxH == kTileH + 8 && kPxW == kTileW + 8 && kPxW % 4 == 0,
              "the guard bands below are sized for the pixel tile detect_tile_kernel loads: 4-px halo on every side");
static void layout_of(const orbfe_config *cfg, DeviceGeom *gp, std::vector<TileDesc> *tiles, size_t *guard)
{
    DeviceGeom &g = *gp;
    memset(&g, 0, sizeof(g));
    g.W = cfg->width;
    g.H = cfg->height;
    g.L = cfg->levels;
    g.cell = cfg->cell;
    g.cells_x = (g.W + g.cell - 1) / g.cell;
    g.cells_y = (g.H + g.cell - 1) / g.cell;
    g.K = g.cells_x * g.cells_y;
    g.cap = (cfg->max_features > 0 && cfg->max_features < g.K) ? cfg->max_features : g.K;
    g.threshold = cfg->fast_threshold;
    g.arc = cfg->min_arc;
    g.max_features = cfg->max_features;
    g.angle_in_radians = cfg->angle_in_radians ? 1 : 0;
    g.descriptor_level = cfg->descriptor_level ? 1 : 0;
    size_t off = 0;
    for (int l = 0; l < g.L; l++) {
        g.lv[l].w = g.W >> l;
        g.lv[l].h = g.H >> l;
        g.lv[l].pitch = (int)align_up((size_t)(g.lv[l].w > 0 ? g.lv[l].w : 1), 64);
        g.lv[l].offset = off;
        off += align_up((size_t)g.lv[l].pitch * (size_t)(g.lv[l].h > 0 ? g.lv[l].h : 1), 256);
    }
    g.frame_stride = off;
    g.Ld = 0;
    while (g.Ld < g.L && (g.cell >> g.Ld) > 0 && g.lv[g.Ld].w > 0 && g.lv[g.Ld].h > 0) g.Ld++;
    tiles->clear();
    for (int l = 0; l < g.Ld; l++) {
        if (g.lv[l].w < 7 || g.lv[l].h < 7) continue; // no pixel is >= 3 from every border
        const int tx = (g.lv[l].w + kTileW - 1) / kTileW, ty = (g.lv[l].h + kTileH - 1) / kTileH;
        for (int y = 0; y < ty; y++)
            for (int x = 0; x < tx; x++) tiles->push_back(TileDesc{(int16_t)l, (int16_t)x, (int16_t)y, 0});
    }
    *guard = align_up((size_t)kPxH * (size_t)g.lv[0].pitch + 256, 256);
}

// lowest and highest byte, relative to frame 0's level 0 (d_pyr), that the unconditional loads of ANY detection tile of
// ANY of n_frames frames touch: dword i < kPxH * kPxDw of a tile is row i / kPxDw, group i % kPxDw, at
// (y0 - 4 + row) * pitch + x0 - 4 + 4 * group of its level (detect_tile_kernel phase A); lanes past the last dword
// re-load their first one
static void tile_load_extent(const DeviceGeom &g, const std::vector<TileDesc> &tiles, size_t n_frames, long long *lo, long long *hi)
{
    long long mn = 0, mx = -1;
    bool any = false;
    for (const TileDesc &t : tiles) {
        const long long P = g.lv[t.level].pitch, base = (long long)g.lv[t.level].offset;
        const long long x0 = (long long)t.tx * kTileW, y0 = (long long)t.ty * kTileH;
        const long long first = base + (y0 - 4) * P + x0 - 4;
        const long long last = base + (y0 - 4 + kPxH - 1) * P + x0 - 4 + 4 * (kPxDw - 1) + 3;
        const long long last_f = last + (long long)(n_frames - 1) * (long long)g.frame_stride;
        if (!any || first < mn) mn = first;
        if (!any || last_f > mx) mx = last_f;
        any = true;
    }
    *lo = any ? mn : 0;
    *hi = any ? mx : -1;
}

#define CTX_FAIL(ctx, code, ...)                                                            \
    do {                                                                                    \
        format_error((ctx) ? (ctx)->err : nullptr, __VA_ARGS__);                            \
        return code;                                                                        \
    } while (0)

#define CTX_LAUNCH_CHECK(ctx, what)                                                         \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) CTX_FAIL(ctx, ORBFE_ERR_HIP, "%s launch failed: %s", what,    \
                                       hipGetErrorString(e_));                              \
    } while (0)

extern "C" {

void orbfe_default_config(orbfe_config *cfg, int width, int height)
{
    if (!cfg) return;
    cfg->width = width;
    cfg->height = height;
    cfg->levels = 1;          // PYRAMID_LEVELS, src/SlamGpuPipeline/defines.h:2
    cfg->cell = 32;           // CELL_SIZE_WIDTH/HEIGHT, defines.h:17-18; nms.cu:266-267
    cfg->fast_threshold = 13; // FAST_EPSILON, defines.h:7
    cfg->min_arc = 12;        // FAST_MIN_ARC_LENGTH, defines.h:8
    cfg->max_features = 0;
    cfg->angle_in_radians = 0;
    cfg->max_batch = 1;
    cfg->device = 0;
    cfg->descriptor_level = 0; // level-0 description, buildStream.cpp:442-460 (Q10)
}

const char *orbfe_last_error(const orbfe_ctx *ctx) { return ctx ? ctx->err : thread_error(); }

int orbfe_layout_bounds(const orbfe_config *cfg, long long *tile_lo, long long *tile_hi, unsigned long long *pyramid_bytes,
                        unsigned long long *guard_bytes, int *n_tiles)
{
    if (!cfg || cfg->width < 8 || cfg->height < 8 || cfg->width > 16384 || cfg->height > 16384 || cfg->levels < 1 ||
        cfg->levels > kMaxLevels || (cfg->cell != 8 && cfg->cell != 16 && cfg->cell != 32 && cfg->cell != 64) || cfg->max_batch < 1)
        return ORBFE_ERR_INVALID_ARG;
    DeviceGeom g;
    std::vector<TileDesc> tiles;
    size_t guard = 0;
    layout_of(cfg, &g, &tiles, &guard);
    long long lo = 0, hi = -1;
    tile_load_extent(g, tiles, (size_t)cfg->max_batch, &lo, &hi);
    if (tile_lo) *tile_lo = lo;
    if (tile_hi) *tile_hi = hi;
    if (pyramid_bytes) *pyramid_bytes = (unsigned long long)cfg->max_batch * g.frame_stride;
    if (guard_bytes) *guard_bytes = guard;
    if (n_tiles) *n_tiles = (int)tiles.size();
    // the tile describe kernel's staging clamps every 16-byte chunk into [0, H - 1] x [0, pitch - 16] of its level
    // (describe_tile_kernel, "stage the tile"): legal iff every level's pitch holds at least one chunk
    for (int l = 0; l < g.L; l++)
        if (g.lv[l].pitch < 16 || g.lv[l].pitch % 16 != 0) return ORBFE_ERR_UNSUPPORTED;
    return ORBFE_OK;
}

int orbfe_create(const orbfe_config *cfg, orbfe_ctx **out)
{
    if (!cfg || !out) CTX_FAIL((orbfe_ctx *)nullptr, ORBFE_ERR_INVALID_ARG, "orbfe_create: null argument");
    *out = nullptr;
    if (cfg->width < 8 || cfg->height < 8 || cfg->width > 16384 || cfg->height > 16384)
        CTX_FAIL((orbfe_ctx *)nullptr, ORBFE_ERR_INVALID_ARG, "orbfe_create: frame size %dx%d out of range", cfg->width, cfg->height);
    if (cfg->levels < 1 || cfg->levels > kMaxLevels)
        CTX_FAIL((orbfe_ctx *)nullptr, ORBFE_ERR_INVALID_ARG, "orbfe_create: levels %d not in 1..%d", cfg->levels, kMaxLevels);
    if (cfg->cell != 8 && cfg->cell != 16 && cfg->cell != 32 && cfg->cell != 64)
        CTX_FAIL((orbfe_ctx *)nullptr, ORBFE_ERR_UNSUPPORTED, "orbfe_create: cell %d not in {8,16,32,64}", cfg->cell);
    if (cfg->min_arc < 9 || cfg->min_arc > 12) // guard upstream vilib had (fast_gpu.cpp:77), Q12
        CTX_FAIL((orbfe_ctx *)nullptr, ORBFE_ERR_UNSUPPORTED, "orbfe_create: min_arc %d not in 9..12", cfg->min_arc);
    if (cfg->fast_threshold < 1 || cfg->fast_threshold > 254)
        CTX_FAIL((orbfe_ctx *)nullptr, ORBFE_ERR_INVALID_ARG, "orbfe_create: fast_threshold %d not in 1..254", cfg->fast_threshold);
    if (cfg->max_features < 0 || cfg->max_batch < 1)
        CTX_FAIL((orbfe_ctx *)nullptr, ORBFE_ERR_INVALID_ARG, "orbfe_create: max_features/max_batch");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        CTX_FAIL((orbfe_ctx *)nullptr, ORBFE_ERR_NO_DEVICE, "orbfe_create: no HIP device (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev)
        CTX_FAIL((orbfe_ctx *)nullptr, ORBFE_ERR_INVALID_ARG, "orbfe_create: device %d of %d", cfg->device, ndev);
    DeviceScope dev(cfg->device);
    if (!dev.ok) CTX_FAIL((orbfe_ctx *)nullptr, ORBFE_ERR_HIP, "orbfe_create: hipSetDevice(%d) failed", cfg->device);

    orbfe_ctx *ctx = new orbfe_ctx();
    ctx->cfg = *cfg;
    DeviceGeom &g = ctx->g;
    std::vector<TileDesc> tiles;
    size_t guard = 0;
    layout_of(cfg, &g, &tiles, &guard);
    ctx->n_tiles = (int)tiles.size();

    const size_t B = (size_t)cfg->max_batch;
    {
        // the invariant behind the unconditional tile loads (detect_tile_kernel phase A), checked per geometry on the
        // host: round 3's work-in-progress form of those loads faulted ("Memory access fault by GPU node-2",
        // gpurun_out/r3i) because the guard bands were not there yet; tests/test_layout.py runs the same check over
        // random geometries without a device
        long long lo = 0, hi = 0;
        tile_load_extent(g, tiles, B, &lo, &hi);
        if (lo < -(long long)guard || hi >= (long long)(B * g.frame_stride + guard)) {
            delete ctx;
            CTX_FAIL((orbfe_ctx *)nullptr, ORBFE_ERR_UNSUPPORTED,
                     "orbfe_create: internal: a detection tile would read bytes %lld .. %lld of a pyramid of %zu bytes with "
                     "%zu-byte guard bands", lo, hi, B * g.frame_stride, guard);
        }
    }
    hipError_t e = hipMalloc((void **)&ctx->d_pyr_alloc, B * g.frame_stride + 2 * guard);
    if (e == hipSuccess) e = hipMemset(ctx->d_pyr_alloc, 0, B * g.frame_stride + 2 * guard);
    if (e == hipSuccess) ctx->d_pyr = ctx->d_pyr_alloc + guard;
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_cellkey, B * g.K * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_sel, B * g.cap * sizeof(uint4));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_selcount, B * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_cellslot, B * g.K * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_mdesc, B * g.cap * 32);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_mpos, B * g.cap * 8);
    if (g.cap <= 65535) { // windowed 256-bit matching: cell buckets of the curr frames
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_bend, B * g.K * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_bsorted, B * g.cap * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_bd32, B * g.cap * sizeof(uint32_t));
    }
    ctx->cap_pad = (g.cap + 15) / 16 * 16;
    {
        const std::vector<int8_t> w = make_moment_weights(g.angle_in_radians ? 19 : 15);
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_momw, w.size());
        if (e == hipSuccess) e = hipMemcpy(ctx->d_momw, w.data(), w.size(), hipMemcpyHostToDevice);
        const std::vector<int8_t> wt = make_tile_moment_weights();
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_momw_tile, wt.size());
        if (e == hipSuccess) e = hipMemcpy(ctx->d_momw_tile, wt.data(), wt.size(), hipMemcpyHostToDevice);
        if (!g.angle_in_radians) { // the rotated pattern as a table: ~230 KB
            std::vector<float> br;
            std::vector<int16_t> off;
            build_steer_table(TileGeom<15>::kPitch, &br, &off, &ctx->steer_central, ctx->steer_sched_mask);
            ctx->n_steer_breaks = (int)br.size();
            if (ctx->n_steer_breaks > kSteerMaxBreaks) e = hipErrorInvalidValue; // (222 today; the kernel's LDS copy holds 224)
            if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_steer_breaks, (br.size() + 1) * sizeof(float));
            if (e == hipSuccess) e = hipMemcpy(ctx->d_steer_breaks, br.data(), br.size() * sizeof(float), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_steer_table, off.size() * sizeof(int16_t));
            if (e == hipSuccess) e = hipMemcpy(ctx->d_steer_table, off.data(), off.size() * sizeof(int16_t), hipMemcpyHostToDevice);
            build_steer_table(PatchGeom<15>::kPitch, &br, &off, &ctx->steer_central, ctx->steer_sched_mask);
            if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_steer_table_patch, off.size() * sizeof(int16_t));
            if (e == hipSuccess) e = hipMemcpy(ctx->d_steer_table_patch, off.data(), off.size() * sizeof(int16_t), hipMemcpyHostToDevice);
        }
    }
    {
        // Few keypoints per 64x64 tile (the reference regime: one per 32-px cell = 4 per tile): staging a private
        // patch per keypoint moves less than staging every tile; from ~8 per tile on the tile kernel wins
        // (reference mode, 256 frames: 0.040 vs 0.053 ms; C4 at 8.3 per tile: 0.078 vs 0.072).  The patch kernel
        // also needs enough keypoints in the call to fill the chip (describe_batch checks that: one 4K frame
        // with 3.9 per tile runs 0.077 ms as patches, 0.060 as 2040 tiles).
        const int n_tiles = ((g.W + 63) / 64) * ((g.H + 63) / 64);
        ctx->describe_patch = g.cap < 8 * n_tiles ? 1 : 0;
        const char *v = getenv("ORBFE_DESCRIBE"); // A/B timing of the two describe kernels on one box
        if (v && !strcmp(v, "patch")) ctx->describe_patch = 2;
        else if (v && !strcmp(v, "tile")) ctx->describe_patch = -1; // (anything else: ignored)
    }
    {
        const char *v = getenv("ORBFE_DETECT_GROUPS"); // detect_tile_kernel's tile groups: by launch size, or forced (the tests run both)
        if (v && !strcmp(v, "single")) ctx->detect_groups = 1;
        else if (v && !strcmp(v, "multi")) ctx->detect_groups = 2;
    }
    if (g.cap <= kMmaMaxKeypoints) { // scratch of the matrix-core matcher
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_mexp, B * ctx->cap_pad * 128);
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_mkey, B * ctx->cap_pad * sizeof(float));
        const char *mf = getenv("ORBFE_MATCH"); // A/B timing of the matrix-core matcher's forms on one box; the tests run both
        if (mf && !strcmp(mf, "stream")) ctx->match_form = 1;
        else if (mf && !strcmp(mf, "tile")) ctx->match_form = 2;
    }
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_tiles, (tiles.size() + 1) * sizeof(TileDesc));
    if (e == hipSuccess && !tiles.empty())
        e = hipMemcpy(ctx->d_tiles, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_thread_error("orbfe_create: device allocation failed: %s", hipGetErrorString(e));
        orbfe_destroy(ctx);
        return ORBFE_ERR_HIP;
    }
    *out = ctx;
    return ORBFE_OK;
}

void orbfe_destroy(orbfe_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->d_pyr_alloc) (void)hipFree(ctx->d_pyr_alloc);
    if (ctx->d_cellkey) (void)hipFree(ctx->d_cellkey);
    if (ctx->d_sel) (void)hipFree(ctx->d_sel);
    if (ctx->d_selcount) (void)hipFree(ctx->d_selcount);
    if (ctx->d_cellslot) (void)hipFree(ctx->d_cellslot);
    if (ctx->d_momw_tile) (void)hipFree(ctx->d_momw_tile);
    if (ctx->d_steer_breaks) (void)hipFree(ctx->d_steer_breaks);
    if (ctx->d_steer_table) (void)hipFree(ctx->d_steer_table);
    if (ctx->d_steer_table_patch) (void)hipFree(ctx->d_steer_table_patch);
    if (ctx->d_tiles) (void)hipFree(ctx->d_tiles);
    if (ctx->d_mdesc) (void)hipFree(ctx->d_mdesc);
    if (ctx->d_mpos) (void)hipFree(ctx->d_mpos);
    if (ctx->d_bend) (void)hipFree(ctx->d_bend);
    if (ctx->d_bsorted) (void)hipFree(ctx->d_bsorted);
    if (ctx->d_bd32) (void)hipFree(ctx->d_bd32);
    if (ctx->d_mexp) (void)hipFree(ctx->d_mexp);
    if (ctx->d_mkey) (void)hipFree(ctx->d_mkey);
    if (ctx->d_momw) (void)hipFree(ctx->d_momw);
    delete ctx;
}

int orbfe_num_cells(const orbfe_ctx *ctx) { return ctx ? ctx->g.K : 0; }
int orbfe_max_keypoints(const orbfe_ctx *ctx) { return ctx ? ctx->g.cap : 0; }
int orbfe_num_levels(const orbfe_ctx *ctx) { return ctx ? ctx->g.L : 0; }

int orbfe_level_info(const orbfe_ctx *ctx, int level, int *width, int *height, size_t *pitch,
                     const uint8_t **d_image, size_t *frame_stride)
{
    if (!ctx) return ORBFE_ERR_INVALID_ARG;
    if (level < 0 || level >= ctx->g.L) return ORBFE_ERR_INVALID_ARG;
    if (width) *width = ctx->g.lv[level].w;
    if (height) *height = ctx->g.lv[level].h;
    if (pitch) *pitch = (size_t)ctx->g.lv[level].pitch;
    if (d_image) *d_image = ctx->d_pyr + ctx->g.lv[level].offset;
    if (frame_stride) *frame_stride = ctx->g.frame_stride;
    return ORBFE_OK;
}

static int build_pyramid_impl(orbfe_ctx *ctx, const uint8_t *d_src, size_t pitch, size_t frame_stride, int n_frames,
                              bool rgb, orbfe_stream_t stream)
{
    if (!ctx) return ORBFE_ERR_INVALID_ARG;
    const DeviceGeom &g = ctx->g;
    const size_t row_bytes = rgb ? 3 * (size_t)g.W : (size_t)g.W;
    if (!d_src || n_frames < 1 || pitch < row_bytes || (n_frames > 1 && frame_stride < pitch * (size_t)g.H))
        CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "build_pyramid: bad input geometry");
    if (n_frames > ctx->cfg.max_batch)
        CTX_FAIL(ctx, ORBFE_ERR_CAPACITY, "build_pyramid: n_frames %d > max_batch %d", n_frames, ctx->cfg.max_batch);
    DeviceScope dev(ctx->cfg.device);
    if (!dev.ok) CTX_FAIL(ctx, ORBFE_ERR_HIP, "build_pyramid: hipSetDevice(%d) failed", ctx->cfg.device);
    const bool vec = (pitch % 4 == 0) && (frame_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_src) & 3u) == 0);
    if (rgb && !(vec && g.W % 4 == 0))
        CTX_FAIL(ctx, ORBFE_ERR_UNSUPPORTED, "build_pyramid_rgb: needs width %% 4 == 0 and a 4-byte aligned source "
                 "(use orbfe_rgb_to_grayscale + orbfe_build_pyramid otherwise)");
    int next_level = 1; // first level still to be produced by the unfused halving kernel
    ctx->cellkey_clean = 0;
    if (vec && g.W % 4 == 0) {
        const int tiles_x = (g.W + 127) / 128, tiles_y = (g.H + 127) / 128;
        if (rgb)
            hipLaunchKernelGGL(pyramid_fused_kernel<true>, frame_grid(tiles_x * tiles_y, n_frames), dim3(256), 0, S(stream),
                               with_frames(g, n_frames), ctx->d_pyr, d_src, (int)pitch, frame_stride, tiles_x, magic_of(tiles_x),
                               (g.K + tiles_x * tiles_y - 1) / (tiles_x * tiles_y), ctx->d_cellkey);
        else
            hipLaunchKernelGGL(pyramid_fused_kernel<false>, frame_grid(tiles_x * tiles_y, n_frames), dim3(256), 0, S(stream),
                               with_frames(g, n_frames), ctx->d_pyr, d_src, (int)pitch, frame_stride, tiles_x, magic_of(tiles_x),
                               (g.K + tiles_x * tiles_y - 1) / (tiles_x * tiles_y), ctx->d_cellkey);
        next_level = 8;
        // the fused kernel cleared these frames' cell keys; valid for a detect_batch issued next on
        // this stream in the same capture mode
        ctx->cellkey_clean = n_frames;
        ctx->clean_stream = S(stream);
        ctx->clean_capture = capture_id_of(S(stream));
    } else {
        dim3 grid(((g.W + 255) / 256) * ((g.H + 3) / 4), n_frames), block(256);
        if (vec)
            hipLaunchKernelGGL(blur_batch_kernel<true>, grid, block, 0, S(stream), ctx->d_pyr + g.lv[0].offset,
                               g.lv[0].pitch, g.frame_stride, d_src, (int)pitch, frame_stride, g.W, g.H);
        else
            hipLaunchKernelGGL(blur_batch_kernel<false>, grid, block, 0, S(stream), ctx->d_pyr + g.lv[0].offset,
                               g.lv[0].pitch, g.frame_stride, d_src, (int)pitch, frame_stride, g.W, g.H);
    }
    for (int l = next_level; l < g.L; l++) {
        const int dw = g.lv[l].w, dh = g.lv[l].h;
        if (dw == 0 || dh == 0) break;
        if (l >= 8 && dw * dh <= 4096) { // the remaining levels are tiny: one workgroup per frame does them all
            hipLaunchKernelGGL(pyramid_tail_kernel, dim3(n_frames), dim3(256), 0, S(stream), g, ctx->d_pyr, l);
            break;
        }
        dim3 grid(((dw + 255) / 256) * ((dh + 3) / 4), n_frames), block(256);
        hipLaunchKernelGGL(halfsample_batch_kernel, grid, block, 0, S(stream), ctx->d_pyr, g.frame_stride,
                           g.lv[l - 1].offset, g.lv[l - 1].pitch, g.lv[l].offset, g.lv[l].pitch, dw, dh);
    }
    CTX_LAUNCH_CHECK(ctx, "build_pyramid");
    return ORBFE_OK;
}

int orbfe_build_pyramid(orbfe_ctx *ctx, const uint8_t *d_gray, size_t pitch, size_t frame_stride, int n_frames,
                        orbfe_stream_t stream)
{
    return build_pyramid_impl(ctx, d_gray, pitch, frame_stride, n_frames, false, stream);
}

int orbfe_build_pyramid_rgb(orbfe_ctx *ctx, const uint8_t *d_rgb, size_t pitch, size_t frame_stride, int n_frames,
                            orbfe_stream_t stream)
{
    return build_pyramid_impl(ctx, d_rgb, pitch, frame_stride, n_frames, true, stream);
}

int orbfe_detect_batch_shard(orbfe_ctx *ctx, int n_frames, int shard_index, int shard_count,
                             orbfe_stream_t stream)
{
    if (!ctx) return ORBFE_ERR_INVALID_ARG;
    if (n_frames < 1) CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "detect_batch: n_frames %d", n_frames);
    if (shard_count < 1 || shard_index < 0 || shard_index >= shard_count)
        CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "detect_batch: shard %d of %d", shard_index, shard_count);
    if (n_frames > ctx->cfg.max_batch)
        CTX_FAIL(ctx, ORBFE_ERR_CAPACITY, "detect_batch: n_frames %d > max_batch %d", n_frames, ctx->cfg.max_batch);
    const DeviceGeom &g = ctx->g;
    DeviceScope dev(ctx->cfg.device);
    if (!dev.ok) CTX_FAIL(ctx, ORBFE_ERR_HIP, "detect_batch: hipSetDevice(%d) failed", ctx->cfg.device);
    const bool clean = ctx->cellkey_clean >= n_frames && ctx->clean_stream == S(stream) &&
                       ctx->clean_capture == capture_id_of(S(stream));
    if (!clean) // not cleared by the pyramid kernel that just ran on this stream, or already used
        ORBFE_HIP_TRY(ctx->err, hipMemsetAsync(ctx->d_cellkey, 0, (size_t)n_frames * g.K * sizeof(uint32_t), S(stream)));
    ctx->cellkey_clean = 0;
    // tiles shard_index, shard_index + shard_count, ... (interleaved: every shard gets a mix of
    // levels and of busy / empty image regions)
    const int n_mine = ctx->n_tiles > shard_index ? (ctx->n_tiles - shard_index + shard_count - 1) / shard_count : 0;
    if (n_mine > 0)
        launch_detect_tiles(with_frames(g, n_frames), n_mine, S(stream), ctx->d_pyr, ctx->d_tiles, ctx->d_cellkey, shard_index,
                            shard_count, ctx->detect_groups);
    CTX_LAUNCH_CHECK(ctx, "detect_batch");
    return ORBFE_OK;
}

int orbfe_detect_batch(orbfe_ctx *ctx, int n_frames, orbfe_stream_t stream)
{
    return orbfe_detect_batch_shard(ctx, n_frames, 0, 1, stream);
}

int orbfe_export_cell_keys(orbfe_ctx *ctx, int n_frames, uint32_t *d_keys, orbfe_stream_t stream)
{
    if (!ctx) return ORBFE_ERR_INVALID_ARG;
    if (!d_keys || n_frames < 1 || n_frames > ctx->cfg.max_batch)
        CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "export_cell_keys: bad argument");
    DeviceScope dev(ctx->cfg.device);
    if (!dev.ok) CTX_FAIL(ctx, ORBFE_ERR_HIP, "export_cell_keys: hipSetDevice(%d) failed", ctx->cfg.device);
    ORBFE_HIP_TRY(ctx->err, hipMemcpyAsync(d_keys, ctx->d_cellkey, (size_t)n_frames * ctx->g.K * sizeof(uint32_t),
                                           hipMemcpyDeviceToDevice, S(stream)));
    return ORBFE_OK;
}

int orbfe_import_cell_keys(orbfe_ctx *ctx, int n_frames, const uint32_t *d_keys, orbfe_stream_t stream)
{
    if (!ctx) return ORBFE_ERR_INVALID_ARG;
    if (!d_keys || n_frames < 1 || n_frames > ctx->cfg.max_batch)
        CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "import_cell_keys: bad argument");
    DeviceScope dev(ctx->cfg.device);
    if (!dev.ok) CTX_FAIL(ctx, ORBFE_ERR_HIP, "import_cell_keys: hipSetDevice(%d) failed", ctx->cfg.device);
    ctx->cellkey_clean = 0;
    ORBFE_HIP_TRY(ctx->err, hipMemcpyAsync(ctx->d_cellkey, d_keys, (size_t)n_frames * ctx->g.K * sizeof(uint32_t),
                                           hipMemcpyDeviceToDevice, S(stream)));
    return ORBFE_OK;
}

int orbfe_describe_batch(orbfe_ctx *ctx, int n_frames, orbfe_keypoint *d_records, int32_t *d_counts,
                         const orbfe_soa *soa, orbfe_stream_t stream)
{
    if (!ctx) return ORBFE_ERR_INVALID_ARG;
    if (!d_records || !d_counts) CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "describe_batch: null output");
    if (n_frames < 1) CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "describe_batch: n_frames %d", n_frames);
    if (n_frames > ctx->cfg.max_batch)
        CTX_FAIL(ctx, ORBFE_ERR_CAPACITY, "describe_batch: n_frames %d > max_batch %d", n_frames, ctx->cfg.max_batch);
    if ((reinterpret_cast<uintptr_t>(d_records) & 3u) || (soa && soa->d_desc && (reinterpret_cast<uintptr_t>(soa->d_desc) & 3u)))
        CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "describe_batch: outputs must be 4-byte aligned");
    const DeviceGeom &g = ctx->g;
    orbfe_soa so = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (soa) so = *soa;
    DeviceScope dev(ctx->cfg.device);
    if (!dev.ok) CTX_FAIL(ctx, ORBFE_ERR_HIP, "describe_batch: hipSetDevice(%d) failed", ctx->cfg.device);
    const bool patch = describe_uses_patch(ctx, n_frames);
    if (so.d_pos || so.d_score || so.d_level || so.d_angle || so.d_desc || so.d_desc32)
        hipLaunchKernelGGL(select_kernel<true>, dim3(n_frames), dim3(kSelThreads), 0, S(stream), g, ctx->d_cellkey,
                           patch ? ctx->d_sel : nullptr, ctx->d_cellslot, ctx->d_selcount, d_counts, so);
    else
        hipLaunchKernelGGL(select_kernel<false>, dim3(n_frames), dim3(kSelThreads), 0, S(stream), g, ctx->d_cellkey,
                           patch ? ctx->d_sel : nullptr, ctx->d_cellslot, ctx->d_selcount, d_counts, so);
    const bool want_soa = so.d_pos || so.d_score || so.d_level || so.d_angle || so.d_desc || so.d_desc32;
#define ORBFE_DESCRIBE_LAUNCH2(KERNEL, DLV, GRID, ...)                                                               \
    do {                                                                                                            \
        const dim3 thr(sizeof(#KERNEL) == sizeof("describe_tile_kernel") ? kDescThreads : 256);                    \
        if (g.angle_in_radians && want_soa) hipLaunchKernelGGL((KERNEL<19, true, DLV>), GRID, thr, 0, S(stream), __VA_ARGS__);  \
        else if (g.angle_in_radians) hipLaunchKernelGGL((KERNEL<19, false, DLV>), GRID, thr, 0, S(stream), __VA_ARGS__);        \
        else if (want_soa) hipLaunchKernelGGL((KERNEL<15, true, DLV>), GRID, thr, 0, S(stream), __VA_ARGS__);                   \
        else hipLaunchKernelGGL((KERNEL<15, false, DLV>), GRID, thr, 0, S(stream), __VA_ARGS__);                                \
    } while (0)
#define ORBFE_DESCRIBE_LAUNCH(KERNEL, GRID, ...)                                                                    \
    do {                                                                                                            \
        if (g.descriptor_level) ORBFE_DESCRIBE_LAUNCH2(KERNEL, true, GRID, __VA_ARGS__);                            \
        else ORBFE_DESCRIBE_LAUNCH2(KERNEL, false, GRID, __VA_ARGS__);                                              \
    } while (0)
    const DeviceGeom gl = with_frames(g, n_frames);
    const SteerArgs steer{ctx->d_steer_breaks, ctx->d_steer_table, ctx->n_steer_breaks, ctx->steer_central,
                          {ctx->steer_sched_mask[0], ctx->steer_sched_mask[1], ctx->steer_sched_mask[2], ctx->steer_sched_mask[3]}};
    SteerArgs steerp = steer;
    steerp.table = ctx->d_steer_table_patch;
    if (patch) {
        ORBFE_DESCRIBE_LAUNCH(describe_kernel, frame_grid((g.cap + 4 * kKpw - 1) / (4 * kKpw), n_frames), gl, ctx->d_pyr, ctx->d_sel,
                              ctx->d_selcount, ctx->d_momw, d_records, so, steerp);
    } else if (g.descriptor_level) { // one workgroup per detection tile: (level, 64x64 tile of that level)
        if (ctx->n_tiles > 0)
            ORBFE_DESCRIBE_LAUNCH2(describe_tile_kernel, true, frame_grid(ctx->n_tiles, n_frames), gl, ctx->d_pyr, ctx->d_cellkey,
                                   ctx->d_cellslot, ctx->d_momw_tile, 0, 0u, ctx->d_tiles, d_records, so, steer);
    } else {
        const int tiles_x = (g.W + kDTile - 1) / kDTile, tiles_y = (g.H + kDTile - 1) / kDTile;
        ORBFE_DESCRIBE_LAUNCH2(describe_tile_kernel, false, frame_grid(tiles_x * tiles_y, n_frames), gl, ctx->d_pyr, ctx->d_cellkey,
                               ctx->d_cellslot, ctx->d_momw_tile, tiles_x, magic_of(tiles_x), (const TileDesc *)nullptr, d_records, so, steer);
    }
#undef ORBFE_DESCRIBE_LAUNCH2
#undef ORBFE_DESCRIBE_LAUNCH
    CTX_LAUNCH_CHECK(ctx, "describe_batch");
    return ORBFE_OK;
}

int orbfe_extract(orbfe_ctx *ctx, const uint8_t *d_gray, size_t pitch, size_t frame_stride,
                  int n_frames, orbfe_keypoint *d_records, int32_t *d_counts, const orbfe_soa *soa,
                  orbfe_stream_t stream)
{
    if (!ctx) return ORBFE_ERR_INVALID_ARG;
    if (!d_records || !d_counts) CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "extract: null output");
    int rc = orbfe_build_pyramid(ctx, d_gray, pitch, frame_stride, n_frames, stream);
    if (rc == ORBFE_OK) rc = orbfe_detect_batch(ctx, n_frames, stream);
    if (rc == ORBFE_OK) rc = orbfe_describe_batch(ctx, n_frames, d_records, d_counts, soa, stream);
    return rc;
}

int orbfe_extract_rgb(orbfe_ctx *ctx, const uint8_t *d_rgb, size_t pitch, size_t frame_stride, int n_frames,
                      orbfe_keypoint *d_records, int32_t *d_counts, const orbfe_soa *soa, orbfe_stream_t stream)
{
    if (!ctx) return ORBFE_ERR_INVALID_ARG;
    if (!d_records || !d_counts) CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "extract_rgb: null output");
    int rc = orbfe_build_pyramid_rgb(ctx, d_rgb, pitch, frame_stride, n_frames, stream);
    if (rc == ORBFE_OK) rc = orbfe_detect_batch(ctx, n_frames, stream);
    if (rc == ORBFE_OK) rc = orbfe_describe_batch(ctx, n_frames, d_records, d_counts, soa, stream);
    return rc;
}

int orbfe_match_pairs(orbfe_ctx *ctx, const orbfe_keypoint *d_records, const int32_t *d_counts, int n_frames,
                      int first, int stride, int mode, int window, int max_distance, int32_t *d_idx, int32_t *d_dist,
                      orbfe_stream_t stream)
{
    if (!ctx) return ORBFE_ERR_INVALID_ARG;
    if (!d_records || !d_counts || !d_idx || n_frames < 1)
        CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "match: null argument");
    if (first < 0 || stride < 1) CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "match: first %d, stride %d", first, stride);
    if (mode != 0 && mode != 1) CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "match: mode %d", mode);
    if (mode == 0 && window < 0) CTX_FAIL(ctx, ORBFE_ERR_INVALID_ARG, "match: reference mode needs window >= 0");
    if (mode == 1 && ctx->g.cap > 65535) // packed (dist << 16 | index) key
        CTX_FAIL(ctx, ORBFE_ERR_UNSUPPORTED, "match: 256-bit mode supports at most 65535 keypoints per frame");
    // pairs (first + k * stride, first + k * stride + 1) with the curr frame inside the batch
    const int n_pairs = n_frames - 2 - first >= 0 ? (n_frames - 2 - first) / stride + 1 : 0;
    if (n_pairs < 1) return ORBFE_OK;
    const int cap = ctx->g.cap;
    DeviceScope dev(ctx->cfg.device);
    if (!dev.ok) CTX_FAIL(ctx, ORBFE_ERR_HIP, "match: hipSetDevice(%d) failed", ctx->cfg.device);
    dim3 grid((cap + 255) / 256, n_pairs), block(256);
    const DeviceGeom &g = ctx->g;
    const float inv_cell = 1.0f / (float)g.cell; // exact: the cell is a power of two
    if (mode == 1 && n_frames > ctx->cfg.max_batch)
        CTX_FAIL(ctx, ORBFE_ERR_CAPACITY, "match: n_frames %d > max_batch %d", n_frames, ctx->cfg.max_batch);
    switch (match_path(ctx, n_frames, mode, window)) {
    case kMatchRefWindow:
        launch_match_bucket(d_records, d_counts, n_pairs, cap, first, stride, g, inv_cell, ctx->d_bend, ctx->d_bsorted,
                            ctx->d_bd32, S(stream));
        hipLaunchKernelGGL(match_window_ref_kernel, grid, block, 0, S(stream), d_records, d_counts, cap, first, stride,
                           g.K, g.cells_x, g.cells_y, inv_cell, ctx->d_bend, ctx->d_bsorted, ctx->d_bd32, window,
                           max_distance, d_idx, d_dist);
        break;
    case kMatchRefLiteral:
        hipLaunchKernelGGL(match_batch_ref_kernel, grid, block, 0, S(stream), d_records, d_counts, cap, first, stride,
                           (float)window, max_distance, d_idx, d_dist);
        break;
    case kMatchMfma:
        launch_match_mfma(d_records, d_counts, n_frames, n_pairs, first, stride, cap, ctx->cap_pad, max_distance, ctx->d_mexp,
                          ctx->d_mkey, ctx->match_form, d_idx, d_dist, S(stream));
        break;
    case kMatchWindow256:
        launch_match_bucket(d_records, d_counts, n_pairs, cap, first, stride, g, inv_cell, ctx->d_bend, ctx->d_bsorted,
                            (uint32_t *)nullptr, S(stream));
        hipLaunchKernelGGL(match_window_kernel, grid, block, 0, S(stream), d_records, d_counts, cap, first, stride, g.K,
                           g.cells_x, g.cells_y, inv_cell, ctx->d_bend, ctx->d_bsorted, window, max_distance, d_idx, d_dist);
        break;
    case kMatchValu256: {
        Desc8 *md = reinterpret_cast<Desc8 *>(ctx->d_mdesc);
        float2 *mp = reinterpret_cast<float2 *>(ctx->d_mpos);
        hipLaunchKernelGGL(match_gather_kernel, dim3((cap + 255) / 256, n_frames), block, 0, S(stream), d_records,
                           d_counts, cap, md, mp);
        if (window >= 0)
            hipLaunchKernelGGL(match_batch_256_kernel<true>, grid, block, 0, S(stream), md, mp, d_counts, cap, first,
                               stride, window, max_distance, d_idx, d_dist);
        else
            hipLaunchKernelGGL(match_batch_256_kernel<false>, grid, block, 0, S(stream), md, mp, d_counts, cap, first,
                               stride, window, max_distance, d_idx, d_dist);
        break;
    }
    }
    CTX_LAUNCH_CHECK(ctx, "match");
    return ORBFE_OK;
}

int orbfe_dispatch_info(const orbfe_ctx *ctx, int n_frames, int mode, int window, char *buf, size_t size)
{
    if (!ctx || !buf || size == 0 || n_frames < 1 || (mode != 0 && mode != 1)) return ORBFE_ERR_INVALID_ARG;
    const DeviceGeom &g = ctx->g;
    const char *desc = describe_uses_patch(ctx, n_frames) ? "describe_kernel" : "describe_tile_kernel";
    const char *match = "";
    switch (match_path(ctx, n_frames, mode, window)) {
    case kMatchRefWindow: match = "match_bucket_kernel+match_window_ref_kernel"; break;
    case kMatchRefLiteral: match = "match_batch_ref_kernel"; break;
    case kMatchMfma: {
        const int np = (n_frames - 2) / 1 + 1; // orbfe_match_batch's pairs (orbfe_match_pairs with another stride has fewer)
        match = match_mfma_uses_tile(np > 0 ? np : 0, ctx->cap_pad, ctx->match_form) ? "match_tile_kernel" : "match_expand_kernel+match_mfma_kernel";
        break;
    }
    case kMatchWindow256: match = "match_bucket_kernel+match_window_kernel"; break;
    case kMatchValu256: match = "match_gather_kernel+match_batch_256_kernel"; break;
    }
    snprintf(buf, size, "pyramid=%s;detect=detect_tile_kernel<%d>%s;describe=select_kernel+%s%s;match=%s;match_examines=%s",
             (g.W % 4 == 0) ? "pyramid_fused_kernel" : "blur_batch_kernel+halfsample_batch_kernel", g.arc,
             detect_uses_groups(ctx->n_tiles, n_frames, ctx->detect_groups) ? "<groups of 4 tiles>" : "", desc,
             g.descriptor_level ? "<descriptor_level>" : "", match,
             match_path(ctx, n_frames, mode, window) == kMatchMfma || match_path(ctx, n_frames, mode, window) == kMatchValu256 ||
                     match_path(ctx, n_frames, mode, window) == kMatchRefLiteral
                 ? "all_pairs" : "window_cells");
    return ORBFE_OK;
}

int orbfe_selfcheck_steer_table(orbfe_ctx *ctx, unsigned long long *n_angles, unsigned long long *n_mismatch)
{
    if (!ctx || !n_angles || !n_mismatch) return ORBFE_ERR_INVALID_ARG;
    *n_angles = *n_mismatch = 0;
    if (!ctx->d_steer_table) return ORBFE_OK; // angle_in_radians: the offsets are computed, there is no table
    DeviceScope dev(ctx->cfg.device);
    if (!dev.ok) CTX_FAIL(ctx, ORBFE_ERR_HIP, "selfcheck_steer_table: hipSetDevice(%d) failed", ctx->cfg.device);
    unsigned long long *d_bad = nullptr;
    ORBFE_HIP_TRY(ctx->err, hipMalloc((void **)&d_bad, sizeof(*d_bad)));
    ORBFE_HIP_TRY(ctx->err, hipMemset(d_bad, 0, sizeof(*d_bad)));
    uint32_t pi_bits;
    const float pi = ORBFE_PI_F; // atan2f's range: [-pi, pi] as floats
    memcpy(&pi_bits, &pi, 4);
    const uint32_t count = pi_bits + 1; // bit patterns 0 .. pi_bits, each with both signs
    const SteerArgs st{ctx->d_steer_breaks, ctx->d_steer_table, ctx->n_steer_breaks, ctx->steer_central,
                       {ctx->steer_sched_mask[0], ctx->steer_sched_mask[1], ctx->steer_sched_mask[2], ctx->steer_sched_mask[3]}};
    hipLaunchKernelGGL(steer_check_kernel, dim3(256 * 32), dim3(256), 0, 0, st, (int)TileGeom<15>::kPitch, 0u, count, d_bad);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(n_mismatch, d_bad, sizeof(*d_bad), hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    if (e != hipSuccess) CTX_FAIL(ctx, ORBFE_ERR_HIP, "selfcheck_steer_table: %s", hipGetErrorString(e));
    *n_angles = 2ull * count;
    return ORBFE_OK;
}

int orbfe_match_batch(orbfe_ctx *ctx, const orbfe_keypoint *d_records, const int32_t *d_counts, int n_frames,
                      int mode, int window, int max_distance, int32_t *d_idx, int32_t *d_dist,
                      orbfe_stream_t stream)
{
    return orbfe_match_pairs(ctx, d_records, d_counts, n_frames, 0, 1, mode, window, max_distance, d_idx, d_dist, stream);
}

} // extern "C"
