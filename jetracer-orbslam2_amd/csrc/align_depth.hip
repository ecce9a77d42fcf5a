// align_depth.hip -- SURVEY.md 8f-2, the producing half: align_depth_to_other (depth image -> the colour camera's
// pixel grid), the step whose output orbfe_keypoint_pixel_to_point reads.  gfx950 only; wave = 64 lanes.
//
// Reference: src/cuda/cuda-align.cu:366-399 issues four launches per frame --
//   kernel_map_depth_to_other (:163-188, :121-161): for both corners (-0.5, +0.5) of every depth pixel deproject, transform,
//       project, round; result to an int2 map of 2 W H entries (16 B per pixel written, then read back);
//   kernel_reset_to_max (:257-267): output = 9999999;
//   kernel_depth_to_other (:224-255): atomicMin of the raw depth over the rectangle [p0, p1] of every depth pixel;
//   kernel_reset_to_zero (:269-280): 9999999 -> 0.
// Here: ONE kernel maps and splats (the map never exists in memory), a 64 x 16 tile of depth pixels per workgroup:
//   - the per-column / per-row quotients (x -+ 0.5 - ppx) / fx, (y -+ 0.5 - ppy) / fy are formed once per tile (81 IEEE
//     divisions for 1024 pixels x 2 corners; corner +0.5 of pixel x IS corner -0.5 of pixel x + 1, bit for bit);
//   - the tile's rectangles meet in an LDS window (ds_min_u32) when their bounding box fits 24 KB -- the usual case: a
//     depth pixel covers 1..9 output pixels and neighbours cover the same ones -- and only the window's touched entries go
//     to memory, one atomic per output pixel and tile instead of one per covered pixel and depth pixel; a tile whose
//     rectangles are scattered (degenerate extrinsics, depth discontinuities across metres) splats straight to memory.
//   - atomicMin is order-free, so the result is deterministic and equals the reference's: min raw depth over the depth
//     pixels whose rectangle covers the output pixel, 0 where none does;
//   - the two quotients of a projection share one reciprocal when that is provably the compiler's own division sequence
//     (div2_in_range / div2_guard below), the whole wave takes the plain `/` otherwise;
//   - with more than one chunk of frames per call, the reset of the next chunk and the close of the previous one are
//     carried by extra workgroups of the splat launch itself (memory-bound roles beside the vector-ALU-bound tiles).
// Two exact forms of the output protocol:
//   literal form (the default of both entry points): reset the grid's part of the output to 9999999, atomicMin WITHOUT
//     return, 9999999 -> 0 on the same part: 3 launches.  Also what reproduces the reference where the intrinsics' sizes
//     exceed the grid made from image_width / image_height (:378-380): output pixels beyond it keep min(what the caller's
//     buffer held, splats).
//   zero-init form (ORBFE_ALIGN_PROTOCOL=zero; whole output inside the launch grid only): output cleared to 0, "0 =
//     nothing yet", a window entry v lands by  old = CAS(p, 0, v); if (old != 0 && v < old) atomicMin(p, v)  -- raw depths
//     that reach a splat are >= 1 (depth 0 is skipped, :140), so 0 is free to mean "empty" and no closing pass is needed:
//     2 launches.  Built first, kept as the A/B: atomics WITH return cost more than the closing pass saves (3.09 against
//     2.10 ms per 1024 frames, 16.1 against 11.4 us for a single frame).
// Arithmetic: float, left to right as written in the reference, no contraction (ORBFE_NO_CONTRACT; the build has
// -ffp-contract=off), IEEE division; static_cast<int>(v + 0.5f) is v_cvt_i32_f32 = truncation, saturating, NaN -> 0,
// exactly CUDA's cvt.rzi.s32.f32.  Parity with the reference is unpinned at the ulp level (nvcc may fuse a*b+c).
#include "orbfe_internal.hpp"
#include "device_common.hpp"

namespace orbfe {

constexpr int kAlignTileW = 64, kAlignTileH = 16; // depth pixels per workgroup: 256 threads x 4 pixels of one row
constexpr int kAlignWin = 6144;                    // LDS window, u32 entries (24 KB: six workgroups per CU)
constexpr uint32_t kAlignMax = 9999999u;           // the reference's sentinel (:265, :277)
// (The phase-ablation builds of the splat kernel -- wrong results by design -- live in tools/experiments/
// profiling_probes.patch, applied to a scratch copy of the sources by tools/build_variant.sh -p.)

struct AlignArgs {
    orbfe_intrinsics d, o;
    orbfe_extrinsics e;
    float scale;
    int mx, my;               // depth pixels the reference's grid reaches: min(grid, depth size)
    int rx, ry;               // output pixels it reaches: min(grid, other size)
    int tiles_x, tiles_y;     // ceil(mx / 64), ceil(my / 16)
    int n_frames, grid8;      // frames of this launch; 1: (8 * items, ceil(n / 8)) grid, blockIdx.x & 7 = frame in its row
    size_t in_stride, out_stride; // elements (u16 / u32) between frames
    // pipelined launches (whole, 16-byte aligned outputs): this launch also CLEARS the next chunk's frames (n_fill of them,
    // fill_value) and CLOSES the previous chunk's (n_close: 9999999 -> 0) -- two memory-bound passes carried by workgroups
    // interleaved with the VALU-bound splat workgroups instead of two launches of their own in between
    int n_slots;                  // max(n_frames, n_fill, n_close): frame slots of the grid
    int mem_items;                // memory work items per frame (kAlignMemQuads 16-byte quads each); 0: no memory roles
    int mem_total;                // memory items per frame slot: mem_items (fill only) or 2 * mem_items (fill, then close)
    int items_total;              // per frame slot: tiles + mem_total
    int n_fill, n_close;
    uint32_t fill_value;
    long long fill_off, close_off; // first frame of the next / previous chunk, in elements from `out`
    int quads;                    // 16-byte quads per output frame
};
constexpr int kAlignMemQuads = 1024; // 16 KB of output per memory work item

__device__ inline bool align_frame_item(const AlignArgs &A, int *frame, int *item)
{
    if (A.grid8) {
        *frame = blockIdx.y * 8 + (blockIdx.x & 7);
        *item = blockIdx.x >> 3;
        return *frame < A.n_slots;
    }
    *frame = blockIdx.y;
    *item = blockIdx.x;
    return true;
}

__device__ inline int cvt_rz_sat(float f)
{
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f)); // truncate, saturate, NaN -> 0 (cvt.rzi.s32.f32 of the reference)
    return r;
}

// x / z and y / z, correctly rounded, with ONE reciprocal.  hipcc expands an IEEE float division into v_div_scale x 2,
// v_rcp, 2 fma (the reciprocal's Newton step), v_mul, 3 fma, v_div_fmas, v_div_fixup: 12 instructions, 24 for the two
// quotients of a projection.  When v_div_scale leaves its operands alone the scaling flag is clear, v_div_fmas is a plain
// fma, v_div_fixup returns the quotient as it is, and the two reciprocals are the same number: the sequence below IS
// hipcc's instruction for instruction (same operations, same order, hence the same bits), 13 instead of 24.  v_div_scale
// is the identity when both operands are non-zero normals, |exponent difference| < 96, 1 / z and x / z are normals and x is
// not tiny (biased exponent > 23) -- all implied by 2^-40 <= |x|, |y|, |z| <= 2^40, the guard the caller tests (any lane
// of the wave outside it: the whole wave takes the plain `/`).  tests/test_gpu_round4.py compares both paths bit for bit.
__device__ inline void div2_in_range(float x, float y, float z, float *qx, float *qy)
{
    const float r0 = __builtin_amdgcn_rcpf(z);
    const float e = __builtin_fmaf(-z, r0, 1.0f);
    const float r = __builtin_fmaf(e, r0, r0);
    float m = x * r;
    float f = __builtin_fmaf(-z, m, x);
    m = __builtin_fmaf(f, r, m);
    f = __builtin_fmaf(-z, m, x);
    *qx = __builtin_fmaf(f, r, m);
    m = y * r;
    f = __builtin_fmaf(-z, m, y);
    m = __builtin_fmaf(f, r, m);
    f = __builtin_fmaf(-z, m, y);
    *qy = __builtin_fmaf(f, r, m);
}
__device__ inline bool div2_guard(float x, float y, float z)
{
    // v_min3 / v_max3 on the magnitudes; NaN fails both compares
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(x), __builtin_fabsf(y)), __builtin_fabsf(z));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x), __builtin_fabsf(y)), __builtin_fabsf(z));
    return lo >= 0x1p-40f && hi <= 0x1p40f;
}

// kernel_transfer_pixels (:121-161) for one corner whose normalised depth-image coordinates (before the inverse
// distortion) are (X, Y): deproject (:57-81), transform (:112-119), project (:23-54), round (:154-155)
template <bool DD>
__device__ inline void to_other_point(const AlignArgs &A, float depth_val, float X, float Y, const float (&rz)[3], float q[3])
{
    ORBFE_NO_CONTRACT
    float x = X, y = Y;
    if (DD) { // RS2_DISTORTION_INVERSE_BROWN_CONRADY on the depth camera
        const float *c = A.d.coeffs;
        const float r2 = x * x + y * y;
        float f = 1 + c[0] * r2;
        f = f + c[1] * r2 * r2;
        f = f + c[4] * r2 * r2 * r2;
        float ux = x * f + 2 * c[2] * x * y;
        ux = ux + c[3] * (r2 + 2 * x * x);
        float uy = y * f + 2 * c[3] * x * y;
        uy = uy + c[2] * (r2 + 2 * y * y);
        x = ux;
        y = uy;
    }
    const float p0 = depth_val * x, p1 = depth_val * y;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        // rz[i] = rotation[6 + i] * depth_val, the same product for both corners of the pixel: formed once by the caller
        float t = A.e.rotation[i] * p0 + A.e.rotation[3 + i] * p1;
        t = t + rz[i];
        t = t + A.e.translation[i];
        q[i] = t;
    }
}
// (x, y) = other_point.xy / other_point.z -> the rounded pixel
// DO = the other camera's rs2_distortion as far as the projection cares: 0 none (models 0, 2, 4), 1 modified
// Brown-Conrady, 3 f-theta
template <int DO>
__device__ inline void to_other_pixel(const AlignArgs &A, float x, float y, int *px, int *py)
{
    ORBFE_NO_CONTRACT
    if (DO == 3) orbfe_ftheta_distort(&x, &y, A.o.coeffs[0]); // RS2_DISTORTION_FTHETA, cuda-align.cu:44-50
    if (DO == 1) { // RS2_DISTORTION_MODIFIED_BROWN_CONRADY on the other camera
        const float *c = A.o.coeffs;
        const float r2 = x * x + y * y;
        float f = 1 + c[0] * r2;
        f = f + c[1] * r2 * r2;
        f = f + c[4] * r2 * r2 * r2;
        x *= f;
        y *= f;
        float dx = x + 2 * c[2] * x * y;
        dx = dx + c[3] * (r2 + 2 * x * x);
        float dy = y + 2 * c[3] * x * y;
        dy = dy + c[2] * (r2 + 2 * y * y);
        x = dx;
        y = dy;
    }
    const float ox = x * A.o.fx + A.o.ppx, oy = y * A.o.fy + A.o.ppy;
    *px = cvt_rz_sat(ox + 0.5f);
    *py = cvt_rz_sat(oy + 0.5f);
}

// one value into the output: literal form = the reference's atomicMin; zero-init form = minimum over the non-zero
template <bool ZERO_INIT>
__device__ inline void out_min(uint32_t *p, uint32_t v)
{
    if (ZERO_INIT) {
        const uint32_t old = atomicCAS(p, 0u, v);
        if (old != 0u && v < old) atomicMin(p, v);
    } else {
        atomicMin(p, v);
    }
}

// (x, y) pairs as two signed 16-bit lanes of a dword (image sizes are < 32768): component-wise minimum / maximum over the
// wave in six DPP steps each -- v_pk_min_i16 / v_pk_max_i16 with a DPP operand, no LDS crossbar -- result in every lane
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pk_min_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ inline uint32_t pk_max_i16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
#define ORBFE_WAVE_PK_REDUCE(name, op)                                                                         \
    __device__ inline uint32_t name(uint32_t v)                                                                \
    {                                                                                                          \
        /* lanes with no source keep their own value (old = v, bound_ctrl off): op(v, v) = v */               \
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xF, 0xF, false));             \
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xF, 0xF, false));             \
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xF, 0xF, false));             \
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xF, 0xF, false));             \
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));             \
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));             \
        return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);                                               \
    }
ORBFE_WAVE_PK_REDUCE(wave_pk_min_i16, pk_min_i16)
ORBFE_WAVE_PK_REDUCE(wave_pk_max_i16, pk_max_i16)
__device__ inline uint32_t pk_xy(int x, int y) { return ((uint32_t)x & 0xFFFFu) | ((uint32_t)y << 16); }
__device__ inline int pk_x(uint32_t v) { return (int)(short)(v & 0xFFFFu); }
__device__ inline int pk_y(uint32_t v) { return (int)v >> 16; }

// VEC: depth rows are 8-byte aligned at every multiple of 4 pixels (width % 4 == 0, aligned base and frame stride)
template <bool DD, int DO, bool ZERO_INIT, bool VEC>
__global__ void __launch_bounds__(256)
align_splat_kernel(uint32_t *__restrict__ out, const uint16_t *__restrict__ depth, AlignArgs A)
{
    ORBFE_NO_CONTRACT
    // the window (row pitch a multiple of 4 entries, left edge at a multiple of 4 output pixels: 16-byte quads), then
    // one dump entry per thread: the branch-free splat sends what a rectangle does not cover there
    __shared__ __attribute__((aligned(16))) uint32_t s_win[kAlignWin + 2 + 256];
    __shared__ float s_tx[kAlignTileW + 1], s_ty[kAlignTileH + 1];
    __shared__ uint32_t s_box[2 * 4]; // per wave: (x0, y0) minima, (x1, y1) maxima, packed

    int frame, item;
    if (!align_frame_item(A, &frame, &item)) return;
    const int tid = threadIdx.x;
    if (A.mem_items) {
        // this frame slot's items_total work items: memory items spread evenly among the tiles (item j is a memory item
        // iff the running count floor(j * M / T) steps at j), so that the dispatcher hands every CU a mix of both kinds
        const uint32_t jm = (uint32_t)item * (uint32_t)A.mem_total;
        const int before = (int)(jm / (uint32_t)A.items_total);
        const bool is_mem = jm - (uint32_t)before * (uint32_t)A.items_total + (uint32_t)A.mem_total >= (uint32_t)A.items_total;
        if (is_mem) { // uniform
            const bool closing = before >= A.mem_items;
            const int mi = closing ? before - A.mem_items : before;
            if (frame >= (closing ? A.n_close : A.n_fill)) return;
            uint4 *base = reinterpret_cast<uint4 *>(out + (closing ? A.close_off : A.fill_off) + (long long)frame * (long long)A.out_stride);
#pragma unroll
            for (int it = 0; it < kAlignMemQuads / 256; it++) {
                const int q = mi * kAlignMemQuads + it * 256 + tid;
                if (q >= A.quads) break;
                if (!closing) {
                    base[q] = make_uint4(A.fill_value, A.fill_value, A.fill_value, A.fill_value);
                } else { // kernel_reset_to_zero (:269-280)
                    uint4 v = base[q];
                    if (v.x == kAlignMax || v.y == kAlignMax || v.z == kAlignMax || v.w == kAlignMax) {
                        v.x = v.x == kAlignMax ? 0u : v.x;
                        v.y = v.y == kAlignMax ? 0u : v.y;
                        v.z = v.z == kAlignMax ? 0u : v.z;
                        v.w = v.w == kAlignMax ? 0u : v.w;
                        base[q] = v;
                    }
                }
            }
            return;
        }
        item -= before;
    }
    if (frame >= A.n_frames) return;
    const int ty_t = item / A.tiles_x, tx_t = item - ty_t * A.tiles_x;
    const int X0 = tx_t * kAlignTileW, Y0 = ty_t * kAlignTileH;
    depth += (size_t)frame * A.in_stride;
    out += (size_t)frame * A.out_stride;

    // (pixel -+ 0.5 - pp) / f per column and per row of the tile; entry i is corner -0.5 of pixel i = corner +0.5 of i - 1
    if (tid <= kAlignTileW) s_tx[tid] = (((float)(X0 + tid) + -0.5f) - A.d.ppx) / A.d.fx;
    else if (tid >= 128 && tid <= 128 + kAlignTileH) s_ty[tid - 128] = (((float)(Y0 + tid - 128) + -0.5f) - A.d.ppy) / A.d.fy;

    // this thread: 4 consecutive pixels of one row
    const int lx = (tid & 15) * 4, ly = tid >> 4;
    const int gx = X0 + lx, gy = Y0 + ly;
    uint32_t raw[4] = {0, 0, 0, 0};
    if (gy < A.my) {
        const uint16_t *row = depth + (size_t)gy * A.d.width + gx;
        if (VEC) {
            if (gx < A.mx) { // width % 4 == 0: all four or none (mx == width here, see the launcher)
                const uint2 v = *reinterpret_cast<const uint2 *>(row);
                raw[0] = v.x & 0xFFFFu;
                raw[1] = v.x >> 16;
                raw[2] = v.y & 0xFFFFu;
                raw[3] = v.y >> 16;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (gx + k < A.mx) raw[k] = row[k];
        }
    }
    __syncthreads();

    // rectangle of pixel k: corner p0 = (x, y) packed, extent (w, h) = p1 - p0; w < 0: nothing to write
    uint32_t p0[4];
    int rw[4], rh[4];
    uint32_t bmin = 0x7FFF7FFFu, bmax = 0x80008000u;
    const float Ym = s_ty[ly], Yp = s_ty[ly + 1];
    float qa[4][3], qb[4][3]; // other_point of the two corners
    bool in_range = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // depth_in[i] * depth_scale (:177): uint16 -> int -> float, one multiply
        const float depth_val = (float)(int)raw[k] * A.scale;
        const float rz[3] = {A.e.rotation[6] * depth_val, A.e.rotation[7] * depth_val, A.e.rotation[8] * depth_val};
        to_other_point<DD>(A, depth_val, s_tx[lx + k], Ym, rz, qa[k]);
        to_other_point<DD>(A, depth_val, s_tx[lx + k + 1], Yp, rz, qb[k]);
        // (a pixel without depth is skipped below whatever its quotients are: it does not hold the wave back)
        in_range &= depth_val == 0 || (div2_guard(qa[k][0], qa[k][1], qa[k][2]) && div2_guard(qb[k][0], qb[k][1], qb[k][2]));
    }
    const bool fast_div = __ballot(!in_range) == 0; // uniform
#pragma unroll
    for (int k = 0; k < 4; k++) {
        p0[k] = 0;
        rw[k] = -1;
        rh[k] = -1;
        int ax, ay, bx, by;
        float ua, va, ub, vb;
        if (fast_div) {
            div2_in_range(qa[k][0], qa[k][1], qa[k][2], &ua, &va);
            div2_in_range(qb[k][0], qb[k][1], qb[k][2], &ub, &vb);
        } else {
            ua = qa[k][0] / qa[k][2];
            va = qa[k][1] / qa[k][2];
            ub = qb[k][0] / qb[k][2];
            vb = qb[k][1] / qb[k][2];
        }
        to_other_pixel<DO>(A, ua, va, &ax, &ay);
        to_other_pixel<DO>(A, ub, vb, &bx, &by);
        // :140: no depth, nothing mapped; :241: skip unless the rectangle's corners are inside; an inverted rectangle
        // writes nothing
        if (raw[k] != 0 && (float)(int)raw[k] * A.scale != 0 && !(ax < 0 || ay < 0 || bx >= A.o.width || by >= A.o.height) &&
            ax <= bx && ay <= by) {
            p0[k] = pk_xy(ax, ay);
            rw[k] = bx - ax;
            rh[k] = by - ay;
            bmin = pk_min_i16(bmin, p0[k]);
            bmax = pk_max_i16(bmax, pk_xy(bx, by));
        }
    }
    // the tile's bounding box on the output
    bmin = wave_pk_min_i16(bmin);
    bmax = wave_pk_max_i16(bmax);
    if ((tid & 63) == 0) {
        s_box[2 * (tid >> 6)] = bmin;
        s_box[2 * (tid >> 6) + 1] = bmax;
    }
    __syncthreads();
    bmin = pk_min_i16(pk_min_i16(s_box[0], s_box[2]), pk_min_i16(s_box[4], s_box[6]));
    bmax = pk_max_i16(pk_max_i16(s_box[1], s_box[3]), pk_max_i16(s_box[5], s_box[7]));
    const int wx0 = pk_x(bmin) & ~3, wy0 = pk_y(bmin), wx1 = pk_x(bmax), wy1 = pk_y(bmax);
    if (wx1 < pk_x(bmin)) return; // no rectangle in this tile (uniform)
    const int bwp = (wx1 - wx0 + 4) & ~3, bh = wy1 - wy0 + 1; // window: bh rows of bwp entries
    const int nq = bwp >> 2;                                    // quads per row
    const bool fits = bwp * bh <= kAlignWin;                    // (both < 32768 / bounded by the image: no overflow)

    if (fits) {
        const int nquads = nq * bh;
        for (int i = tid; i < nquads; i += 256) reinterpret_cast<uint4 *>(s_win)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
        __syncthreads();
        // byte addresses inside s_win; entry (dx, dy) of the 3 x 3 block at p0 lands at row_dy + 4 dx.  What the rectangle
        // does not cover goes to the thread's dump word instead -- the select is on the ROW address and the + 4 dx rides in
        // the instruction's offset field (the dump address is pre-biased by - 4 dx; two words of padding keep it inside
        // the dump area): one v_cndmask per entry, no branch, no EXEC change, no per-entry shift or add
        unsigned char *wb = reinterpret_cast<unsigned char *>(s_win);
        const uint32_t dump4 = (uint32_t)(kAlignWin + 2 + tid) * 4u, bwp4 = (uint32_t)bwp * 4u;
        int ext = -1; // largest extent of this thread's rectangles
#pragma unroll
        for (int k = 0; k < 4; k++) ext = max(ext, max(rw[k], rh[k]));
        const bool big = ext > 2;
        const bool any3 = __ballot(ext >= 2) != 0; // uniform: someone in the wave is 3 wide or high
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t row0 = (uint32_t)((pk_y(p0[k]) - wy0) * bwp + (pk_x(p0[k]) - wx0)) * 4u, row1 = row0 + bwp4;
            const int w = rw[k], h = rh[k];
            const uint32_t v = raw[k];
            auto put = [&](bool covered, uint32_t row, int dx) {
                atomicMin(reinterpret_cast<uint32_t *>(wb + (covered ? row : dump4 - 4u * (uint32_t)dx)) + dx, v);
            };
            put(w >= 0, row0, 0); // (w >= 0 implies h >= 0)
            put(w >= 1, row0, 1);
            put(w >= 0 && h >= 1, row1, 0);
            put(w >= 1 && h >= 1, row1, 1);
            if (any3) {
                const uint32_t row2 = row1 + bwp4;
                put(w >= 2, row0, 2);
                put(w >= 2 && h >= 1, row1, 2);
                put(w >= 0 && h >= 2, row2, 0);
                put(w >= 1 && h >= 2, row2, 1);
                put(w >= 2 && h >= 2, row2, 2);
            }
        }
        if (big) { // rectangles beyond 3 x 3 (output much finer than the depth image): the part the block above left out
#pragma unroll
            for (int k = 0; k < 4; k++)
                for (int dy = 0; dy <= rh[k]; dy++)
                    for (int dx = dy > 2 ? 0 : 3; dx <= rw[k]; dx++)
                        atomicMin(&s_win[(pk_y(p0[k]) - wy0 + dy) * bwp + (pk_x(p0[k]) - wx0) + dx], raw[k]);
        }
        __syncthreads();
        // a wave per window row, CONSECUTIVE lanes on consecutive entries: the touched entries of a row leave as one or
        // two fully used cache lines per atomic instruction (a lane per quad -- four strided atomics -- visits every line
        // four times: measured 2x slower, the L2's atomic rate is per line visit)
        // (the row is wave-uniform: scalar loop control, a scalar row base for the atomics' addresses; two entries per lane
        // and trip are read before the first atomic goes out; a read past the row's end lands in the next row or in the
        // dump area -- valid LDS, dropped by the column test)
        const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        for (int r = wv; r < bh; r += 4) {
            uint32_t *orow = out + (size_t)(wy0 + r) * A.o.width + wx0;
            const uint32_t *wrow = s_win + r * bwp;
            // lane 0 sits on a 128-byte line of the output: every atomic instruction then visits exactly two lines (the
            // L2's atomic rate is per line visit; an unaligned 64-lane run touches three)
            const int mis = (int)((reinterpret_cast<uintptr_t>(orow) >> 2) & 31u);
            for (int c0 = -mis; c0 < bwp; c0 += 128) {
                const int ca = c0 + lane, cb = ca + 64;
                const uint32_t va = wrow[ca < 0 ? 0 : ca], vb = wrow[cb];
                if (ca >= 0 && ca < bwp && va != ~0u) out_min<ZERO_INIT>(orow + ca, va);
                if (cb < bwp && vb != ~0u) out_min<ZERO_INIT>(orow + cb, vb);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
            for (int dy = 0; dy <= rh[k]; dy++)
                for (int dx = 0; dx <= rw[k]; dx++)
                    out_min<ZERO_INIT>(out + (size_t)(pk_y(p0[k]) + dy) * A.o.width + pk_x(p0[k]) + dx, raw[k]);
    }
}

// kernel_reset_to_max (:257-267) / the zero-init form's clear: the grid's part of the output := value
__global__ void __launch_bounds__(256) align_fill_kernel(uint32_t *__restrict__ out, uint32_t value, AlignArgs A)
{
    const int frame = blockIdx.z;
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= A.ry || x >= A.rx) return;
    uint32_t *p = out + (size_t)frame * A.out_stride + (size_t)y * A.o.width + x;
    if (x + 3 < A.rx && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
        *reinterpret_cast<uint4 *>(p) = make_uint4(value, value, value, value);
    } else {
        for (int k = 0; k < 4 && x + k < A.rx; k++) p[k] = value;
    }
}

// kernel_reset_to_zero (:269-280): 9999999 -> 0 on the same part
__global__ void __launch_bounds__(256) align_unmax_kernel(uint32_t *__restrict__ out, AlignArgs A)
{
    const int frame = blockIdx.z;
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= A.ry || x >= A.rx) return;
    uint32_t *p = out + (size_t)frame * A.out_stride + (size_t)y * A.o.width + x;
    if (x + 3 < A.rx && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
        uint4 v = *reinterpret_cast<const uint4 *>(p);
        if (v.x == kAlignMax || v.y == kAlignMax || v.z == kAlignMax || v.w == kAlignMax) {
            v.x = v.x == kAlignMax ? 0u : v.x;
            v.y = v.y == kAlignMax ? 0u : v.y;
            v.z = v.z == kAlignMax ? 0u : v.z;
            v.w = v.w == kAlignMax ? 0u : v.w;
            *reinterpret_cast<uint4 *>(p) = v;
        }
    } else {
        for (int k = 0; k < 4 && x + k < A.rx; k++)
            if (p[k] == kAlignMax) p[k] = 0u;
    }
}

} // namespace orbfe

using namespace orbfe;

static inline hipStream_t S(orbfe_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define ARG_CHECK(cond)                                                                     \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            set_thread_error("%s: invalid argument: %s", __func__, #cond);                  \
            return ORBFE_ERR_INVALID_ARG;                                                   \
        }                                                                                   \
    } while (0)

template <bool ZI, bool VEC>
static void launch_splat(uint32_t *out, const uint16_t *depth, const AlignArgs &A, dim3 grid, hipStream_t s)
{
    const bool dd = A.d.model == 2;
    const int dO = A.o.model == 1 ? 1 : A.o.model == 3 ? 3 : 0;
    if (dd && dO == 1) hipLaunchKernelGGL((align_splat_kernel<true, 1, ZI, VEC>), grid, dim3(256), 0, s, out, depth, A);
    else if (dd && dO == 3) hipLaunchKernelGGL((align_splat_kernel<true, 3, ZI, VEC>), grid, dim3(256), 0, s, out, depth, A);
    else if (dd) hipLaunchKernelGGL((align_splat_kernel<true, 0, ZI, VEC>), grid, dim3(256), 0, s, out, depth, A);
    else if (dO == 1) hipLaunchKernelGGL((align_splat_kernel<false, 1, ZI, VEC>), grid, dim3(256), 0, s, out, depth, A);
    else if (dO == 3) hipLaunchKernelGGL((align_splat_kernel<false, 3, ZI, VEC>), grid, dim3(256), 0, s, out, depth, A);
    else hipLaunchKernelGGL((align_splat_kernel<false, 0, ZI, VEC>), grid, dim3(256), 0, s, out, depth, A);
}

// frames_per_launch: how many frames go through clear -> splat (-> close) together; the output of a chunk is cleared
// and then hit by the splat's atomics, so a chunk that fits the 256 MB Infinity Cache pays HBM once per output byte
static int align_frames(uint32_t *d_out, size_t out_stride, const uint16_t *d_depth, size_t in_stride, int n_frames,
                        float depth_scale, int image_width, int image_height, const orbfe_intrinsics *din,
                        const orbfe_intrinsics *oin, const orbfe_extrinsics *ext, int frames_per_launch, bool prefer_zero_init,
                        hipStream_t stream, const char *what)
{
    if (din->model == 1 || din->model == 3) {
        set_thread_error("%s: cannot deproject a forward-distorted depth image (model %d; the reference asserts, "
                         "cuda-align.cu:62-63)", what, din->model);
        return ORBFE_ERR_UNSUPPORTED;
    }
    AlignArgs A;
    memset(&A, 0, sizeof(A));
    A.d = *din;
    A.o = *oin;
    A.e = *ext;
    A.scale = depth_scale;
    const long long gx = 32ll * ((image_width + 31) / 32), gy = 32ll * ((image_height + 31) / 32);
    A.mx = (int)(gx < din->width ? gx : din->width);
    A.my = (int)(gy < din->height ? gy : din->height);
    A.rx = (int)(gx < oin->width ? gx : oin->width);
    A.ry = (int)(gy < oin->height ? gy : oin->height);
    A.tiles_x = (A.mx + kAlignTileW - 1) / kAlignTileW;
    A.tiles_y = (A.my + kAlignTileH - 1) / kAlignTileH;
    A.in_stride = in_stride;
    A.out_stride = out_stride;
    const bool whole = A.rx == oin->width && A.ry == oin->height;
    bool zero_init = whole && prefer_zero_init;
    if (const char *v = getenv("ORBFE_ALIGN_PROTOCOL")) { // A/B timing and the tests: "zero" | "literal"
        if (!strcmp(v, "literal")) zero_init = false;
        else if (!strcmp(v, "zero")) zero_init = whole;
    }
    const bool vec = din->width % 4 == 0 && A.mx == din->width && (reinterpret_cast<uintptr_t>(d_depth) & 7u) == 0 &&
                     in_stride % 4 == 0;
    const int items = A.tiles_x * A.tiles_y;
    // pipelined form: the clear of chunk c + 1 and the close of chunk c - 1 ride inside chunk c's splat launch as
    // memory-role workgroups (align_splat_kernel); needs whole frames that are arrays of 16-byte quads
    const long long px = (long long)oin->width * oin->height;
    // (the kernel finds a work item's role with 32-bit arithmetic: item * memory items per slot must stay below 2^32 --
    // frames beyond ~8000 x 8000 take the unpipelined launches)
    const unsigned long long mem_items_ll = (unsigned long long)((px / 4 + kAlignMemQuads - 1) / kAlignMemQuads);
    const bool piped = whole && px % 4 == 0 && out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15u) == 0 &&
                       n_frames > frames_per_launch && !getenv("ORBFE_ALIGN_NO_PIPE") &&
                       ((unsigned long long)items + 2 * mem_items_ll) * (2 * mem_items_ll) < 0xFFFFFFFFull;
    const uint32_t fill_value = zero_init ? 0u : kAlignMax;
    const int n_chunks = (n_frames + frames_per_launch - 1) / frames_per_launch;
    for (int c = 0; c < n_chunks; c++) {
        const int f0 = c * frames_per_launch;
        const int n = n_frames - f0 < frames_per_launch ? n_frames - f0 : frames_per_launch;
        uint32_t *o = d_out + (size_t)f0 * out_stride;
        const uint16_t *i = d_depth + (size_t)f0 * in_stride;
        A.n_frames = n;
        A.n_slots = n;
        A.mem_items = A.mem_total = A.n_fill = A.n_close = 0;
        A.items_total = items;
        const dim3 fgrid((A.rx + 255) / 256, (A.ry + 3) / 4, n);
        if (!piped || c == 0) hipLaunchKernelGGL(align_fill_kernel, fgrid, dim3(256), 0, stream, o, fill_value, A);
        if (piped) {
            A.quads = (int)(px / 4);
            A.mem_items = (A.quads + kAlignMemQuads - 1) / kAlignMemQuads;
            A.mem_total = zero_init ? A.mem_items : 2 * A.mem_items;
            A.items_total = items + A.mem_total;
            A.fill_value = fill_value;
            const int left = n_frames - (f0 + n);
            A.n_fill = left < frames_per_launch ? left : frames_per_launch; // the next chunk (0: none)
            A.n_close = (!zero_init && c > 0) ? frames_per_launch : 0;      // the previous chunk is always a full one
            A.fill_off = (long long)n * (long long)out_stride;
            A.close_off = -(long long)frames_per_launch * (long long)out_stride;
            A.n_slots = n > A.n_fill ? n : A.n_fill;
            A.n_slots = A.n_slots > A.n_close ? A.n_slots : A.n_close;
        }
        A.grid8 = A.n_slots >= 8;
        const dim3 grid = A.grid8 ? dim3(8 * A.items_total, (A.n_slots + 7) / 8) : dim3(A.items_total, A.n_slots);
        if (zero_init) {
            if (vec) launch_splat<true, true>(o, i, A, grid, stream);
            else launch_splat<true, false>(o, i, A, grid, stream);
        } else {
            if (vec) launch_splat<false, true>(o, i, A, grid, stream);
            else launch_splat<false, false>(o, i, A, grid, stream);
            if (!piped || c == n_chunks - 1) hipLaunchKernelGGL(align_unmax_kernel, fgrid, dim3(256), 0, stream, o, A);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_thread_error("%s launch failed: %s", what, hipGetErrorString(e));
        return ORBFE_ERR_HIP;
    }
    return ORBFE_OK;
}

static int align_env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    const int n = atoi(v);
    return n > 0 ? n : dflt;
}

extern "C" {

int orbfe_align_depth_to_other(uint32_t *d_aligned_out, const uint16_t *d_depth_in, void *d_pixel_map, float depth_scale,
                               int image_width, int image_height, const orbfe_intrinsics *depth_intrin,
                               const orbfe_intrinsics *other_intrin, const orbfe_extrinsics *depth_to_other,
                               orbfe_stream_t stream)
{
    (void)d_pixel_map; // the reference's int2 scratch (:382-396): never read by anyone else, not touched here
    ARG_CHECK(d_aligned_out && d_depth_in && depth_intrin && other_intrin && depth_to_other);
    // the three camera structs: host pointers, or the reference's device copies (buildStream.cpp:391-393)
    orbfe_intrinsics di_tmp, oi_tmp;
    orbfe_extrinsics ex_tmp;
    depth_intrin = host_view(depth_intrin, &di_tmp);
    other_intrin = host_view(other_intrin, &oi_tmp);
    depth_to_other = host_view(depth_to_other, &ex_tmp);
    ARG_CHECK(depth_intrin && other_intrin && depth_to_other); // a device struct that could not be copied back
    ARG_CHECK(image_width > 0 && image_height > 0 && depth_intrin->width > 0 && depth_intrin->height > 0 &&
              other_intrin->width > 0 && other_intrin->height > 0);
    ARG_CHECK(other_intrin->width <= 32767 && other_intrin->height <= 32767); // output coordinates travel as int16 pairs
    ARG_CHECK(depth_scale == depth_scale && depth_scale - depth_scale == 0.0f); // finite
    return align_frames(d_aligned_out, 0, d_depth_in, 0, 1, depth_scale, image_width, image_height, depth_intrin,
                        other_intrin, depth_to_other, 1, /* literal protocol: 11.4 us per 848x480 frame against 16.1 for the 2-launch
                        zero-init form (tools/stage_latency.py) */ false, S(stream), "align_depth_to_other");
}

int orbfe_align_depth_batch(uint32_t *d_aligned_out, size_t out_frame_stride, const uint16_t *d_depth_in,
                            size_t in_frame_stride, int n_frames, float depth_scale,
                            const orbfe_intrinsics *depth_intrin, const orbfe_intrinsics *other_intrin,
                            const orbfe_extrinsics *depth_to_other, orbfe_stream_t stream)
{
    ARG_CHECK(n_frames >= 0 && depth_intrin && other_intrin && depth_to_other);
    if (n_frames == 0) return ORBFE_OK;
    orbfe_intrinsics di_tmp, oi_tmp;
    orbfe_extrinsics ex_tmp;
    depth_intrin = host_view(depth_intrin, &di_tmp);
    other_intrin = host_view(other_intrin, &oi_tmp);
    depth_to_other = host_view(depth_to_other, &ex_tmp);
    ARG_CHECK(depth_intrin && other_intrin && depth_to_other);
    ARG_CHECK(d_aligned_out && d_depth_in);
    ARG_CHECK(depth_intrin->width > 0 && depth_intrin->height > 0 && other_intrin->width > 0 && other_intrin->height > 0);
    ARG_CHECK(other_intrin->width <= 32767 && other_intrin->height <= 32767);
    ARG_CHECK(in_frame_stride >= (size_t)depth_intrin->width * depth_intrin->height);
    ARG_CHECK(out_frame_stride >= (size_t)other_intrin->width * other_intrin->height);
    ARG_CHECK(depth_scale == depth_scale && depth_scale - depth_scale == 0.0f);
    // the launch grid covers both images: image_width / height of the per-frame call = the larger of the two sizes
    const int w = depth_intrin->width > other_intrin->width ? depth_intrin->width : other_intrin->width;
    const int h = depth_intrin->height > other_intrin->height ? depth_intrin->height : other_intrin->height;
    return align_frames(d_aligned_out, out_frame_stride, d_depth_in, in_frame_stride, n_frames, depth_scale, w, h,
                        depth_intrin, other_intrin, depth_to_other, align_env_int("ORBFE_ALIGN_CHUNK", 128),
                        /* many frames: the literal protocol's no-return atomicMin beats the compare-and-swap (measured) */ false,
                        S(stream), "align_depth_batch");
}

} // extern "C"
