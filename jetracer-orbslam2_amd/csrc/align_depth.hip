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
//     pixels whose rectangle covers the output pixel, 0 where none does.
// Two exact forms of the output protocol:
//   zero-init form (whole output inside the launch grid, the normal case): output cleared to 0, "0 = nothing yet", a
//     window entry v lands by  old = CAS(p, 0, v); if (old != 0 && v < old) atomicMin(p, v)  -- raw depths that reach a
//     splat are >= 1 (depth 0 is skipped, :140), so 0 is free to mean "empty" and no closing pass is needed: 2 launches;
//   literal form: reset the grid's part of the output to 9999999, atomicMin, 9999999 -> 0 on the same part: 3 launches;
//     needed only when the intrinsics' sizes exceed the grid made from image_width / image_height (:378-380), where the
//     reference leaves what the caller's buffer held (min'ed with any splat) -- reproduced.
// Arithmetic: float, left to right as written in the reference, no contraction (ORBFE_NO_CONTRACT; the build has
// -ffp-contract=off), IEEE division; static_cast<int>(v + 0.5f) is v_cvt_i32_f32 = truncation, saturating, NaN -> 0,
// exactly CUDA's cvt.rzi.s32.f32.  Parity with the reference is unpinned at the ulp level (nvcc may fuse a*b+c).
#include "orbfe_internal.hpp"
#include "device_common.hpp"

namespace orbfe {

constexpr int kAlignTileW = 64, kAlignTileH = 16; // depth pixels per workgroup: 256 threads x 4 pixels of one row
constexpr int kAlignWin = 6144;                    // LDS window, u32 entries (24 KB: six workgroups per CU)
constexpr uint32_t kAlignMax = 9999999u;           // the reference's sentinel (:265, :277)

struct AlignArgs {
    orbfe_intrinsics d, o;
    orbfe_extrinsics e;
    float scale;
    int mx, my;               // depth pixels the reference's grid reaches: min(grid, depth size)
    int rx, ry;               // output pixels it reaches: min(grid, other size)
    int tiles_x, tiles_y;     // ceil(mx / 64), ceil(my / 16)
    int n_frames, grid8;      // frames of this launch; 1: (8 * items, ceil(n / 8)) grid, blockIdx.x & 7 = frame in its row
    size_t in_stride, out_stride; // elements (u16 / u32) between frames
};

__device__ inline bool align_frame_item(const AlignArgs &A, int *frame, int *item)
{
    if (A.grid8) {
        *frame = blockIdx.y * 8 + (blockIdx.x & 7);
        *item = blockIdx.x >> 3;
        return *frame < A.n_frames;
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

// kernel_transfer_pixels (:121-161) for one corner whose normalised depth-image coordinates (before the inverse
// distortion) are (X, Y): deproject (:57-81), transform (:112-119), project (:23-54), round (:154-155)
template <bool DD, bool DO>
__device__ inline void map_corner(const AlignArgs &A, float depth_val, float X, float Y, int *px, int *py)
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
    const float p0 = depth_val * x, p1 = depth_val * y, p2 = depth_val;
    float q[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float t = A.e.rotation[i] * p0 + A.e.rotation[3 + i] * p1;
        t = t + A.e.rotation[6 + i] * p2;
        t = t + A.e.translation[i];
        q[i] = t;
    }
    x = q[0] / q[2];
    y = q[1] / q[2];
    if (DO) { // RS2_DISTORTION_MODIFIED_BROWN_CONRADY on the other camera
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

__device__ inline int wave_min_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}
__device__ inline int wave_max_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}

// VEC: depth rows are 8-byte aligned at every multiple of 4 pixels (width % 4 == 0, aligned base and frame stride)
template <bool DD, bool DO, bool ZERO_INIT, bool VEC>
__global__ void __launch_bounds__(256)
align_splat_kernel(uint32_t *__restrict__ out, const uint16_t *__restrict__ depth, AlignArgs A)
{
    ORBFE_NO_CONTRACT
    __shared__ uint32_t s_win[kAlignWin];
    __shared__ float s_tx[kAlignTileW + 1], s_ty[kAlignTileH + 1];
    __shared__ int s_box[4]; // x0, y0 (minima), x1, y1 (maxima)

    int frame, item;
    if (!align_frame_item(A, &frame, &item)) return;
    const int ty_t = item / A.tiles_x, tx_t = item - ty_t * A.tiles_x;
    const int X0 = tx_t * kAlignTileW, Y0 = ty_t * kAlignTileH;
    const int tid = threadIdx.x;
    depth += (size_t)frame * A.in_stride;
    out += (size_t)frame * A.out_stride;

    // (pixel -+ 0.5 - pp) / f per column and per row of the tile; entry i is corner -0.5 of pixel i = corner +0.5 of i - 1
    if (tid <= kAlignTileW) s_tx[tid] = (((float)(X0 + tid) + -0.5f) - A.d.ppx) / A.d.fx;
    else if (tid >= 128 && tid <= 128 + kAlignTileH) s_ty[tid - 128] = (((float)(Y0 + tid - 128) + -0.5f) - A.d.ppy) / A.d.fy;
    if (tid == 0) {
        s_box[0] = 0x7FFFFFFF;
        s_box[1] = 0x7FFFFFFF;
        s_box[2] = -0x7FFFFFFF;
        s_box[3] = -0x7FFFFFFF;
    }

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

    int p0x[4], p0y[4], p1x[4], p1y[4];
    int bx0 = 0x7FFFFFFF, by0 = 0x7FFFFFFF, bx1 = -0x7FFFFFFF, by1 = -0x7FFFFFFF;
    const float Ym = s_ty[ly], Yp = s_ty[ly + 1];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // depth_in[i] * depth_scale (:177): uint16 -> int -> float, one multiply
        const float depth_val = (float)(int)raw[k] * A.scale;
        p0x[k] = 0;
        p1x[k] = -1; // empty rectangle
        p0y[k] = 0;
        p1y[k] = -1;
        if (depth_val != 0) {
            int ax, ay, bx, by;
            map_corner<DD, DO>(A, depth_val, s_tx[lx + k], Ym, &ax, &ay);
            map_corner<DD, DO>(A, depth_val, s_tx[lx + k + 1], Yp, &bx, &by);
            // :241: skip unless the rectangle's corners are inside; an inverted rectangle writes nothing
            if (!(ax < 0 || ay < 0 || bx >= A.o.width || by >= A.o.height) && ax <= bx && ay <= by) {
                p0x[k] = ax;
                p0y[k] = ay;
                p1x[k] = bx;
                p1y[k] = by;
                bx0 = ax < bx0 ? ax : bx0;
                by0 = ay < by0 ? ay : by0;
                bx1 = bx > bx1 ? bx : bx1;
                by1 = by > by1 ? by : by1;
            }
        }
    }
    // the tile's bounding box on the output
    bx0 = wave_min_i32(bx0);
    by0 = wave_min_i32(by0);
    bx1 = wave_max_i32(bx1);
    by1 = wave_max_i32(by1);
    if ((tid & 63) == 0 && bx1 >= bx0) {
        atomicMin(&s_box[0], bx0);
        atomicMin(&s_box[1], by0);
        atomicMax(&s_box[2], bx1);
        atomicMax(&s_box[3], by1);
    }
    __syncthreads();
    const int wx0 = s_box[0], wy0 = s_box[1], wx1 = s_box[2], wy1 = s_box[3];
    if (wx1 < wx0) return; // nothing to write (uniform)
    const int bw = wx1 - wx0 + 1, bh = wy1 - wy0 + 1;
    const bool fits = (long long)bw * bh <= kAlignWin;

    if (fits) {
        const int n = bw * bh;
        for (int i = tid; i < n; i += 256) s_win[i] = 0xFFFFFFFFu;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++)
            for (int v = p0y[k]; v <= p1y[k]; v++)
                for (int u = p0x[k]; u <= p1x[k]; u++) atomicMin(&s_win[(v - wy0) * bw + (u - wx0)], raw[k]);
        __syncthreads();
        // a wave per window row, lanes along it: the touched entries of a row leave as neighbouring atomics
        for (int r = tid >> 6; r < bh; r += 4) {
            uint32_t *orow = out + (size_t)(wy0 + r) * A.o.width + wx0;
            for (int c = tid & 63; c < bw; c += 64) {
                const uint32_t v = s_win[r * bw + c];
                if (v != 0xFFFFFFFFu) out_min<ZERO_INIT>(orow + c, v);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
            for (int v = p0y[k]; v <= p1y[k]; v++)
                for (int u = p0x[k]; u <= p1x[k]; u++) out_min<ZERO_INIT>(out + (size_t)v * A.o.width + u, raw[k]);
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
    const bool dd = A.d.model == 2, dO = A.o.model == 1;
    if (dd && dO) hipLaunchKernelGGL((align_splat_kernel<true, true, ZI, VEC>), grid, dim3(256), 0, s, out, depth, A);
    else if (dd) hipLaunchKernelGGL((align_splat_kernel<true, false, ZI, VEC>), grid, dim3(256), 0, s, out, depth, A);
    else if (dO) hipLaunchKernelGGL((align_splat_kernel<false, true, ZI, VEC>), grid, dim3(256), 0, s, out, depth, A);
    else hipLaunchKernelGGL((align_splat_kernel<false, false, ZI, VEC>), grid, dim3(256), 0, s, out, depth, A);
}

// frames_per_launch: how many frames go through clear -> splat (-> close) together; the output of a chunk is cleared
// and then hit by the splat's atomics, so a chunk that fits the 256 MB Infinity Cache pays HBM once per output byte
static int align_frames(uint32_t *d_out, size_t out_stride, const uint16_t *d_depth, size_t in_stride, int n_frames,
                        float depth_scale, int image_width, int image_height, const orbfe_intrinsics *din,
                        const orbfe_intrinsics *oin, const orbfe_extrinsics *ext, int frames_per_launch, int force_literal,
                        hipStream_t stream, const char *what)
{
    if (din->model == 1 || din->model == 3) {
        set_thread_error("%s: cannot deproject a forward-distorted depth image (model %d; the reference asserts, "
                         "cuda-align.cu:62-63)", what, din->model);
        return ORBFE_ERR_UNSUPPORTED;
    }
    if (oin->model == 3) {
        set_thread_error("%s: f-theta projection (cuda-align.cu:44-50) needs libdevice's double atan / tan: not reproducible",
                         what);
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
    const bool zero_init = whole && !force_literal;
    const bool vec = din->width % 4 == 0 && A.mx == din->width && (reinterpret_cast<uintptr_t>(d_depth) & 7u) == 0 &&
                     in_stride % 4 == 0;
    const int items = A.tiles_x * A.tiles_y;
    for (int f0 = 0; f0 < n_frames; f0 += frames_per_launch) {
        const int n = n_frames - f0 < frames_per_launch ? n_frames - f0 : frames_per_launch;
        uint32_t *o = d_out + (size_t)f0 * out_stride;
        const uint16_t *i = d_depth + (size_t)f0 * in_stride;
        A.n_frames = n;
        A.grid8 = n >= 8;
        const dim3 fgrid((A.rx + 255) / 256, (A.ry + 3) / 4, n);
        hipLaunchKernelGGL(align_fill_kernel, fgrid, dim3(256), 0, stream, o, zero_init ? 0u : kAlignMax, A);
        const dim3 grid = A.grid8 ? dim3(8 * items, (n + 7) / 8) : dim3(items, n);
        if (zero_init) {
            if (vec) launch_splat<true, true>(o, i, A, grid, stream);
            else launch_splat<true, false>(o, i, A, grid, stream);
        } else {
            if (vec) launch_splat<false, true>(o, i, A, grid, stream);
            else launch_splat<false, false>(o, i, A, grid, stream);
            hipLaunchKernelGGL(align_unmax_kernel, fgrid, dim3(256), 0, stream, o, A);
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
    ARG_CHECK(image_width > 0 && image_height > 0 && depth_intrin->width > 0 && depth_intrin->height > 0 &&
              other_intrin->width > 0 && other_intrin->height > 0);
    ARG_CHECK(depth_scale == depth_scale && depth_scale - depth_scale == 0.0f); // finite
    return align_frames(d_aligned_out, 0, d_depth_in, 0, 1, depth_scale, image_width, image_height, depth_intrin,
                        other_intrin, depth_to_other, 1, getenv("ORBFE_ALIGN_LITERAL") != nullptr, S(stream),
                        "align_depth_to_other");
}

int orbfe_align_depth_batch(uint32_t *d_aligned_out, size_t out_frame_stride, const uint16_t *d_depth_in,
                            size_t in_frame_stride, int n_frames, float depth_scale,
                            const orbfe_intrinsics *depth_intrin, const orbfe_intrinsics *other_intrin,
                            const orbfe_extrinsics *depth_to_other, orbfe_stream_t stream)
{
    ARG_CHECK(n_frames >= 0 && depth_intrin && other_intrin && depth_to_other);
    if (n_frames == 0) return ORBFE_OK;
    ARG_CHECK(d_aligned_out && d_depth_in);
    ARG_CHECK(depth_intrin->width > 0 && depth_intrin->height > 0 && other_intrin->width > 0 && other_intrin->height > 0);
    ARG_CHECK(in_frame_stride >= (size_t)depth_intrin->width * depth_intrin->height);
    ARG_CHECK(out_frame_stride >= (size_t)other_intrin->width * other_intrin->height);
    ARG_CHECK(depth_scale == depth_scale && depth_scale - depth_scale == 0.0f);
    // the launch grid covers both images: image_width / height of the per-frame call = the larger of the two sizes
    const int w = depth_intrin->width > other_intrin->width ? depth_intrin->width : other_intrin->width;
    const int h = depth_intrin->height > other_intrin->height ? depth_intrin->height : other_intrin->height;
    return align_frames(d_aligned_out, out_frame_stride, d_depth_in, in_frame_stride, n_frames, depth_scale, w, h,
                        depth_intrin, other_intrin, depth_to_other, align_env_int("ORBFE_ALIGN_CHUNK", 64),
                        getenv("ORBFE_ALIGN_LITERAL") != nullptr, S(stream), "align_depth_batch");
}

} // extern "C"
