// stage_kernels.hip -- the per-frame "stage" entry points of include/orbfe.h: drop-in
// replacements for the free functions of the reference's src/cuda/*.cuh, with the
// reference's buffer contracts (caller-owned pitched device buffers, any pitch).
// gfx950 only; wave = 64 lanes.  Semantics follow SURVEY.md Appendix C; quirk numbers Qn
// refer to its Appendix A.  These are the correctness-first forms; the throughput path is
// batch_kernels.hip.
#include "orbfe_internal.hpp"
#include "device_common.hpp"

namespace orbfe {

static thread_local char t_err[512];

void set_thread_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
}
const char *thread_error() { return t_err; }

void format_error(char *ctx_err, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx_err ? ctx_err : t_err, 512, fmt, ap);
    va_end(ap);
}

// f1  RGB8 -> gray (cuda_RGB_to_Grayscale.cu:10-23), one thread per pixel
__global__ void rgb_to_gray_px_kernel(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int cols,
                                      int rows, int dst_pitch, int src_pitch)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows) return;
    const uint8_t *p = src + (size_t)y * src_pitch + 3 * x;
    dst[(size_t)y * dst_pitch + x] = (uint8_t)rgb_to_gray1(p[0], p[1], p[2]);
}

// ------------------------------------------------------------------------------------
// a2  3x3 Gaussian with the reference's 32-column seams (gaussian_blur_3x3.cu:15-53)
// One thread = one output pixel; rows 0, h-2, h-1 are written as 0 (Q1).
// ------------------------------------------------------------------------------------
__global__ void blur3x3_px_kernel(uint8_t *__restrict__ dst, int dst_pitch,
                                  const uint8_t *__restrict__ src, int src_pitch, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    uint8_t *o = dst + (size_t)y * dst_pitch + x;
    if (y == 0 || y >= h - 2) {
        *o = 0;
        return;
    }
    const int xl = ((x & 31) == 0) ? x : x - 1;
    const int xr = ((x & 31) == 31 || x == w - 1) ? x : x + 1;
    const uint8_t *ra = src + (size_t)(y - 1) * src_pitch;
    const uint8_t *rb = ra + src_pitch;
    const uint8_t *rc = rb + src_pitch;
    const int s = 2 * ra[x] + 4 * rb[x] + 2 * rc[x] + ra[xl] + ra[xr] + 2 * rb[xl] + 2 * rb[xr] +
                  rc[xl] + rc[xr];
    *o = (uint8_t)((s + 8) >> 4); // == floor(s / 16 + 0.5), s >= 0
}

// a3  2x2 box halving (pyramid.cu:6-29)
__global__ void halfsample_px_kernel(const uint8_t *__restrict__ src, int src_pitch,
                                     uint8_t *__restrict__ dst, int dst_pitch, int dw, int dh)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const uint8_t *t = src + (size_t)(2 * y) * src_pitch + 2 * x;
    const uint8_t *b = t + src_pitch;
    dst[(size_t)y * dst_pitch + x] = (uint8_t)(((unsigned)t[0] + t[1] + b[0] + b[1]) >> 2);
}

// a4  corner LUT (fast.cu:11-39): cyclic run of >= min_arc ones in the 16-bit mask
__global__ void fast_lut_kernel(uint8_t *__restrict__ lut, int min_arc)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= 65536u) return;
    const uint32_t dup = m | (m << 16);
    const uint32_t want = min_arc >= 32 ? 0xFFFFFFFFu : ((1u << min_arc) - 1u);
    int hit = 0;
    if (min_arc <= 16)
        for (int s = 0; s < 16; s++) hit |= (((dup >> s) & want) == want);
    lut[m] = (uint8_t)hit;
}

// a5  corner response, float arithmetic as in fast.cu:150-287 (any float threshold)
// SCORE = the reference's enum fast_score (fast.cuh:18-23): 0 SUM_OF_ABS_DIFF_ALL (fast.cu:233-241), 1 SUM_OF_ABS_DIFF_ON_ARC
// (:243-255, the live one), 2 MAX_THRESHOLD (:256-283: bisection over the threshold with fast_gpu_is_corner_quick, :126-148)
template <int SCORE>
__global__ void fast_response_px_kernel(int w, int h, int pitch, const uint8_t *__restrict__ img,
                                        int hb, int vb, const uint8_t *__restrict__ lut,
                                        float threshold, int resp_pitch,
                                        float *__restrict__ resp)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    float out = 0.0f;
    if (x >= hb && y >= vb && x < w - hb && y < h - vb) {
        const uint8_t *p = img + (size_t)y * pitch + x;
        const float c = (float)p[0];
        const float ct = c + threshold, c_t = c - threshold;
        float a = (float)p[-3], b = (float)p[3];
        bool similar = !(a < c_t) && !(b < c_t) && !(ct < a) && !(ct < b);
        if (!similar) {
            a = (float)p[3 * pitch];
            b = (float)p[-3 * pitch];
            similar = !(a < c_t) && !(b < c_t) && !(ct < a) && !(ct < b);
        }
        if (!similar) {
            uint32_t dark = 0, bright = 0;
            float px[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                px[i] = (float)p[ring_dy(i) * pitch + ring_dx(i)];
                dark |= (uint32_t)(px[i] < c_t) << i;
                bright |= (uint32_t)(ct < px[i]) << i;
            }
            if (lut[dark] | lut[bright]) {
                if (SCORE == ORBFE_SUM_OF_ABS_DIFF_ALL) {
                    float r = 0.0f;
#pragma unroll
                    for (int i = 0; i < 16; i++) r += fabsf(px[i] - c);
                    out = r;
                } else if (SCORE == ORBFE_MAX_THRESHOLD) {
                    // every value here is a multiple of 1/2 below 512 when the threshold is: exact in float whatever
                    // the evaluation order, as in the oracle
                    float min_thr = threshold + 1.0f, max_thr = 255.0f;
                    while (min_thr <= max_thr) {
                        const float med = floorf((min_thr + max_thr) * 0.5f);
                        const float mt = c + med, m_t = c - med;
                        uint32_t dk = 0, br = 0;
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            dk |= (uint32_t)(px[i] < m_t) << i;
                            br |= (uint32_t)(mt < px[i]) << i;
                        }
                        if (lut[dk] | lut[br])
                            min_thr = med + 1.0f;
                        else
                            max_thr = med - 1.0f;
                    }
                    out = max_thr;
                } else {
                    float rb = 0.0f, rd = 0.0f;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float ad = fabsf(px[i] - c) - threshold;
                        rd += (dark >> i & 1u) ? ad : 0.0f;
                        rb += (bright >> i & 1u) ? ad : 0.0f;
                    }
                    out = fmaxf(rb, rd);
                }
            }
        }
    }
    resp[(size_t)y * resp_pitch + x] = out;
}

// a6  grid NMS over all levels (nms.cu:86-296): one wave per level-0 cell.  The wave walks
// the cell's pixels on every level, keeps the unsigned maximum of nms_key() and decodes it.
struct NmsLevels {
    int n;
    int w[8], h[8], pitch[8]; // pitch in floats
    const float *resp[8];
};

__global__ void grid_nms_kernel(NmsLevels lv, int cell0, int cells_x, int K,
                                float *__restrict__ d_pos, float *__restrict__ d_score,
                                int *__restrict__ d_level)
{
    const int lane = threadIdx.x & 63;
    const int cell = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (cell >= K) return; // whole wave
    const int cx = cell % cells_x, cy = cell / cells_x;
    // Responses are floats here (any float threshold, fast.cuh:28-40: with t = 7.5 they are x.5), so the
    // key is 64 bits: the response's bit pattern (positive floats order like unsigned integers) above
    // the level / tie-rank field of nms_key().  Its unsigned maximum is the reference's winner.
    uint64_t best = 0;
    for (int l = 0; l < lv.n; l++) {
        const int c = cell0 >> l;
        if (c == 0) break;
        const int W = lv.w[l], H = lv.h[l], P = lv.pitch[l];
        const float *R = lv.resp[l];
        for (int i = lane; i < c * c; i += 64) {
            const int x = c * cx + (i % c), y = c * cy + (i / c);
            if (x < 3 || y < 3 || x >= W - 3 || y >= H - 3) continue;
            const float *q = R + (size_t)y * P + x;
            const float v = q[0];
            if (!(v > 0.0f)) continue;
            const bool is_max = v > q[-P] && v > q[-P + 1] && v > q[1] && v > q[P + 1] &&
                                v > q[P] && v > q[P - 1] && v > q[-1] && v > q[-P - 1];
            if (!is_max) continue;
            const uint64_t key = ((uint64_t)__float_as_uint(v) << 32) | nms_key(0, l, x, y, cell0);
            best = key > best ? key : best;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best, off), hi = (uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), off);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        best = o > best ? o : best;
    }
    if (lane == 0) {
        int s = 0, l = 0, x = 0, y = 0;
        // position and level from the low field (a non-zero score field makes nms_decode treat the cell as taken)
        if (best) nms_decode((1u << 15) | (uint32_t)(best & 0x7FFFu), cx, cy, cell0, &s, &l, &x, &y);
        d_score[cell] = __uint_as_float((uint32_t)(best >> 32));
        d_pos[2 * cell] = (float)x;
        d_pos[2 * cell + 1] = (float)y;
        d_level[cell] = l;
    }
}

// ------------------------------------------------------------------------------------
// a8  intensity-centroid orientation (orb.cu:77-142).  One wave per keypoint: lane =
// patch column (0..30) + 32 * half; half 0 sums the rows above and the centre row, half 1
// the rows below.  Integer sums (exact), wave add-reduction, deterministic atan2f.
// ------------------------------------------------------------------------------------
__global__ void fast_angle_kernel(float *__restrict__ d_angle, const float *__restrict__ d_pos,
                                  const uint8_t *__restrict__ img, int pitch, int w, int h, int n)
{
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= n) return;
    const int kx = (int)floor((double)d_pos[2 * idx] + 0.5); // double, as orb.cu:86-87
    const int ky = (int)floor((double)d_pos[2 * idx + 1] + 0.5);
    int m10 = 0, m01 = 0;
    // a keypoint the detector produced is >= 3 px inside; an arbitrary caller position may
    // not be: rows outside the image are skipped instead of read (the reference would read
    // out of bounds for the centre row).
    if (ky >= 0 && ky < h) patch_moments(GlobalPx{img, pitch}, w, h, kx, ky, lane, &m10, &m01);
    if (lane == 0) d_angle[idx] = orbfe_atan2f((float)m01, (float)m10);
}

// ------------------------------------------------------------------------------------
// a9 + a10  rBRIEF (orb.cu:17-75) + 32-bit "compression" (orb.cu:145-169).
// One wave per keypoint.  In round r (0..3) lane t evaluates pattern test 64 r + t, and the
// 64-bit __ballot of the outcomes IS descriptor bytes 8r .. 8r+7 (bit j of byte b is test
// 8b + j).  No shared memory, no byte assembly.
// ------------------------------------------------------------------------------------
__global__ void calc_orb_kernel(const float *__restrict__ d_angle, const float *__restrict__ d_pos,
                                uint8_t *__restrict__ d_desc, uint32_t *__restrict__ d_desc32,
                                const uint8_t *__restrict__ img, int pitch, int w, int h, int n,
                                int radians)
{
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= n) return;
    const int lx = (int)(short)d_pos[2 * idx], ly = (int)(short)d_pos[2 * idx + 1];
    uint64_t d[4] = {0, 0, 0, 0};
    if (!orb_border_zero(lx, ly, w, h, radians)) {
        float a, b;
        orb_steer(d_angle[idx], radians, &a, &b);
        orb_describe(GlobalPx{img, pitch}, lx, ly, a, b, lane, d);
    }
    if (lane < 4) reinterpret_cast<uint64_t *>(d_desc + (size_t)idx * 32)[lane] = d[lane];
    if (lane == 0 && d_desc32) d_desc32[idx] = orb_compress(d);
}

// ------------------------------------------------------------------------------------
// f2  keypoint filter + ordered compaction + deprojection (cuda-align.cu:85-112, :282-364).
// One workgroup walks the keypoints in chunks of 256 with a running offset, so the output order
// is the keypoint order.  f64 arithmetic as written in the reference, no contraction.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
keypoint_pixel_to_point_kernel(const uint32_t *__restrict__ depth_img, orbfe_intrinsics K, int W,
                               float *__restrict__ pos_out, const float *__restrict__ pos_in,
                               const float *__restrict__ score_in, double *__restrict__ points,
                               uint32_t *__restrict__ desc_out, const uint32_t *__restrict__ desc_in, int n,
                               int32_t *__restrict__ n_valid, int fix_col)
{
    ORBFE_NO_CONTRACT
    __shared__ int s_wave[4];
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int idx = i0 + threadIdx.x;
        bool ok = false;
        float px = 0.f, py = 0.f;
        int depth = 0;
        if (idx < n) {
            px = pos_in[2 * idx];
            py = pos_in[2 * idx + 1];
            const int row = (int)((double)py + 0.5);
            const int col = fix_col ? (int)((double)px + 0.5) : (int)((double)py + 0.5);
            depth = (int)depth_img[(size_t)row * W + col];
            ok = depth > 1 && score_in[idx] > 1.0f;
        }
        int total;
        const int slot = base + block_excl_scan(ok, s_wave, &total);
        if (ok) {
            double x = (double)((px - K.ppx) / K.fx);
            double y = (double)((py - K.ppy) / K.fy);
            if (K.model == 2) {
                const double c0 = K.coeffs[0], c1 = K.coeffs[1], c2 = K.coeffs[2], c3 = K.coeffs[3], c4 = K.coeffs[4];
                const double r2 = x * x + y * y;
                double f = 1 + c0 * r2;
                f = f + c1 * r2 * r2;
                f = f + c4 * r2 * r2 * r2;
                double ux = x * f + 2 * c2 * x * y;
                ux = ux + c3 * (r2 + 2 * x * x);
                double uy = y * f + 2 * c3 * x * y;
                uy = uy + c2 * (r2 + 2 * y * y);
                x = ux;
                y = uy;
            }
            const double dd = (double)(float)depth;
            points[3 * (size_t)slot + 0] = dd * x;
            points[3 * (size_t)slot + 1] = dd * y;
            points[3 * (size_t)slot + 2] = dd;
            desc_out[slot] = desc_in[idx];
            pos_out[2 * slot] = px;
            pos_out[2 * slot + 1] = py;
        }
        base += total;
    }
    if (threadIdx.x == 0) *n_valid = base;
}

// ------------------------------------------------------------------------------------
// a11 / a12  the compacted outputs of kernel_match_keypoints (post_processing.cu:176-198): matched
// prev / curr 3-D points (double3) and the matched curr positions as uint16 x / y (the frame's
// keypoints_x / keypoints_y, :300-331).  One workgroup walks the prev keypoints in chunks of 256
// with a running offset, so the lists are in prev order (the reference's atomics give any order).
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
match_compact_kernel(const int32_t *__restrict__ match_idx, int n_prev, const double *__restrict__ points_prev,
                     const double *__restrict__ points_curr, const float *__restrict__ pos_curr,
                     double *__restrict__ prev_matched, double *__restrict__ curr_matched,
                     uint16_t *__restrict__ kx, uint16_t *__restrict__ ky, int32_t *__restrict__ n_matched)
{
    __shared__ int s_wave[4];
    int base = 0;
    for (int i0 = 0; i0 < n_prev; i0 += 256) {
        const int idx = i0 + threadIdx.x;
        const int pair = idx < n_prev ? match_idx[idx] : -1;
        const bool ok = pair >= 0;
        int total;
        const int slot = base + block_excl_scan(ok, s_wave, &total);
        if (ok) {
            if (points_prev && prev_matched) {
#pragma unroll
                for (int k = 0; k < 3; k++) prev_matched[3 * (size_t)slot + k] = points_prev[3 * (size_t)idx + k];
            }
            if (points_curr && curr_matched) {
#pragma unroll
                for (int k = 0; k < 3; k++) curr_matched[3 * (size_t)slot + k] = points_curr[3 * (size_t)pair + k];
            }
            kx[slot] = (uint16_t)pos_curr[2 * (size_t)pair];
            ky[slot] = (uint16_t)pos_curr[2 * (size_t)pair + 1];
        }
        base += total;
    }
    if (threadIdx.x == 0 && n_matched) *n_matched = base;
}

// f4 (part)  kernel_reproject_prev_points + project_point_to_pixel_double (post_processing.cu:11-43, :72-90)
struct Mat4 {
    double m[16]; // column-major, as Eigen::Matrix4d stores it
};
__global__ void reproject_points_kernel(float *__restrict__ pos_out, const double *__restrict__ points, int n, Mat4 T,
                                        orbfe_intrinsics K)
{
    ORBFE_NO_CONTRACT
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const double px = points[3 * (size_t)idx], py = points[3 * (size_t)idx + 1], pz = points[3 * (size_t)idx + 2];
    double e[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double t = T.m[i] * px + T.m[4 + i] * py;
        t = t + T.m[8 + i] * pz;
        t = t + T.m[12 + i];
        e[i] = t;
    }
    float x = (float)(e[0] / e[2]), y = (float)(e[1] / e[2]);
    if (K.model == 1) {
        const float r2 = x * x + y * y;
        float f = 1 + K.coeffs[0] * r2;
        f = f + K.coeffs[1] * r2 * r2;
        f = f + K.coeffs[4] * r2 * r2 * r2;
        x *= f;
        y *= f;
        float dx = x + 2 * K.coeffs[2] * x * y;
        dx = dx + K.coeffs[3] * (r2 + 2 * x * x);
        float dy = y + 2 * K.coeffs[3] * x * y;
        dy = dy + K.coeffs[2] * (r2 + 2 * y * y);
        x = dx;
        y = dy;
    }
    if (K.model == 3) orbfe_ftheta_distort(&x, &y, K.coeffs[0]); // RS2_DISTORTION_FTHETA, post_processing.cu:32-38
    pos_out[2 * (size_t)idx] = x * K.fx + K.ppx;
    pos_out[2 * (size_t)idx + 1] = y * K.fy + K.ppy;
}

// ------------------------------------------------------------------------------------
// a11  reference matcher (post_processing.cu:92-200).  Thread i = prev keypoint i; its
// "tid" in the reference's 32-thread block is i & 31.  Curr keypoints are staged in LDS in
// tiles of 32; inside a tile of m entries thread tid visits j = (s + tid) % m, s = 0..m-1,
// and skips the tile when tid >= m (Q8).  First strictly smaller distance wins.
// ------------------------------------------------------------------------------------
__global__ void match_ref_kernel(const float *__restrict__ pos_prev,
                                 const uint32_t *__restrict__ desc_prev, int n_prev,
                                 const float *__restrict__ pos_curr,
                                 const uint32_t *__restrict__ desc_curr, int n_curr, float win,
                                 int max_ham, int32_t *__restrict__ match_idx,
                                 int32_t *__restrict__ num_matched)
{
    __shared__ float s_x[32], s_y[32];
    __shared__ uint32_t s_d[32];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int tid = i & 31;
    const bool live = i < n_prev;
    float px = 0.f, py = 0.f;
    uint32_t d = 0;
    if (live) {
        px = pos_prev[2 * i];
        py = pos_prev[2 * i + 1];
        d = desc_prev[i];
    }
    int best = 9999999, pair = -1;
    for (int base = 0; base < n_curr; base += 32) {
        __syncthreads();
        if (threadIdx.x < 32 && base + (int)threadIdx.x < n_curr) {
            s_x[threadIdx.x] = pos_curr[2 * (base + threadIdx.x)];
            s_y[threadIdx.x] = pos_curr[2 * (base + threadIdx.x) + 1];
            s_d[threadIdx.x] = desc_curr[base + threadIdx.x];
        }
        __syncthreads();
        const int m = (base + 32 >= n_curr) ? n_curr - base : 32;
        if (live && tid < m) {
            int j = tid; // (s + tid) % m for s = 0
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
    if (live) match_idx[i] = pair;
    const uint64_t hit = __ballot(live && pair >= 0);
    if ((threadIdx.x & 63) == 0 && hit) atomicAdd(num_matched, (int)__popcll(hit));
}

// ------------------------------------------------------------------------------------
// EXT  brute-force 256-bit Hamming.  Thread = one query (8 dwords in registers); B is
// staged through LDS in tiles and read as wave-uniform (broadcast) b128 pairs.
// ------------------------------------------------------------------------------------
constexpr int kMatchTile = 256;

__global__ void __launch_bounds__(256)
match256_kernel(const uint8_t *__restrict__ descA, const float *__restrict__ posA, int nA,
                const uint8_t *__restrict__ descB, const float *__restrict__ posB, int nB,
                int window, int max_dist, int32_t *__restrict__ out_idx,
                int32_t *__restrict__ out_dist)
{
    __shared__ uint4 s_b[kMatchTile * 2];
    __shared__ float2 s_p[kMatchTile];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < nA;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    float ax = 0.f, ay = 0.f;
    if (live) {
        const uint4 *pa = reinterpret_cast<const uint4 *>(descA + (size_t)i * 32);
        a0 = pa[0];
        a1 = pa[1];
        if (window >= 0) {
            ax = posA[2 * i];
            ay = posA[2 * i + 1];
        }
    }
    const float win = (float)window;
    int best = 1 << 30, best_j = -1;
    for (int base = 0; base < nB; base += kMatchTile) {
        const int m = min(kMatchTile, nB - base);
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * m; t += blockDim.x)
            s_b[t] = reinterpret_cast<const uint4 *>(descB + (size_t)base * 32)[t];
        if (window >= 0)
            for (int t = threadIdx.x; t < m; t += blockDim.x)
                s_p[t] = make_float2(posB[2 * (base + t)], posB[2 * (base + t) + 1]);
        __syncthreads();
        for (int j = 0; j < m; j++) {
            const uint4 b0 = s_b[2 * j], b1 = s_b[2 * j + 1];
            int dist = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) +
                       __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) +
                       __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
            if (window >= 0) {
                const float2 pb = s_p[j];
                if (fabsf(ax - pb.x) > win || fabsf(ay - pb.y) > win) dist = 1 << 30;
            }
            if (dist < best) { // strict: ties keep the lower index
                best = dist;
                best_j = base + j;
            }
        }
    }
    if (live) {
        const bool ok = best_j >= 0 && best <= max_dist;
        out_idx[i] = ok ? best_j : -1;
        if (out_dist) out_dist[i] = ok ? best : -1;
    }
}

} // namespace orbfe

// ======================================================================================
// C ABI
// ======================================================================================
using namespace orbfe;

static inline hipStream_t S(orbfe_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static int launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_thread_error("%s launch failed: %s", what, hipGetErrorString(e));
        return ORBFE_ERR_HIP;
    }
    return ORBFE_OK;
}

#define ARG_CHECK(cond)                                                                     \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            set_thread_error("%s: invalid argument: %s", __func__, #cond);                  \
            return ORBFE_ERR_INVALID_ARG;                                                   \
        }                                                                                   \
    } while (0)

extern "C" {

int orbfe_version(void) { return ORBFE_VERSION; }

int orbfe_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int orbfe_load_pattern(void) { return ORBFE_OK; }

int orbfe_rgb_to_grayscale(unsigned char *d_dst, const unsigned char *d_src, int cols, int rows, int dst_pitch,
                           int src_pitch, orbfe_stream_t stream)
{
    ARG_CHECK(d_dst && d_src && cols > 0 && rows > 0 && dst_pitch >= cols && src_pitch >= 3 * cols);
    dim3 block(64, 4), grid((cols + 63) / 64, (rows + 3) / 4);
    hipLaunchKernelGGL(rgb_to_gray_px_kernel, grid, block, 0, S(stream), d_dst, d_src, cols, rows, dst_pitch,
                       src_pitch);
    return launch_status("rgb_to_grayscale");
}

int orbfe_gaussian_blur_3x3(unsigned char *d_blurred, int blurred_pitch,
                            const unsigned char *d_image, int image_pitch, int w, int h,
                            orbfe_stream_t stream)
{
    ARG_CHECK(d_blurred && d_image && w > 0 && h > 0 && blurred_pitch >= w && image_pitch >= w);
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(blur3x3_px_kernel, grid, block, 0, S(stream), d_blurred, blurred_pitch,
                       d_image, image_pitch, w, h);
    return launch_status("gaussian_blur_3x3");
}

int orbfe_pyramid_create_levels(const orbfe_pyramid_level *lv, int n_levels,
                                orbfe_stream_t stream)
{
    ARG_CHECK(lv && n_levels >= 1);
    for (int i = 1; i < n_levels; i++) {
        ARG_CHECK(lv[i].image && lv[i - 1].image);
        ARG_CHECK(lv[i].image_width == lv[i - 1].image_width / 2 &&
                  lv[i].image_height == lv[i - 1].image_height / 2);
        ARG_CHECK(lv[i].image_pitch >= lv[i].image_width);
        const int dw = (int)lv[i].image_width, dh = (int)lv[i].image_height;
        if (dw == 0 || dh == 0) continue;
        dim3 block(64, 4), grid((dw + 63) / 64, (dh + 3) / 4);
        hipLaunchKernelGGL(halfsample_px_kernel, grid, block, 0, S(stream), lv[i - 1].image,
                           (int)lv[i - 1].image_pitch, lv[i].image, (int)lv[i].image_pitch, dw, dh);
    }
    return launch_status("pyramid_create_levels");
}

int orbfe_fast_calculate_lut(unsigned char *d_lut, int min_arc, orbfe_stream_t stream)
{
    ARG_CHECK(d_lut && min_arc >= 1 && min_arc <= 16);
    hipLaunchKernelGGL(fast_lut_kernel, dim3(256), dim3(256), 0, S(stream), d_lut, min_arc);
    return launch_status("fast_calculate_lut");
}

int orbfe_fast_calc_corner_response(int w, int h, int pitch, const unsigned char *d_image, int hb,
                                    int vb, const unsigned char *d_lut, float threshold,
                                    int min_arc_length, int score, int resp_pitch_elems,
                                    float *d_response, orbfe_stream_t stream)
{
    (void)min_arc_length; // unused by the reference kernel too (fast.cu:160): the LUT decides
    ARG_CHECK(d_image && d_lut && d_response && w > 0 && h > 0 && pitch >= w &&
              resp_pitch_elems >= w && hb >= 3 && vb >= 3);
    ARG_CHECK(score == ORBFE_SUM_OF_ABS_DIFF_ALL || score == ORBFE_SUM_OF_ABS_DIFF_ON_ARC || score == ORBFE_MAX_THRESHOLD);
    ARG_CHECK(threshold - threshold == 0.0f); // finite: a NaN or infinite threshold would never end MAX_THRESHOLD's bisection
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4);
    // fast.cu:325-384: one instantiation per score
    if (score == ORBFE_SUM_OF_ABS_DIFF_ALL)
        hipLaunchKernelGGL(fast_response_px_kernel<ORBFE_SUM_OF_ABS_DIFF_ALL>, grid, block, 0, S(stream), w, h, pitch, d_image, hb,
                           vb, d_lut, threshold, resp_pitch_elems, d_response);
    else if (score == ORBFE_MAX_THRESHOLD)
        hipLaunchKernelGGL(fast_response_px_kernel<ORBFE_MAX_THRESHOLD>, grid, block, 0, S(stream), w, h, pitch, d_image, hb,
                           vb, d_lut, threshold, resp_pitch_elems, d_response);
    else
        hipLaunchKernelGGL(fast_response_px_kernel<ORBFE_SUM_OF_ABS_DIFF_ON_ARC>, grid, block, 0, S(stream), w, h, pitch, d_image, hb,
                           vb, d_lut, threshold, resp_pitch_elems, d_response);
    return launch_status("fast_calc_corner_response");
}

int orbfe_grid_nms(const orbfe_pyramid_level *lv, int n_levels, float *d_pos, float *d_score,
                   int *d_level, orbfe_stream_t stream)
{
    ARG_CHECK(lv && n_levels >= 1 && d_pos && d_score && d_level);
    if (n_levels > 6) { // 32 >> 6 == 0: the reference divides by zero (nms.cu:273), Q12
        set_thread_error("grid_nms: at most 6 levels with 32-pixel cells");
        return ORBFE_ERR_UNSUPPORTED;
    }
    NmsLevels L;
    L.n = n_levels;
    for (int i = 0; i < n_levels; i++) {
        ARG_CHECK(lv[i].response && lv[i].response_pitch >= lv[i].image_width * sizeof(float));
        L.w[i] = (int)lv[i].image_width;
        L.h[i] = (int)lv[i].image_height;
        L.pitch[i] = (int)(lv[i].response_pitch / sizeof(float));
        L.resp[i] = lv[i].response;
    }
    const int W = (int)lv[0].image_width, H = (int)lv[0].image_height;
    const int cx = (W + 31) / 32, cy = (H + 31) / 32, K = cx * cy;
    hipLaunchKernelGGL(grid_nms_kernel, dim3((K + 3) / 4), dim3(256), 0, S(stream), L, 32, cx, K,
                       d_pos, d_score, d_level);
    return launch_status("grid_nms");
}

int orbfe_detect(const orbfe_pyramid_level *lv, int n_levels, const unsigned char *d_lut,
                 float threshold, float *d_pos, float *d_score, int *d_level,
                 orbfe_stream_t stream)
{
    ARG_CHECK(lv && n_levels >= 1 && d_lut && d_pos && d_score && d_level);
    if (n_levels > 6) { // 32 >> 6 == 0: the reference divides by zero (nms.cu:273), Q12
        set_thread_error("detect: at most 6 levels with 32-pixel cells");
        return ORBFE_ERR_UNSUPPORTED;
    }
    // Fused path: FAST score + 3x3 NMS + cell maximum of every level in ONE launch (the batch path's tile
    // kernel, instantiated table-driven: the reference's own prechecks, then lut[dark] | lut[bright] per
    // candidate, so d_lut stays the opaque table of fast.cuh:42-48 -- nothing is remembered about how or
    // where it was built), the scores also written to the caller's response maps; needs an integer threshold
    // (the kernel's packed-u16 arithmetic) and dword-aligned levels.
    bool fused = threshold >= 1.0f && threshold <= 254.0f && threshold == (float)(int)threshold;
    for (int i = 0; i < n_levels && fused; i++) {
        ARG_CHECK(lv[i].image && lv[i].image_pitch >= lv[i].image_width);
        ARG_CHECK(!lv[i].response || lv[i].response_pitch >= lv[i].image_width * sizeof(float));
        fused = lv[i].image_pitch % 4 == 0 && (reinterpret_cast<uintptr_t>(lv[i].image) & 3u) == 0 &&
                lv[i].image_pitch < (1u << 24) && lv[i].image_width > 0 && lv[i].image_height > 0;
    }
    if (fused) {
        const int rc = launch_detect_stage(lv, n_levels, (int)threshold, d_lut, d_pos, d_score, d_level, S(stream));
        if (rc != ORBFE_OK) set_thread_error("detect: launch failed");
        return rc;
    }
    for (int i = 0; i < n_levels; i++) {
        int rc = orbfe_fast_calc_corner_response(
            (int)lv[i].image_width, (int)lv[i].image_height, (int)lv[i].image_pitch, lv[i].image, 3,
            3, d_lut, threshold, 0, ORBFE_SUM_OF_ABS_DIFF_ON_ARC,
            (int)(lv[i].response_pitch / sizeof(float)), lv[i].response, stream);
        if (rc != ORBFE_OK) return rc;
    }
    return orbfe_grid_nms(lv, n_levels, d_pos, d_score, d_level, stream);
}

int orbfe_compute_fast_angle(float *d_angle, const float *d_pos, const unsigned char *d_image,
                             int pitch, int w, int h, int n, orbfe_stream_t stream)
{
    ARG_CHECK(d_angle && d_pos && d_image && w > 0 && h > 0 && pitch >= w && n >= 0);
    if (n == 0) return ORBFE_OK;
    hipLaunchKernelGGL(fast_angle_kernel, dim3((n + 3) / 4), dim3(256), 0, S(stream), d_angle, d_pos,
                       d_image, pitch, w, h, n);
    return launch_status("compute_fast_angle");
}

int orbfe_calc_orb(const float *d_angle, const float *d_pos, unsigned char *d_desc_tmp,
                   uint32_t *d_desc, const unsigned char *d_image, int pitch, int w, int h, int n,
                   orbfe_stream_t stream)
{
    ARG_CHECK(d_angle && d_pos && d_desc_tmp && d_image && w > 0 && h > 0 && pitch >= w && n >= 0);
    ARG_CHECK((reinterpret_cast<uintptr_t>(d_desc_tmp) & 7u) == 0);
    if (n == 0) return ORBFE_OK;
    hipLaunchKernelGGL(calc_orb_kernel, dim3((n + 3) / 4), dim3(256), 0, S(stream), d_angle, d_pos,
                       d_desc_tmp, d_desc, d_image, pitch, w, h, n, 0);
    return launch_status("calc_orb");
}

int orbfe_match_keypoints(const float *d_pos_prev, const uint32_t *d_desc_prev, int n_prev,
                          const float *d_pos_curr, const uint32_t *d_desc_curr, int n_curr,
                          int max_px, int max_ham, int32_t *d_match_idx, int32_t *d_num_matched,
                          orbfe_stream_t stream)
{
    ARG_CHECK(n_prev >= 0 && n_curr >= 0 && d_num_matched);
    hipError_t e = hipMemsetAsync(d_num_matched, 0, sizeof(int32_t), S(stream));
    if (e != hipSuccess) {
        set_thread_error("match_keypoints: hipMemsetAsync: %s", hipGetErrorString(e));
        return ORBFE_ERR_HIP;
    }
    if (n_prev == 0) return ORBFE_OK;
    ARG_CHECK(d_pos_prev && d_desc_prev && d_match_idx && (n_curr == 0 || (d_pos_curr && d_desc_curr)));
    hipLaunchKernelGGL(match_ref_kernel, dim3((n_prev + 255) / 256), dim3(256), 0, S(stream),
                       d_pos_prev, d_desc_prev, n_prev, d_pos_curr, d_desc_curr, n_curr,
                       (float)max_px, max_ham, d_match_idx, d_num_matched);
    return launch_status("match_keypoints");
}

int orbfe_match_compact(const int32_t *d_match_idx, int n_prev, const double *d_points_prev,
                        const double *d_points_curr, const float *d_pos_curr, double *d_prev_matched,
                        double *d_curr_matched, uint16_t *d_keypoints_x, uint16_t *d_keypoints_y,
                        int32_t *d_num_matched, orbfe_stream_t stream)
{
    ARG_CHECK(n_prev >= 0 && d_num_matched);
    ARG_CHECK(n_prev == 0 || (d_match_idx && d_pos_curr && d_keypoints_x && d_keypoints_y));
    ARG_CHECK((d_points_prev == nullptr) == (d_prev_matched == nullptr));
    ARG_CHECK((d_points_curr == nullptr) == (d_curr_matched == nullptr));
    hipLaunchKernelGGL(match_compact_kernel, dim3(1), dim3(256), 0, S(stream), d_match_idx, n_prev, d_points_prev,
                       d_points_curr, d_pos_curr, d_prev_matched, d_curr_matched, d_keypoints_x, d_keypoints_y,
                       d_num_matched);
    return launch_status("match_compact");
}

int orbfe_reproject_points(float *d_pos_out, const double *d_points_prev, int n, const double *T_w2c_prev_curr,
                           const orbfe_intrinsics *intrin, orbfe_stream_t stream)
{
    ARG_CHECK(n >= 0 && T_w2c_prev_curr && intrin);
    orbfe_intrinsics k_tmp; // host struct, or the reference's device copy (_d_rgb_intrinsics, post_processing.cuh:45)
    intrin = host_view(intrin, &k_tmp);
    ARG_CHECK(intrin);
    // model 2 (inverse Brown-Conrady, what a D4xx colour stream reports) and 4 project WITHOUT distortion in the
    // reference: its assert against model 2 is commented out (post_processing.cu:15) and only models 1 and 3 have a
    // branch (:19-38).  Model 3 (f-theta, :32-38) is orbfe_ftheta_distort: float atanf / tanf, the build's deterministic
    // versions (include/orbfe_math.h).
    if (n == 0) return ORBFE_OK;
    ARG_CHECK(d_pos_out && d_points_prev);
    Mat4 T;
    memcpy(T.m, T_w2c_prev_curr, sizeof(T.m));
    hipLaunchKernelGGL(reproject_points_kernel, dim3((n + 255) / 256), dim3(256), 0, S(stream), d_pos_out, d_points_prev,
                       n, T, *intrin);
    return launch_status("reproject_points");
}

int orbfe_keypoint_pixel_to_point(const uint32_t *d_aligned_depth, const orbfe_intrinsics *intrin, int image_width,
                                  int image_height, float *d_pos_out, const float *d_pos_in, const float *d_score,
                                  double *d_points, uint32_t *d_descriptors_out, const uint32_t *d_descriptors_in,
                                  int keypoints_num, int32_t *d_valid_keypoints_num, int fix_depth_index,
                                  orbfe_stream_t stream)
{
    ARG_CHECK(intrin && d_valid_keypoints_num && image_width > 0 && image_height > 0 && keypoints_num >= 0);
    orbfe_intrinsics k_tmp; // host struct, or the reference's device copy (_d_rgb_intrinsics, buildStream.cpp:469)
    intrin = host_view(intrin, &k_tmp);
    ARG_CHECK(intrin);
    if (intrin->model == 1 || intrin->model == 3) {
        set_thread_error("keypoint_pixel_to_point: cannot deproject a forward-distorted image (model %d)",
                         intrin->model);
        return ORBFE_ERR_UNSUPPORTED;
    }
    ARG_CHECK(fix_depth_index || image_height <= image_width); // the reference's index uses y as the column
    ARG_CHECK(keypoints_num == 0 || (d_aligned_depth && d_pos_out && d_pos_in && d_score && d_points &&
                                      d_descriptors_out && d_descriptors_in));
    hipLaunchKernelGGL(keypoint_pixel_to_point_kernel, dim3(1), dim3(256), 0, S(stream), d_aligned_depth, *intrin,
                       image_width, d_pos_out, d_pos_in, d_score, d_points, d_descriptors_out, d_descriptors_in,
                       keypoints_num, d_valid_keypoints_num, fix_depth_index ? 1 : 0);
    return launch_status("keypoint_pixel_to_point");
}

int orbfe_match256(const unsigned char *d_descA, const float *d_posA, int nA,
                   const unsigned char *d_descB, const float *d_posB, int nB, int window,
                   int max_distance, int32_t *d_idx, int32_t *d_dist, orbfe_stream_t stream)
{
    ARG_CHECK(nA >= 0 && nB >= 0);
    if (nA == 0) return ORBFE_OK;
    ARG_CHECK(d_descA && d_idx && (nB == 0 || d_descB));
    ARG_CHECK(window < 0 || (d_posA && (nB == 0 || d_posB)));
    ARG_CHECK((reinterpret_cast<uintptr_t>(d_descA) & 15u) == 0 &&
              (reinterpret_cast<uintptr_t>(d_descB) & 15u) == 0);
    hipLaunchKernelGGL(match256_kernel, dim3((nA + 255) / 256), dim3(256), 0, S(stream), d_descA,
                       d_posA, nA, d_descB, d_posB, nB, window, max_distance, d_idx, d_dist);
    return launch_status("match256");
}

int orbfe_memcpy_d2h(void *dst, const void *d_src, size_t bytes, orbfe_stream_t stream)
{
    hipError_t e = hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, S(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
    if (e != hipSuccess) {
        set_thread_error("memcpy_d2h: %s", hipGetErrorString(e));
        return ORBFE_ERR_HIP;
    }
    return ORBFE_OK;
}

int orbfe_memcpy_h2d(void *d_dst, const void *src, size_t bytes, orbfe_stream_t stream)
{
    hipError_t e = hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, S(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
    if (e != hipSuccess) {
        set_thread_error("memcpy_h2d: %s", hipGetErrorString(e));
        return ORBFE_ERR_HIP;
    }
    return ORBFE_OK;
}

int orbfe_stream_sync(orbfe_stream_t stream)
{
    hipError_t e = hipStreamSynchronize(S(stream));
    if (e != hipSuccess) {
        set_thread_error("stream_sync: %s", hipGetErrorString(e));
        return ORBFE_ERR_HIP;
    }
    return ORBFE_OK;
}

} // extern "C"
