/* orbfe_oracle.c -- CPU oracle for the ORB front-end hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * See orbfe_oracle.h for who may call this and for the "PARITY UNPINNED" statement.
 * Every function cites the reference lines (relative to /root/reference) it restates.
 * The restatement is deliberately literal (thread ids, warps, shuffles, shared-memory
 * scans are simulated as written) so that it is an independent check of the closed forms
 * the HIP kernels use.  Quirk numbers Qn refer to SURVEY.md Appendix A.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  -ffp-contract=off is
 * REQUIRED: include/orbfe_math.h must see no fused multiply-add.
 */
#include "orbfe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/orbfe_math.h"
#include "../include/orbfe_pattern.h"

#define WARP 32 /* CUDA_WARP_SIZE, src/cuda_common.h:57-59 */

static const int8_t g_pattern[ORBFE_PATTERN_TESTS * 4] = {ORBFE_PATTERN_VALUES};

const int8_t *oracle_pattern(void) { return g_pattern; }
float oracle_atan2f(float y, float x) { return orbfe_atan2f(y, x); }
void oracle_sincosf(float x, float *s, float *c) { orbfe_sincosf(x, s, c); }
int oracle_has_arc(uint32_t mask, int arc) { return orbfe_has_arc(mask, arc); }

/* ------------------------------------------------------------------------------------
 * f1  kernel_rgb_to_grayscale     src/cuda/cuda_RGB_to_Grayscale.cu:10-23
 * dst = floor((B*0.07 + G*0.72 + R*0.21) + 0.5): R, G, B are floats, the constants doubles, so
 * the expression is evaluated in double, left to right.  Restated literally with separate
 * IEEE multiplies and adds (no contraction, as everywhere in this build; nvcc's default
 * -fmad=true may fuse some of them, which changes the result only when 7B+72G+21R is within
 * rounding error of k + 1/2 -- unobservable here, hence unpinned like the trig functions).
 * ------------------------------------------------------------------------------------ */
void oracle_rgb_to_grayscale(uint8_t *dst, const uint8_t *src, int cols, int rows, int dst_pitch,
                             int src_pitch)
{
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float R = (float)src[(size_t)y * src_pitch + x * 3 + 0];
            float G = (float)src[(size_t)y * src_pitch + x * 3 + 1];
            float B = (float)src[(size_t)y * src_pitch + x * 3 + 2];
            double t = (double)B * 0.07;
            t = t + (double)G * 0.72;
            t = t + (double)R * 0.21;
            dst[(size_t)y * dst_pitch + x] = (uint8_t)floor(t + 0.5);
        }
}

/* ------------------------------------------------------------------------------------
 * a2  gaussian_blur_3x3        src/cuda/gaussian_blur_3x3.cu:15-53 (kernel), :55-73 (host)
 * One 32-lane warp per 32 columns of one row; horizontal taps come from __shfl_up/_down
 * inside the warp, so lane 0 has no left and lane 31 no right neighbour and gets its own
 * value back (Q2).  Rows 0, h-2, h-1 are never written (:22-23) -> decided 0 (Q1).
 * If w % 32 != 0 the last active lane shuffles from an exited lane: decided own value (Q2).
 * ------------------------------------------------------------------------------------ */
void oracle_gaussian_blur_3x3(uint8_t *blurred, int blurred_pitch, const uint8_t *image,
                              int image_pitch, int w, int h)
{
    for (int y = 0; y < h; y++) {
        uint8_t *out = blurred + (size_t)y * blurred_pitch;
        if (y == 0 || y >= h - 2) { /* :22 */
            memset(out, 0, (size_t)w);
            continue;
        }
        for (int x = 0; x < w; x++) {
            int lane = x & (WARP - 1);
            int xl = (lane == 0) ? x : x - 1;                  /* __shfl_up, delta 1  */
            int xr = (lane == WARP - 1 || x == w - 1) ? x : x + 1; /* __shfl_down, delta 1 */
            const uint8_t *ra = image + (size_t)(y - 1) * image_pitch;
            const uint8_t *rb = image + (size_t)y * image_pitch;
            const uint8_t *rc = image + (size_t)(y + 1) * image_pitch;
            int a = ra[x], b = rb[x], c = rc[x];
            int blur_sum = 0;
            blur_sum += a * 2; /* :37-45 */
            blur_sum += b * 4;
            blur_sum += c * 2;
            blur_sum += ra[xl];
            blur_sum += ra[xr];
            blur_sum += rb[xl] * 2;
            blur_sum += rb[xr] * 2;
            blur_sum += rc[xl];
            blur_sum += rc[xr];
            out[x] = (uint8_t)floor((double)((float)blur_sum / 16.0f) + 0.5); /* :49 */
        }
    }
}

/* ------------------------------------------------------------------------------------
 * a3  image_halfsample_gpu_kernel / pyramid_create_levels   src/cuda/pyramid.cu:6-29, :31-84
 * dst(x,y) = (s(2x,2y)+s(2x+1,2y)+s(2x,2y+1)+s(2x+1,2y+1)) >> 2.  The vector width N only
 * changes how many outputs one thread writes, not their values (:37-40).
 * ------------------------------------------------------------------------------------ */
void oracle_halfsample(const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch,
                       int dst_w, int dst_h)
{
    for (int y = 0; y < dst_h; y++) {
        const uint8_t *t = src + (size_t)(2 * y) * src_pitch;
        const uint8_t *b = t + src_pitch;
        uint8_t *o = dst + (size_t)y * dst_pitch;
        for (int x = 0; x < dst_w; x++)
            o[x] = (uint8_t)(((unsigned)t[2 * x] + (unsigned)t[2 * x + 1] + (unsigned)b[2 * x] +
                              (unsigned)b[2 * x + 1]) >> 2); /* :26 */
    }
}

void oracle_pyramid_create_levels(const oracle_level *lv, int n_levels)
{
    for (int i = 1; i < n_levels; i++) /* :33 */
        oracle_halfsample(lv[i - 1].image, lv[i - 1].image_pitch, lv[i].image, lv[i].image_pitch,
                          lv[i].width, lv[i].height);
}

/* ------------------------------------------------------------------------------------
 * a4  fast_gpu_is_corner / fast_gpu_calculate_lut     src/cuda/fast.cu:11-32, :34-39, :292-302
 * ------------------------------------------------------------------------------------ */
static int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; } /* CUDA __clz(0) = 32 */

int oracle_fast_is_corner(uint32_t address, int min_arc_length)
{
    int ones = __builtin_popcount(address);
    if (ones < min_arc_length) return 0; /* :15-18 */
    uint32_t address_dup = address | (address << 16);
    while (ones > 0) {
        int sh = clz32(address_dup);
        address_dup = sh >= 32 ? 0u : address_dup << sh; /* shift out the high order zeros */
        int lones = clz32(~address_dup);                 /* count the leading ones */
        if (lones >= min_arc_length) return 1;
        address_dup = lones >= 32 ? 0u : address_dup << lones;
        ones -= lones;
    }
    return 0;
}

void oracle_fast_calculate_lut(uint8_t *lut, int min_arc)
{
    for (uint32_t m = 0; m < 65536u; m++) lut[m] = (uint8_t)oracle_fast_is_corner(m, min_arc);
}

/* ------------------------------------------------------------------------------------
 * a5  fast_gpu_calc_corner_response_kernel<SUM_OF_ABS_DIFF_ON_ARC>
 *     src/cuda/fast.cu:150-287; ring offsets :41-96; prechecks :98-124.
 * The reference's `min_arc_length` kernel argument is unused and the score enum is the
 * one selected by defines.h:9 (Q4).  All values are small integers held in floats.
 * ------------------------------------------------------------------------------------ */
static int ring_offset(int i, int pitch)
{
    switch (i) { /* :62-95 */
    case 0: return 3 * pitch;
    case 1: return 3 * pitch - 1;
    case 2: return 2 * pitch - 2;
    case 3: return pitch - 3;
    case 4: return -3;
    case 5: return -pitch - 3;
    case 6: return -2 * pitch - 2;
    case 7: return -3 * pitch - 1;
    case 8: return -3 * pitch;
    case 9: return -3 * pitch + 1;
    case 10: return -2 * pitch + 2;
    case 11: return -pitch + 3;
    case 12: return 3;
    case 13: return pitch + 3;
    case 14: return 2 * pitch + 2;
    default: return 3 * pitch + 1;
    }
}

static int sgnbit(float v) { return signbit(v) ? 1 : 0; }

/* fast_gpu_is_corner_quick, src/cuda/fast.cu:126-148: the labels re-made at another threshold */
static int is_corner_quick(const uint8_t *lut, const float *px, float c, float threshold)
{
    const float ct = c + threshold;
    const float c_t = c - threshold;
    unsigned dark = 0, bright = 0;
    for (int i = 0; i < 16; i++) {
        dark += sgnbit(px[i] - c_t) ? (1u << i) : 0u;
        bright += sgnbit(ct - px[i]) ? (1u << i) : 0u;
    }
    return lut[dark] || lut[bright];
}

void oracle_fast_calc_corner_response(int w, int h, int pitch, const uint8_t *img, int hb,
                                      int vb, const uint8_t *lut, float threshold,
                                      int resp_pitch_elems, float *resp)
{
    oracle_fast_calc_corner_response_score(w, h, pitch, img, hb, vb, lut, threshold, 1, resp_pitch_elems, resp);
}

/* the same with the reference's `score` argument (enum fast_score, fast.cuh:18-23): 0 SUM_OF_ABS_DIFF_ALL (:233-241),
 * 1 SUM_OF_ABS_DIFF_ON_ARC (:243-255, the live one), 2 MAX_THRESHOLD (:256-283) */
void oracle_fast_calc_corner_response_score(int w, int h, int pitch, const uint8_t *img, int hb,
                                            int vb, const uint8_t *lut, float threshold, int score,
                                            int resp_pitch_elems, float *resp)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float *out = resp + (size_t)y * resp_pitch_elems + x;
            *out = 0.0f; /* :169 */
            if (!(x >= hb && y >= vb && x < w - hb && y < h - vb)) continue;
            const uint8_t *p = img + (size_t)y * pitch + x;
            const float c = (float)*p;
            const float ct = c + threshold;
            const float c_t = c - threshold;
            /* prechecks :98-124 */
            {
                float px0 = (float)p[ring_offset(4, pitch)];
                float px1 = (float)p[ring_offset(12, pitch)];
                if ((sgnbit(px0 - c_t) | sgnbit(px1 - c_t) | sgnbit(ct - px0) |
                     sgnbit(ct - px1)) == 0)
                    continue;
                px0 = (float)p[ring_offset(0, pitch)];
                px1 = (float)p[ring_offset(8, pitch)];
                if ((sgnbit(px0 - c_t) | sgnbit(px1 - c_t) | sgnbit(ct - px0) |
                     sgnbit(ct - px1)) == 0)
                    continue;
            }
            float px[16];
            unsigned dark = 0, bright = 0;
            for (int i = 0; i < 16; i++) { /* :214-222 */
                px[i] = (float)p[ring_offset(i, pitch)];
                dark += sgnbit(px[i] - c_t) ? (1u << i) : 0u;
                bright += sgnbit(ct - px[i]) ? (1u << i) : 0u;
            }
            if ((lut[dark] || lut[bright]) && score == 0) { /* :233-241 */
                float response = 0.0f;
                for (int i = 0; i < 16; i++) response += fabsf(px[i] - c);
                *out = response;
            } else if ((lut[dark] || lut[bright]) && score == 2) { /* :256-283: the largest threshold at which the
                                                                      table still accepts the pixel, by bisection */
                float min_thr = threshold + 1;
                float max_thr = 255.0f;
                while (min_thr <= max_thr) {
                    float med_thr = floorf((min_thr + max_thr) * 0.5f);
                    if (is_corner_quick(lut, px, c, med_thr))
                        min_thr = med_thr + 1.0f;
                    else
                        max_thr = med_thr - 1.0f;
                }
                *out = max_thr;
            } else if (lut[dark] || lut[bright]) { /* :225 */
                float response_bright = 0.0f, response_dark = 0.0f;
                for (int i = 0; i < 16; i++) { /* :248-253 */
                    float absdiff = fabsf(px[i] - c) - threshold;
                    response_dark += (dark & (1u << i)) ? absdiff : 0.0f;
                    response_bright += (bright & (1u << i)) ? absdiff : 0.0f;
                }
                *out = fmaxf(response_bright, response_dark); /* :254 */
            }
        }
}

/* ------------------------------------------------------------------------------------
 * a6  detector_base_gpu_grid_nms_kernel<true> / grid_nms     src/cuda/nms.cu:86-254, :256-296
 * Literal simulation of one thread block per cell: per-thread column scan with the
 * copysign suppression trick, 32-lane __shfl_down tournament, shared-memory scan by
 * thread (0,0), compare-and-replace into d_score/d_pos/d_level.
 * Decisions: lanes that do not exist in a partial warp never win (Q6); level 0 also
 * resets pos and level so that empty cells read (0,0)/0 (Q5).  `cell` generalises the
 * reference's fixed 32 (EXT ii); levels with (cell >> level) == 0 are skipped (EXT i; the
 * reference divides by zero there, nms.cu:273).
 * ------------------------------------------------------------------------------------ */
static int nms_offset(int i, int pitch)
{
    switch (i) { /* :58-84, spiral N,NE,E,SE,S,SW,W,NW */
    case 0: return -pitch;
    case 1: return -pitch + 1;
    case 2: return 1;
    case 3: return pitch + 1;
    case 4: return pitch;
    case 5: return pitch - 1;
    case 6: return -1;
    default: return -pitch - 1;
    }
}

static void nms_block(int level, int bx_idx, int by_idx, int grid_w, int image_width,
                      int image_height, int cell_w, int cell_h, int bdx, int bdy,
                      int response_pitch_elements, const float *d_response, float *d_pos,
                      float *d_score, int32_t *d_level)
{
    const int hb = 3, vb = 3; /* nms.cu:285-286 */
    const int nthreads = bdx * bdy;
    const int warp_cnt = (nthreads + 31) >> 5;
    float t_resp[128], t_x[128], t_y[128];
    const int cell_id = grid_w * by_idx + bx_idx;

    for (int ty = 0; ty < bdy; ty++)
        for (int tx = 0; tx < bdx; tx++) {
            const int x = cell_w * bx_idx + tx;
            const int y = cell_h * by_idx + ty;
            const int thread_id = tx + bdx * ty;
            float max_x = (float)x, max_y = 0.f, max_resp = 0.0f;
            if (x < image_width && y < image_height) {
                if (tx == 0 && ty == 0 && level == 0) { /* :116-123 (+ Q5 reset) */
                    d_score[cell_id] = 0.0f;
                    d_pos[2 * cell_id] = 0.0f;
                    d_pos[2 * cell_id + 1] = 0.0f;
                    d_level[cell_id] = 0;
                }
                int max_y_tmp = 0;
                int image_width_m_border = image_width - hb;
                int image_height_m_border = image_height - vb;
                if (x >= hb && x < image_width_m_border) {
                    int cell_top_to_border = vb - (cell_h * by_idx);
                    int y_offset = cell_top_to_border > 0 ? cell_top_to_border : 0;
                    int gy = y + y_offset;
                    int box_line = ty + y_offset;
                    for (; (box_line < cell_h) && (gy < image_height_m_border);
                         box_line += bdy, gy += bdy) {
                        const float *rp = d_response + (size_t)gy * response_pitch_elements + x;
                        float center_value = rp[0];
                        for (int i = 0; i < 8; i++) { /* :161-185 */
                            int j = nms_offset(i, response_pitch_elements);
                            center_value *=
                                -0.5f * (-1.0f + copysignf(1.0f, rp[j] - center_value));
                            if (center_value == 0.0f) break;
                        }
                        if (center_value > max_resp) { /* :188 */
                            max_resp = center_value;
                            max_y_tmp = gy;
                        }
                    }
                }
                max_y = (float)max_y_tmp;
            }
            t_resp[thread_id] = max_resp;
            t_x[thread_id] = max_x;
            t_y[thread_id] = max_y;
        }

    /* warp tournament :201-212 -- all lanes read before any lane writes */
    for (int offset = WARP / 2; offset > 0; offset /= 2) {
        float n_resp[128], n_x[128], n_y[128];
        for (int t = 0; t < nthreads; t++) {
            int lane = t & 31;
            int src = t + offset;
            if (lane + offset < WARP && src < nthreads) {
                n_resp[t] = t_resp[src];
                n_x[t] = t_x[src];
                n_y[t] = t_y[src];
            } else { /* out of warp: own value; non-existent lane: never wins (Q6) */
                n_resp[t] = t_resp[t];
                n_x[t] = t_x[t];
                n_y[t] = t_y[t];
            }
        }
        for (int t = 0; t < nthreads; t++)
            if (n_resp[t] > t_resp[t]) {
                t_resp[t] = n_resp[t];
                t_x[t] = n_x[t];
                t_y[t] = n_y[t];
            }
    }
    /* shared memory, lane 0 of each warp :218-227; thread (0,0) scans :230-244 */
    float max_resp = t_resp[0], max_x = t_x[0], max_y = t_y[0];
    for (int i = 1; i < warp_cnt; i++) {
        int t = i * 32;
        if (t_resp[t] > max_resp) {
            max_resp = t_resp[t];
            max_x = t_x[t];
            max_y = t_y[t];
        }
    }
    float scale = (float)(1 << level);
    if (d_score[cell_id] < max_resp) { /* :246-252 */
        d_score[cell_id] = max_resp;
        d_pos[2 * cell_id] = max_x * scale;
        d_pos[2 * cell_id + 1] = max_y * scale;
        d_level[cell_id] = level;
    }
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

void oracle_grid_nms(const oracle_level *lv, int n_levels, int cell, float *pos, float *score,
                     int32_t *level_out)
{
    /* :262-263 with 32 -> cell */
    const int hcells = (lv[0].width % cell == 0) ? lv[0].width / cell : lv[0].width / cell + 1;
    const int vcells = (lv[0].height % cell == 0) ? lv[0].height / cell : lv[0].height / cell + 1;
    for (int level = 0; level < n_levels; level++) {
        const int cw = cell >> level, ch = cell >> level; /* :266-267 */
        if (cw == 0) break;                              /* EXT i */
        const int bdx = cw;
        const int bdy = imax(1, imin(128 / cw, ch)); /* :272-273 */
        for (int by = 0; by < vcells; by++)
            for (int bx = 0; bx < hcells; bx++)
                nms_block(level, bx, by, hcells, lv[level].width, lv[level].height, cw, ch, bdx,
                          bdy, lv[level].response_pitch, lv[level].response, pos, score,
                          level_out);
    }
}

/* a7  detect   src/cuda/fast.cu:374-407 */
void oracle_detect(const oracle_level *lv, int n_levels, int cell, const uint8_t *lut,
                   float threshold, float *pos, float *score, int32_t *level_out)
{
    for (int l = 0; l < n_levels; l++)
        oracle_fast_calc_corner_response(lv[l].width, lv[l].height, lv[l].image_pitch,
                                         lv[l].image, 3, 3, lut, threshold,
                                         lv[l].response_pitch, lv[l].response);
    oracle_grid_nms(lv, n_levels, cell, pos, score, level_out);
}

/* ------------------------------------------------------------------------------------
 * a8  compute_fast_angle_kernel     src/cuda/orb.cu:77-142
 * 32 threads = 31 patch columns (+1 idle); per-thread float partial sums (exact: every
 * partial and the total are integers below 2^24), 32-lane __shfl_down sum, atan2f.
 * Empty cells (score == 0): angle 0 (Q5).  `score` may be NULL (= compute all).
 * ------------------------------------------------------------------------------------ */
void oracle_compute_fast_angle(float *angle, const float *pos, const float *score,
                               const uint8_t *image, int image_pitch, int image_width,
                               int image_height, int n)
{
    for (int idx = 0; idx < n; idx++) {
        if (score && !(score[idx] > 0.0f)) {
            angle[idx] = 0.0f;
            continue;
        }
        int k_x = (int)floor((double)pos[2 * idx] + 0.5);
        int k_y = (int)floor((double)pos[2 * idx + 1] + 0.5);
        int r2 = 15 * 15;
        float m10_t[32], m01_t[32];
        for (int tid = 0; tid < 32; tid++) {
            float m10 = 0, m01 = 0;
            if (tid < 31) { /* :94-102 */
                int mult_dx = tid - 15;
                int tdx = tid + k_x - 15;
                if (tdx > 0 && tdx < image_width)
                    m10 = (float)(mult_dx * image[(size_t)k_y * image_pitch + tdx]);
            }
            for (int dy = 1; dy < 16; dy++) { /* :104-126 */
                int dx = (int)floor((double)sqrtf((float)r2 - (float)(dy * dy)) + 0.5);
                if (tid > 14 - dx && tid < 16 + dx) {
                    int mult_dx = tid - 15;
                    int tdx = k_x + tid - 15;
                    if (k_y - dy > 0 && tdx > 0 && tdx < image_width) {
                        float i = image[(size_t)(k_y - dy) * image_pitch + tdx];
                        m01 -= dy * i;
                        m10 += mult_dx * i;
                    }
                    if (k_y + dy < image_height && tdx > 0 && tdx < image_width) {
                        float i = image[(size_t)(k_y + dy) * image_pitch + tdx];
                        m01 += dy * i;
                        m10 += mult_dx * i;
                    }
                }
            }
            m10_t[tid] = m10;
            m01_t[tid] = m01;
        }
        for (int offset = 16; offset > 0; offset /= 2) /* :130-134 */
            for (int t = 0; t + offset < 32; t++) {
                /* lanes with t+offset >= 32 receive their own value; lane 0 never depends on
                 * them, and only lane 0 is read (:138). */
                if (t < offset) {
                    m01_t[t] += m01_t[t + offset];
                    m10_t[t] += m10_t[t + offset];
                }
            }
        angle[idx] = orbfe_atan2f(m01_t[0], m10_t[0]); /* :140 */
    }
}

/* ------------------------------------------------------------------------------------
 * a9  calc_orb_kernel (+ GET_VALUE)   src/cuda/orb.cu:12-14, :17-75
 * a10 compress_descriptors_kernel     src/cuda/orb.cu:145-169
 * 32 threads = 32 descriptor bytes; pattern viewed as 512 int2 points, thread t uses
 * points 16t .. 16t+15 (:40).  Products and the single add/sub are separate f32
 * operations (no contraction, Q7 decision); __float2int_rn = round half to even.
 * EXT v (angle_in_radians): the stored angle is used directly and the border guard grows
 * from 17 to 19 because the rotated pattern then reaches 18 px.
 * ------------------------------------------------------------------------------------ */
static void calc_orb_impl(const float *d_angle, const float *d_pos, uint8_t *desc_tmp,
                          uint32_t *desc32, const uint8_t *image, int image_pitch, int image_width,
                          int image_height, int n, int angle_in_radians, int fma)
{
    for (int id = 0; id < n; id++) {
        uint8_t *desc = desc_tmp + (size_t)id * 32;
        short lx = (short)d_pos[2 * id], ly = (short)d_pos[2 * id + 1]; /* :32 */
        int zero;
        if (!angle_in_radians)
            zero = (lx < 17 || lx > image_width - 17 || ly < 17 || ly > image_height - 17);
        else
            zero = (lx < 19 || lx > image_width - 20 || ly < 19 || ly > image_height - 20);
        if (zero) {
            memset(desc, 0, 32); /* :34-38 */
        } else {
            const float factorPI = (float)(3.141592654f / 180.f); /* :42 */
            float ang = angle_in_radians ? d_angle[id] : d_angle[id] * factorPI;
            float a, b;
            orbfe_sincosf(ang, &b, &a); /* a = cos, b = sin :45-46 */
            for (int tid = 0; tid < 32; tid++) {
                const int8_t *pt = g_pattern + 2 * (16 * tid); /* int2 index 16*tid */
                int val = 0;
                for (int k = 0; k < 8; k++) {
                    int t[2];
                    for (int e = 0; e < 2; e++) {
                        float pxf = (float)pt[2 * (2 * k + e)];
                        float pyf = (float)pt[2 * (2 * k + e) + 1];
                        float m1 = pxf * b, m2 = pyf * a, m3 = pxf * a, m4 = pyf * b;
                        float fr = m1 + m2, fc = m3 - m4;
                        if (fma) { /* what nvcc's default contraction most likely makes of orb.cu:13-14 */
                            fr = fmaf(pxf, b, m2);
                            fc = fmaf(pxf, a, -m4);
                        }
                        int row = ly + orbfe_rn_int(fr);
                        int col = lx + orbfe_rn_int(fc);
                        t[e] = image[(size_t)row * image_pitch + col];
                    }
                    val |= (t[0] < t[1]) << k;
                }
                desc[tid] = (uint8_t)val;
            }
        }
        if (desc32) { /* :149-167 */
            uint32_t d = 0;
            for (int i = 0; i < 32; i++)
                if (desc[i] == 1) d |= (1u << i);
            desc32[id] = d;
        }
    }
}

void oracle_calc_orb(const float *d_angle, const float *d_pos, uint8_t *desc_tmp, uint32_t *desc32,
                     const uint8_t *image, int image_pitch, int image_width, int image_height, int n,
                     int angle_in_radians)
{
    calc_orb_impl(d_angle, d_pos, desc_tmp, desc32, image, image_pitch, image_width, image_height, n, angle_in_radians, 0);
}

/* The same with GET_VALUE's sums contracted into FMAs, fmaf(x, b, y * a) and fmaf(x, a, -(y * b)): nvcc
 * fuses by default (the reference's CMakeLists sets no -fmad=false), which products it fuses is not
 * observable here.  NOT the parity definition (that is oracle_calc_orb, no contraction); it exists so
 * that tests/test_fma_exposure.py can count how many descriptor bits the decision can move. */
void oracle_calc_orb_fma(const float *d_angle, const float *d_pos, uint8_t *desc_tmp, uint32_t *desc32,
                         const uint8_t *image, int image_pitch, int image_width, int image_height, int n,
                         int angle_in_radians)
{
    calc_orb_impl(d_angle, d_pos, desc_tmp, desc32, image, image_pitch, image_width, image_height, n, angle_in_radians, 1);
}

/* ------------------------------------------------------------------------------------
 * a11  kernel_match_keypoints   src/cuda/post_processing.cu:92-200 (host :234-341)
 * Blocks of 32 threads over prev; curr scanned in shared-memory tiles of 32; in a tile of
 * m entries thread tid visits j = (s + tid) % m for s = 0..m-1, only if tid < m (Q8).
 * Decisions: the tile is taken as fully loaded even when the last prev block has fewer
 * than 32 live threads (the reference then reads stale shared memory: UB, Q9); output is
 * match_idx[i] (curr index or -1) instead of atomically compacted lists; returns the count.
 * pos_prev is the reprojected prev position (kernel_reproject_prev_points, out of scope).
 * ------------------------------------------------------------------------------------ */
int oracle_match_keypoints(const float *pos_prev, const uint32_t *desc_prev, int n_prev,
                           const float *pos_curr, const uint32_t *desc_curr, int n_curr,
                           int max_pixel_distance, int max_hamming_distance,
                           int32_t *match_idx)
{
    int matched = 0;
    for (int idx = 0; idx < n_prev; idx++) {
        const int tid = idx & 31;
        int pair_idx = 0, is_matched = 0;
        int hamming_distance_curr = 9999999;
        const uint32_t descriptor = desc_prev[idx];
        const float ppx = pos_prev[2 * idx], ppy = pos_prev[2 * idx + 1];
        for (int i = 0; i < n_curr; i += WARP) {
            int max_j_loop = WARP;
            if (i + WARP >= n_curr) max_j_loop = n_curr - i;
            for (int j = 0; j < max_j_loop; j++) {
                if (tid < max_j_loop) {
                    int j_idx = (j + tid) % max_j_loop;
                    const float cx = pos_curr[2 * (i + j_idx)], cy = pos_curr[2 * (i + j_idx) + 1];
                    if (fabsf(ppx - cx) <= (float)max_pixel_distance &&
                        fabsf(ppy - cy) <= (float)max_pixel_distance) {
                        int hd = __builtin_popcount(descriptor ^ desc_curr[i + j_idx]);
                        if (hd < max_hamming_distance && hd < hamming_distance_curr) {
                            hamming_distance_curr = hd;
                            is_matched = 1;
                            pair_idx = i + j_idx;
                        }
                    }
                }
            }
        }
        match_idx[idx] = is_matched ? pair_idx : -1;
        matched += is_matched;
    }
    return matched;
}

/* ------------------------------------------------------------------------------------
 * a11 / a12 output contract of kernel_match_keypoints    src/cuda/post_processing.cu:176-198
 * A matched prev keypoint idx with partner pair_idx appends
 *     previous_matched_points[slot] = previous_points[idx]            (double3)
 *     current_matched_points[slot]  = current_points[pair_idx]        (double3)
 *     d_pos_frame[slot]             = uint16_t(d_pos[pair_idx].x)
 *     d_pos_frame[keypoints_num + slot] = uint16_t(d_pos[pair_idx].y)
 * where slot comes from two atomicAdds (arbitrary order).  Decision: slots are taken in ascending
 * prev index.  The two halves of d_pos_frame are the frame's keypoints_x / keypoints_y
 * (post_processing.cu:300-331, types.h:29-30).  Points may be NULL (RGB-D only).  Returns the count.
 * ------------------------------------------------------------------------------------ */
int oracle_match_compact(const int32_t *match_idx, int n_prev, const double *points_prev,
                         const double *points_curr, const float *pos_curr, double *prev_matched,
                         double *curr_matched, uint16_t *keypoints_x, uint16_t *keypoints_y)
{
    int slot = 0;
    for (int idx = 0; idx < n_prev; idx++) {
        const int pair_idx = match_idx[idx];
        if (pair_idx < 0) continue;
        if (points_prev && prev_matched)
            for (int k = 0; k < 3; k++) prev_matched[3 * slot + k] = points_prev[3 * idx + k];
        if (points_curr && curr_matched)
            for (int k = 0; k < 3; k++) curr_matched[3 * slot + k] = points_curr[3 * pair_idx + k];
        keypoints_x[slot] = (uint16_t)pos_curr[2 * pair_idx];
        keypoints_y[slot] = (uint16_t)pos_curr[2 * pair_idx + 1];
        slot++;
    }
    return slot;
}

/* ------------------------------------------------------------------------------------
 * f4 (part)  kernel_reproject_prev_points + project_point_to_pixel_double
 *            src/cuda/post_processing.cu:11-43, :72-90
 * e = T * (x, y, z, 1) in double (T column-major as Eigen::Matrix4d stores it), then the
 * librealsense projection in FLOAT: x = e0 / e2, y = e1 / e2 (double quotients narrowed),
 * optional modified-Brown-Conrady / f-theta distortion, pixel = x * fx + ppx.
 * PARITY UNPINNED at the ulp level: Eigen's product order and nvcc's FMA contraction are not
 * observable here; decided as ((T_i0 x + T_i1 y) + T_i2 z) + T_i3, no contraction, and the float
 * polynomial evaluated left to right as written.
 * ------------------------------------------------------------------------------------ */
void oracle_reproject_points(float *pos_out, const double *points_prev, int n, const double *T,
                             const oracle_intrinsics *intrin)
{
    ORBFE_NO_CONTRACT
    for (int idx = 0; idx < n; idx++) {
        const double px = points_prev[3 * idx], py = points_prev[3 * idx + 1], pz = points_prev[3 * idx + 2];
        double e[3];
        for (int i = 0; i < 3; i++) {
            double t = T[i] * px + T[4 + i] * py;
            t = t + T[8 + i] * pz;
            t = t + T[12 + i];
            e[i] = t;
        }
        float x = (float)(e[0] / e[2]), y = (float)(e[1] / e[2]);
        if (intrin->model == 1) { /* RS2_DISTORTION_MODIFIED_BROWN_CONRADY */
            const float r2 = x * x + y * y;
            float f = 1 + intrin->coeffs[0] * r2;
            f = f + intrin->coeffs[1] * r2 * r2;
            f = f + intrin->coeffs[4] * r2 * r2 * r2;
            x *= f;
            y *= f;
            float dx = x + 2 * intrin->coeffs[2] * x * y;
            dx = dx + intrin->coeffs[3] * (r2 + 2 * x * x);
            float dy = y + 2 * intrin->coeffs[3] * x * y;
            dy = dy + intrin->coeffs[2] * (r2 + 2 * y * y);
            x = dx;
            y = dy;
        }
        if (intrin->model == 3) orbfe_ftheta_distort(&x, &y, intrin->coeffs[0]); /* RS2_DISTORTION_FTHETA :32-38 */
        pos_out[2 * idx] = x * intrin->fx + intrin->ppx;
        pos_out[2 * idx + 1] = y * intrin->fy + intrin->ppy;
    }
}

/* ------------------------------------------------------------------------------------
 * f2  kernel_keypoint_pixel_to_point    src/cuda/cuda-align.cu:282-364
 *     deproject_pixel_to_point_double   src/cuda/cuda-align.cu:85-112
 * Keep keypoints with depth > 1 and score > 1.0f; deproject in double.  The pixel offsets
 * (pixel - pp) / f are FLOAT operations (float operands) widened afterwards; the distortion
 * polynomial is double with float coefficients, evaluated left to right without contraction.
 * The depth lookup uses int(pos.y + 0.5) for BOTH row and column (:332, a reference bug kept in
 * parity mode).  Decision: compaction is by ascending keypoint index (the reference's two-level
 * atomicAdd gives an arbitrary order).  Returns the number of valid keypoints.
 * ------------------------------------------------------------------------------------ */
int oracle_keypoint_pixel_to_point(const uint32_t *aligned_depth, const oracle_intrinsics *intrin,
                                   int image_width, int image_height, float *pos_out,
                                   const float *pos_in, const float *score_in, double *points,
                                   uint32_t *desc_out, const uint32_t *desc_in, int n,
                                   int fix_depth_index)
{
    int count = 0;
    (void)image_height;
    for (int idx = 0; idx < n; idx++) {
        const float px = pos_in[2 * idx], py = pos_in[2 * idx + 1];
        const float score = score_in[idx];
        const int row = (int)((double)py + 0.5);
        const int col = fix_depth_index ? (int)((double)px + 0.5) : (int)((double)py + 0.5);
        const int depth = (int)aligned_depth[(size_t)row * image_width + col];
        if (!(depth > 1 && score > 1.0f)) continue;
        const float fdepth = (float)depth;
        double x = (double)((px - intrin->ppx) / intrin->fx);
        double y = (double)((py - intrin->ppy) / intrin->fy);
        if (intrin->model == 2) { /* RS2_DISTORTION_INVERSE_BROWN_CONRADY */
            const double c0 = intrin->coeffs[0], c1 = intrin->coeffs[1], c2 = intrin->coeffs[2],
                         c3 = intrin->coeffs[3], c4 = intrin->coeffs[4];
            double r2 = x * x + y * y;
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
        const double depth_d = (double)fdepth;
        points[3 * count + 0] = depth_d * x;
        points[3 * count + 1] = depth_d * y;
        points[3 * count + 2] = depth_d;
        desc_out[count] = desc_in[idx];
        pos_out[2 * count] = px;
        pos_out[2 * count + 1] = py;
        count++;
    }
    return count;
}

/* ------------------------------------------------------------------------------------
 * f2 (the producing half)  align_depth_to_other            src/cuda/cuda-align.cu:366-399
 *     kernel_map_depth_to_other + kernel_transfer_pixels   :163-188, :121-161
 *     deproject / transform / project (float)              :57-81, :112-119, :23-54
 *     kernel_reset_to_max, kernel_depth_to_other, kernel_reset_to_zero   :257-267, :224-255, :269-280
 * The four launches are restated one after the other, the int2 map included (the reference's
 * scratch, returned here so that the tests can look at it).  All launches use ONE grid made from
 * image_width / image_height (32 x 32 threads, ceil-div, :378-380) while every bound inside the
 * kernels is an intrinsics field, so:
 *   - a depth pixel is mapped and splatted iff x < min(gx, depth.width), y < min(gy, depth.height),
 *     gx = 32 ceil(image_width / 32), gy likewise; depth_in and the map are indexed with depth.width;
 *   - an output pixel is reset to 9999999 (and back to 0 at the end) iff x < min(gx, other.width),
 *     y < min(gy, other.height); output pixels outside keep min(what the caller left there, splats).
 * Arithmetic: float, left to right as written, NO contraction (nvcc's default -fmad may fuse; not
 * observable here: PARITY UNPINNED at the ulp level, as for a8 / a9); `static_cast<int>(v + 0.5f)`
 * is CUDA's cvt.rzi.s32.f32 -- truncation that SATURATES and maps NaN to 0 (other_point z = 0 gives
 * inf / NaN pixels, which the reference then treats like any other pair of corners).
 * The f-theta branch of project_point_to_pixel (:44-50) is orbfe_ftheta_distort (include/orbfe_math.h: float atanf /
 * tanf, the overloads nvcc picks for float operands; the build's deterministic versions, unpinned at the ulp level);
 * deprojecting from model 1 or 3 trips the reference's device asserts (:62-63): return -1.
 * kernel_depth_to_other's atomicMin over a rectangle is order-free: the result is the minimum raw
 * depth over all depth pixels whose mapped rectangle [p0, p1] covers the output pixel.
 * ------------------------------------------------------------------------------------ */
static int cvt_rzi_s32_f32(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int)(-2147483647 - 1);
    return (int)f;
}

/* kernel_transfer_pixels for one corner (block_index 0: shift -0.5, 1: +0.5), cuda-align.cu:121-161 */
static void oracle_transfer_pixel(int32_t *mapped_xy, const oracle_intrinsics *din,
                                  const oracle_intrinsics *oin, const oracle_extrinsics *e,
                                  float depth_val, int depth_x, int depth_y, int block_index)
{
    ORBFE_NO_CONTRACT
    const float shift = block_index ? 0.5f : -0.5f;
    mapped_xy[0] = -1;
    mapped_xy[1] = -1;
    if (depth_val != 0) {
        const float depth_pixel[2] = {(float)depth_x + shift, (float)depth_y + shift};
        float depth_point[3], other_point[3], other_pixel[2];
        /* deproject_pixel_to_point :57-81 */
        float x = (depth_pixel[0] - din->ppx) / din->fx;
        float y = (depth_pixel[1] - din->ppy) / din->fy;
        if (din->model == 2) { /* RS2_DISTORTION_INVERSE_BROWN_CONRADY */
            const float *c = din->coeffs;
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
        depth_point[0] = depth_val * x;
        depth_point[1] = depth_val * y;
        depth_point[2] = depth_val;
        /* transform_point_to_point :112-119 (rotation column-major) */
        for (int i = 0; i < 3; i++) {
            float t = e->rotation[i] * depth_point[0] + e->rotation[3 + i] * depth_point[1];
            t = t + e->rotation[6 + i] * depth_point[2];
            t = t + e->translation[i];
            other_point[i] = t;
        }
        /* project_point_to_pixel :23-54 */
        x = other_point[0] / other_point[2];
        y = other_point[1] / other_point[2];
        if (oin->model == 1) { /* RS2_DISTORTION_MODIFIED_BROWN_CONRADY */
            const float *c = oin->coeffs;
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
        if (oin->model == 3) orbfe_ftheta_distort(&x, &y, oin->coeffs[0]); /* RS2_DISTORTION_FTHETA :44-50 */
        other_pixel[0] = x * oin->fx + oin->ppx;
        other_pixel[1] = y * oin->fy + oin->ppy;
        mapped_xy[0] = cvt_rzi_s32_f32(other_pixel[0] + 0.5f);
        mapped_xy[1] = cvt_rzi_s32_f32(other_pixel[1] + 0.5f);
    }
}

int oracle_align_depth_to_other(uint32_t *aligned_out, const uint16_t *depth_in, int32_t *pixel_map,
                                float depth_scale, int image_width, int image_height,
                                const oracle_intrinsics *din, const oracle_intrinsics *oin,
                                const oracle_extrinsics *e)
{
    if (din->model == 1 || din->model == 3) return -1;
    const int gx = 32 * ((image_width + 31) / 32), gy = 32 * ((image_height + 31) / 32);
    const int dw = din->width, dh = din->height, ow = oin->width, oh = oin->height;
    const int mx = gx < dw ? gx : dw, my = gy < dh ? gy : dh; /* depth pixels the grid reaches */
    const int rx = gx < ow ? gx : ow, ry = gy < oh ? gy : oh; /* output pixels the grid reaches */
    const size_t depth_size = (size_t)dw * dh;
    int32_t *map = pixel_map ? pixel_map : (int32_t *)malloc(depth_size * 2 * 2 * sizeof(int32_t));
    /* kernel_map_depth_to_other, blockIdx.z = 0, 1 */
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < my; y++)
            for (int x = 0; x < mx; x++) {
                const size_t i = (size_t)y * dw + x;
                const float depth_val = (float)depth_in[i] * depth_scale;
                oracle_transfer_pixel(map + 2 * ((size_t)z * depth_size + i), din, oin, e, depth_val, x, y, z);
            }
    /* kernel_reset_to_max */
    for (int y = 0; y < ry; y++)
        for (int x = 0; x < rx; x++) aligned_out[(size_t)y * ow + x] = 9999999u;
    /* kernel_depth_to_other */
    for (int y = 0; y < my; y++)
        for (int x = 0; x < mx; x++) {
            const size_t i = (size_t)y * dw + x;
            const int32_t *p0 = map + 2 * i, *p1 = map + 2 * (depth_size + i);
            if (p0[0] < 0 || p0[1] < 0 || p1[0] >= ow || p1[1] >= oh) continue;
            const uint32_t new_val = depth_in[i];
            for (int v = p0[1]; v <= p1[1]; v++)
                for (int u = p0[0]; u <= p1[0]; u++) {
                    uint32_t *o = aligned_out + (size_t)v * ow + u;
                    if (new_val < *o) *o = new_val; /* atomicMin */
                }
        }
    /* kernel_reset_to_zero */
    for (int y = 0; y < ry; y++)
        for (int x = 0; x < rx; x++)
            if (aligned_out[(size_t)y * ow + x] == 9999999u) aligned_out[(size_t)y * ow + x] = 0;
    if (!pixel_map) free(map);
    return 0;
}

/* EXT C.9: brute-force 256-bit Hamming, lexicographic (dist, idx) minimum.
 * 256 bits = 4 x 64-bit popcounts.  The loop is instantiated twice, once for the hardware popcnt
 * instruction (picked at run time when the host has it) so that the CPU baseline is a fair port
 * wherever the library lands; both give the same results. */
#define ORACLE_MATCH256_BODY                                                                       \
    for (int i = 0; i < nA; i++) {                                                                 \
        int best = 1 << 30, best_j = -1;                                                           \
        uint64_t wa[4];                                                                            \
        memcpy(wa, descA + 32 * (size_t)i, 32);                                                    \
        for (int j = 0; j < nB; j++) {                                                             \
            if (window >= 0) {                                                                     \
                if (fabsf(posA[2 * i] - posB[2 * j]) > (float)window ||                            \
                    fabsf(posA[2 * i + 1] - posB[2 * j + 1]) > (float)window)                      \
                    continue;                                                                      \
            }                                                                                      \
            uint64_t wb[4];                                                                        \
            memcpy(wb, descB + 32 * (size_t)j, 32);                                                \
            const int d = __builtin_popcountll(wa[0] ^ wb[0]) + __builtin_popcountll(wa[1] ^ wb[1]) + \
                          __builtin_popcountll(wa[2] ^ wb[2]) + __builtin_popcountll(wa[3] ^ wb[3]);  \
            if (d < best) {                                                                        \
                best = d;                                                                          \
                best_j = j;                                                                        \
            }                                                                                      \
        }                                                                                          \
        if (best_j >= 0 && best <= max_dist) {                                                     \
            idx[i] = best_j;                                                                       \
            dist[i] = best;                                                                        \
        } else {                                                                                   \
            idx[i] = -1;                                                                           \
            dist[i] = -1;                                                                          \
        }                                                                                          \
    }

static void match256_generic(const uint8_t *descA, const float *posA, int nA, const uint8_t *descB,
                             const float *posB, int nB, int window, int max_dist, int32_t *idx, int32_t *dist)
{
    ORACLE_MATCH256_BODY
}
#if defined(__x86_64__)
__attribute__((target("popcnt"))) static void match256_popcnt(const uint8_t *descA, const float *posA, int nA,
                                                              const uint8_t *descB, const float *posB, int nB,
                                                              int window, int max_dist, int32_t *idx, int32_t *dist)
{
    ORACLE_MATCH256_BODY
}
#endif

void oracle_match256(const uint8_t *descA, const float *posA, int nA, const uint8_t *descB,
                     const float *posB, int nB, int window, int max_dist, int32_t *idx,
                     int32_t *dist)
{
#if defined(__x86_64__)
    __builtin_cpu_init();
    if (__builtin_cpu_supports("popcnt")) {
        match256_popcnt(descA, posA, nA, descB, posB, nB, window, max_dist, idx, dist);
        return;
    }
#endif
    match256_generic(descA, posA, nA, descB, posB, nB, window, max_dist, idx, dist);
}

/* ------------------------------------------------------------------------------------
 * Whole pipeline on one frame, in the call order of
 * src/SlamGpuPipeline/buildStream.cpp:424-460 (blur -> levels -> detect -> angle -> orb).
 * Level sizes halve with floor (buildStream.cpp:312-336).
 * ------------------------------------------------------------------------------------ */
int oracle_num_cells(const oracle_config *cfg)
{
    int gc = (cfg->width + cfg->cell - 1) / cfg->cell; /* buildStream.cpp:233-235 */
    int gr = (cfg->height + cfg->cell - 1) / cfg->cell;
    return gc * gr;
}

size_t oracle_level_dims(const oracle_config *cfg, int level, int *w, int *h)
{
    int ww = cfg->width, hh = cfg->height;
    for (int i = 0; i < level; i++) {
        ww /= 2;
        hh /= 2;
    }
    if (w) *w = ww;
    if (h) *h = hh;
    return (size_t)ww * hh;
}

typedef struct {
    float score;
    int cell;
} sel_t;

static int sel_cmp(const void *a, const void *b)
{
    const sel_t *x = (const sel_t *)a, *y = (const sel_t *)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1; /* score desc */
    return x->cell - y->cell;                                      /* cell asc  */
}

int oracle_extract_frame(const oracle_config *cfg, const uint8_t *gray, int gray_pitch,
                         uint8_t **pyr_out, float *pos_o, float *score_o, int32_t *level_o,
                         float *angle_o, uint8_t *desc_o, uint32_t *desc32_o,
                         oracle_keypoint *records)
{
    const int L = cfg->levels, K = oracle_num_cells(cfg);
    oracle_level *lv = (oracle_level *)calloc((size_t)L, sizeof(oracle_level));
    for (int l = 0; l < L; l++) {
        int w, h;
        oracle_level_dims(cfg, l, &w, &h);
        lv[l].width = w;
        lv[l].height = h;
        lv[l].image_pitch = w > 0 ? w : 1;
        lv[l].response_pitch = w > 0 ? w : 1;
        lv[l].image = (uint8_t *)calloc((size_t)(w * h) + 1, 1);
        lv[l].response = (float *)calloc((size_t)(w * h) + 1, sizeof(float));
    }
    uint8_t *lut = (uint8_t *)malloc(65536);
    float *pos = (float *)calloc((size_t)K * 2, sizeof(float));
    float *score = (float *)calloc((size_t)K, sizeof(float));
    int32_t *level = (int32_t *)calloc((size_t)K, sizeof(int32_t));
    float *angle = (float *)calloc((size_t)K, sizeof(float));
    uint8_t *desc = (uint8_t *)calloc((size_t)K * 32, 1);
    uint32_t *desc32 = (uint32_t *)calloc((size_t)K, sizeof(uint32_t));
    float *sel_score = (float *)calloc((size_t)K, sizeof(float));

    oracle_gaussian_blur_3x3(lv[0].image, lv[0].image_pitch, gray, gray_pitch, cfg->width,
                             cfg->height);
    oracle_pyramid_create_levels(lv, L);
    oracle_fast_calculate_lut(lut, cfg->min_arc);
    /* detection only on levels whose cell is >= 1 px (EXT i) and that hold a pixel */
    int Ld = 0;
    while (Ld < L && (cfg->cell >> Ld) > 0 && lv[Ld].width > 0 && lv[Ld].height > 0) Ld++;
    oracle_detect(lv, Ld, cfg->cell, lut, cfg->fast_threshold, pos, score, level);

    /* selection: every non-empty cell, or the top-N by (score desc, cell asc) (EXT ii) */
    memcpy(sel_score, score, (size_t)K * sizeof(float));
    if (cfg->max_features > 0) {
        sel_t *s = (sel_t *)malloc((size_t)K * sizeof(sel_t));
        int n = 0;
        for (int k = 0; k < K; k++)
            if (score[k] > 0.0f) {
                s[n].score = score[k];
                s[n].cell = k;
                n++;
            }
        qsort(s, (size_t)n, sizeof(sel_t), sel_cmp);
        for (int i = cfg->max_features; i < n; i++) sel_score[s[i].cell] = 0.0f;
        free(s);
    }
    /* orientation + descriptor, selected cells only: on the level-0 image at the level-0 position (Q10,
     * buildStream.cpp:442-460), or -- EXT iv, descriptor_level -- on the level that won the cell at
     * pos / 2^level (exact: pos = level coordinate << level, nms.cu:246-252), with that level's width and
     * height in every bound and guard band of orb.cu:77-142 and :17-75 */
    for (int k = 0; k < K; k++) {
        if (!(sel_score[k] > 0.0f)) continue; /* zero descriptor, zero desc32, zero angle (Q5) -- calloc */
        const int l = cfg->descriptor_level ? level[k] : 0;
        const float pl[2] = {pos[2 * k] / (float)(1 << l), pos[2 * k + 1] / (float)(1 << l)};
        oracle_compute_fast_angle(angle + k, pl, NULL, lv[l].image, lv[l].image_pitch, lv[l].width,
                                  lv[l].height, 1);
        oracle_calc_orb(angle + k, pl, desc + 32 * (size_t)k, desc32 + k, lv[l].image,
                        lv[l].image_pitch, lv[l].width, lv[l].height, 1, cfg->angle_in_radians);
    }
    int count = 0;
    for (int k = 0; k < K; k++)
        if (sel_score[k] > 0.0f) {
            if (records) {
                oracle_keypoint *r = records + count;
                r->x = pos[2 * k];
                r->y = pos[2 * k + 1];
                r->score = score[k];
                r->level = level[k];
                r->angle = angle[k];
                memcpy(r->desc, desc + 32 * (size_t)k, 32);
            }
            count++;
        }

    if (pyr_out)
        for (int l = 0; l < L; l++)
            if (pyr_out[l]) memcpy(pyr_out[l], lv[l].image, (size_t)lv[l].width * lv[l].height);
    if (pos_o) memcpy(pos_o, pos, (size_t)K * 2 * sizeof(float));
    if (score_o) memcpy(score_o, score, (size_t)K * sizeof(float));
    if (level_o) memcpy(level_o, level, (size_t)K * sizeof(int32_t));
    if (angle_o) memcpy(angle_o, angle, (size_t)K * sizeof(float));
    if (desc_o) memcpy(desc_o, desc, (size_t)K * 32);
    if (desc32_o) memcpy(desc32_o, desc32, (size_t)K * sizeof(uint32_t));

    for (int l = 0; l < L; l++) {
        free(lv[l].image);
        free(lv[l].response);
    }
    free(lv);
    free(lut);
    free(pos);
    free(score);
    free(level);
    free(angle);
    free(desc);
    free(desc32);
    free(sel_score);
    return count;
}
