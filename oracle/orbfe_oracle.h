/* orbfe_oracle.h -- CPU oracle for the ORB front-end hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 * The product library (liborbfe.so) never links, loads or falls back to it.
 *
 * PARITY UNPINNED BY THE REFERENCE: dsvua/jetracer-orbslam2 holds no tests, golden
 * vectors or fixtures for this path and its only implementation
 * is CUDA (src/cuda/ *.cu), which cannot be compiled in this image (needs nvcc,
 * helper_cuda.h, librealsense2, Eigen).  This file is therefore a statement-by-statement
 * CPU restatement of those kernels with the determinisation decisions of SURVEY.md
 * Appendix A, pinned by the known answers derivable from the source (Appendix B) in
 * tests/test_kat.py, by definition-level numpy / scipy restatements of blur, pyramid, FAST,
 * orientation and rBRIEF that share nothing with this file (same tests file), and by committed
 * digests in tests/golden/.
 */
#ifndef ORBFE_ORACLE_H
#define ORBFE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_level {
    int width, height;
    int image_pitch;    /* bytes */
    uint8_t *image;
    int response_pitch; /* elements (floats) */
    float *response;
} oracle_level;

typedef struct oracle_config {
    int width, height;
    int levels;           /* pyramid levels built */
    int cell;             /* level-0 NMS cell; reference 32 */
    float fast_threshold; /* reference 13.0f */
    int min_arc;          /* reference 12 */
    int max_features;     /* 0 = every non-empty cell (reference); >0 = top-N */
    int angle_in_radians; /* 0 = reference quirk Q7 */
    int descriptor_level; /* 0 = reference quirk Q10 (orientation + descriptor always on level 0,
                             buildStream.cpp:442-460); 1 = EXT iv: on the pyramid level that won the cell,
                             at pos / 2^level, image bounds and guard bands in that level's coordinates */
} oracle_config;

/* 52-byte keypoint record */
typedef struct oracle_keypoint {
    float x, y;
    float score;
    int32_t level;
    float angle;
    uint8_t desc[32];
} oracle_keypoint;

/* ---- one function per reference kernel / host function ---- */
/* "next" row f1: kernel_rgb_to_grayscale, src/cuda/cuda_RGB_to_Grayscale.cu:10-23 */
void oracle_rgb_to_grayscale(uint8_t *dst, const uint8_t *src, int cols, int rows, int dst_pitch,
                             int src_pitch);
void oracle_gaussian_blur_3x3(uint8_t *blurred, int blurred_pitch, const uint8_t *image,
                              int image_pitch, int w, int h);
void oracle_halfsample(const uint8_t *src, int src_pitch, uint8_t *dst, int dst_pitch,
                       int dst_w, int dst_h);
void oracle_pyramid_create_levels(const oracle_level *levels, int n_levels);
int oracle_fast_is_corner(uint32_t mask, int min_arc);
void oracle_fast_calculate_lut(uint8_t *lut, int min_arc);
void oracle_fast_calc_corner_response(int w, int h, int pitch, const uint8_t *img, int hb,
                                      int vb, const uint8_t *lut, float threshold,
                                      int resp_pitch_elems, float *resp);
/* ... with the reference's fast_score argument: 0 SUM_OF_ABS_DIFF_ALL (fast.cu:233-241), 1 SUM_OF_ABS_DIFF_ON_ARC
 * (:243-255), 2 MAX_THRESHOLD (:256-283, bisection with fast_gpu_is_corner_quick :126-148) */
void oracle_fast_calc_corner_response_score(int w, int h, int pitch, const uint8_t *img, int hb,
                                            int vb, const uint8_t *lut, float threshold, int score,
                                            int resp_pitch_elems, float *resp);
void oracle_grid_nms(const oracle_level *levels, int n_levels, int cell, float *pos /*x,y*/,
                     float *score, int32_t *level);
void oracle_detect(const oracle_level *levels, int n_levels, int cell, const uint8_t *lut,
                   float threshold, float *pos, float *score, int32_t *level);
void oracle_compute_fast_angle(float *angle, const float *pos, const float *score,
                               const uint8_t *img, int pitch, int w, int h, int n);
void oracle_calc_orb(const float *angle, const float *pos, uint8_t *desc_tmp, uint32_t *desc32,
                     const uint8_t *img, int pitch, int w, int h, int n, int angle_in_radians);
/* calc_orb with GET_VALUE's sums as FMAs (nvcc's default contraction): exposure probe, not the parity definition */
void oracle_calc_orb_fma(const float *angle, const float *pos, uint8_t *desc_tmp, uint32_t *desc32,
                         const uint8_t *img, int pitch, int w, int h, int n, int angle_in_radians);
/* f2: kernel_keypoint_pixel_to_point + deproject_pixel_to_point_double, cuda-align.cu:85-112, :282-364 */
typedef struct oracle_intrinsics {
    int32_t width, height;
    float ppx, ppy, fx, fy;
    int32_t model;
    float coeffs[5];
} oracle_intrinsics;
int oracle_keypoint_pixel_to_point(const uint32_t *aligned_depth, const oracle_intrinsics *intrin,
                                   int image_width, int image_height, float *pos_out,
                                   const float *pos_in, const float *score, double *points,
                                   uint32_t *desc_out, const uint32_t *desc_in, int n,
                                   int fix_depth_index);
/* f2, the producing half: align_depth_to_other, cuda-align.cu:121-188, :224-280, :366-399.
 * rs2_extrinsics as the kernels read it (:112-119): column-major 3x3 rotation, translation.
 * pixel_map (optional): int2[2 * depth.width * depth.height], the reference's scratch.
 * Returns 0, or -1 for the models the reference asserts on / needs libdevice double atan for. */
typedef struct oracle_extrinsics {
    float rotation[9];
    float translation[3];
} oracle_extrinsics;
int oracle_align_depth_to_other(uint32_t *aligned_out, const uint16_t *depth_in, int32_t *pixel_map,
                                float depth_scale, int image_width, int image_height,
                                const oracle_intrinsics *depth_intrin, const oracle_intrinsics *other_intrin,
                                const oracle_extrinsics *depth_to_other);
int oracle_match_keypoints(const float *pos_prev, const uint32_t *desc_prev, int n_prev,
                           const float *pos_curr, const uint32_t *desc_curr, int n_curr,
                           int max_px, int max_ham, int32_t *match_idx /*[n_prev], -1 = none*/);
/* a11/a12: the compacted lists kernel_match_keypoints writes (post_processing.cu:176-198), in prev order */
int oracle_match_compact(const int32_t *match_idx, int n_prev, const double *points_prev,
                         const double *points_curr, const float *pos_curr, double *prev_matched,
                         double *curr_matched, uint16_t *keypoints_x, uint16_t *keypoints_y);
/* f4 (part): kernel_reproject_prev_points, post_processing.cu:11-43, :72-90; T column-major 4x4 */
void oracle_reproject_points(float *pos_out, const double *points_prev, int n, const double *T,
                             const oracle_intrinsics *intrin);
/* EXT C.9: brute-force 256-bit; (dist, idx) lexicographic minimum; window < 0 = none */
void oracle_match256(const uint8_t *descA, const float *posA, int nA, const uint8_t *descB,
                     const float *posB, int nB, int window, int max_dist, int32_t *idx,
                     int32_t *dist);

/* ---- whole pipeline on one frame ---- */
int oracle_num_cells(const oracle_config *cfg);
size_t oracle_level_dims(const oracle_config *cfg, int level, int *w, int *h);
/* All outputs optional (NULL to skip).  pyr_out[l] must hold w_l*h_l bytes (tight pitch).
 * Returns the number of records written (score > 0, after top-N), in cell order. */
int oracle_extract_frame(const oracle_config *cfg, const uint8_t *gray, int gray_pitch,
                         uint8_t **pyr_out, float *pos, float *score, int32_t *level,
                         float *angle, uint8_t *desc, uint32_t *desc32,
                         oracle_keypoint *records);

/* deterministic-math probes for the known-answer tests */
float oracle_atan2f(float y, float x);
void oracle_sincosf(float x, float *s, float *c);
int oracle_has_arc(uint32_t mask, int arc);
const int8_t *oracle_pattern(void);

#ifdef __cplusplus
}
#endif
#endif
