/* orbfe.h -- C ABI of the MI355X-native ORB front end (liborbfe.so).
 *
 * This is the drop-in boundary for the GPU feature path of dsvua/jetracer-orbslam2: the
 * free functions declared in the reference's src/cuda/ *.cuh and called by
 * SlamGpuPipeline::buildStream (src/SlamGpuPipeline/buildStream.cpp:338, :424-460,
 * :545-556).  Each entry point below names the reference declaration it replaces.
 *
 * Conventions (differences from the reference are deliberate and listed in DESIGN.md):
 *   - extern "C", plain pointers and sizes.  Every `d_` pointer is DEVICE memory owned by
 *     the caller; the library allocates nothing per call (only orbfe_create allocates).
 *   - every call is asynchronous on the caller's HIP stream and in-order on it; no call
 *     synchronises the stream or the device.
 *   - int status return (ORBFE_OK = 0) instead of the reference's abort-on-error
 *     (checkCudaErrors / exit(1), src/cuda_common.h:69-77); orbfe_last_error() gives text.
 *   - no global mutable state: the rBRIEF pattern is a compile-time constant (the
 *     reference uploads it with loadPattern(), src/cuda/orb.cu:218-225) and the FAST LUT
 *     is a caller buffer as in the reference, read as an opaque table by every call that
 *     takes it -- nothing is remembered about how, where or for which arc it was built.
 *     (The only writable static storage in the library is the thread-local text buffer
 *     behind orbfe_last_error(NULL).)  A context is thread-compatible, not
 *     thread-safe: ONE context = ONE host thread + ONE stream at a time (its pyramid, cell
 *     keys and matcher scratch are single copies), as the reference runs one buildStream
 *     thread per stream (src/SlamGpuPipeline/SlamGpuPipeline.cpp:43-50).  Calls leave the
 *     caller's current HIP device unchanged.
 *   - there is NO CPU fallback: without a HIP device every call returns ORBFE_ERR_NO_DEVICE
 *     or ORBFE_ERR_HIP.
 */
#ifndef ORBFE_H
#define ORBFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBFE_VERSION 2 /* 2: orbfe_config gained descriptor_level (appended) */

enum {
    ORBFE_OK = 0,
    ORBFE_ERR_INVALID_ARG = 1, /* null pointer, non-positive size, bad pitch ...          */
    ORBFE_ERR_UNSUPPORTED = 2, /* configuration outside the reference regime + EXT rules  */
    ORBFE_ERR_HIP = 3,         /* a HIP runtime call failed; see orbfe_last_error()       */
    ORBFE_ERR_NO_DEVICE = 4,
    ORBFE_ERR_CAPACITY = 5     /* n_frames > max_batch, output too small ...              */
};

/* hipStream_t without the HIP headers (hipStream_t is `struct ihipStream_t *`). */
typedef struct ihipStream_t *orbfe_stream_t;

/* Mirror of the reference's pyramid_t (src/cuda/pyramid.cuh:9-18), passed as pointer +
 * count instead of a std::vector by value. */
typedef struct orbfe_pyramid_level {
    size_t image_width;
    size_t image_height;
    size_t image_pitch; /* bytes */
    unsigned char *image;
    size_t response_pitch; /* bytes, as in the reference (divided by sizeof(float) inside) */
    float *response;
} orbfe_pyramid_level;

/* enum fast_score, src/cuda/fast.cuh:18-23: what orbfe_fast_calc_corner_response writes for an accepted pixel --
 * the sum of |p - c| over the ring (fast.cu:233-241), the larger of the two sums of |p - c| - t over the darker /
 * brighter ring pixels (:243-255; the live path's, src/SlamGpuPipeline/defines.h:9, and the only one orbfe_detect and
 * the batch API use), or the largest threshold at which the table still accepts the pixel (:256-283). */
enum {
    ORBFE_SUM_OF_ABS_DIFF_ALL = 0,
    ORBFE_SUM_OF_ABS_DIFF_ON_ARC = 1,
    ORBFE_MAX_THRESHOLD = 2
};

/* ================= stage API: one frame, caller-owned buffers, reference semantics ===== */

/* rgb_to_grayscale, src/cuda/cuda_RGB_to_Grayscale.cuh:10-16 (the step right
 * before the path): dst = floor((B*0.07 + G*0.72 + R*0.21) + 0.5) in double, interleaved RGB8. */
int orbfe_rgb_to_grayscale(unsigned char *d_dst, const unsigned char *d_src, int cols, int rows,
                           int dst_pitch, int src_pitch, orbfe_stream_t stream);

/* gaussian_blur_3x3, src/cuda/orb.cuh:29-35.  Rows 0, h-2, h-1 are written as 0. */
int orbfe_gaussian_blur_3x3(unsigned char *d_blurred, int blurred_pitch,
                            const unsigned char *d_image, int image_pitch, int image_width,
                            int image_height, orbfe_stream_t stream);

/* pyramid_create_levels, src/cuda/pyramid.cuh:20-21: level i (i >= 1) = 2x2 box average of
 * level i-1.  levels[i].image_width/height must equal floor(previous / 2). */
int orbfe_pyramid_create_levels(const orbfe_pyramid_level *levels, int n_levels,
                                orbfe_stream_t stream);

/* fast_gpu_calculate_lut, src/cuda/fast.cuh:25-26: 65536-byte table, lut[m] = 1 iff the
 * 16-bit ring mask m holds a cyclic run of >= min_arc ones.  Unlike the reference (legacy
 * default stream, fast.cu:299) it runs on `stream`. */
int orbfe_fast_calculate_lut(unsigned char *d_corner_lut, int min_arc_length,
                             orbfe_stream_t stream);

/* fast_gpu_calc_corner_response, src/cuda/fast.cuh:28-40. */
int orbfe_fast_calc_corner_response(int image_width, int image_height, int image_pitch,
                                    const unsigned char *d_image, int horizontal_border,
                                    int vertical_border, const unsigned char *d_corner_lut,
                                    float threshold, int min_arc_length, int score,
                                    int response_pitch_elements, float *d_response,
                                    orbfe_stream_t stream);

/* grid_nms, src/cuda/nms.cuh:11-15: one keypoint per 32x32 level-0 cell over all levels.
 * d_pos is float2[K], d_score float[K], d_level int[K], K = ceil(W/32)*ceil(H/32).
 * Empty cells get score 0, pos (0,0), level 0. */
int orbfe_grid_nms(const orbfe_pyramid_level *levels, int n_levels, float *d_pos,
                   float *d_score, int *d_level, orbfe_stream_t stream);

/* detect, src/cuda/fast.cuh:42-48 = corner response on every level, then grid_nms.
 * d_corner_lut is the opaque 65536-byte table of the reference: a corner is what the
 * reference's prechecks (fast.cu:98-124) let through AND lut[dark] | lut[bright] accepts,
 * for any table contents (one built by orbfe_fast_calculate_lut, copied, or filled by the
 * caller).  levels[i].response may be NULL (the maps are then not written). */
int orbfe_detect(const orbfe_pyramid_level *levels, int n_levels,
                 const unsigned char *d_corner_lut, float threshold, float *d_pos,
                 float *d_score, int *d_level, orbfe_stream_t stream);

/* compute_fast_angle, src/cuda/orb.cuh:9-16.  Intensity-centroid angle in radians for all
 * keypoints_num entries of d_keypoints_pos (float2[]). */
int orbfe_compute_fast_angle(float *d_keypoints_angle, const float *d_keypoints_pos,
                             const unsigned char *d_image, int image_pitch, int image_width,
                             int image_height, int keypoints_num, orbfe_stream_t stream);

/* calc_orb, src/cuda/orb.cuh:18-27: 32-byte rBRIEF into d_descriptors_tmp[K][32] and the
 * reference's 32-bit "compressed" word (bit i = byte i == 1) into d_descriptors[K]. */
int orbfe_calc_orb(const float *d_keypoints_angle, const float *d_keypoints_pos,
                   unsigned char *d_descriptors_tmp, uint32_t *d_descriptors,
                   const unsigned char *d_image, int image_pitch, int image_width,
                   int image_height, int keypoints_num, orbfe_stream_t stream);

/* loadPattern, src/cuda/orb.cuh:37.  The pattern is compiled in; kept so that a caller's
 * init sequence (SlamGpuPipeline.cpp:52) ports line for line.  Always ORBFE_OK. */
int orbfe_load_pattern(void);

/* match_keypoints, src/cuda/post_processing.cuh:40-51, without the RGB-D parts: the caller
 * passes the (re-projected) prev positions directly.  For every prev keypoint i the best
 * curr keypoint within +-max_pixel_distance in x and y and Hamming distance
 * < max_hamming_distance on the 32-bit words; d_match_idx[i] = curr index or -1;
 * *d_num_matched = number of matches.  The reference's visiting order (tiles of 32,
 * rotated by i % 32, partial-tile skip) is reproduced. */
int orbfe_match_keypoints(const float *d_pos_prev, const uint32_t *d_descriptors_prev,
                          int keypoints_num_prev, const float *d_pos_curr,
                          const uint32_t *d_descriptors_curr, int keypoints_num_curr,
                          int max_pixel_distance, int max_hamming_distance,
                          int32_t *d_match_idx, int32_t *d_num_matched,
                          orbfe_stream_t stream);

/* The compacted outputs kernel_match_keypoints writes for every matched prev keypoint
 * (src/cuda/post_processing.cu:176-198): previous_matched_points / current_matched_points (double3)
 * and d_pos_frame = uint16 x | uint16 y of the matched CURRENT keypoint, which match_keypoints then
 * copies into slam_frame_t::keypoints_x / keypoints_y (:300-331, src/SlamGpuPipeline/types.h:29-30).
 * d_match_idx is orbfe_match_keypoints' output.  The lists are written in ascending prev index
 * (the reference's atomicAdd slots come in arbitrary order) and *d_num_matched receives their
 * length.  The point arrays are optional (RGB-D only): pass d_points_* and d_*_matched NULL together. */
int orbfe_match_compact(const int32_t *d_match_idx, int keypoints_num_prev, const double *d_points_prev,
                        const double *d_points_curr, const float *d_pos_curr, double *d_prev_matched,
                        double *d_curr_matched, uint16_t *d_keypoints_x, uint16_t *d_keypoints_y,
                        int32_t *d_num_matched, orbfe_stream_t stream);

/* rs2_intrinsics (librealsense2 rs_types.h) as the reference's kernels read it
 * (src/cuda/cuda-align.cu:57-112): same field order and meaning. */
typedef struct orbfe_intrinsics {
    int32_t width, height;
    float ppx, ppy, fx, fy;
    int32_t model; /* rs2_distortion: 0 none, 1 modified Brown-Conrady, 2 inverse Brown-Conrady, 3 f-theta, 4 Brown-Conrady */
    float coeffs[5];
} orbfe_intrinsics;

/* keypoint_pixel_to_point, src/cuda/cuda-align.cuh (kernel cuda-align.cu:282-364), SURVEY.md 8f-2,
 * the step right after the path: keep keypoints with aligned depth > 1 and score > 1, compact
 * them and deproject to 3-D (3 doubles per point).  Differences: `intrin` may be a HOST pointer (read
 * at call time) or, as in the reference, a DEVICE pointer (copied back once per call); the compacted order is by keypoint index (the reference's atomics give an
 * arbitrary order); *d_valid_keypoints_num is written by the kernel, nothing is copied to the host.
 * fix_depth_index = 0 reproduces the reference's depth lookup depth[int(y+.5) * W + int(y+.5)]
 * (it uses y for the column, cuda-align.cu:332; needs height <= width); 1 uses int(x+.5).
 * Forward-distorted models (1, 3) cannot be deprojected (the reference asserts): UNSUPPORTED. */
int orbfe_keypoint_pixel_to_point(const uint32_t *d_aligned_depth, const orbfe_intrinsics *intrin,
                                  int image_width, int image_height, float *d_pos_out,
                                  const float *d_pos_in, const float *d_score, double *d_points,
                                  uint32_t *d_descriptors_out, const uint32_t *d_descriptors_in,
                                  int keypoints_num, int32_t *d_valid_keypoints_num,
                                  int fix_depth_index, orbfe_stream_t stream);

/* rs2_extrinsics (librealsense2 rs_types.h) as the reference's kernels read it
 * (src/cuda/cuda-align.cu:112-119): column-major 3x3 rotation, then the translation. */
typedef struct orbfe_extrinsics {
    float rotation[9];
    float translation[3];
} orbfe_extrinsics;

/* align_depth_to_other, src/cuda/cuda-align.cuh:37-46 (host :366-399; kernels :121-188, :224-280), SURVEY.md 8f-2:
 * the depth image resampled onto the other (colour) camera's pixel grid -- the d_aligned_out that
 * orbfe_keypoint_pixel_to_point reads (buildStream.cpp:385, :468).  Every depth pixel with depth != 0 is mapped
 * twice (corners -0.5 and +0.5: deproject, transform, project, int(v + 0.5f)), and its raw 16-bit depth goes by
 * minimum into every output pixel of the rectangle between the two images; pixels no rectangle covers read 0.
 * Same arguments as the reference, with these differences: the three camera structs may be HOST pointers read at
 * call time or the reference's DEVICE copies (_d_depth_intrinsics ..., SlamGpuPipeline.cpp:53-55: detected with
 * hipPointerGetAttributes and copied back with one small synchronous hipMemcpy per call); d_pixel_map, the reference's int2[2 W H] scratch, is not
 * touched and may be NULL (map and splat are one kernel; nothing else ever read the map); depth_scale must be
 * finite.  As in the reference the launch grid is made from image_width / image_height (32 x 32 blocks) while
 * all bounds come from the intrinsics: depth pixels / output pixels beyond the grid are not read / not reset.
 * d_depth_in: depth_intrin->width * height uint16, d_aligned_out: other_intrin->width * height uint32, both
 * contiguous.  Depth model 1 / 3 (the reference asserts) returns ORBFE_ERR_UNSUPPORTED; other model 3 (f-theta,
 * cuda-align.cu:44-50) is evaluated in float with the build's deterministic atanf / tanf (orbfe_ftheta_distort,
 * include/orbfe_math.h).  Float arithmetic as written, no contraction: parity with the reference unpinned at the ulp
 * level, bit-exact against the oracle. */
int orbfe_align_depth_to_other(uint32_t *d_aligned_out, const uint16_t *d_depth_in, void *d_pixel_map,
                               float depth_scale, int image_width, int image_height,
                               const orbfe_intrinsics *depth_intrin, const orbfe_intrinsics *other_intrin,
                               const orbfe_extrinsics *depth_to_other, orbfe_stream_t stream);

/* The same for n_frames depth frames in one call (frame f at d_depth_in + f * in_frame_stride uint16 elements,
 * its output at d_aligned_out + f * out_frame_stride uint32 elements); the grid covers both images.  One camera
 * rig, i.e. one set of intrinsics / extrinsics for the whole batch. */
int orbfe_align_depth_batch(uint32_t *d_aligned_out, size_t out_frame_stride, const uint16_t *d_depth_in,
                            size_t in_frame_stride, int n_frames, float depth_scale,
                            const orbfe_intrinsics *depth_intrin, const orbfe_intrinsics *other_intrin,
                            const orbfe_extrinsics *depth_to_other, orbfe_stream_t stream);

/* kernel_reproject_prev_points, src/cuda/post_processing.cu:72-90 (with project_point_to_pixel_double,
 * :11-43): the prev frame's 3-D points moved by T_w2c_prev_curr (HOST pointer to 16 doubles,
 * column-major as Eigen::Matrix4d) and projected to pixels -- the positions orbfe_match_keypoints takes
 * as d_pos_prev.  Model 1 (modified Brown-Conrady) applies its polynomial; 0, 2 and 4 project
 * without distortion, exactly as the reference (its assert against model 2 is commented out, :15, and a D4xx colour
 * stream reports model 2); 3 (f-theta, :32-38) as orbfe_ftheta_distort (include/orbfe_math.h).  `intrin` host or device pointer.  Parity unpinned at the ulp level (Eigen's
 * product order and nvcc's FMA contraction are not observable): ((T_i0 x + T_i1 y) + T_i2 z) + T_i3. */
int orbfe_reproject_points(float *d_pos_out, const double *d_points_prev, int keypoints_num_prev,
                           const double *T_w2c_prev_curr, const orbfe_intrinsics *intrin, orbfe_stream_t stream);

/* EXT: brute-force 256-bit Hamming matcher.  For every descriptor i of A the
 * lexicographic minimum (distance, index) over B; window < 0 disables the position gate
 * (then d_posA/d_posB may be NULL); matches with distance > max_distance give -1/-1. */
int orbfe_match256(const unsigned char *d_descA, const float *d_posA, int nA,
                   const unsigned char *d_descB, const float *d_posB, int nB, int window,
                   int max_distance, int32_t *d_idx, int32_t *d_dist, orbfe_stream_t stream);

/* ================= batch API: many frames per call, context-owned scratch ============== */

typedef struct orbfe_config {
    int32_t width, height;     /* level-0 frame size                                       */
    int32_t levels;            /* pyramid levels built, 1..16                              */
    int32_t cell;              /* level-0 NMS cell: 8, 16, 32 (reference) or 64            */
    int32_t fast_threshold;    /* FAST epsilon, 1..254 (reference 13, defines.h:7)         */
    int32_t min_arc;           /* 9..12 (reference 12, defines.h:8)                        */
    int32_t max_features;      /* 0 = every non-empty cell (reference); N = top-N by       */
                               /* (score desc, cell asc)                                   */
    int32_t angle_in_radians;  /* 0 = reference (radians used as degrees); 1 = fixed       */
    int32_t max_batch;         /* frames per orbfe_extract call the context is sized for   */
    int32_t device;            /* HIP device ordinal                                       */
    int32_t descriptor_level;  /* 0 = reference (orientation + descriptor always sample    */
                               /* level 0 at the level-0 position, buildStream.cpp:442-460,*/
                               /* quirk Q10); 1 = EXT iv: they sample the pyramid level    */
                               /* that won the cell, at position / 2^level, with that      */
                               /* level's size in every bound and guard band (scale-aware  */
                               /* ORB).  Records keep level-0 coordinates either way.      */
} orbfe_config;

/* 52-byte keypoint record, little endian. */
typedef struct orbfe_keypoint {
    float x, y;      /* level-0 pixel coordinates                                 */
    float score;     /* FAST sum-of-abs-diff-on-arc score (an integer)            */
    int32_t level;   /* pyramid level that won the cell                           */
    float angle;     /* radians, as stored by compute_fast_angle                  */
    uint8_t desc[32];
} orbfe_keypoint;

/* Optional cell-indexed struct-of-arrays view, the layout the reference keeps
 * (src/SlamGpuPipeline/buildStream.cpp:289-296, :255-258).  Each non-NULL pointer is
 * device memory for n_frames * K elements (pos: 2 floats, desc: 32 bytes per cell). */
typedef struct orbfe_soa {
    float *d_pos;
    float *d_score;
    int32_t *d_level;
    float *d_angle;
    uint8_t *d_desc;
    uint32_t *d_desc32;
} orbfe_soa;

typedef struct orbfe_ctx orbfe_ctx;

void orbfe_default_config(orbfe_config *cfg, int width, int height);
int orbfe_create(const orbfe_config *cfg, orbfe_ctx **out);
void orbfe_destroy(orbfe_ctx *ctx);
/* Text of the last error on this context (ctx == NULL: last error of orbfe_create or of a
 * stage call on the calling thread). */
const char *orbfe_last_error(const orbfe_ctx *ctx);

int orbfe_num_cells(const orbfe_ctx *ctx);     /* K                                        */
int orbfe_max_keypoints(const orbfe_ctx *ctx); /* records per frame: min(K, max_features)  */
int orbfe_num_levels(const orbfe_ctx *ctx);
/* Geometry and device address of pyramid level `level` of frame 0 inside the context;
 * frame f lives *frame_stride bytes further per frame. */
int orbfe_level_info(const orbfe_ctx *ctx, int level, int *width, int *height, size_t *pitch,
                     const uint8_t **d_image, size_t *frame_stride);

/* a2+a3: blur + pyramid for n_frames frames. d_gray: frame f at d_gray + f*frame_stride. */
int orbfe_build_pyramid(orbfe_ctx *ctx, const uint8_t *d_gray, size_t pitch,
                        size_t frame_stride, int n_frames, orbfe_stream_t stream);

/* The same from interleaved RGB8 frames (pitch >= 3 * width): the gray conversion is fused
 * into the pyramid kernel's row loads, the gray frame is never written.  Needs width % 4 == 0
 * and a 4-byte aligned source, else ORBFE_ERR_UNSUPPORTED. */
int orbfe_build_pyramid_rgb(orbfe_ctx *ctx, const uint8_t *d_rgb, size_t pitch, size_t frame_stride,
                            int n_frames, orbfe_stream_t stream);

/* a4..a7 on the context's pyramid: fused FAST score + 3x3 NMS + per-cell maximum over all
 * detection levels (levels with cell >> level >= 1) of n_frames frames. */
int orbfe_detect_batch(orbfe_ctx *ctx, int n_frames, orbfe_stream_t stream);

/* Sharded detection (one large frame over several GPUs, SURVEY.md 8e "per-level shard"):
 * run only the detection tiles shard_index, shard_index + shard_count, ... (tiles of all
 * levels, interleaved).  Each shard yields a partial per-cell key array; because the key
 * encodes score, level and the reference's tie order, the exact result is the element-wise
 * unsigned MAXIMUM of the shards' arrays, in any order: export, all-reduce(MAX) over RCCL,
 * import, then orbfe_describe_batch.  Keys are < 2^27, so a signed-int32 MAX is equivalent. */
int orbfe_detect_batch_shard(orbfe_ctx *ctx, int n_frames, int shard_index, int shard_count,
                             orbfe_stream_t stream);
/* Copy the n_frames * K cell keys of the last detection out of / into the context (device to
 * device, on `stream`). */
int orbfe_export_cell_keys(orbfe_ctx *ctx, int n_frames, uint32_t *d_keys, orbfe_stream_t stream);
int orbfe_import_cell_keys(orbfe_ctx *ctx, int n_frames, const uint32_t *d_keys,
                           orbfe_stream_t stream);

/* selection (all non-empty cells or top-N) + a8 orientation + a9/a10 descriptors for the
 * frames last detected; outputs as orbfe_extract. */
int orbfe_describe_batch(orbfe_ctx *ctx, int n_frames, orbfe_keypoint *d_records,
                         int32_t *d_counts, const orbfe_soa *soa, orbfe_stream_t stream);

/* Whole extraction (a2..a10) for n_frames frames: pyramid, fused FAST + grid NMS,
 * selection, orientation, descriptors.  d_records holds n_frames * orbfe_max_keypoints()
 * records (frame f starts at f * max_keypoints), in cell order; d_counts[f] = number of
 * valid records of frame f.  soa may be NULL. */
int orbfe_extract(orbfe_ctx *ctx, const uint8_t *d_gray, size_t pitch, size_t frame_stride,
                  int n_frames, orbfe_keypoint *d_records, int32_t *d_counts,
                  const orbfe_soa *soa, orbfe_stream_t stream);

/* orbfe_extract with interleaved RGB8 input (orbfe_build_pyramid_rgb + detect + describe). */
int orbfe_extract_rgb(orbfe_ctx *ctx, const uint8_t *d_rgb, size_t pitch, size_t frame_stride,
                      int n_frames, orbfe_keypoint *d_records, int32_t *d_counts,
                      const orbfe_soa *soa, orbfe_stream_t stream);

/* a11 over a batch: for f = 1 .. n_frames-1 match the records of frame f-1 (prev) against
 * frame f (curr).  mode 0 = reference semantics (32-bit word, +-window px, distance <
 * max_distance); mode 1 = EXT 256-bit brute force (window < 0: none; distance <=
 * max_distance).  d_idx / d_dist: (n_frames-1) * max_keypoints int32 (d_dist may be NULL). */
int orbfe_match_batch(orbfe_ctx *ctx, const orbfe_keypoint *d_records, const int32_t *d_counts,
                      int n_frames, int mode, int window, int max_distance, int32_t *d_idx,
                      int32_t *d_dist, orbfe_stream_t stream);

/* The same for an arbitrary pair list inside the batch: pair k = (prev frame first + k * stride,
 * curr frame first + k * stride + 1) for every k whose curr frame is < n_frames; results of
 * pair k at d_idx / d_dist + k * max_keypoints.  orbfe_match_batch == first 0, stride 1;
 * stereo pairs (left, right, left, right ...) == first 0, stride 2 (config C3). */
int orbfe_match_pairs(orbfe_ctx *ctx, const orbfe_keypoint *d_records, const int32_t *d_counts,
                      int n_frames, int first, int stride, int mode, int window, int max_distance,
                      int32_t *d_idx, int32_t *d_dist, orbfe_stream_t stream);

/* ================= harness helpers (tests, bench) ===================================== */
int orbfe_device_count(void);
int orbfe_memcpy_d2h(void *dst, const void *d_src, size_t bytes, orbfe_stream_t stream);
int orbfe_memcpy_h2d(void *d_dst, const void *src, size_t bytes, orbfe_stream_t stream);
int orbfe_stream_sync(orbfe_stream_t stream);
int orbfe_version(void);
/* Which kernels the batch calls run for this context, call size and matcher mode, as
 * "pyramid=...;detect=...;describe=...;match=...;match_examines=all_pairs|window_cells" (the library picks the
 * describe and match kernels by keypoint density, call size and window; bench.py labels its stages from this
 * instead of repeating the conditions).  Host-side only. */
int orbfe_dispatch_info(const orbfe_ctx *ctx, int n_frames, int mode, int window, char *buf, size_t size);
/* The pyramid layout orbfe_create builds for `cfg` and the byte range the detection kernel's UNCONDITIONAL tile loads
 * touch in it (every tile of every level of frames 0 .. max_batch - 1 reads rows y0 - 4 .. y0 + 67, columns
 * x0 - 4 .. x0 + 67 of its level without range tests): *tile_lo / *tile_hi = lowest / highest byte relative to the
 * first byte of frame 0, *pyramid_bytes = max_batch frames, *guard_bytes = the band allocated before and after them.
 * Legal iff -guard <= lo and hi < pyramid + guard; orbfe_create refuses a geometry that violates it.  Host code, needs
 * no device: tests/test_layout.py runs it over random geometries (a round-3 work-in-progress build faulted exactly
 * here before the guard bands existed). */
int orbfe_layout_bounds(const orbfe_config *cfg, long long *tile_lo, long long *tile_hi,
                        unsigned long long *pyramid_bytes, unsigned long long *guard_bytes, int *n_tiles);
/* Exhaustive self-check of the orientation -> rotated-pattern table the tile describe kernel uses in the reference's
 * degrees-as-radians regime (angle_in_radians = 0): for EVERY float orientation in [-pi, pi] (both
 * signs, ~2.2e9 values) the table's 512 sample offsets are compared with the arithmetic of orb.cu:12-14, :42-46 as the
 * oracle and the other kernels evaluate it.  Synchronous (~1 s on an MI355X); *n_mismatch must come back 0.  With
 * angle_in_radians = 1 there is no table: *n_angles = 0. */
int orbfe_selfcheck_steer_table(orbfe_ctx *ctx, unsigned long long *n_angles, unsigned long long *n_mismatch);

#ifdef __cplusplus
}
#endif
#endif /* ORBFE_H */
