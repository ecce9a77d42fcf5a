// jetracer_compat.hpp -- the reference's own function names over the orbfe C ABI.
//
// Host-side mirror of the operator interface the reference's pipeline uses for this path:
// the free functions of namespace Jetracer declared in src/cuda/orb.cuh, pyramid.cuh, fast.cuh,
// nms.cuh and post_processing.cuh (dsvua/jetracer-orbslam2).  Same names, same argument order
// and meaning, so SlamGpuPipeline::buildStream (src/SlamGpuPipeline/buildStream.cpp:338,
// :424-460) compiles against liborbfe.so after replacing cudaStream_t by hipStream_t.
// Error behaviour is the reference's: a failing call prints and aborts (checkCudaErrors,
// src/cuda_common.h:68-118); use the C ABI directly for status codes.
//
// Header-only C++17; needs only <hip/hip_runtime_api.h> for hipStream_t and float2.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/orbfe.h"

namespace Jetracer {

typedef orbfe_pyramid_level pyramid_t; // src/cuda/pyramid.cuh:9-18, same fields

enum fast_score { SUM_OF_ABS_DIFF_ALL = 0, SUM_OF_ABS_DIFF_ON_ARC, MAX_THRESHOLD }; // fast.cuh:18-23

namespace detail {
inline void check(int rc, const char *what)
{
    if (rc != ORBFE_OK) {
        std::fprintf(stderr, "orbfe: %s failed (%d): %s\n", what, rc, orbfe_last_error(nullptr));
        std::exit(EXIT_FAILURE);
    }
}
inline orbfe_stream_t S(hipStream_t s) { return reinterpret_cast<orbfe_stream_t>(s); }
} // namespace detail

// src/cuda/cuda_RGB_to_Grayscale.cuh:10-16
inline void rgb_to_grayscale(unsigned char *dst, unsigned char *src, int cols, int rows, int dst_pitch,
                             int src_pitch, hipStream_t stream)
{
    detail::check(orbfe_rgb_to_grayscale(dst, src, cols, rows, dst_pitch, src_pitch, detail::S(stream)),
                  "rgb_to_grayscale");
}

// src/cuda/orb.cuh:29-35
inline void gaussian_blur_3x3(unsigned char *blurred_image, int blurred_image_pitch, unsigned char *image,
                              int image_pitch, int image_width, int image_height, hipStream_t stream)
{
    detail::check(orbfe_gaussian_blur_3x3(blurred_image, blurred_image_pitch, image, image_pitch, image_width,
                                          image_height, detail::S(stream)), "gaussian_blur_3x3");
}

// src/cuda/pyramid.cuh:20-21
inline void pyramid_create_levels(std::vector<pyramid_t> pyramid, hipStream_t stream)
{
    detail::check(orbfe_pyramid_create_levels(pyramid.data(), (int)pyramid.size(), detail::S(stream)),
                  "pyramid_create_levels");
}

// src/cuda/fast.cuh:25-26 (the reference runs this on the legacy default stream, fast.cu:299)
inline void fast_gpu_calculate_lut(unsigned char *d_corner_lut, const int &min_arc_length,
                                   hipStream_t stream = nullptr)
{
    detail::check(orbfe_fast_calculate_lut(d_corner_lut, min_arc_length, detail::S(stream)),
                  "fast_gpu_calculate_lut");
}

// src/cuda/fast.cuh:28-40
inline void fast_gpu_calc_corner_response(const int image_width, const int image_height, const int image_pitch,
                                          const unsigned char *d_image, const int horizontal_border,
                                          const int vertical_border, const unsigned char *d_corner_lut,
                                          const float threshold, const int min_arc_length, const fast_score score,
                                          const int response_pitch_elements, float *d_response, hipStream_t stream)
{
    detail::check(orbfe_fast_calc_corner_response(image_width, image_height, image_pitch, d_image, horizontal_border,
                                                  vertical_border, d_corner_lut, threshold, min_arc_length,
                                                  (int)score, response_pitch_elements, d_response,
                                                  detail::S(stream)), "fast_gpu_calc_corner_response");
}

// src/cuda/nms.cuh:11-15
inline void grid_nms(std::vector<pyramid_t> pyramid, float2 *d_pos, float *d_score, int *d_level,
                     hipStream_t stream)
{
    detail::check(orbfe_grid_nms(pyramid.data(), (int)pyramid.size(), reinterpret_cast<float *>(d_pos), d_score,
                                 d_level, detail::S(stream)), "grid_nms");
}

// src/cuda/fast.cuh:42-48
inline void detect(std::vector<pyramid_t> pyramid, const unsigned char *d_corner_lut, const float threshold,
                   float2 *d_pos, float *d_score, int *d_level, hipStream_t stream)
{
    detail::check(orbfe_detect(pyramid.data(), (int)pyramid.size(), d_corner_lut, threshold,
                               reinterpret_cast<float *>(d_pos), d_score, d_level, detail::S(stream)), "detect");
}

// src/cuda/orb.cuh:9-16
inline void compute_fast_angle(float *d_keypoints_angle, float2 *d_keypoints_pos, unsigned char *image,
                               int image_pitch, int image_width, int image_height, int keypoints_num,
                               hipStream_t stream)
{
    detail::check(orbfe_compute_fast_angle(d_keypoints_angle, reinterpret_cast<const float *>(d_keypoints_pos), image,
                                           image_pitch, image_width, image_height, keypoints_num,
                                           detail::S(stream)), "compute_fast_angle");
}

// src/cuda/orb.cuh:18-27
inline void calc_orb(float *d_keypoints_angle, float2 *d_keypoints_pos, unsigned char *d_descriptors_tmp,
                     uint32_t *d_descriptors, unsigned char *image, int image_pitch, int image_width,
                     int image_height, int keypoints_num, hipStream_t stream)
{
    detail::check(orbfe_calc_orb(d_keypoints_angle, reinterpret_cast<const float *>(d_keypoints_pos),
                                 d_descriptors_tmp, d_descriptors, image, image_pitch, image_width, image_height,
                                 keypoints_num, detail::S(stream)), "calc_orb");
}

// src/cuda/orb.cuh:37
inline void loadPattern() { detail::check(orbfe_load_pattern(), "loadPattern"); }

// src/cuda/post_processing.cuh:40-51 without the RGB-D arguments (slam_frame_t, Eigen pose,
// rs2_intrinsics are outside this path): the caller passes device arrays directly.
inline void match_keypoints(const float2 *d_pos_prev_reprojected, const uint32_t *d_descriptors_prev,
                            int keypoints_num_prev, const float2 *d_pos_curr, const uint32_t *d_descriptors_curr,
                            int keypoints_num_curr, int max_pixel_distance, int max_hamming_distance,
                            int32_t *d_match_idx, int32_t *d_keypoints_num_matched, hipStream_t stream)
{
    detail::check(orbfe_match_keypoints(reinterpret_cast<const float *>(d_pos_prev_reprojected), d_descriptors_prev,
                                        keypoints_num_prev, reinterpret_cast<const float *>(d_pos_curr),
                                        d_descriptors_curr, keypoints_num_curr, max_pixel_distance,
                                        max_hamming_distance, d_match_idx, d_keypoints_num_matched,
                                        detail::S(stream)), "match_keypoints");
}

} // namespace Jetracer
