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
#include <cstring>
#include <memory>
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

// src/cuda/cuda-align.cuh:37-46 (rs2_intrinsics / rs2_extrinsics -> the layout-identical orbfe structs).  The three
// camera structs may be the reference's DEVICE copies (_d_depth_intrinsics, _d_rgb_intrinsics, _d_depth_rgb_extrinsics:
// SlamGpuPipeline.cpp:53-55, passed at buildStream.cpp:391-393 -- a port keeps those arguments as they are: the library
// detects device memory and copies the 48 bytes back once per call) or plain host structs; either way the values reach
// the kernels as launch arguments.  d_pixel_map is accepted and ignored (the fused kernel has no map).
inline void align_depth_to_other(unsigned int *d_aligned_out, const uint16_t *d_depth_in, int2 *d_pixel_map,
                                 float depth_scale, int image_width, int image_height,
                                 const orbfe_intrinsics *depth_intrin, const orbfe_intrinsics *other_intrin,
                                 const orbfe_extrinsics *depth_to_other, hipStream_t stream)
{
    detail::check(orbfe_align_depth_to_other(d_aligned_out, d_depth_in, d_pixel_map, depth_scale, image_width,
                                             image_height, depth_intrin, other_intrin, depth_to_other,
                                             detail::S(stream)), "align_depth_to_other");
}

// src/cuda/cuda-align.cuh:48-61 (host :401-443): like the reference it zeroes the counter, launches, and queues the
// copy of the count to *h_valid_keypoints_num on `stream` (valid after the caller's next synchronisation, as in
// buildStream.cpp:468-487).  fix_depth_index = 0: the reference's depth[int(y+.5) * W + int(y+.5)] lookup (:332).
// (rgb_intrin: host or device pointer, as above)
inline void keypoint_pixel_to_point(unsigned int *d_aligned_depth, const orbfe_intrinsics *rgb_intrin, int image_width,
                                    int image_height, float2 *d_pos_out, float2 *d_pos_in, float *d_score,
                                    double *d_points, uint32_t *d_descriptors_out, uint32_t *d_descriptors_in,
                                    int keypoints_num, int *h_valid_keypoints_num, int *d_valid_keypoints_num,
                                    hipStream_t stream)
{
    detail::check(orbfe_keypoint_pixel_to_point(d_aligned_depth, rgb_intrin, image_width, image_height,
                                                reinterpret_cast<float *>(d_pos_out),
                                                reinterpret_cast<const float *>(d_pos_in), d_score, d_points,
                                                d_descriptors_out, d_descriptors_in, keypoints_num,
                                                d_valid_keypoints_num, 0, detail::S(stream)), "keypoint_pixel_to_point");
    if (hipMemcpyAsync(h_valid_keypoints_num, d_valid_keypoints_num, sizeof(int), hipMemcpyDeviceToHost, stream) !=
        hipSuccess) {
        std::fprintf(stderr, "orbfe: keypoint_pixel_to_point: copying the count failed\n");
        std::exit(EXIT_FAILURE);
    }
}

// ---- slam_frame_t and the reference's match_keypoints host function -------------------------------
// src/SlamGpuPipeline/types.h:25-65: the same fields with the same names and types, minus what this path
// does not own (the rgbd_frame handle) and with Eigen::Matrix4d as its 16 column-major doubles
// (Eigen::Matrix4d::data() is exactly that).  Buffers are released as the reference's destructor does.
static_assert(sizeof(double3) == 24, "double3 is three packed doubles, as in CUDA");

typedef struct slam_frame {
    unsigned char *image = nullptr;
    size_t image_length = 0;
    uint16_t *keypoints_x = nullptr; // matched current keypoints, filled by match_keypoints (post_processing.cu:300-331)
    uint16_t *keypoints_y = nullptr;
    std::shared_ptr<double[]> h_points;
    std::shared_ptr<uint32_t[]> h_descriptors;

    float2 *d_pos = nullptr;       // compacted valid keypoints (cuda-align.cu:282-364 / orbfe_keypoint_pixel_to_point)
    double *d_points = nullptr;    // their 3-D points, 3 doubles each
    uint32_t *d_descriptors = nullptr;

    int keypoints_count = 0;
    int h_valid_keypoints_num = 0;
    int h_matched_keypoints_num = 0;
    float3 theta = {0.f, 0.f, 0.f};

    double T_c2w[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double T_w2c[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

    ~slam_frame()
    {
        if (image) free(image);
        if (keypoints_x) free(keypoints_x);
        if (keypoints_y) free(keypoints_y);
        if (d_pos) (void)hipFree(d_pos);
        if (d_points) (void)hipFree(d_points);
        if (d_descriptors) (void)hipFree(d_descriptors);
    }
} slam_frame_t;

namespace detail {
inline void hip_check(hipError_t e, const char *what)
{
    if (e != hipSuccess) {
        std::fprintf(stderr, "orbfe: %s failed: %s\n", what, hipGetErrorString(e));
        std::exit(EXIT_FAILURE);
    }
}
} // namespace detail

// src/cuda/post_processing.cuh:40-51 / post_processing.cu:234-341, argument for argument:
//   Eigen::Matrix4d T_w2c_prev_curr  -> its 16 doubles (T.data());
//   const rs2_intrinsics *d_rgb_intrin -> the same pointer: orbfe_intrinsics has rs2_intrinsics' layout, and the
//     library takes the reference's device copy (_d_rgb_intrinsics) or a host struct alike;
//   d_valid_keypoints_num (a device int the reference's kernel dereferences) is unused: the count is
//     current_frame->h_valid_keypoints_num.
// Same observable results: h_*_matched_points, *h_keypoints_num_matched, current_frame->keypoints_x / _y
// (malloc'ed here, freed by ~slam_frame), in prev-keypoint order instead of atomic order.  Like the
// reference it allocates scratch and synchronises the stream; the C ABI underneath does neither.
inline void match_keypoints(std::shared_ptr<slam_frame_t> current_frame, std::shared_ptr<slam_frame_t> previous_frame,
                            int max_pixel_distance, int max_hamming_distance, const double *T_w2c_prev_curr,
                            int * /*d_valid_keypoints_num*/, int *d_keypoints_num_matched, int *h_keypoints_num_matched,
                            double3 *h_current_matched_points, double3 *h_previous_matched_points,
                            const orbfe_intrinsics *d_rgb_intrin, hipStream_t stream)
{
    const int n_prev = previous_frame->h_valid_keypoints_num, n_curr = current_frame->h_valid_keypoints_num;
    const int cap = previous_frame->keypoints_count > n_prev ? previous_frame->keypoints_count : n_prev;
    float2 *d_pos_tmp = nullptr;
    int32_t *d_match_idx = nullptr;
    uint16_t *d_pos_frame = nullptr;
    double3 *d_prev_m = nullptr, *d_curr_m = nullptr;
    const size_t n = (size_t)(cap > 0 ? cap : 1);
    detail::hip_check(hipMalloc((void **)&d_pos_tmp, sizeof(float2) * n), "hipMalloc");
    detail::hip_check(hipMalloc((void **)&d_match_idx, sizeof(int32_t) * n), "hipMalloc");
    detail::hip_check(hipMalloc((void **)&d_pos_frame, sizeof(uint16_t) * n * 2), "hipMalloc");
    detail::hip_check(hipMalloc((void **)&d_prev_m, sizeof(double3) * n), "hipMalloc");
    detail::hip_check(hipMalloc((void **)&d_curr_m, sizeof(double3) * n), "hipMalloc");
    detail::check(orbfe_reproject_points(reinterpret_cast<float *>(d_pos_tmp), previous_frame->d_points, n_prev,
                                         T_w2c_prev_curr, d_rgb_intrin, detail::S(stream)), "reproject_points");
    detail::check(orbfe_match_keypoints(reinterpret_cast<const float *>(d_pos_tmp), previous_frame->d_descriptors, n_prev,
                                        reinterpret_cast<const float *>(current_frame->d_pos),
                                        current_frame->d_descriptors, n_curr, max_pixel_distance, max_hamming_distance,
                                        d_match_idx, d_keypoints_num_matched, detail::S(stream)), "match_keypoints");
    detail::check(orbfe_match_compact(d_match_idx, n_prev, previous_frame->d_points, current_frame->d_points,
                                      reinterpret_cast<const float *>(current_frame->d_pos),
                                      reinterpret_cast<double *>(d_prev_m), reinterpret_cast<double *>(d_curr_m),
                                      d_pos_frame, d_pos_frame + n, d_keypoints_num_matched, detail::S(stream)),
                  "match_compact");
    detail::hip_check(hipMemcpyAsync(h_keypoints_num_matched, d_keypoints_num_matched, sizeof(int), hipMemcpyDeviceToHost,
                                     stream), "hipMemcpyAsync");
    detail::hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize"); // the reference syncs here too (:315)
    const size_t m = (size_t)*h_keypoints_num_matched;
    if (h_previous_matched_points)
        detail::hip_check(hipMemcpyAsync(h_previous_matched_points, d_prev_m, sizeof(double3) * m, hipMemcpyDeviceToHost,
                                         stream), "hipMemcpyAsync");
    if (h_current_matched_points)
        detail::hip_check(hipMemcpyAsync(h_current_matched_points, d_curr_m, sizeof(double3) * m, hipMemcpyDeviceToHost,
                                         stream), "hipMemcpyAsync");
    if (current_frame->keypoints_x) free(current_frame->keypoints_x);
    if (current_frame->keypoints_y) free(current_frame->keypoints_y);
    current_frame->keypoints_x = (uint16_t *)malloc(sizeof(uint16_t) * (m ? m : 1));
    current_frame->keypoints_y = (uint16_t *)malloc(sizeof(uint16_t) * (m ? m : 1));
    detail::hip_check(hipMemcpyAsync(current_frame->keypoints_x, d_pos_frame, sizeof(uint16_t) * m, hipMemcpyDeviceToHost,
                                     stream), "hipMemcpyAsync");
    detail::hip_check(hipMemcpyAsync(current_frame->keypoints_y, d_pos_frame + n, sizeof(uint16_t) * m,
                                     hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    detail::hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    current_frame->h_matched_keypoints_num = (int)m;
    (void)hipFree(d_pos_tmp);
    (void)hipFree(d_match_idx);
    (void)hipFree(d_pos_frame);
    (void)hipFree(d_prev_m);
    (void)hipFree(d_curr_m);
}

} // namespace Jetracer
