// match_port.cpp -- the two-frame part of the reference's SlamGpuPipeline::buildStream written against
// compat/jetracer_compat.hpp: per frame align_depth_to_other on its own stream (buildStream.cpp:376-394),
// rgb_to_grayscale .. calc_orb (:399-466) and keypoint_pixel_to_point on the aligned depth (:468-481), then
// match_keypoints(current, previous, 2, 4, T, ...) exactly as :545-556 calls it, with the reference's slam_frame_t
// (types.h:25-65).
//
//   match_port <width> <height> <rgbA.bin> <rgbB.bin> <depth_u16.bin> <rig.bin> <out.bin>
//
// rig.bin = orbfe_intrinsics depth | orbfe_intrinsics rgb | orbfe_extrinsics depth->rgb | float depth_scale (the
// rs2_* structs and get_units() the reference takes from the camera, SlamGpuPipeline.cpp:75-88).
//
// out = [int32 n_prev_valid | int32 n_curr_valid | int32 n_matched | keypoints_x u16[n] | keypoints_y u16[n] |
//        previous_matched double3[n] | current_matched double3[n]] for tests/test_gpu_round2.py to compare
// with the CPU oracle.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "../jetracer-orbslam2_amd/compat/jetracer_compat.hpp"

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            std::exit(2);                                                                \
        }                                                                                \
    } while (0)

#define FAST_EPSILON (13.0f)     // src/SlamGpuPipeline/defines.h:7
#define FAST_MIN_ARC_LENGTH 12   // defines.h:8

using namespace Jetracer;

struct Pipeline { // the per-stream buffers of buildStream.cpp:233-336
    int cam_w, cam_h;
    std::size_t keypoints_num;
    unsigned char *d_rgb_image, *d_gray_image, *d_descriptors_tmp, *d_corner_lut;
    std::size_t rgb_pitch, gray_pitch;
    float *d_keypoints_angle, *d_feature_grid;
    uint32_t *d_descriptors;
    unsigned int *d_aligned_out;
    uint16_t *d_depth_in;
    int2 *d_pixel_map;
    float2 *d_pos;
    float *d_score;
    int *d_level, *d_valid_keypoints_num;
    std::vector<pyramid_t> pyramid;
    // the reference's DEVICE copies of the camera structs (SlamGpuPipeline.cpp:53-55 cudaMalloc, upload_intristics :58-86):
    // passed on exactly as buildStream.cpp:391-393, :469 and post_processing.cuh:45 do
    orbfe_intrinsics *_d_depth_intrinsics, *_d_rgb_intrinsics;
    orbfe_extrinsics *_d_depth_rgb_extrinsics;
    float depth_scale;
    hipStream_t stream, align_stream;
};

static std::shared_ptr<slam_frame_t> process(Pipeline &p, const std::vector<unsigned char> &h_rgb,
                                             const std::vector<uint16_t> &h_depth)
{
    // --------------- align depth to RGB (buildStream.cpp:376-394): its own stream, in parallel with the keypoints
    CHECK(hipMemcpyAsync(p.d_depth_in, h_depth.data(), h_depth.size() * sizeof(uint16_t), hipMemcpyHostToDevice,
                         p.align_stream));
    align_depth_to_other(p.d_aligned_out, p.d_depth_in, p.d_pixel_map, p.depth_scale, p.cam_w, p.cam_h, p._d_depth_intrinsics,
                         p._d_rgb_intrinsics, p._d_depth_rgb_extrinsics, p.align_stream);
    CHECK(hipMemcpy2DAsync(p.d_rgb_image, p.rgb_pitch, h_rgb.data(), (size_t)p.cam_w * 3, (size_t)p.cam_w * 3, p.cam_h,
                           hipMemcpyHostToDevice, p.stream));
    rgb_to_grayscale(p.d_gray_image, p.d_rgb_image, p.cam_w, p.cam_h, (int)p.gray_pitch, (int)p.rgb_pitch, p.stream);
    gaussian_blur_3x3(p.pyramid[0].image, (int)p.pyramid[0].image_pitch, p.d_gray_image, (int)p.gray_pitch, p.cam_w,
                      p.cam_h, p.stream);
    pyramid_create_levels(p.pyramid, p.stream);
    detect(p.pyramid, p.d_corner_lut, FAST_EPSILON, p.d_pos, p.d_score, p.d_level, p.stream);
    compute_fast_angle(p.d_keypoints_angle, p.d_pos, p.pyramid[0].image, (int)p.pyramid[0].image_pitch, p.cam_w, p.cam_h,
                       (int)p.keypoints_num, p.stream);
    calc_orb(p.d_keypoints_angle, p.d_pos, p.d_descriptors_tmp, p.d_descriptors, p.pyramid[0].image,
             (int)p.pyramid[0].image_pitch, p.cam_w, p.cam_h, (int)p.keypoints_num, p.stream);
    // buildStream.cpp:468-521: valid keypoints -> the frame's own compacted buffers
    auto frame = std::make_shared<slam_frame_t>();
    frame->keypoints_count = (int)p.keypoints_num;
    CHECK(hipMalloc((void **)&frame->d_pos, sizeof(float2) * p.keypoints_num));
    CHECK(hipMalloc((void **)&frame->d_points, sizeof(double) * 3 * p.keypoints_num));
    CHECK(hipMalloc((void **)&frame->d_descriptors, sizeof(uint32_t) * p.keypoints_num));
    // the reference reads d_aligned_out on `stream` and only synchronises align_stream afterwards (:468-487), a race
    // it gets away with; ordered here
    CHECK(hipStreamSynchronize(p.align_stream));
    keypoint_pixel_to_point(p.d_aligned_out, p._d_rgb_intrinsics, p.cam_w, p.cam_h, frame->d_pos, p.d_pos, p.d_score,
                            frame->d_points, frame->d_descriptors, p.d_descriptors, (int)p.keypoints_num,
                            &frame->h_valid_keypoints_num, p.d_valid_keypoints_num, p.stream);
    CHECK(hipStreamSynchronize(p.stream));
    return frame;
}

int main(int argc, char **argv)
{
    if (argc != 8) return 1;
    Pipeline p;
    p.cam_w = std::atoi(argv[1]);
    p.cam_h = std::atoi(argv[2]);
    auto read_file = [](const char *path, void *dst, size_t bytes) {
        FILE *f = std::fopen(path, "rb");
        if (!f || std::fread(dst, 1, bytes, f) != bytes) std::exit(1);
        std::fclose(f);
    };
    std::vector<unsigned char> rgb[2];
    for (int i = 0; i < 2; i++) {
        rgb[i].resize((size_t)p.cam_w * p.cam_h * 3);
        read_file(argv[3 + i], rgb[i].data(), rgb[i].size());
    }
    std::vector<uint16_t> depth((size_t)p.cam_w * p.cam_h);
    read_file(argv[5], depth.data(), depth.size() * sizeof(uint16_t));
    struct {
        orbfe_intrinsics depth, rgb;
        orbfe_extrinsics depth_to_rgb;
        float depth_scale;
    } rig;
    read_file(argv[6], &rig, sizeof(rig));
    CHECK(hipMalloc((void **)&p._d_depth_intrinsics, sizeof(orbfe_intrinsics)));
    CHECK(hipMalloc((void **)&p._d_rgb_intrinsics, sizeof(orbfe_intrinsics)));
    CHECK(hipMalloc((void **)&p._d_depth_rgb_extrinsics, sizeof(orbfe_extrinsics)));
    CHECK(hipMemcpy(p._d_depth_intrinsics, &rig.depth, sizeof(orbfe_intrinsics), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(p._d_rgb_intrinsics, &rig.rgb, sizeof(orbfe_intrinsics), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(p._d_depth_rgb_extrinsics, &rig.depth_to_rgb, sizeof(orbfe_extrinsics), hipMemcpyHostToDevice));
    p.depth_scale = rig.depth_scale;

    CHECK(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&p.align_stream, hipStreamNonBlocking));
    const int grid_cols = (p.cam_w + 31) / 32, grid_rows = (p.cam_h + 31) / 32;
    p.keypoints_num = (std::size_t)grid_cols * grid_rows;
    CHECK(hipMallocPitch((void **)&p.d_rgb_image, &p.rgb_pitch, (size_t)p.cam_w * 3, p.cam_h));
    CHECK(hipMallocPitch((void **)&p.d_gray_image, &p.gray_pitch, p.cam_w, p.cam_h));
    CHECK(hipMalloc((void **)&p.d_keypoints_angle, p.keypoints_num * sizeof(float)));
    CHECK(hipMalloc((void **)&p.d_descriptors_tmp, p.keypoints_num * 32));
    CHECK(hipMalloc((void **)&p.d_descriptors, p.keypoints_num * sizeof(uint32_t)));
    CHECK(hipMalloc((void **)&p.d_corner_lut, 64 * 1024));
    CHECK(hipMalloc((void **)&p.d_feature_grid, p.keypoints_num * sizeof(float) * 4));
    CHECK(hipMalloc((void **)&p.d_aligned_out, (size_t)p.cam_w * sizeof(unsigned int) * p.cam_h)); // buildStream.cpp:250-252
    CHECK(hipMalloc((void **)&p.d_depth_in, (size_t)p.cam_w * sizeof(uint16_t) * p.cam_h));
    CHECK(hipMalloc((void **)&p.d_pixel_map, (size_t)p.cam_w * sizeof(int2) * p.cam_h * 2));
    CHECK(hipMalloc((void **)&p.d_valid_keypoints_num, sizeof(int)));
    p.d_pos = (float2 *)p.d_feature_grid;
    p.d_score = p.d_feature_grid + p.keypoints_num * 2;
    p.d_level = (int *)(p.d_feature_grid + p.keypoints_num * 3);
    pyramid_t level; // PYRAMID_LEVELS 1, defines.h:2
    level.image_width = p.cam_w;
    level.image_height = p.cam_h;
    CHECK(hipMallocPitch((void **)&level.image, &level.image_pitch, level.image_width, level.image_height));
    CHECK(hipMallocPitch((void **)&level.response, &level.response_pitch, level.image_width * sizeof(float),
                         level.image_height));
    p.pyramid.push_back(level);
    loadPattern();
    fast_gpu_calculate_lut(p.d_corner_lut, FAST_MIN_ARC_LENGTH, p.stream);

    std::shared_ptr<slam_frame_t> previous_frame = process(p, rgb[0], depth);
    std::shared_ptr<slam_frame_t> slam_frame = process(p, rgb[1], depth);

    // buildStream.cpp:523-556
    const double T_w2c_prev_curr[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; // Eigen::Matrix4d::Identity().data()
    int *d_keypoints_num_matched, h_keypoints_num_matched = 0;
    CHECK(hipMalloc((void **)&d_keypoints_num_matched, sizeof(int)));
    std::vector<double3> h_curr(p.keypoints_num), h_prev(p.keypoints_num);
    match_keypoints(slam_frame, previous_frame, 2, 4, T_w2c_prev_curr, p.d_valid_keypoints_num, d_keypoints_num_matched,
                    &h_keypoints_num_matched, h_curr.data(), h_prev.data(), p._d_rgb_intrinsics, p.stream);

    const int n = h_keypoints_num_matched;
    FILE *f = std::fopen(argv[7], "wb");
    if (!f) return 1;
    const int32_t head[3] = {previous_frame->h_valid_keypoints_num, slam_frame->h_valid_keypoints_num, n};
    std::fwrite(head, 4, 3, f);
    std::fwrite(slam_frame->keypoints_x, 2, n, f);
    std::fwrite(slam_frame->keypoints_y, 2, n, f);
    std::fwrite(h_prev.data(), sizeof(double3), n, f);
    std::fwrite(h_curr.data(), sizeof(double3), n, f);
    std::fclose(f);
    std::printf("valid %d -> %d, matched %d (slam_frame->h_matched_keypoints_num %d)\n", head[0], head[1], n,
                slam_frame->h_matched_keypoints_num);
    return 0;
}
