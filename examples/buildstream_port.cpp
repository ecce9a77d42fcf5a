// buildstream_port.cpp -- the frame path of the reference's SlamGpuPipeline::buildStream
// (src/SlamGpuPipeline/buildStream.cpp:233-336 buffer setup, :338 LUT, :399-466 per-frame calls)
// written against compat/jetracer_compat.hpp, i.e. with the reference's own function names.
// It is what a maintainer's port looks like: cuda* -> hip*, nothing else changes.
//
//   buildstream_port <width> <height> <levels> <rgb_in.bin> <out.bin>
//
// reads one interleaved RGB8 frame, runs rgb_to_grayscale -> gaussian_blur_3x3 ->
// pyramid_create_levels -> detect -> compute_fast_angle -> calc_orb, and writes
// [feature grid 16*K bytes | angle 4*K | desc 32*K | desc32 4*K] for the test to compare with the
// CPU oracle (tests/test_gpu_parity.py::test_cpp_buildstream_port).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../jetracer-orbslam2_amd/compat/jetracer_compat.hpp"

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

// src/SlamGpuPipeline/defines.h:7-8
#define FAST_EPSILON (13.0f)
#define FAST_MIN_ARC_LENGTH 12

int main(int argc, char **argv)
{
    if (argc != 6) return 1;
    const int cam_w = std::atoi(argv[1]), cam_h = std::atoi(argv[2]), levels = std::atoi(argv[3]);
    std::vector<unsigned char> h_rgb((size_t)cam_w * cam_h * 3);
    FILE *f = std::fopen(argv[4], "rb");
    if (!f || std::fread(h_rgb.data(), 1, h_rgb.size(), f) != h_rgb.size()) return 1;
    std::fclose(f);

    hipStream_t stream;
    CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); // buildStream.cpp:208-213

    const int grid_cols = (cam_w + 32 - 1) / 32, grid_rows = (cam_h + 32 - 1) / 32; // :233-235
    const std::size_t keypoints_num = (std::size_t)grid_cols * grid_rows;

    unsigned char *d_rgb_image, *d_gray_image, *d_descriptors_tmp, *d_corner_lut;
    std::size_t rgb_pitch, gray_pitch;
    float *d_keypoints_angle, *d_feature_grid;
    uint32_t *d_descriptors;
    CHECK(hipMallocPitch((void **)&d_rgb_image, &rgb_pitch, (size_t)cam_w * 3, cam_h)); // :247
    CHECK(hipMallocPitch((void **)&d_gray_image, &gray_pitch, cam_w, cam_h));           // :248
    CHECK(hipMalloc((void **)&d_keypoints_angle, keypoints_num * sizeof(float)));       // :255
    CHECK(hipMalloc((void **)&d_descriptors_tmp, keypoints_num * 32));                  // :257
    CHECK(hipMalloc((void **)&d_descriptors, keypoints_num * sizeof(uint32_t)));        // :258
    CHECK(hipMalloc((void **)&d_corner_lut, 64 * 1024));                                // :262-263
    const std::size_t feature_grid_bytes = keypoints_num * sizeof(float) * 4;          // :289-291
    CHECK(hipMalloc((void **)&d_feature_grid, feature_grid_bytes));
    float2 *d_pos = (float2 *)d_feature_grid;                                           // :294-296
    float *d_score = d_feature_grid + keypoints_num * 2;
    int *d_level = (int *)(d_feature_grid + keypoints_num * 3);

    std::vector<Jetracer::pyramid_t> pyramid; // :309-336
    int prev_width = 0, prev_height = 0;
    for (int i = 0; i < levels; i++) {
        Jetracer::pyramid_t level;
        level.image_width = i ? prev_width / 2 : cam_w;
        level.image_height = i ? prev_height / 2 : cam_h;
        CHECK(hipMallocPitch((void **)&level.image, &level.image_pitch, level.image_width, level.image_height));
        CHECK(hipMallocPitch((void **)&level.response, &level.response_pitch, level.image_width * sizeof(float),
                             level.image_height));
        pyramid.push_back(level);
        prev_width = (int)level.image_width;
        prev_height = (int)level.image_height;
    }

    Jetracer::loadPattern();                                                    // SlamGpuPipeline.cpp:52
    Jetracer::fast_gpu_calculate_lut(d_corner_lut, FAST_MIN_ARC_LENGTH, stream); // :338

    // ---- one frame, buildStream.cpp:399-466
    CHECK(hipMemcpy2DAsync(d_rgb_image, rgb_pitch, h_rgb.data(), (size_t)cam_w * 3, (size_t)cam_w * 3, cam_h,
                           hipMemcpyHostToDevice, stream));
    Jetracer::rgb_to_grayscale(d_gray_image, d_rgb_image, cam_w, cam_h, (int)gray_pitch, (int)rgb_pitch, stream);
    Jetracer::gaussian_blur_3x3(pyramid[0].image, (int)pyramid[0].image_pitch, d_gray_image, (int)gray_pitch, cam_w,
                                cam_h, stream);
    Jetracer::pyramid_create_levels(pyramid, stream);
    Jetracer::detect(pyramid, d_corner_lut, FAST_EPSILON, d_pos, d_score, d_level, stream);
    Jetracer::compute_fast_angle(d_keypoints_angle, d_pos, pyramid[0].image, (int)pyramid[0].image_pitch, cam_w,
                                 cam_h, (int)keypoints_num, stream);
    Jetracer::calc_orb(d_keypoints_angle, d_pos, d_descriptors_tmp, d_descriptors, pyramid[0].image,
                       (int)pyramid[0].image_pitch, cam_w, cam_h, (int)keypoints_num, stream);

    std::vector<unsigned char> out(feature_grid_bytes + keypoints_num * (4 + 32 + 4));
    unsigned char *o = out.data();
    CHECK(hipMemcpyAsync(o, d_feature_grid, feature_grid_bytes, hipMemcpyDeviceToHost, stream)); // :462-466
    o += feature_grid_bytes;
    CHECK(hipMemcpyAsync(o, d_keypoints_angle, keypoints_num * 4, hipMemcpyDeviceToHost, stream));
    o += keypoints_num * 4;
    CHECK(hipMemcpyAsync(o, d_descriptors_tmp, keypoints_num * 32, hipMemcpyDeviceToHost, stream));
    o += keypoints_num * 32;
    CHECK(hipMemcpyAsync(o, d_descriptors, keypoints_num * 4, hipMemcpyDeviceToHost, stream));
    CHECK(hipStreamSynchronize(stream));

    f = std::fopen(argv[5], "wb");
    if (!f || std::fwrite(out.data(), 1, out.size(), f) != out.size()) return 1;
    std::fclose(f);
    std::printf("K = %zu, wrote %zu bytes\n", keypoints_num, out.size());
    return 0;
}
