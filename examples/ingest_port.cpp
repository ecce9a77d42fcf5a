// ingest_port.cpp -- the HOST side of the reference's frame loop carried over to the staging ring of
// include/orbfe_ingest.h.  The reference (src/SlamGpuPipeline/buildStream.cpp) takes every camera frame from host memory
// and returns host results, with four copies per frame on its work streams:
//     :376-381  cudaMemcpyAsync(d_depth_in, rgbd_frame->depth_image, ...)           H2D, pageable source
//     :399-406  cudaMemcpy2DAsync(d_rgb_image, rgb_pitch, rgbd_frame->rgb_image, ...) H2D, pageable source
//     :462-466  cudaMemcpyAsync(h_feature_grid, d_feature_grid, ...)                 D2H
//     :483-487  cudaMemcpyAsync(h_points, slam_frame->d_points, ...)                 D2H
// Here the same traffic moves in batches through a ring of pinned slots: while slot s computes (orbfe_extract +
// orbfe_match_batch on the ring's compute stream), slot s + 1 uploads and slot s - 1 downloads.  The caller's loop is the
// one below: fill / submit, and collect the slot submitted `slots` submits ago.
//
//   ingest_port <width> <height> <n_frames> <frames_per_slot> <frames_u8.bin> <out.bin> [rgb]
//
// frames_u8.bin: n_frames gray (or interleaved RGB8 with `rgb`) frames in ordinary memory, as a camera driver hands them over.
// out = [int32 n_frames | int32 cap | int32 counts[n_frames] | records n_frames * cap * 52 bytes | int32 match_idx[(n_frames) * cap]]
// (match_idx of frame f = index in frame f + 1 of its best 256-bit match INSIDE ITS SLOT, -1 for the last frame of a slot).
// tests/test_gpu_round5.py::test_cpp_ingest_port compares the file with the oracle.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/orbfe.h"
#include "../include/orbfe_ingest.h"

#define CHECK(x, what)                                                                       \
    do {                                                                                     \
        int rc_ = (x);                                                                       \
        if (rc_ != ORBFE_OK) {                                                               \
            std::fprintf(stderr, "%s: %d %s\n", #x, rc_, what);                              \
            return 2;                                                                        \
        }                                                                                    \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 7) return 1;
    const int w = std::atoi(argv[1]), h = std::atoi(argv[2]), n = std::atoi(argv[3]), per_slot = std::atoi(argv[4]);
    const bool rgb = argc > 7 && !std::strcmp(argv[7], "rgb");
    const size_t frame_bytes = (size_t)w * h * (rgb ? 3 : 1);
    std::vector<uint8_t> frames(frame_bytes * n); // pageable, like rgbd_frame->rgb_image
    FILE *f = std::fopen(argv[5], "rb");
    if (!f || std::fread(frames.data(), 1, frames.size(), f) != frames.size()) return 1;
    std::fclose(f);

    orbfe_config cfg;
    orbfe_default_config(&cfg, w, h);
    cfg.levels = 8;
    cfg.cell = 8;
    cfg.min_arc = 9;
    cfg.max_features = 2000;
    cfg.max_batch = per_slot;
    orbfe_ctx *ctx = nullptr;
    CHECK(orbfe_create(&cfg, &ctx), orbfe_last_error(nullptr));
    const int cap = orbfe_max_keypoints(ctx);

    orbfe_ingest_config icfg;
    orbfe_ingest_default_config(&icfg, per_slot);
    icfg.channels = rgb ? 3 : 1;
    icfg.match_mode = 1; // 256-bit brute force, frame f -> f + 1 inside a slot
    icfg.match_window = -1;
    icfg.match_max_distance = 256;
    icfg.download_matches = 1;
    orbfe_ingest *ing = nullptr;
    CHECK(orbfe_ingest_create(ctx, &icfg, &ing), orbfe_ingest_last_error(nullptr));
    const int slots = orbfe_ingest_slots(ing);

    std::vector<int32_t> counts(n), match_idx((size_t)n * cap, -1);
    std::vector<orbfe_keypoint> records((size_t)n * cap);
    std::memset(records.data(), 0, records.size() * sizeof(orbfe_keypoint));
    const int n_batches = (n + per_slot - 1) / per_slot;
    auto frames_of = [&](int b) { return b == n_batches - 1 ? n - b * per_slot : per_slot; };
    auto collect = [&](int b) -> int {
        const orbfe_keypoint *rec;
        const int32_t *cnt, *idx;
        CHECK(orbfe_ingest_wait(ing, b % slots, &rec, &cnt, &idx, nullptr), orbfe_ingest_last_error(ing));
        const int k = frames_of(b), f0 = b * per_slot;
        std::memcpy(counts.data() + f0, cnt, sizeof(int32_t) * k);
        std::memcpy(records.data() + (size_t)f0 * cap, rec, sizeof(orbfe_keypoint) * (size_t)k * cap);
        if (idx) std::memcpy(match_idx.data() + (size_t)f0 * cap, idx, sizeof(int32_t) * (size_t)(k - 1) * cap);
        return 0;
    };
    for (int b = 0; b < n_batches; b++) {
        if (b >= slots && collect(b - slots)) return 2; // the ring is full: take the oldest slot's results first
        // the copy the CUDA driver does behind cudaMemcpy2DAsync from pageable memory, then upload / extract / match / download
        CHECK(orbfe_ingest_submit_from(ing, b % slots, frames_of(b), frames.data() + frame_bytes * (size_t)b * per_slot,
                                       (size_t)w * (rgb ? 3 : 1), frame_bytes),
              orbfe_ingest_last_error(ing));
    }
    for (int b = n_batches > slots ? n_batches - slots : 0; b < n_batches; b++)
        if (collect(b)) return 2;
    float up = 0, cmp = 0, down = 0;
    size_t ub = 0, db = 0;
    CHECK(orbfe_ingest_timing(ing, 0, &up, &cmp, &down, &ub, &db), orbfe_ingest_last_error(ing));
    orbfe_ingest_destroy(ing);
    orbfe_destroy(ctx);

    f = std::fopen(argv[6], "wb");
    if (!f) return 1;
    const int32_t head[2] = {n, cap};
    std::fwrite(head, 4, 2, f);
    std::fwrite(counts.data(), 4, counts.size(), f);
    std::fwrite(records.data(), sizeof(orbfe_keypoint), records.size(), f);
    std::fwrite(match_idx.data(), 4, match_idx.size(), f);
    std::fclose(f);
    long total = 0;
    for (int c : counts) total += c;
    std::printf("%d frames in %d batches of <= %d through %d pinned slots, %ld keypoints; slot 0's last pass: upload %.3f ms (%zu B), "
                "compute %.3f ms, download %.3f ms (%zu B)\n", n, n_batches, per_slot, slots, total, up, ub, cmp, down, db);
    return 0;
}
