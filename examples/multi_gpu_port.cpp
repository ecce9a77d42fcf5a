// multi_gpu_port.cpp -- the reference's thread-per-stream model (SlamGpuPipeline.cpp:43-50: one buildStream
// thread, one CUDA stream and one set of device buffers per slam_frames slot) carried over to one thread
// per GPU, in C++ against the two C ABIs only (no torch, no MPI):
//
//   thread r:  hipSetDevice(r) -> orbfe_create -> orbfe_dist_create(id, r, N, r)
//              frames [f0, f1) of the batch = orbfe_dist_shard_range(n_total, r, N)
//              orbfe_extract (its shard) -> orbfe_match_batch (pairs inside the shard)
//              orbfe_dist_gather_keypoints(... root 0 ...)   // RCCL over xGMI, keypoint records only
//   rank 0 then holds every frame's records in frame order and writes them to <out.bin>.
//
//   multi_gpu_port <n_gpus> <width> <height> <n_frames> <frames_u8.bin> <out.bin> [exact] [devices=a,b,...]
//
// devices= picks the HIP device of every rank (default: rank r on device r).  Naming one device twice is what RCCL itself
// refuses ("Duplicate GPU detected"); tests/test_gpu_round5.py does it on a one-GPU box with a loopback transport in
// place of librccl.so.1 to execute the world > 1 branches of liborbfe_dist.so.
//
// out = [int32 n_frames | int32 cap | int32 counts[n_frames] | records n_frames * cap * 52 bytes] (fixed stride)
// tests/test_gpu_round2.py::test_cpp_multi_gpu_port runs it with one GPU and compares with the oracle; on a
// node with N GPUs the same binary shards over all of them.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/orbfe.h"
#include "../include/orbfe_dist.h"

#define CHECK_HIP(x)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "rank %d: %s: %s\n", rank, #x, hipGetErrorString(e_));  \
            std::exit(2);                                                                \
        }                                                                                \
    } while (0)
#define CHECK_ORBFE(x, ctx)                                                              \
    do {                                                                                 \
        int rc_ = (x);                                                                   \
        if (rc_ != ORBFE_OK) {                                                           \
            std::fprintf(stderr, "rank %d: %s: %d %s\n", rank, #x, rc_, orbfe_last_error(ctx)); \
            std::exit(3);                                                                \
        }                                                                                \
    } while (0)
#define CHECK_DIST(x, d)                                                                 \
    do {                                                                                 \
        int rc_ = (x);                                                                   \
        if (rc_ != ORBFE_OK) {                                                           \
            std::fprintf(stderr, "rank %d: %s: %d %s\n", rank, #x, rc_, orbfe_dist_last_error(d)); \
            std::exit(4);                                                                \
        }                                                                                \
    } while (0)

struct Job {
    int world, width, height, n_frames, exact;
    std::vector<int> devices;   // HIP device of rank r
    const uint8_t *frames;      // host, n_frames * width * height
    uint8_t id[ORBFE_DIST_ID_BYTES];
    std::vector<int32_t> counts; // filled by rank 0
    std::vector<uint8_t> records;
    int cap = 0;
};

static void rank_main(Job *job, int rank)
{
    const int w = job->width, h = job->height;
    const int device = job->devices[rank];
    CHECK_HIP(hipSetDevice(device));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    int f0, f1;
    CHECK_DIST(orbfe_dist_shard_range(job->n_frames, rank, job->world, &f0, &f1), nullptr);
    const int n = f1 - f0; // equal on every rank here (the gather wants equal shards): n_frames % world == 0
    orbfe_config cfg;
    orbfe_default_config(&cfg, w, h);
    cfg.levels = 8;
    cfg.cell = 8;
    cfg.min_arc = 9;
    cfg.max_features = 2000;
    cfg.max_batch = n;
    cfg.device = device;
    orbfe_ctx *ctx = nullptr;
    CHECK_ORBFE(orbfe_create(&cfg, &ctx), nullptr);
    const int cap = orbfe_max_keypoints(ctx);
    orbfe_dist *dist = nullptr;
    CHECK_DIST(orbfe_dist_create(job->id, rank, job->world, device, &dist), nullptr);

    uint8_t *d_gray;
    orbfe_keypoint *d_records, *d_all_records = nullptr;
    int32_t *d_counts, *d_all_counts = nullptr, *d_idx;
    const size_t frame_bytes = (size_t)w * h;
    CHECK_HIP(hipMalloc((void **)&d_gray, frame_bytes * n));
    CHECK_HIP(hipMalloc((void **)&d_records, sizeof(orbfe_keypoint) * (size_t)n * cap));
    CHECK_HIP(hipMalloc((void **)&d_counts, sizeof(int32_t) * n));
    CHECK_HIP(hipMalloc((void **)&d_idx, sizeof(int32_t) * (size_t)(n > 1 ? n - 1 : 1) * cap));
    if (rank == 0) {
        CHECK_HIP(hipMalloc((void **)&d_all_records, sizeof(orbfe_keypoint) * (size_t)job->n_frames * cap));
        CHECK_HIP(hipMalloc((void **)&d_all_counts, sizeof(int32_t) * job->n_frames));
        CHECK_HIP(hipMemsetAsync(d_all_records, 0, sizeof(orbfe_keypoint) * (size_t)job->n_frames * cap, stream));
    }
    CHECK_HIP(hipMemcpyAsync(d_gray, job->frames + frame_bytes * f0, frame_bytes * n, hipMemcpyHostToDevice, stream));
    CHECK_ORBFE(orbfe_extract(ctx, d_gray, w, frame_bytes, n, d_records, d_counts, nullptr, (orbfe_stream_t)stream), ctx);
    CHECK_ORBFE(orbfe_match_batch(ctx, d_records, d_counts, n, 1, -1, 256, d_idx, nullptr, (orbfe_stream_t)stream), ctx);
    CHECK_DIST(orbfe_dist_gather_keypoints(dist, d_records, d_counts, n, cap, d_all_records, d_all_counts, 0, job->exact,
                                           (orbfe_stream_t)stream), dist);
    CHECK_DIST(orbfe_dist_sync(dist), dist); // "gather complete on rank 0"
    CHECK_HIP(hipStreamSynchronize(stream));
    if (rank == 0) {
        job->cap = cap;
        job->counts.resize(job->n_frames);
        job->records.assign(sizeof(orbfe_keypoint) * (size_t)job->n_frames * cap, 0);
        CHECK_HIP(hipMemcpy(job->counts.data(), d_all_counts, sizeof(int32_t) * job->n_frames, hipMemcpyDeviceToHost));
        std::vector<uint8_t> raw(job->records.size());
        CHECK_HIP(hipMemcpy(raw.data(), d_all_records, raw.size(), hipMemcpyDeviceToHost));
        if (!job->exact) {
            job->records = raw;
        } else { // exact-length form: rank r's block is dense inside; re-expand to the fixed stride for the file
            for (int r = 0; r < job->world; r++) {
                size_t src = sizeof(orbfe_keypoint) * (size_t)r * n * cap;
                for (int f = 0; f < n; f++) {
                    const size_t bytes = sizeof(orbfe_keypoint) * (size_t)job->counts[r * n + f];
                    std::memcpy(job->records.data() + sizeof(orbfe_keypoint) * (size_t)(r * n + f) * cap, raw.data() + src, bytes);
                    src += bytes;
                }
            }
        }
    }
    orbfe_dist_destroy(dist);
    orbfe_destroy(ctx);
    (void)hipFree(d_gray);
    (void)hipFree(d_records);
    (void)hipFree(d_counts);
    (void)hipFree(d_idx);
    if (d_all_records) (void)hipFree(d_all_records);
    if (d_all_counts) (void)hipFree(d_all_counts);
    (void)hipStreamDestroy(stream);
}

int main(int argc, char **argv)
{
    if (argc < 7) return 1;
    Job job;
    job.world = std::atoi(argv[1]);
    job.width = std::atoi(argv[2]);
    job.height = std::atoi(argv[3]);
    job.n_frames = std::atoi(argv[4]);
    job.exact = 0;
    for (int a = 7; a < argc; a++) {
        if (!std::strcmp(argv[a], "exact")) job.exact = 1;
        if (!std::strncmp(argv[a], "devices=", 8))
            for (const char *p = argv[a] + 8; *p;) {
                job.devices.push_back(std::atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
    }
    const int ndev = orbfe_device_count();
    if (job.devices.empty())
        for (int r = 0; r < job.world; r++) job.devices.push_back(r);
    bool devices_ok = (int)job.devices.size() == job.world;
    for (int dv : job.devices) devices_ok = devices_ok && dv >= 0 && dv < ndev;
    if (job.world < 1 || !devices_ok || job.n_frames % job.world != 0) {
        std::fprintf(stderr, "need n_gpus >= 1 ranks on devices 0..%d (one per rank) and n_frames %% n_gpus == 0\n", ndev - 1);
        return 1;
    }
    std::vector<uint8_t> frames((size_t)job.width * job.height * job.n_frames);
    FILE *f = std::fopen(argv[5], "rb");
    if (!f || std::fread(frames.data(), 1, frames.size(), f) != frames.size()) return 1;
    std::fclose(f);
    job.frames = frames.data();
    if (orbfe_dist_unique_id(job.id) != ORBFE_OK) { // ncclGetUniqueId; the threads share it through the Job
        std::fprintf(stderr, "unique id: %s\n", orbfe_dist_last_error(nullptr));
        return 1;
    }
    std::vector<std::thread> threads;
    for (int r = 0; r < job.world; r++) threads.emplace_back(rank_main, &job, r);
    for (auto &t : threads) t.join();
    f = std::fopen(argv[6], "wb");
    if (!f) return 1;
    const int32_t head[2] = {job.n_frames, job.cap};
    std::fwrite(head, 4, 2, f);
    std::fwrite(job.counts.data(), 4, job.counts.size(), f);
    std::fwrite(job.records.data(), 1, job.records.size(), f);
    std::fclose(f);
    long total = 0;
    for (int c : job.counts) total += c;
    std::printf("%d GPU(s), %d frames, %ld keypoints gathered on rank 0\n", job.world, job.n_frames, total);
    return 0;
}
