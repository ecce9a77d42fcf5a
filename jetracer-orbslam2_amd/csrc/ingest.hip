// ingest.hip -- include/orbfe_ingest.h: the pinned host <-> device staging ring around orbfe_extract (SURVEY.md 8f-1,
// the staging half).  Replaces the four per-frame copies of the reference's frame loop
// (src/SlamGpuPipeline/buildStream.cpp:376-381 depth H2D, :399-406 colour H2D, :462-466 feature grid D2H, :483-487
// points D2H), which go through pageable memory on the work stream, by batches that move on their own copy streams
// while another batch computes.  Host code only: no kernel lives here; the arithmetic is orbfe_extract /
// orbfe_extract_rgb / orbfe_match_batch, called on the ring's compute stream.
//
// Ordering is by events alone (no host synchronisation inside submit):
//   copy-in  : [h2d0] memcpy frames -> d_frames [h2d1]
//   compute  : wait h2d1; [cmp0] extract (+ match) [cmp1]
//   copy-out : wait cmp1; [d2h0] memcpy counts, records (, idx, dist) -> pinned [d2h1]
// A slot is FREE or IN FLIGHT on the host side; it becomes free only in orbfe_ingest_wait (which synchronises d2h1),
// so when a slot is submitted again its device input, its device records and its pinned result buffers are all idle
// and need no further event.  What serialises consecutive slots is the compute stream (one context = one pyramid).
#include "orbfe_internal.hpp"

#include "../../include/orbfe_ingest.h"

#include <new>

namespace {

using orbfe::format_error;
using orbfe::set_thread_error;

struct Slot {
    uint8_t *h_frames = nullptr, *d_frames = nullptr;
    orbfe_keypoint *h_records = nullptr, *d_records = nullptr;
    int32_t *h_counts = nullptr, *d_counts = nullptr;
    int32_t *h_idx = nullptr, *d_idx = nullptr, *h_dist = nullptr, *d_dist = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // h2d0 h2d1 cmp0 cmp1 d2h0 d2h1
    int in_flight = 0;
    int n_frames = 0;      // of the last submit
    int completed = 0;     // a pass has finished: the timing events are valid
    size_t up_bytes = 0, down_bytes = 0;
};

} // namespace

struct orbfe_ingest {
    orbfe_ctx *ctx = nullptr;
    orbfe_ingest_config cfg{};
    int device = 0;
    int W = 0, H = 0, cap = 0;
    size_t frame_bytes = 0;
    hipStream_t s_in = nullptr, s_cmp = nullptr, s_out = nullptr;
    std::vector<Slot> slots;
    char err[512] = {0};
};

#define ING_FAIL(ing, code, ...)                                                              \
    do {                                                                                      \
        format_error((ing) ? (ing)->err : nullptr, __VA_ARGS__);                              \
        return code;                                                                          \
    } while (0)

#define ING_HIP(ing, expr)                                                                    \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            ING_FAIL(ing, ORBFE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static void free_all(orbfe_ingest *g)
{
    for (Slot &s : g->slots) {
        if (s.h_frames) (void)hipHostFree(s.h_frames);
        if (s.h_records) (void)hipHostFree(s.h_records);
        if (s.h_counts) (void)hipHostFree(s.h_counts);
        if (s.h_idx) (void)hipHostFree(s.h_idx);
        if (s.h_dist) (void)hipHostFree(s.h_dist);
        if (s.d_frames) (void)hipFree(s.d_frames);
        if (s.d_records) (void)hipFree(s.d_records);
        if (s.d_counts) (void)hipFree(s.d_counts);
        if (s.d_idx) (void)hipFree(s.d_idx);
        if (s.d_dist) (void)hipFree(s.d_dist);
        for (hipEvent_t e : s.ev)
            if (e) (void)hipEventDestroy(e);
    }
    if (g->s_in) (void)hipStreamDestroy(g->s_in);
    if (g->s_cmp) (void)hipStreamDestroy(g->s_cmp);
    if (g->s_out) (void)hipStreamDestroy(g->s_out);
}

extern "C" {

void orbfe_ingest_default_config(orbfe_ingest_config *cfg, int frames_per_slot)
{
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->slots = 3;
    cfg->frames_per_slot = frames_per_slot;
    cfg->channels = 1;
    cfg->match_mode = -1;
    cfg->match_window = -1;
    cfg->match_max_distance = 256;
}

const char *orbfe_ingest_last_error(const orbfe_ingest *ing) { return ing ? ing->err : orbfe::thread_error(); }

int orbfe_ingest_create(orbfe_ctx *ctx, const orbfe_ingest_config *cfg, orbfe_ingest **out)
{
    if (out) *out = nullptr;
    if (!ctx || !cfg || !out) {
        set_thread_error("orbfe_ingest_create: null argument");
        return ORBFE_ERR_INVALID_ARG;
    }
    if (cfg->slots < 2 || cfg->slots > 16 || cfg->frames_per_slot < 1 || (cfg->channels != 1 && cfg->channels != 3) ||
        cfg->match_mode < -1 || cfg->match_mode > 1 || cfg->download_matches < 0 || cfg->download_matches > 2 || cfg->reserved != 0) {
        set_thread_error("orbfe_ingest_create: slots 2..16, frames_per_slot >= 1, channels 1 or 3, match_mode -1..1, "
                         "download_matches 0..2, reserved 0");
        return ORBFE_ERR_INVALID_ARG;
    }
    if (cfg->frames_per_slot > ctx->cfg.max_batch) {
        set_thread_error("orbfe_ingest_create: frames_per_slot %d exceeds the context's max_batch %d", cfg->frames_per_slot,
                         ctx->cfg.max_batch);
        return ORBFE_ERR_CAPACITY;
    }
    if (cfg->channels == 3 && (ctx->cfg.width & 3)) { // orbfe_build_pyramid_rgb's own precondition, refused up front
        set_thread_error("orbfe_ingest_create: RGB8 input needs width %% 4 == 0 (orbfe_extract_rgb)");
        return ORBFE_ERR_UNSUPPORTED;
    }
    if (cfg->download_matches > 0 && cfg->match_mode < 0) {
        set_thread_error("orbfe_ingest_create: download_matches without match_mode");
        return ORBFE_ERR_INVALID_ARG;
    }
    orbfe::DeviceScope dev(ctx->cfg.device);
    if (!dev.ok) {
        set_thread_error("orbfe_ingest_create: cannot select HIP device %d", ctx->cfg.device);
        return ORBFE_ERR_NO_DEVICE;
    }
    orbfe_ingest *g = new (std::nothrow) orbfe_ingest;
    if (!g) {
        set_thread_error("orbfe_ingest_create: out of host memory");
        return ORBFE_ERR_HIP;
    }
    g->ctx = ctx;
    g->cfg = *cfg;
    g->device = ctx->cfg.device;
    g->W = ctx->cfg.width;
    g->H = ctx->cfg.height;
    g->cap = ctx->g.cap;
    g->frame_bytes = (size_t)g->W * g->H * cfg->channels;
    g->slots.resize(cfg->slots);
    const size_t F = (size_t)cfg->frames_per_slot;
    const size_t in_bytes = F * g->frame_bytes;
    const size_t rec_bytes = F * g->cap * sizeof(orbfe_keypoint);
    const size_t cnt_bytes = F * sizeof(int32_t);
    const size_t m_bytes = (F > 1 ? F - 1 : 1) * (size_t)g->cap * sizeof(int32_t);
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
        return e == hipSuccess;
    };
    ok(hipStreamCreateWithFlags(&g->s_in, hipStreamNonBlocking));
    ok(hipStreamCreateWithFlags(&g->s_cmp, hipStreamNonBlocking));
    ok(hipStreamCreateWithFlags(&g->s_out, hipStreamNonBlocking));
    for (Slot &s : g->slots) {
        if (e != hipSuccess) break;
        // pinned and mapped into every device's address space of this process; the default (coherent) flavour: the
        // producer's plain stores are visible to the copy engine without a flush
        ok(hipHostMalloc((void **)&s.h_frames, in_bytes, hipHostMallocDefault));
        ok(hipHostMalloc((void **)&s.h_records, rec_bytes, hipHostMallocDefault));
        ok(hipHostMalloc((void **)&s.h_counts, cnt_bytes, hipHostMallocDefault));
        ok(hipMalloc((void **)&s.d_frames, in_bytes));
        ok(hipMalloc((void **)&s.d_records, rec_bytes));
        ok(hipMalloc((void **)&s.d_counts, cnt_bytes));
        if (cfg->match_mode >= 0) {
            ok(hipMalloc((void **)&s.d_idx, m_bytes));
            ok(hipMalloc((void **)&s.d_dist, m_bytes));
            if (cfg->download_matches >= 1) ok(hipHostMalloc((void **)&s.h_idx, m_bytes, hipHostMallocDefault));
            if (cfg->download_matches >= 2) ok(hipHostMalloc((void **)&s.h_dist, m_bytes, hipHostMallocDefault));
        }
        for (hipEvent_t &ev : s.ev) ok(hipEventCreate(&ev));
    }
    if (e != hipSuccess) {
        set_thread_error("orbfe_ingest_create: allocation failed: %s (%d slots x %zu + %zu bytes pinned)", hipGetErrorString(e),
                         cfg->slots, in_bytes, rec_bytes);
        free_all(g);
        delete g;
        return ORBFE_ERR_HIP;
    }
    *out = g;
    return ORBFE_OK;
}

void orbfe_ingest_destroy(orbfe_ingest *g)
{
    if (!g) return;
    orbfe::DeviceScope dev(g->device);
    if (g->s_in) (void)hipStreamSynchronize(g->s_in);
    if (g->s_cmp) (void)hipStreamSynchronize(g->s_cmp);
    if (g->s_out) (void)hipStreamSynchronize(g->s_out);
    free_all(g);
    delete g;
}

int orbfe_ingest_slots(const orbfe_ingest *g) { return g ? (int)g->slots.size() : 0; }
size_t orbfe_ingest_frame_bytes(const orbfe_ingest *g) { return g ? g->frame_bytes : 0; }

uint8_t *orbfe_ingest_host_frames(orbfe_ingest *g, int slot)
{
    if (!g || slot < 0 || slot >= (int)g->slots.size()) return nullptr;
    return g->slots[slot].h_frames;
}

int orbfe_ingest_submit(orbfe_ingest *g, int slot, int n_frames)
{
    if (!g) return ORBFE_ERR_INVALID_ARG;
    if (slot < 0 || slot >= (int)g->slots.size()) ING_FAIL(g, ORBFE_ERR_INVALID_ARG, "ingest_submit: slot %d of %zu", slot, g->slots.size());
    if (n_frames < 1 || n_frames > g->cfg.frames_per_slot)
        ING_FAIL(g, ORBFE_ERR_CAPACITY, "ingest_submit: %d frames, a slot holds 1..%d", n_frames, g->cfg.frames_per_slot);
    Slot &s = g->slots[slot];
    if (s.in_flight)
        ING_FAIL(g, ORBFE_ERR_CAPACITY, "ingest_submit: slot %d is still in flight (the ring is full: orbfe_ingest_wait it first)", slot);
    orbfe::DeviceScope dev(g->device);
    if (!dev.ok) ING_FAIL(g, ORBFE_ERR_NO_DEVICE, "ingest_submit: cannot select HIP device %d", g->device);
    const size_t up = (size_t)n_frames * g->frame_bytes;
    const size_t recs = (size_t)n_frames * g->cap * sizeof(orbfe_keypoint);
    const size_t mb = (size_t)(n_frames - 1) * g->cap * sizeof(int32_t);
    const bool match = g->cfg.match_mode >= 0 && n_frames >= 2;
    // copy-in
    ING_HIP(g, hipEventRecord(s.ev[0], g->s_in));
    ING_HIP(g, hipMemcpyAsync(s.d_frames, s.h_frames, up, hipMemcpyHostToDevice, g->s_in));
    ING_HIP(g, hipEventRecord(s.ev[1], g->s_in));
    // compute
    ING_HIP(g, hipStreamWaitEvent(g->s_cmp, s.ev[1], 0));
    ING_HIP(g, hipEventRecord(s.ev[2], g->s_cmp));
    int rc;
    if (g->cfg.channels == 3)
        rc = orbfe_extract_rgb(g->ctx, s.d_frames, (size_t)g->W * 3, g->frame_bytes, n_frames, s.d_records, s.d_counts, nullptr, g->s_cmp);
    else
        rc = orbfe_extract(g->ctx, s.d_frames, (size_t)g->W, g->frame_bytes, n_frames, s.d_records, s.d_counts, nullptr, g->s_cmp);
    if (rc == ORBFE_OK && match)
        rc = orbfe_match_batch(g->ctx, s.d_records, s.d_counts, n_frames, g->cfg.match_mode, g->cfg.match_window,
                               g->cfg.match_max_distance, s.d_idx, s.d_dist, g->s_cmp);
    if (rc != ORBFE_OK) ING_FAIL(g, rc, "ingest_submit: %s", orbfe_last_error(g->ctx));
    ING_HIP(g, hipEventRecord(s.ev[3], g->s_cmp));
    // copy-out
    ING_HIP(g, hipStreamWaitEvent(g->s_out, s.ev[3], 0));
    ING_HIP(g, hipEventRecord(s.ev[4], g->s_out));
    ING_HIP(g, hipMemcpyAsync(s.h_counts, s.d_counts, (size_t)n_frames * sizeof(int32_t), hipMemcpyDeviceToHost, g->s_out));
    ING_HIP(g, hipMemcpyAsync(s.h_records, s.d_records, recs, hipMemcpyDeviceToHost, g->s_out));
    size_t down = recs + (size_t)n_frames * sizeof(int32_t);
    if (match && s.h_idx) {
        ING_HIP(g, hipMemcpyAsync(s.h_idx, s.d_idx, mb, hipMemcpyDeviceToHost, g->s_out));
        down += mb;
    }
    if (match && s.h_dist) {
        ING_HIP(g, hipMemcpyAsync(s.h_dist, s.d_dist, mb, hipMemcpyDeviceToHost, g->s_out));
        down += mb;
    }
    ING_HIP(g, hipEventRecord(s.ev[5], g->s_out));
    s.in_flight = 1;
    s.n_frames = n_frames;
    s.up_bytes = up;
    s.down_bytes = down;
    return ORBFE_OK;
}

int orbfe_ingest_submit_from(orbfe_ingest *g, int slot, int n_frames, const uint8_t *frames, size_t pitch, size_t frame_stride)
{
    if (!g) return ORBFE_ERR_INVALID_ARG;
    if (slot < 0 || slot >= (int)g->slots.size()) ING_FAIL(g, ORBFE_ERR_INVALID_ARG, "ingest_submit_from: slot %d of %zu", slot, g->slots.size());
    if (n_frames < 1 || n_frames > g->cfg.frames_per_slot)
        ING_FAIL(g, ORBFE_ERR_CAPACITY, "ingest_submit_from: %d frames, a slot holds 1..%d", n_frames, g->cfg.frames_per_slot);
    const size_t row = (size_t)g->W * g->cfg.channels;
    if (!frames || pitch < row || (n_frames > 1 && frame_stride < (size_t)(g->H - 1) * pitch + row))
        ING_FAIL(g, ORBFE_ERR_INVALID_ARG, "ingest_submit_from: null frames, pitch < width * channels or overlapping frames");
    Slot &s = g->slots[slot];
    if (s.in_flight)
        ING_FAIL(g, ORBFE_ERR_CAPACITY, "ingest_submit_from: slot %d is still in flight (orbfe_ingest_wait it first)", slot);
    for (int f = 0; f < n_frames; ++f) {
        const uint8_t *src = frames + (size_t)f * frame_stride;
        uint8_t *dst = s.h_frames + (size_t)f * g->frame_bytes;
        if (pitch == row)
            std::memcpy(dst, src, g->frame_bytes);
        else
            for (int y = 0; y < g->H; ++y) std::memcpy(dst + (size_t)y * row, src + (size_t)y * pitch, row);
    }
    return orbfe_ingest_submit(g, slot, n_frames);
}

int orbfe_ingest_ready(orbfe_ingest *g, int slot)
{
    if (!g || slot < 0 || slot >= (int)g->slots.size()) return 0;
    Slot &s = g->slots[slot];
    if (!s.in_flight) return 1;
    orbfe::DeviceScope dev(g->device);
    return hipEventQuery(s.ev[5]) == hipSuccess ? 1 : 0;
}

int orbfe_ingest_wait(orbfe_ingest *g, int slot, const orbfe_keypoint **records, const int32_t **counts,
                      const int32_t **match_idx, const int32_t **match_dist)
{
    if (!g) return ORBFE_ERR_INVALID_ARG;
    if (slot < 0 || slot >= (int)g->slots.size()) ING_FAIL(g, ORBFE_ERR_INVALID_ARG, "ingest_wait: slot %d of %zu", slot, g->slots.size());
    Slot &s = g->slots[slot];
    if (!s.in_flight) ING_FAIL(g, ORBFE_ERR_INVALID_ARG, "ingest_wait: slot %d is not in flight", slot);
    orbfe::DeviceScope dev(g->device);
    ING_HIP(g, hipEventSynchronize(s.ev[5]));
    s.in_flight = 0;
    s.completed = 1;
    const bool match = g->cfg.match_mode >= 0 && s.n_frames >= 2;
    if (records) *records = s.h_records;
    if (counts) *counts = s.h_counts;
    if (match_idx) *match_idx = match ? s.h_idx : nullptr;
    if (match_dist) *match_dist = match ? s.h_dist : nullptr;
    return ORBFE_OK;
}

int orbfe_ingest_device_buffers(orbfe_ingest *g, int slot, const uint8_t **d_frames, const orbfe_keypoint **d_records,
                                const int32_t **d_counts, const int32_t **d_match_idx, const int32_t **d_match_dist)
{
    if (!g) return ORBFE_ERR_INVALID_ARG;
    if (slot < 0 || slot >= (int)g->slots.size()) ING_FAIL(g, ORBFE_ERR_INVALID_ARG, "ingest_device_buffers: slot %d of %zu", slot, g->slots.size());
    Slot &s = g->slots[slot];
    if (d_frames) *d_frames = s.d_frames;
    if (d_records) *d_records = s.d_records;
    if (d_counts) *d_counts = s.d_counts;
    if (d_match_idx) *d_match_idx = s.d_idx;
    if (d_match_dist) *d_match_dist = s.d_dist;
    return ORBFE_OK;
}

orbfe_stream_t orbfe_ingest_compute_stream(orbfe_ingest *g) { return g ? (orbfe_stream_t)g->s_cmp : nullptr; }

int orbfe_ingest_timing(orbfe_ingest *g, int slot, float *upload_ms, float *compute_ms, float *download_ms,
                        size_t *upload_bytes, size_t *download_bytes)
{
    if (!g) return ORBFE_ERR_INVALID_ARG;
    if (slot < 0 || slot >= (int)g->slots.size()) ING_FAIL(g, ORBFE_ERR_INVALID_ARG, "ingest_timing: slot %d of %zu", slot, g->slots.size());
    Slot &s = g->slots[slot];
    if (s.in_flight || !s.completed) ING_FAIL(g, ORBFE_ERR_INVALID_ARG, "ingest_timing: slot %d has no completed pass (or is in flight)", slot);
    orbfe::DeviceScope dev(g->device);
    float t[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < 3; ++i) ING_HIP(g, hipEventElapsedTime(&t[i], s.ev[2 * i], s.ev[2 * i + 1]));
    if (upload_ms) *upload_ms = t[0];
    if (compute_ms) *compute_ms = t[1];
    if (download_ms) *download_ms = t[2];
    if (upload_bytes) *upload_bytes = s.up_bytes;
    if (download_bytes) *download_bytes = s.down_bytes;
    return ORBFE_OK;
}

} // extern "C"
