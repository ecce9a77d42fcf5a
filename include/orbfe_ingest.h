/* orbfe_ingest.h -- C ABI of the host <-> device staging ring of the ORB front end (inside liborbfe.so).
 *
 * SURVEY.md 8f-1, the half that is not arithmetic: the reference's buildStream thread feeds HOST frames and reads
 * HOST results, one frame at a time, with four copies per frame on the streams of its pipeline
 *   - depth H2D   cudaMemcpyAsync(d_depth_in, rgbd_frame->depth_image, ...)        src/SlamGpuPipeline/buildStream.cpp:376-381
 *   - colour H2D  cudaMemcpy2DAsync(d_rgb_image, rgb_pitch, rgbd_frame->rgb_image) src/SlamGpuPipeline/buildStream.cpp:399-406
 *   - grid D2H    cudaMemcpyAsync(h_feature_grid, d_feature_grid, ...)             src/SlamGpuPipeline/buildStream.cpp:462-466
 *   - points D2H  cudaMemcpyAsync(h_points, slam_frame->d_points, ...)             src/SlamGpuPipeline/buildStream.cpp:483-487
 * from pageable memory (`new float[]`, the camera's frame buffer), so every copy is staged by the driver and the
 * stream stalls on it.  Here the same traffic is a ring of SLOTS, each a batch of frames:
 *
 *      host, pinned (hipHostMalloc)          device                                host, pinned
 *      frames[slot]  --- copy-in stream -->  d_frames[slot]
 *                                            orbfe_extract / orbfe_extract_rgb
 *                                            (+ orbfe_match_batch)   compute stream
 *                                            d_records[slot], d_counts[slot] --- copy-out stream --> records[slot] ...
 *
 * Three streams owned by the object, ordered by events only: while slot s computes, slot s + 1 uploads and slot
 * s - 1 downloads (PCIe is full duplex and the copy engines take no compute unit).  The caller's loop is
 *
 *      for (;;) {
 *          fill orbfe_ingest_host_frames(ing, s) ...          // the camera / decoder writes straight into pinned memory
 *          orbfe_ingest_submit(ing, s, n);                    // returns at once
 *          s = (s + 1) % slots;
 *          if (slot s is in flight) orbfe_ingest_wait(ing, s, &rec, &cnt, &idx, &dist);   // results of `slots` submits ago
 *      }
 *
 * The device-resident entry points of orbfe.h stay as they are (caller-owned device buffers); this object only owns
 * the buffers of its ring and calls those entry points on its compute stream.  One orbfe_ingest = one orbfe_ctx = one
 * host thread at a time.  No CPU fallback: without a HIP device orbfe_ingest_create fails.
 */
#ifndef ORBFE_INGEST_H
#define ORBFE_INGEST_H

#include <stddef.h>
#include <stdint.h>

#include "orbfe.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orbfe_ingest orbfe_ingest;

typedef struct orbfe_ingest_config {
    int32_t slots;              /* ring depth, 2..16 (3 = upload / compute / download all busy)                  */
    int32_t frames_per_slot;    /* 1 .. the context's max_batch                                                  */
    int32_t channels;           /* 1: gray frames -> orbfe_extract; 3: interleaved RGB8 -> orbfe_extract_rgb     */
    int32_t match_mode;         /* -1: extraction only; 0 / 1: orbfe_match_batch(mode) over the slot's frames    */
                                /* (frame f - 1 -> f inside the slot, as bench.py's step)                        */
    int32_t match_window;       /* orbfe_match_batch's window                                                    */
    int32_t match_max_distance; /* ... and max_distance                                                          */
    int32_t download_matches;   /* 0: records + counts only; 1: + match indices; 2: + distances                  */
    int32_t reserved;           /* 0                                                                             */
} orbfe_ingest_config;

/* {3 slots, frames_per_slot, gray, no matching}. */
void orbfe_ingest_default_config(orbfe_ingest_config *cfg, int frames_per_slot);

/* Allocates, per slot: pinned host frames + device frames (frames_per_slot * width * height * channels bytes),
 * device and pinned host records (frames_per_slot * orbfe_max_keypoints() * 52 bytes) and counts, and when matching
 * the index / distance arrays ((frames_per_slot - 1) * max_keypoints int32 each); three non-blocking streams and
 * six events per slot.  `ctx` must outlive the object and must not be used by other calls while slots are in flight. */
int orbfe_ingest_create(orbfe_ctx *ctx, const orbfe_ingest_config *cfg, orbfe_ingest **out);
/* Drains the three streams, then frees everything. */
void orbfe_ingest_destroy(orbfe_ingest *ing);
/* ing == NULL: last error of orbfe_ingest_create on the calling thread. */
const char *orbfe_ingest_last_error(const orbfe_ingest *ing);

int orbfe_ingest_slots(const orbfe_ingest *ing);
size_t orbfe_ingest_frame_bytes(const orbfe_ingest *ing); /* width * height * channels, frames are contiguous */

/* The slot's pinned input buffer: frame f at + f * orbfe_ingest_frame_bytes(), rows contiguous (pitch = width *
 * channels).  The producer writes here; it may do so from the moment the slot is free until orbfe_ingest_submit. */
uint8_t *orbfe_ingest_host_frames(orbfe_ingest *ing, int slot);

/* Enqueue slot `slot` with its first n_frames frames: upload, extraction (+ matching), download; returns at once.
 * The slot must be free: never submitted, or waited for since (ORBFE_ERR_CAPACITY otherwise: the ring is full). */
int orbfe_ingest_submit(orbfe_ingest *ing, int slot, int n_frames);

/* The same for frames that live in the caller's own (pageable) memory, as rgbd_frame->rgb_image does in the
 * reference: rows of width * channels bytes at `pitch`, frames `frame_stride` bytes apart, are copied into the
 * slot's pinned buffer by the calling thread (this is the copy the CUDA driver does behind cudaMemcpy2DAsync from
 * pageable memory, buildStream.cpp:399-406), then submitted. */
int orbfe_ingest_submit_from(orbfe_ingest *ing, int slot, int n_frames, const uint8_t *frames, size_t pitch,
                             size_t frame_stride);

/* 1 when the slot's results have arrived on the host (or the slot is free), 0 while in flight; never blocks. */
int orbfe_ingest_ready(orbfe_ingest *ing, int slot);

/* Block the calling thread until the slot's results are on the host and hand out the pinned result buffers
 * (any of the four may be NULL): records frame-major, orbfe_max_keypoints() per frame, counts per frame, and the
 * matcher's outputs for pair f - 1 -> f at (f - 1) * max_keypoints (NULL unless downloaded).  The slot is free
 * afterwards; the buffers stay valid until it is submitted again.  A slot that is not in flight: ORBFE_ERR_INVALID_ARG. */
int orbfe_ingest_wait(orbfe_ingest *ing, int slot, const orbfe_keypoint **records, const int32_t **counts,
                      const int32_t **match_idx, const int32_t **match_dist);

/* For device-side consumers (orbfe_keypoint_pixel_to_point, a gather over RCCL ...): the slot's device buffers and
 * the compute stream; a consumer enqueued on that stream after orbfe_ingest_submit(slot) sees the slot's results.  The
 * buffers are rewritten by the next submit of the same slot. */
int orbfe_ingest_device_buffers(orbfe_ingest *ing, int slot, const uint8_t **d_frames, const orbfe_keypoint **d_records,
                                const int32_t **d_counts, const int32_t **d_match_idx, const int32_t **d_match_dist);
orbfe_stream_t orbfe_ingest_compute_stream(orbfe_ingest *ing);

/* Durations of the slot's last completed pass, from the events that order it (milliseconds; upload, extraction +
 * matching, download), and the bytes its upload and download moved.  The slot must not be in flight. */
int orbfe_ingest_timing(orbfe_ingest *ing, int slot, float *upload_ms, float *compute_ms, float *download_ms,
                        size_t *upload_bytes, size_t *download_bytes);

#ifdef __cplusplus
}
#endif
#endif /* ORBFE_INGEST_H */
