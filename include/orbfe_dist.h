/* orbfe_dist.h -- C ABI of the multi-GPU leg of the ORB front end (liborbfe_dist.so).
 *
 * Frames are independent, so a batch shards across the GPUs of a node in contiguous blocks:
 * one host thread (or process) + one orbfe_ctx + one stream per device, exactly as the
 * reference runs one buildStream thread per stream (src/SlamGpuPipeline/SlamGpuPipeline.cpp:43-50).
 * Images never cross GPUs.  The only exchange steps are
 *   - the gather of the keypoint records (52 B each) and per-frame counts on one rank
 *     (grouped ncclSend / ncclRecv over xGMI: every peer uses its own link to the root), and
 *   - for ONE large frame sharded by detection tiles (orbfe_detect_batch_shard), an
 *     all-reduce(MAX) of the K per-cell keys before selection / description.
 * This library is RCCL + HIP only (no torch, no dependency on liborbfe.so); device pointers are
 * plain addresses.  A communicator is created from an RCCL unique id that the application hands
 * from rank 0 to the other ranks (any side channel: a file, a socket, torch.distributed's store).
 *
 * Stream model: collectives run on a communication stream owned by the orbfe_dist object.  A
 * call orders them behind everything already enqueued on the caller's stream (event), returns at
 * once, and the caller's later kernels overlap the transfer; orbfe_dist_wait() makes a stream
 * wait for the outstanding collectives, orbfe_dist_sync() blocks the host.
 *
 * There is NO CPU fallback: without a HIP device / RCCL every call returns an error status.
 *
 * VERIFICATION STATUS: with one rank (world = 1) every entry point is exercised on hardware by the GPU tests.  The code
 * under `world > 1` (grouped ncclSend / ncclRecv, root placement, exact-length packing, the all-reduce) runs in the GPU
 * tests with 2 and 3 ranks on ONE device over a test-only loopback stand-in for librccl.so.1 (tests/fake_rccl: real RCCL
 * refuses two ranks on one GPU), against the oracle.  What has not executed on any multi-GPU node available to the build
 * is RCCL's own xGMI transport -- see DESIGN.md section 5.
 */
#ifndef ORBFE_DIST_H
#define ORBFE_DIST_H

#include <stddef.h>
#include <stdint.h>

#include "orbfe.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORBFE_DIST_ID_BYTES 128 /* == NCCL_UNIQUE_ID_BYTES */

typedef struct orbfe_dist orbfe_dist;

/* ncclGetUniqueId: call on ONE rank, hand the 128 bytes to every rank. */
int orbfe_dist_unique_id(uint8_t *id);

/* ncclCommInitRank on HIP device `device` (collective: every rank of `world` must call it). */
int orbfe_dist_create(const uint8_t *id, int rank, int world, int device, orbfe_dist **out);
void orbfe_dist_destroy(orbfe_dist *d);
int orbfe_dist_rank(const orbfe_dist *d);
int orbfe_dist_world(const orbfe_dist *d);
/* d == NULL: last error of orbfe_dist_unique_id / orbfe_dist_create on the calling thread. */
const char *orbfe_dist_last_error(const orbfe_dist *d);

/* Contiguous block [*begin, *end) of n_total frames owned by `rank` of `world` (block sizes differ
 * by at most one; matching pairs (t-1, t) stay inside a block).  Pure host arithmetic. */
int orbfe_dist_shard_range(int n_total, int rank, int world, int *begin, int *end);

/* Gather of keypoint records on `root`, asynchronous.
 *   d_records / d_counts : this rank's n_frames * cap records and n_frames counts (as written by
 *                          orbfe_extract).  Every rank passes the same n_frames and cap.
 *   d_all_records        : root only: world * n_frames * cap records, rank-major (== frame order,
 *                          because shards are contiguous); d_all_counts: world * n_frames ints.
 *   exact == 0 : fixed stride -- every rank ships n_frames * cap * 52 bytes; nothing touches the host.  A root
 *                that extracted straight into its own block (d_records == d_all_records + root * n_frames * cap,
 *                d_counts likewise) copies nothing.
 *   exact != 0 : variable length -- counts travel first and the host reads them (this call blocks
 *                until this rank's extraction has finished); every rank packs its valid records
 *                densely on the device and ships exactly sum(counts) * 52 bytes.  On the root,
 *                rank r's block still starts at record r * n_frames * cap, but inside the block
 *                the frames are dense: frame f starts at sum(counts_r[0 .. f-1]).
 * Ordered after the work already enqueued on `stream`; runs on the communicator's stream. */
int orbfe_dist_gather_keypoints(orbfe_dist *d, const orbfe_keypoint *d_records, const int32_t *d_counts,
                                int n_frames, int cap, orbfe_keypoint *d_all_records,
                                int32_t *d_all_counts, int root, int exact, orbfe_stream_t stream);

/* Root side of the exact-length form: where does frame f of rank r start?  Writes world * n_frames record
 * indices into d_all_records (int64): offsets[r * n_frames + f] = r * n_frames * cap + sum(counts_r[0 .. f-1]),
 * from the gathered d_all_counts (counts clamped to [0, cap] as everywhere).  A kernel on `stream` (which must
 * already wait for the gather: orbfe_dist_wait); nothing touches the host.  With these and d_all_counts a consumer
 * walks the dense layout without re-deriving it; for the fixed-stride form the offsets are simply
 * (r * n_frames + f) * cap and no call is needed. */
int orbfe_dist_exact_offsets(orbfe_dist *d, const int32_t *d_all_counts, int n_frames, int cap,
                             int64_t *d_offsets, orbfe_stream_t stream);

/* All-reduce(MAX) of n unsigned 32-bit cell keys in place (tile-sharded detection of one frame;
 * keys are < 2^27).  Ordered after `stream`, runs on the communicator's stream; follow with
 * orbfe_dist_wait(d, stream) before orbfe_import_cell_keys / orbfe_describe_batch. */
int orbfe_dist_allreduce_max_keys(orbfe_dist *d, uint32_t *d_keys, size_t n, orbfe_stream_t stream);

/* `stream` waits (on the device, no host block) for every collective issued so far. */
int orbfe_dist_wait(orbfe_dist *d, orbfe_stream_t stream);
/* Finer grain: orbfe_dist_ticket() after a gather / all-reduce call names that collective;
 * orbfe_dist_wait_ticket() makes `stream` wait for it (and, the communication stream being in
 * order, for everything issued before it) but not for later ones -- e.g. step i + 2 may reuse the
 * record buffer of step i while the gather of step i + 1 is still in flight. */
int64_t orbfe_dist_ticket(const orbfe_dist *d);
int orbfe_dist_wait_ticket(orbfe_dist *d, int64_t ticket, orbfe_stream_t stream);
/* The host blocks until every collective issued so far has completed. */
int orbfe_dist_sync(orbfe_dist *d);

/* Host-side reductions for the harness (bench timing): values[] is replaced by the MAX (op 0) or
 * SUM (op 1) over ranks.  Blocking.  orbfe_dist_barrier = a 1-element all-reduce. */
int orbfe_dist_host_allreduce(orbfe_dist *d, double *values, int n, int op);
int orbfe_dist_barrier(orbfe_dist *d);

#ifdef __cplusplus
}
#endif
#endif /* ORBFE_DIST_H */
