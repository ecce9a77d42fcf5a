// dist_rccl.hip -- liborbfe_dist.so: the multi-GPU leg behind include/orbfe_dist.h.
// RCCL (ncclCommInitRank, grouped ncclSend / ncclRecv, ncclAllReduce) + HIP only; gfx950 only.
// One orbfe_dist = one rank = one device, as the reference runs one buildStream thread per stream
// (src/SlamGpuPipeline/SlamGpuPipeline.cpp:43-50).  Collectives run on a stream owned by the
// object, ordered behind the caller's stream by an event, so the caller's next kernels overlap the
// transfer (xGMI is point to point: every peer -> root transfer rides its own link).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/orbfe_dist.h"

namespace {

thread_local char t_err[512];

void set_err(char *dst, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst ? dst : t_err, 512, fmt, ap);
    va_end(ap);
}

// Current-device guard: RCCL and stream calls need the rank's device current; the caller's
// current device is restored on return.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// Dense packing of the valid records of a rank's frames (exact-length gather): frame f's
// counts[f] records go to offset sum(counts[0..f-1]).  One workgroup per frame; the prefix is a
// block reduction over the f preceding counts (n_frames is at most a few thousand).
__global__ void __launch_bounds__(256)
pack_records_kernel(const uint32_t *__restrict__ records, const int32_t *__restrict__ counts, int cap,
                    uint32_t *__restrict__ dense)
{
    __shared__ int s_part[4];
    const int f = blockIdx.x, tid = threadIdx.x;
    int part = 0;
    for (int i = tid; i < f; i += 256) {
        const int c = counts[i];
        part += c < 0 ? 0 : (c > cap ? cap : c);
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if ((tid & 63) == 0) s_part[tid >> 6] = part;
    __syncthreads();
    const size_t base = (size_t)(s_part[0] + s_part[1] + s_part[2] + s_part[3]);
    int n = counts[f];
    n = n < 0 ? 0 : (n > cap ? cap : n);
    const uint32_t *src = records + (size_t)f * cap * 13;
    uint32_t *dst = dense + base * 13;
    for (int i = tid; i < n * 13; i += 256) dst[i] = src[i];
}

} // namespace

struct orbfe_dist {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t cs = nullptr;      // communication stream
    hipEvent_t ev_in = nullptr;    // caller's stream -> cs
    static constexpr int kRing = 16;
    hipEvent_t ev_done[kRing] = {}; // cs -> waiters; collective number t completes at ev_done[t % kRing]
    int64_t ticket = 0;             // number of collectives issued so far (0 = none)
    double *d_red = nullptr;       // host_allreduce staging (device)
    int red_cap = 0;
    int32_t *h_counts = nullptr;   // pinned: counts of every rank (exact-length gather)
    size_t h_counts_cap = 0;
    uint8_t *d_pack = nullptr;     // dense records of this rank (exact-length gather)
    size_t pack_cap = 0;
    char err[512] = {0};
};

#define D_FAIL(d, code, ...)                                                                \
    do {                                                                                    \
        set_err((d) ? (d)->err : nullptr, __VA_ARGS__);                                     \
        return code;                                                                        \
    } while (0)

#define D_HIP(d, expr)                                                                      \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            D_FAIL(d, ORBFE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                   __FILE__, __LINE__);                                                     \
    } while (0)

#define D_NCCL(d, expr)                                                                     \
    do {                                                                                    \
        ncclResult_t r_ = (expr);                                                           \
        if (r_ != ncclSuccess)                                                              \
            D_FAIL(d, ORBFE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), \
                   __FILE__, __LINE__);                                                     \
    } while (0)

// Inside an ncclGroupStart() / ncclGroupEnd() pair: a failed call must still close the group before the
// function returns, or every later call on this communicator is queued into a group that never ends.
#define D_NCCL_G(d, expr)                                                                   \
    do {                                                                                    \
        ncclResult_t r_ = (expr);                                                           \
        if (r_ != ncclSuccess) {                                                            \
            (void)ncclGroupEnd();                                                           \
            D_FAIL(d, ORBFE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), \
                   __FILE__, __LINE__);                                                     \
        }                                                                                   \
    } while (0)

#define D_GUARD(d, g, what)                                                                 \
    do {                                                                                    \
        if (!(g).ok) D_FAIL(d, ORBFE_ERR_HIP, "%s: hipSetDevice(%d) failed", what, (d)->device); \
    } while (0)

static inline hipStream_t S(orbfe_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Exact-length layout on the root: record index at which frame f of rank r starts (see orbfe_dist_exact_offsets)
__global__ void __launch_bounds__(256)
exact_offsets_kernel(const int32_t *__restrict__ all_counts, int n_frames, int cap, int64_t *__restrict__ offsets)
{
    __shared__ int s_run;
    const int r = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int f0 = 0; f0 < n_frames; f0 += 256) { // chunks of 256 frames with a running base
        const int f = f0 + tid;
        int c = 0;
        if (f < n_frames) {
            c = all_counts[(size_t)r * n_frames + f];
            c = c < 0 ? 0 : (c > cap ? cap : c);
        }
        // inclusive scan inside the wave, then across the four waves through LDS
        int incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if ((tid & 63) >= off) incl += v;
        }
        __shared__ int s_w[4];
        if ((tid & 63) == 63) s_w[tid >> 6] = incl;
        __syncthreads();
        int before = s_run;
        for (int w = 0; w < (tid >> 6); w++) before += s_w[w];
        if (f < n_frames) offsets[(size_t)r * n_frames + f] = (int64_t)r * n_frames * cap + before + incl - c;
        __syncthreads();
        if (tid == 255) s_run = before + incl;
        __syncthreads();
    }
}

// cs waits for everything enqueued on the caller's stream so far
static int order_after(orbfe_dist *d, orbfe_stream_t stream)
{
    D_HIP(d, hipEventRecord(d->ev_in, S(stream)));
    D_HIP(d, hipStreamWaitEvent(d->cs, d->ev_in, 0));
    return ORBFE_OK;
}

static int mark_done(orbfe_dist *d)
{
    d->ticket++;
    D_HIP(d, hipEventRecord(d->ev_done[d->ticket % orbfe_dist::kRing], d->cs));
    return ORBFE_OK;
}

extern "C" {

const char *orbfe_dist_last_error(const orbfe_dist *d) { return d ? d->err : t_err; }

int orbfe_dist_unique_id(uint8_t *id)
{
    static_assert(ORBFE_DIST_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id) D_FAIL((orbfe_dist *)nullptr, ORBFE_ERR_INVALID_ARG, "orbfe_dist_unique_id: null");
    ncclUniqueId u;
    D_NCCL((orbfe_dist *)nullptr, ncclGetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return ORBFE_OK;
}

int orbfe_dist_shard_range(int n_total, int rank, int world, int *begin, int *end)
{
    if (n_total < 0 || world < 1 || rank < 0 || rank >= world || !begin || !end)
        D_FAIL((orbfe_dist *)nullptr, ORBFE_ERR_INVALID_ARG, "orbfe_dist_shard_range: rank %d of %d, %d frames", rank,
               world, n_total);
    const int base = n_total / world, rem = n_total % world;
    *begin = rank * base + (rank < rem ? rank : rem);
    *end = *begin + base + (rank < rem ? 1 : 0);
    return ORBFE_OK;
}

int orbfe_dist_create(const uint8_t *id, int rank, int world, int device, orbfe_dist **out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world)
        D_FAIL((orbfe_dist *)nullptr, ORBFE_ERR_INVALID_ARG, "orbfe_dist_create: rank %d of %d", rank, world);
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        D_FAIL((orbfe_dist *)nullptr, ORBFE_ERR_NO_DEVICE, "orbfe_dist_create: no HIP device (no CPU fallback)");
    if (device < 0 || device >= ndev)
        D_FAIL((orbfe_dist *)nullptr, ORBFE_ERR_INVALID_ARG, "orbfe_dist_create: device %d of %d", device, ndev);
    DeviceGuard g(device);
    if (!g.ok) D_FAIL((orbfe_dist *)nullptr, ORBFE_ERR_HIP, "orbfe_dist_create: hipSetDevice(%d) failed", device);
    orbfe_dist *d = new orbfe_dist();
    d->rank = rank;
    d->world = world;
    d->device = device;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&d->comm, world, u, rank);
    hipError_t e = hipSuccess;
    if (r == ncclSuccess) e = hipStreamCreateWithFlags(&d->cs, hipStreamNonBlocking);
    if (r == ncclSuccess && e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_in, hipEventDisableTiming);
    for (int i = 0; i < orbfe_dist::kRing; i++)
        if (r == ncclSuccess && e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_done[i], hipEventDisableTiming);
    if (r != ncclSuccess || e != hipSuccess) {
        set_err(nullptr, "orbfe_dist_create: %s", r != ncclSuccess ? ncclGetErrorString(r) : hipGetErrorString(e));
        orbfe_dist_destroy(d);
        return ORBFE_ERR_HIP;
    }
    *out = d;
    return ORBFE_OK;
}

void orbfe_dist_destroy(orbfe_dist *d)
{
    if (!d) return;
    DeviceGuard g(d->device);
    if (d->cs) (void)hipStreamSynchronize(d->cs);
    if (d->comm) (void)ncclCommDestroy(d->comm);
    if (d->ev_in) (void)hipEventDestroy(d->ev_in);
    for (int i = 0; i < orbfe_dist::kRing; i++)
        if (d->ev_done[i]) (void)hipEventDestroy(d->ev_done[i]);
    if (d->cs) (void)hipStreamDestroy(d->cs);
    if (d->d_red) (void)hipFree(d->d_red);
    if (d->h_counts) (void)hipHostFree(d->h_counts);
    if (d->d_pack) (void)hipFree(d->d_pack);
    delete d;
}

int orbfe_dist_rank(const orbfe_dist *d) { return d ? d->rank : -1; }
int orbfe_dist_world(const orbfe_dist *d) { return d ? d->world : 0; }

int64_t orbfe_dist_ticket(const orbfe_dist *d) { return d ? d->ticket : 0; }

int orbfe_dist_wait_ticket(orbfe_dist *d, int64_t ticket, orbfe_stream_t stream)
{
    if (!d) return ORBFE_ERR_INVALID_ARG;
    if (ticket <= 0 || d->ticket == 0) return ORBFE_OK; // nothing was issued
    if (ticket > d->ticket) D_FAIL(d, ORBFE_ERR_INVALID_ARG, "wait_ticket: ticket %lld was never issued", (long long)ticket);
    // the communication stream is in order: a ticket that has left the ring is covered by the
    // oldest event still in it
    const int64_t oldest = d->ticket - orbfe_dist::kRing + 1;
    const int64_t t = ticket < oldest ? oldest : ticket;
    DeviceGuard g(d->device);
    D_GUARD(d, g, "wait_ticket");
    D_HIP(d, hipStreamWaitEvent(S(stream), d->ev_done[t % orbfe_dist::kRing], 0));
    return ORBFE_OK;
}

int orbfe_dist_wait(orbfe_dist *d, orbfe_stream_t stream)
{
    return d ? orbfe_dist_wait_ticket(d, d->ticket, stream) : ORBFE_ERR_INVALID_ARG;
}

int orbfe_dist_sync(orbfe_dist *d)
{
    if (!d) return ORBFE_ERR_INVALID_ARG;
    DeviceGuard g(d->device);
    D_GUARD(d, g, "sync");
    D_HIP(d, hipStreamSynchronize(d->cs));
    return ORBFE_OK;
}

int orbfe_dist_exact_offsets(orbfe_dist *d, const int32_t *d_all_counts, int n_frames, int cap, int64_t *d_offsets,
                             orbfe_stream_t stream)
{
    if (!d) return ORBFE_ERR_INVALID_ARG;
    if (!d_all_counts || !d_offsets || n_frames < 1 || cap < 1)
        D_FAIL(d, ORBFE_ERR_INVALID_ARG, "exact_offsets: bad argument (n_frames %d, cap %d)", n_frames, cap);
    DeviceGuard g(d->device);
    D_GUARD(d, g, "exact_offsets");
    hipLaunchKernelGGL(exact_offsets_kernel, dim3(d->world), dim3(256), 0, S(stream), d_all_counts, n_frames, cap, d_offsets);
    D_HIP(d, hipGetLastError());
    return ORBFE_OK;
}

int orbfe_dist_gather_keypoints(orbfe_dist *d, const orbfe_keypoint *d_records, const int32_t *d_counts,
                                int n_frames, int cap, orbfe_keypoint *d_all_records, int32_t *d_all_counts,
                                int root, int exact, orbfe_stream_t stream)
{
    if (!d) return ORBFE_ERR_INVALID_ARG;
    if (!d_records || !d_counts || n_frames < 1 || cap < 1 || root < 0 || root >= d->world)
        D_FAIL(d, ORBFE_ERR_INVALID_ARG, "gather_keypoints: bad argument (n_frames %d, cap %d, root %d)", n_frames, cap, root);
    const bool is_root = d->rank == root;
    if (is_root && (!d_all_records || !d_all_counts))
        D_FAIL(d, ORBFE_ERR_INVALID_ARG, "gather_keypoints: the root needs d_all_records and d_all_counts");
    static_assert(sizeof(orbfe_keypoint) == 52, "record size");
    DeviceGuard g(d->device);
    if (!g.ok) D_FAIL(d, ORBFE_ERR_HIP, "gather_keypoints: hipSetDevice(%d) failed", d->device);
    int rc = order_after(d, stream);
    if (rc != ORBFE_OK) return rc;
    const size_t cbytes = (size_t)n_frames * sizeof(int32_t);
    const size_t rstride = (size_t)n_frames * cap; // records per rank block
    const size_t rbytes = rstride * sizeof(orbfe_keypoint);

    // ---- counts (always fixed size)
    // (a root that extracts straight into its own block of the gathered arrays passes those addresses: no copy)
    if (is_root && d_all_counts + (size_t)root * n_frames != d_counts)
        D_HIP(d, hipMemcpyAsync(d_all_counts + (size_t)root * n_frames, d_counts, cbytes, hipMemcpyDeviceToDevice, d->cs));
    if (d->world > 1) {
        D_NCCL(d, ncclGroupStart());
        if (is_root) {
            for (int r = 0; r < d->world; r++)
                if (r != root) D_NCCL_G(d, ncclRecv(d_all_counts + (size_t)r * n_frames, cbytes, ncclUint8, r, d->comm, d->cs));
        } else {
            D_NCCL_G(d, ncclSend(d_counts, cbytes, ncclUint8, root, d->comm, d->cs));
        }
        D_NCCL(d, ncclGroupEnd());
    }

    if (!exact) {
        // ---- fixed stride: nothing touches the host
        if (is_root && d_all_records + (size_t)root * rstride != d_records)
            D_HIP(d, hipMemcpyAsync(d_all_records + (size_t)root * rstride, d_records, rbytes, hipMemcpyDeviceToDevice, d->cs));
        if (d->world > 1) {
            D_NCCL(d, ncclGroupStart());
            if (is_root) {
                for (int r = 0; r < d->world; r++)
                    if (r != root) D_NCCL_G(d, ncclRecv(d_all_records + (size_t)r * rstride, rbytes, ncclUint8, r, d->comm, d->cs));
            } else {
                D_NCCL_G(d, ncclSend(d_records, rbytes, ncclUint8, root, d->comm, d->cs));
            }
            D_NCCL(d, ncclGroupEnd());
        }
        return mark_done(d);
    }

    // ---- exact length: the host needs the totals (sender: its own; root: everybody's)
    const size_t n_host = (size_t)(is_root ? d->world : 1) * n_frames;
    if (d->h_counts_cap < n_host) {
        if (d->h_counts) (void)hipHostFree(d->h_counts);
        d->h_counts = nullptr;
        d->h_counts_cap = 0;
        D_HIP(d, hipHostMalloc((void **)&d->h_counts, n_host * sizeof(int32_t), hipHostMallocDefault));
        d->h_counts_cap = n_host;
    }
    D_HIP(d, hipMemcpyAsync(d->h_counts, is_root ? d_all_counts : d_counts, n_host * sizeof(int32_t),
                            hipMemcpyDeviceToHost, d->cs));
    D_HIP(d, hipStreamSynchronize(d->cs));
    auto total_of = [&](const int32_t *c) {
        size_t t = 0;
        for (int f = 0; f < n_frames; f++) t += (size_t)(c[f] < 0 ? 0 : (c[f] > cap ? cap : c[f]));
        return t;
    };
    uint32_t *pack_dst;
    if (is_root && d_all_records + (size_t)root * rstride == d_records)
        D_FAIL(d, ORBFE_ERR_INVALID_ARG, "gather_keypoints: the exact-length form packs the root's records into its block "
               "of d_all_records, which therefore cannot be the buffer they come from");
    if (is_root) {
        pack_dst = reinterpret_cast<uint32_t *>(d_all_records + (size_t)root * rstride);
    } else {
        if (d->pack_cap < rbytes) {
            if (d->d_pack) (void)hipFree(d->d_pack);
            d->d_pack = nullptr;
            d->pack_cap = 0;
            D_HIP(d, hipMalloc((void **)&d->d_pack, rbytes));
            d->pack_cap = rbytes;
        }
        pack_dst = reinterpret_cast<uint32_t *>(d->d_pack);
    }
    hipLaunchKernelGGL(pack_records_kernel, dim3(n_frames), dim3(256), 0, d->cs,
                       reinterpret_cast<const uint32_t *>(d_records), d_counts, cap, pack_dst);
    D_HIP(d, hipGetLastError());
    if (d->world > 1) {
        D_NCCL(d, ncclGroupStart());
        if (is_root) {
            for (int r = 0; r < d->world; r++) {
                if (r == root) continue;
                const size_t bytes = total_of(d->h_counts + (size_t)r * n_frames) * sizeof(orbfe_keypoint);
                if (bytes) D_NCCL_G(d, ncclRecv(d_all_records + (size_t)r * rstride, bytes, ncclUint8, r, d->comm, d->cs));
            }
        } else {
            const size_t bytes = total_of(d->h_counts) * sizeof(orbfe_keypoint);
            if (bytes) D_NCCL_G(d, ncclSend(d->d_pack, bytes, ncclUint8, root, d->comm, d->cs));
        }
        D_NCCL(d, ncclGroupEnd());
    }
    return mark_done(d);
}

int orbfe_dist_allreduce_max_keys(orbfe_dist *d, uint32_t *d_keys, size_t n, orbfe_stream_t stream)
{
    if (!d) return ORBFE_ERR_INVALID_ARG;
    if (!d_keys || n == 0) D_FAIL(d, ORBFE_ERR_INVALID_ARG, "allreduce_max_keys: bad argument");
    DeviceGuard g(d->device);
    D_GUARD(d, g, "allreduce_max_keys");
    int rc = order_after(d, stream);
    if (rc != ORBFE_OK) return rc;
    if (d->world > 1) D_NCCL(d, ncclAllReduce(d_keys, d_keys, n, ncclUint32, ncclMax, d->comm, d->cs));
    return mark_done(d);
}

int orbfe_dist_host_allreduce(orbfe_dist *d, double *values, int n, int op)
{
    if (!d) return ORBFE_ERR_INVALID_ARG;
    if (!values || n < 1 || (op != 0 && op != 1)) D_FAIL(d, ORBFE_ERR_INVALID_ARG, "host_allreduce: bad argument");
    if (d->world == 1) return ORBFE_OK;
    DeviceGuard g(d->device);
    D_GUARD(d, g, "host_allreduce");
    if (d->red_cap < n) {
        if (d->d_red) (void)hipFree(d->d_red);
        d->d_red = nullptr;
        d->red_cap = 0;
        D_HIP(d, hipMalloc((void **)&d->d_red, (size_t)n * sizeof(double)));
        d->red_cap = n;
    }
    D_HIP(d, hipMemcpyAsync(d->d_red, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, d->cs));
    D_NCCL(d, ncclAllReduce(d->d_red, d->d_red, (size_t)n, ncclDouble, op == 0 ? ncclMax : ncclSum, d->comm, d->cs));
    D_HIP(d, hipMemcpyAsync(values, d->d_red, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, d->cs));
    D_HIP(d, hipStreamSynchronize(d->cs));
    return ORBFE_OK;
}

int orbfe_dist_barrier(orbfe_dist *d)
{
    double one = 1.0;
    return orbfe_dist_host_allreduce(d, &one, 1, 1);
}

} // extern "C"
