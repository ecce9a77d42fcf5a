// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a loopback stand-in for librccl.so.1 so that the `world > 1` branches of
// liborbfe_dist.so (csrc/dist_rccl.hip: grouped ncclSend / ncclRecv, per-rank offsets, root placement, the all-reduce)
// can execute on a box with ONE GPU.  Real RCCL refuses two ranks on one device ("Duplicate GPU detected"), and the build
// pool never offers a multi-GPU node.  It is never linked into the product: tests put the directory that holds the built
// `librccl.so.1` in front of LD_LIBRARY_PATH of a CHILD process (liborbfe_dist.so resolves its DT_NEEDED librccl.so.1
// there; its RUNPATH comes after LD_LIBRARY_PATH), so the product binary under test is byte-for-byte the shipped one.
//
// It exports exactly the nine functions liborbfe_dist.so imports (nm -D --undefined-only): ncclGetUniqueId,
// ncclCommInitRank, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv, ncclAllReduce, ncclGetErrorString,
// with the signatures of <rccl/rccl.h>.  Semantics kept: point-to-point operations between a group's start and end are
// issued together and may complete in any order; every operation is ordered after the work already enqueued on its
// stream and its effects are visible to work enqueued on that stream afterwards.  Not kept: asynchrony (a call returns
// when the bytes have moved) and speed.  Ranks may be processes or threads: the transport is a POSIX shared-memory
// segment named after the unique id, one mailbox per ordered (source, destination) pair, bytes staged through the host.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr size_t kChunk = 1u << 20;      // bytes per mailbox slot
constexpr size_t kRedBytes = 4u << 20;   // all-reduce staging per rank
constexpr int kMaxWorld = 8;
constexpr double kTimeoutS = 60.0;       // a peer that never arrives fails the call instead of hanging the test

struct Mailbox {
    std::atomic<uint64_t> written; // chunks published by the source
    std::atomic<uint64_t> read;    // chunks consumed by the destination
    uint64_t bytes;                // of the chunk in flight
    uint8_t data[kChunk];
};

struct Segment {
    std::atomic<uint32_t> arrived;   // ranks that have mapped the segment
    std::atomic<uint32_t> left;      // ranks that have destroyed their communicator
    std::atomic<uint32_t> bar_count; // sense-reversing barrier
    std::atomic<uint32_t> bar_gen;
    uint32_t world;
    uint32_t pad[11];
    Mailbox box[kMaxWorld][kMaxWorld]; // [src][dst]
    uint8_t red[kMaxWorld][kRedBytes];
};

struct Op {
    bool send;
    void *dev;
    size_t bytes, done;
    int peer;
    hipStream_t stream;
    std::vector<uint8_t> host;
    bool staged; // send: device bytes are in `host`
};

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

thread_local int t_group_depth = 0;
thread_local std::vector<std::pair<struct ncclComm *, Op>> t_pending;

} // namespace

struct ncclComm {
    Segment *seg = nullptr;
    int rank = 0, world = 1;
    char name[NCCL_UNIQUE_ID_BYTES] = {0};
};

namespace {

ncclResult_t barrier(ncclComm *c)
{
    Segment *s = c->seg;
    const uint32_t gen = s->bar_gen.load();
    if (s->bar_count.fetch_add(1) + 1 == (uint32_t)c->world) {
        s->bar_count.store(0);
        s->bar_gen.fetch_add(1);
        return ncclSuccess;
    }
    const double t0 = now_s();
    while (s->bar_gen.load() == gen) {
        sched_yield();
        if (now_s() - t0 > kTimeoutS) return ncclSystemError;
    }
    return ncclSuccess;
}

// Progress loop over a set of point-to-point operations: every pass pushes / pulls at most one chunk per operation and
// never blocks on one, so any mixture of sends and receives between any ranks completes (no ordering assumptions).
ncclResult_t run_ops(std::vector<std::pair<ncclComm *, Op>> &ops)
{
    for (auto &co : ops) { // order after the stream's earlier work; stage the sources
        Op &o = co.second;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        o.host.resize(o.bytes);
        if (o.send && o.bytes) {
            if (hipMemcpy(o.host.data(), o.dev, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        }
        o.staged = true;
    }
    size_t open = 0;
    for (auto &co : ops)
        if (co.second.bytes) open++;
    const double t0 = now_s();
    while (open) {
        bool moved = false;
        for (size_t k = 0; k < ops.size(); k++) {
            auto &co = ops[k];
            ncclComm *c = co.first;
            Op &o = co.second;
            if (o.done == o.bytes) continue;
            bool earlier = false; // messages of one (source, destination) pair stay in posting order
            for (size_t q = 0; q < k && !earlier; q++)
                earlier = ops[q].first == c && ops[q].second.send == o.send && ops[q].second.peer == o.peer &&
                          ops[q].second.done != ops[q].second.bytes;
            if (earlier) continue;
            Mailbox &m = o.send ? c->seg->box[c->rank][o.peer] : c->seg->box[o.peer][c->rank];
            if (o.send) {
                if (m.written.load(std::memory_order_acquire) != m.read.load(std::memory_order_acquire)) continue; // slot busy
                const size_t n = o.bytes - o.done < kChunk ? o.bytes - o.done : kChunk;
                std::memcpy(m.data, o.host.data() + o.done, n);
                m.bytes = n;
                m.written.fetch_add(1, std::memory_order_release);
                o.done += n;
            } else {
                if (m.written.load(std::memory_order_acquire) == m.read.load(std::memory_order_acquire)) continue; // nothing yet
                const size_t n = m.bytes;
                if (n > o.bytes - o.done) return ncclInvalidUsage; // the peer sent more than this receive was posted for
                std::memcpy(o.host.data() + o.done, m.data, n);
                m.read.fetch_add(1, std::memory_order_release);
                o.done += n;
            }
            moved = true;
            if (o.done == o.bytes) open--;
        }
        if (!moved) {
            sched_yield();
            if (now_s() - t0 > kTimeoutS) return ncclSystemError;
        }
    }
    for (auto &co : ops) { // deliver the receives; visible to later work on the stream because the copy is synchronous
        Op &o = co.second;
        if (!o.send && o.bytes)
            if (hipMemcpy(o.dev, o.host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

ncclResult_t post(ncclComm *c, bool send, void *dev, size_t count, ncclDataType_t type, int peer, hipStream_t stream)
{
    if (!c || !c->seg || peer < 0 || peer >= c->world || peer == c->rank || (!dev && count)) return ncclInvalidArgument;
    const size_t ts = type_size(type);
    if (!ts) return ncclInvalidArgument;
    Op o;
    o.send = send;
    o.dev = dev;
    o.bytes = count * ts;
    o.done = 0;
    o.peer = peer;
    o.stream = stream;
    o.staged = false;
    t_pending.emplace_back(c, std::move(o));
    if (t_group_depth == 0) {
        std::vector<std::pair<ncclComm *, Op>> ops;
        ops.swap(t_pending);
        return run_ops(ops);
    }
    return ncclSuccess;
}

template <typename T>
void reduce_into(T *acc, const T *x, size_t n, ncclRedOp_t op)
{
    for (size_t i = 0; i < n; i++) {
        if (op == ncclMax) acc[i] = x[i] > acc[i] ? x[i] : acc[i];
        else if (op == ncclMin) acc[i] = x[i] < acc[i] ? x[i] : acc[i];
        else if (op == ncclSum) acc[i] = acc[i] + x[i];
        else acc[i] = acc[i] * x[i];
    }
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof(*id));
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    static std::atomic<uint32_t> serial{0};
    std::snprintf(id->internal, sizeof(id->internal), "/orbfe_fake_rccl_%d_%lx_%u", (int)getpid(),
                  (unsigned long)(ts.tv_sec * 1000000000L + ts.tv_nsec), serial.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > kMaxWorld || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    if (id.internal[0] != '/') return ncclInvalidArgument; // not an id of this library
    ncclComm *c = new ncclComm;
    c->rank = rank;
    c->world = nranks;
    std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(Segment)) != 0) { // a fresh segment is zero-filled: all counters start at 0
        if (fd >= 0) close(fd);
        delete c;
        return ncclSystemError;
    }
    void *p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->seg = static_cast<Segment *>(p);
    c->seg->world = (uint32_t)nranks;
    c->seg->arrived.fetch_add(1);
    const double t0 = now_s();
    while (c->seg->arrived.load() < (uint32_t)nranks) { // ncclCommInitRank is collective
        sched_yield();
        if (now_s() - t0 > kTimeoutS) {
            munmap(p, sizeof(Segment));
            delete c;
            return ncclSystemError;
        }
    }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclSuccess;
    if (comm->seg) {
        const bool last = comm->seg->left.fetch_add(1) + 1 == (uint32_t)comm->world;
        munmap(comm->seg, sizeof(Segment));
        if (last) shm_unlink(comm->name);
    }
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    t_group_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth > 0) return ncclSuccess;
    std::vector<std::pair<ncclComm *, Op>> ops;
    ops.swap(t_pending);
    return ops.empty() ? ncclSuccess : run_ops(ops);
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(comm, true, const_cast<void *>(sendbuff), count, datatype, peer, stream);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(comm, false, recvbuff, count, datatype, peer, stream);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream)
{
    if (!comm || !comm->seg || !sendbuff || !recvbuff) return ncclInvalidArgument;
    if (datatype != ncclUint32 && datatype != ncclInt32 && datatype != ncclFloat64) return ncclInvalidArgument; // what the library uses
    const size_t bytes = count * type_size(datatype);
    if (bytes > kRedBytes) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    Segment *s = comm->seg;
    if (hipMemcpy(s->red[comm->rank], sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t r = barrier(comm); // every contribution is in place
    if (r != ncclSuccess) return r;
    std::vector<uint8_t> acc(s->red[0], s->red[0] + bytes);
    for (int k = 1; k < comm->world; k++) {
        if (datatype == ncclUint32) reduce_into((uint32_t *)acc.data(), (const uint32_t *)s->red[k], count, op);
        else if (datatype == ncclInt32) reduce_into((int32_t *)acc.data(), (const int32_t *)s->red[k], count, op);
        else reduce_into((double *)acc.data(), (const double *)s->red[k], count, op);
    }
    r = barrier(comm); // everybody has read: the staging may be overwritten by the next call
    if (r != ncclSuccess) return r;
    if (hipMemcpy(recvbuff, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t result)
{
    switch (result) {
    case ncclSuccess: return "fake rccl: success";
    case ncclUnhandledCudaError: return "fake rccl: a HIP call failed";
    case ncclSystemError: return "fake rccl: shared-memory transport error or a peer did not arrive within 60 s";
    case ncclInvalidArgument: return "fake rccl: invalid argument";
    case ncclInvalidUsage: return "fake rccl: invalid usage";
    default: return "fake rccl: error";
    }
}

} // extern "C"
