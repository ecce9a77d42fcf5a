"""Frame sharding across ranks and the keypoint gather.

Frames are independent units, so a batch is split into contiguous blocks, one per rank (one
process per GPU); images never cross GPUs.  The only exchange step is a gather of the
keypoint records and per-frame counts to rank 0.  Matching pairs (t-1, t) stay on one rank
because shards are cut on frame boundaries and each rank matches inside its block.

Two implementations of the same exchange:
  * RcclComm -- the product path: liborbfe_dist.so (include/orbfe_dist.h), C++ host code on RCCL
    (ncclCommInitRank, grouped ncclSend / ncclRecv, ncclAllReduce) with its own communication
    stream.  bench.py uses it on GPUs.
  * gather_keypoints / merge_cell_keys over torch.distributed -- the rehearsal: backend "gloo" in
    the CPU tests and when several ranks share one GPU (RCCL refuses two ranks on one device).
"""
import ctypes as C
import os

# torch is imported only by the torch.distributed rehearsal functions below: RcclComm and dist_lib() are ctypes over
# liborbfe_dist.so and must stay usable in a process that never loads torch (tests/fake_rccl/world2_worker.py runs the
# library's world > 1 branches against a loopback transport and must not have torch's own librccl in the process)

DIST_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "liborbfe_dist.so")
DIST_ID_BYTES = 128

_DIST_SIGS = {
    "orbfe_dist_unique_id": (C.c_int, [C.c_void_p]),
    "orbfe_dist_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "orbfe_dist_destroy": (None, [C.c_void_p]),
    "orbfe_dist_rank": (C.c_int, [C.c_void_p]),
    "orbfe_dist_world": (C.c_int, [C.c_void_p]),
    "orbfe_dist_last_error": (C.c_char_p, [C.c_void_p]),
    "orbfe_dist_shard_range": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "orbfe_dist_gather_keypoints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "orbfe_dist_exact_offsets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "orbfe_dist_allreduce_max_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "orbfe_dist_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orbfe_dist_ticket": (C.c_int64, [C.c_void_p]),
    "orbfe_dist_wait_ticket": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "orbfe_dist_sync": (C.c_int, [C.c_void_p]),
    "orbfe_dist_host_allreduce": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]),
    "orbfe_dist_barrier": (C.c_int, [C.c_void_p]),
}
DIST_EXPORTS = tuple(_DIST_SIGS)  # every symbol include/orbfe_dist.h declares
_dist_lib = None


def dist_lib():
    """Load liborbfe_dist.so (RCCL + HIP); raises if it has not been built."""
    global _dist_lib
    if _dist_lib is None:
        if not os.path.exists(DIST_LIB_PATH):
            raise RuntimeError("liborbfe_dist.so not found at %s: build it with `make -C jetracer-orbslam2_amd/csrc`"
                               % DIST_LIB_PATH)
        L = C.CDLL(DIST_LIB_PATH)
        for name, (res, args) in _DIST_SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _dist_lib = L
    return _dist_lib


class RcclComm:
    """One rank of the C++/RCCL communicator of include/orbfe_dist.h.

    `exchange_id(id_bytes_or_None) -> bytes` hands rank 0's unique id to every rank (the side
    channel is the application's: bench.py uses torch.distributed's store)."""

    def __init__(self, rank, world, device, exchange_id):
        L = dist_lib()
        ident = None
        if rank == 0:
            buf = (C.c_uint8 * DIST_ID_BYTES)()
            self._check(L.orbfe_dist_unique_id(buf), None)
            ident = bytes(buf)
        ident = exchange_id(ident)
        assert len(ident) == DIST_ID_BYTES
        h = C.c_void_p()
        idbuf = (C.c_uint8 * DIST_ID_BYTES).from_buffer_copy(ident)
        self._check(L.orbfe_dist_create(idbuf, rank, world, device, C.byref(h)), None)
        self.handle, self.rank, self.world = h, rank, world

    @staticmethod
    def _check(code, handle):
        if code != 0:
            raise RuntimeError("orbfe_dist error %d: %s" % (code, dist_lib().orbfe_dist_last_error(handle).decode()))

    def gather_keypoints(self, records, counts, n_frames, cap, all_records, all_counts, root, exact, stream):
        """Pointers are device addresses (ints); all_* may be None off the root."""
        self._check(dist_lib().orbfe_dist_gather_keypoints(self.handle, records, counts, n_frames, cap, all_records,
                                                           all_counts, root, int(bool(exact)), stream), self.handle)

    def exact_offsets(self, all_counts, n_frames, cap, offsets, stream):
        """Root: record index where each (rank, frame) starts in the exact-length layout -> int64[world * n_frames]."""
        self._check(dist_lib().orbfe_dist_exact_offsets(self.handle, all_counts, n_frames, cap, offsets, stream), self.handle)

    def allreduce_max_keys(self, keys, n, stream):
        self._check(dist_lib().orbfe_dist_allreduce_max_keys(self.handle, keys, n, stream), self.handle)

    def wait(self, stream):
        self._check(dist_lib().orbfe_dist_wait(self.handle, stream), self.handle)

    def ticket(self):
        return int(dist_lib().orbfe_dist_ticket(self.handle))

    def wait_ticket(self, ticket, stream):
        self._check(dist_lib().orbfe_dist_wait_ticket(self.handle, ticket, stream), self.handle)

    def sync(self):
        self._check(dist_lib().orbfe_dist_sync(self.handle), self.handle)

    def host_allreduce(self, values, op="max"):
        arr = (C.c_double * len(values))(*values)
        self._check(dist_lib().orbfe_dist_host_allreduce(self.handle, arr, len(values), 0 if op == "max" else 1),
                    self.handle)
        return list(arr)

    def barrier(self):
        self._check(dist_lib().orbfe_dist_barrier(self.handle), self.handle)

    def close(self):
        if self.handle:
            dist_lib().orbfe_dist_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_range(n_total, rank, world):
    """Contiguous block [begin, end) of `n_total` frames owned by `rank` (sizes differ by <= 1)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    base, rem = divmod(n_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class AsyncGather:
    """Handle of an in-flight gather; wait() makes the current stream wait for it."""

    def __init__(self, works, rec_all, cnt_all):
        self.works, self.records, self.counts = works, rec_all, cnt_all

    def wait(self):
        for w in self.works:
            w.wait()
        return self.records, self.counts


def gather_keypoints_async(records, counts, out=None, dst=0, group=None):
    """Like gather_keypoints but returns at once (async_op=True): the collective is ordered
    after the work already queued on the current stream and overlaps what is queued next
    (the following step's kernels).  `out` = (records_all, counts_all) buffers to reuse on dst."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return AsyncGather([], records.unsqueeze(0), counts.unsqueeze(0))
    world = dist.get_world_size(group)
    if dist.get_rank(group) == dst:
        if out is None:
            out = (torch.empty((world,) + tuple(records.shape), dtype=records.dtype, device=records.device),
                   torch.empty((world,) + tuple(counts.shape), dtype=counts.dtype, device=counts.device))
        w1 = dist.gather(counts, list(out[1].unbind(0)), dst=dst, group=group, async_op=True)
        w2 = dist.gather(records, list(out[0].unbind(0)), dst=dst, group=group, async_op=True)
        return AsyncGather([w1, w2], out[0], out[1])
    w1 = dist.gather(counts, None, dst=dst, group=group, async_op=True)
    w2 = dist.gather(records, None, dst=dst, group=group, async_op=True)
    return AsyncGather([w1, w2], None, None)


def gather_keypoints(records, counts, dst=0, group=None):
    """Gather fixed-stride record blocks and counts of every rank on `dst`.

    records: uint8 tensor [frames_local * cap * 52]; counts: int32 tensor [frames_local].
    All ranks must pass equally sized tensors (weak scaling: same frames per rank).
    Returns (records_all [world, ...], counts_all [world, frames_local]) on dst, (None, None)
    elsewhere.  Without an initialised process group (single process) it is the identity.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return records.unsqueeze(0), counts.unsqueeze(0)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == dst:
        rec_all = torch.empty((world,) + tuple(records.shape), dtype=records.dtype, device=records.device)
        cnt_all = torch.empty((world,) + tuple(counts.shape), dtype=counts.dtype, device=counts.device)
        dist.gather(counts, list(cnt_all.unbind(0)), dst=dst, group=group)
        dist.gather(records, list(rec_all.unbind(0)), dst=dst, group=group)
        return rec_all, cnt_all
    dist.gather(counts, None, dst=dst, group=group)
    dist.gather(records, None, dst=dst, group=group)
    return None, None


def merge_cell_keys(keys, group=None):
    """All-reduce(MAX) of per-cell detection keys across ranks (tile-sharded detection of one
    large frame, orbfe_detect_batch_shard).  keys: int32 tensor viewing the uint32 keys (all
    < 2**27, so signed MAX == unsigned MAX).  In place; identity without a process group."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(keys, op=dist.ReduceOp.MAX, group=group)
    return keys
