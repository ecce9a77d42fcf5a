"""Frame sharding across ranks and the keypoint gather.

Frames are independent units, so a batch is split into contiguous blocks, one per rank (one
process per GPU); images never cross GPUs.  The only exchange step is a gather of the
fixed-stride keypoint records and per-frame counts to rank 0 -- torch.distributed backend
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  Matching pairs (t-1, t) stay on
one rank because shards are cut on frame boundaries and each rank matches inside its block.
"""
import torch
import torch.distributed as dist


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
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(keys, op=dist.ReduceOp.MAX, group=group)
    return keys
