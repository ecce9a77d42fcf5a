"""Multi-rank path on CPU: frame sharding + the keypoint gather over torch.distributed with
the gloo backend, world_size 2 (the GPU run uses the same code with backend nccl = RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from orbfe.dist import shard_range
    for n in (0, 1, 7, 64, 65, 255):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_path):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "jetracer-orbslam2_amd")):
        sys.path.insert(0, p)
    import oracle  # the CPU oracle stands in for the HIP extractor in this CPU-only test
    from orbfe import synth
    from orbfe.dist import gather_keypoints, gather_keypoints_async, merge_cell_keys, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, cap = 160, 120, 20
    cfg = oracle.make_config(w, h, levels=3)
    begin, end = shard_range(n_total, rank, world)
    rec = np.zeros((end - begin, cap), dtype=oracle.KEYPOINT_DTYPE)
    cnt = np.zeros(end - begin, dtype=np.int32)
    for i, f in enumerate(range(begin, end)):
        r = oracle.extract_frame(synth.frame(w, h, f, "rects", n_rects=40, min_size=6, max_size=30), cfg)["records"]
        cnt[i] = len(r)
        rec[i, :len(r)] = r
    rec_t = torch.from_numpy(rec.view(np.uint8).reshape(-1))
    cnt_t = torch.from_numpy(cnt)
    all_rec, all_cnt = gather_keypoints(rec_t, cnt_t, dst=0)
    # the asynchronous form bench.py overlaps with the next step must deliver the same bytes
    a_rec, a_cnt = gather_keypoints_async(rec_t, cnt_t, dst=0).wait()
    if rank == 0:
        assert torch.equal(a_rec, all_rec) and torch.equal(a_cnt, all_cnt)
    else:
        assert a_rec is None and a_cnt is None
    if rank == 0:
        np.savez(out_path, rec=all_rec.numpy(), cnt=all_cnt.numpy())
    else:
        assert all_rec is None and all_cnt is None
    # tile-sharded detection: partial per-cell keys merge by all-reduce(MAX)
    g = torch.Generator().manual_seed(7)
    full = torch.randint(0, 2 ** 27, (world, 500), generator=g, dtype=torch.int32)
    mine = full[rank].clone()
    mine[torch.arange(500) % world != rank] //= 3  # each rank holds the winner of "its" cells
    want = torch.stack([full[r].clone() for r in range(world)])
    for r in range(world):
        want[r][torch.arange(500) % world != r] //= 3
    merged = merge_cell_keys(mine)
    assert torch.equal(merged, want.max(0).values)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_keypoints_gloo_world2(tmp_path, oracle_mod):
    from orbfe import synth
    world, n_total = 2, 6
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(world, _free_port(), n_total, out), nprocs=world, join=True)
    got = np.load(out)
    w, h, cap = 160, 120, 20
    cfg = oracle_mod.make_config(w, h, levels=3)
    rec = got["rec"].reshape(world, n_total // world, cap * 52).view(oracle_mod.KEYPOINT_DTYPE)
    rec = rec.reshape(n_total, cap)
    cnt = got["cnt"].reshape(n_total)
    total = 0
    for f in range(n_total):  # rank-major order == frame order because shards are contiguous
        ref = oracle_mod.extract_frame(synth.frame(w, h, f, "rects", n_rects=40, min_size=6, max_size=30), cfg)
        assert cnt[f] == ref["count"]
        assert rec[f, :cnt[f]].tobytes() == ref["records"].tobytes()
        total += cnt[f]
    assert total > 10


def test_gather_single_process_is_identity():
    from orbfe.dist import gather_keypoints
    r, c = torch.arange(8, dtype=torch.uint8), torch.tensor([3], dtype=torch.int32)
    rr, cc = gather_keypoints(r, c)
    assert rr.shape == (1, 8) and cc.shape == (1, 1) and (rr[0] == r).all()
