"""GPU tests of the round-2 entry points: orbfe_match_pairs (pair lists / stereo stride), the
count clamps, the C++/RCCL communicator of include/orbfe_dist.h (world size 1 on the single test
GPU: self send / recv through RCCL), and the capture-safe cell-key shortcut.  Checked against the
CPU oracle or against byte-exact expectations, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from orbfe import synth
from test_gpu_parity import _run_extract, dev, stream

pytestmark = pytest.mark.gpu


def _match_ref(oracle_mod, A, B, mode, window, maxd):
    pa = np.stack([A["x"], A["y"]], 1)
    pb = np.stack([B["x"], B["y"]], 1)
    if mode == 0:
        comp = lambda d: ((d == 1).astype(np.uint32) << np.arange(32, dtype=np.uint32)).sum(1).astype(np.uint32)
        idx, _ = oracle_mod.match_keypoints(pa, comp(A["desc"]), pb, comp(B["desc"]), window, maxd)
        return idx, None
    return oracle_mod.match256(A["desc"], B["desc"], pa, pb, window, maxd)


@pytest.mark.parametrize("mode,window,maxd", [(0, 4, 6), (1, -1, 256), (1, 16, 64)])
@pytest.mark.parametrize("first,stride", [(0, 2), (1, 2), (0, 3), (2, 1)])
def test_match_pairs_strided(gpu, oracle_mod, mode, window, maxd, first, stride):
    """Pair k = frames (first + k * stride, first + k * stride + 1); stereo = (0, 2)."""
    torch, orbfe = gpu
    w, h = 320, 240
    fr = []
    for i in range(3):
        a, b = synth.shifted_pair(w, h, 20 + i, dx=3, dy=0, n_rects=200, min_size=6, max_size=32)
        fr += [a, b]
    fr.append(synth.frame(w, h, 5, "uniform"))
    frames = np.stack(fr)  # 7 frames
    n = len(frames)
    cfg = dict(levels=4, cell=16, min_arc=9)
    ctx, rec, cnt, _ = _run_extract(torch, orbfe, frames, want_soa=False, **cfg)
    n_pairs = (n - 2 - first) // stride + 1
    d_rec, d_cnt = dev(torch, rec.view(np.uint8).reshape(-1)), dev(torch, cnt)
    d_idx = torch.full(((n - 1) * ctx.cap,), -7, dtype=torch.int32, device="cuda")
    d_dist = torch.full(((n - 1) * ctx.cap,), -7, dtype=torch.int32, device="cuda")
    ctx.match_pairs(d_rec.data_ptr(), d_cnt.data_ptr(), n, first, stride, mode, window, maxd, d_idx.data_ptr(),
                    d_dist.data_ptr(), stream(torch))
    idx = d_idx.cpu().numpy().reshape(n - 1, ctx.cap)
    dist = d_dist.cpu().numpy().reshape(n - 1, ctx.cap)
    matched = 0
    for k in range(n_pairs):
        p = first + k * stride
        A, B = rec[p, :cnt[p]], rec[p + 1, :cnt[p + 1]]
        ref_idx, ref_dist = _match_ref(oracle_mod, A, B, mode, window, maxd)
        np.testing.assert_array_equal(idx[k, :cnt[p]], ref_idx, err_msg="pair %d" % k)
        if ref_dist is not None:
            np.testing.assert_array_equal(dist[k, :cnt[p]], ref_dist)
        assert (idx[k, cnt[p]:] == -1).all()
        matched += int((ref_idx >= 0).sum())
    assert (idx[n_pairs:] == -7).all(), "rows beyond the pair list must stay untouched"
    assert matched > 0


@pytest.mark.parametrize("mode,window", [(0, 3), (1, -1), (1, 12)])
def test_counts_beyond_cap_are_clamped(gpu, oracle_mod, mode, window):
    """A stale / corrupt counts buffer must not make the matcher read past a frame's records."""
    torch, orbfe = gpu
    w, h = 160, 120
    frames = np.stack([synth.frame(w, h, i, "uniform") for i in range(3)])
    cfg = dict(levels=2, cell=16, min_arc=9)
    ctx, rec, cnt, _ = _run_extract(torch, orbfe, frames, want_soa=False, **cfg)
    assert (cnt == ctx.cap).all(), "uniform noise fills every cell"
    bad = cnt.copy()
    bad[:] = [ctx.cap + 1000, 2 ** 30, -5]
    d_rec, d_bad = dev(torch, rec.view(np.uint8).reshape(-1)), dev(torch, bad)
    d_idx = torch.full((2 * ctx.cap,), -7, dtype=torch.int32, device="cuda")
    d_dist = torch.full((2 * ctx.cap,), -7, dtype=torch.int32, device="cuda")
    maxd = 4 if mode == 0 else 256
    ctx.match_batch(d_rec.data_ptr(), d_bad.data_ptr(), 3, mode, window, maxd, d_idx.data_ptr(), d_dist.data_ptr(),
                    stream(torch))
    torch.cuda.synchronize()
    idx = d_idx.cpu().numpy().reshape(2, ctx.cap)
    # pair 0: both counts clamp to cap -> identical to the honest run
    ref_idx, _ = _match_ref(oracle_mod, rec[0], rec[1], mode, window, maxd)
    np.testing.assert_array_equal(idx[0], ref_idx)
    # pair 1: the curr frame has a negative count -> no candidates
    assert (idx[1] == -1).all()


def _comm(orbfe_dist_mod):
    return orbfe_dist_mod.RcclComm(0, 1, 0, lambda ident: ident)


@pytest.mark.parametrize("exact", [0, 1])
def test_rccl_gather_world1(gpu, oracle_mod, exact):
    """The C++/RCCL gather with one rank: counts and records must arrive in the root's buffers
    (fixed stride: byte-identical; exact: densely packed in frame order)."""
    torch, orbfe = gpu
    from orbfe import dist as od
    w, h = 320, 240
    frames = np.stack([synth.frame(w, h, 3, "rects", n_rects=150, min_size=6, max_size=30),
                       synth.frame(w, h, 0, "const"), synth.frame(w, h, 4, "uniform"),
                       synth.frame(w, h, 5, "rects", n_rects=40, min_size=6, max_size=60)])
    cfg = dict(levels=4, cell=16, min_arc=9, max_features=120)
    ctx, rec, cnt, _ = _run_extract(torch, orbfe, frames, want_soa=False, **cfg)
    n = len(frames)
    assert cnt[1] == 0 and cnt[2] == 120 and 0 < cnt[0] and 0 < cnt[3]
    d_rec, d_cnt = dev(torch, rec.view(np.uint8).reshape(-1)), dev(torch, cnt)
    all_rec = torch.full((n * ctx.cap * 52,), 0xEE, dtype=torch.uint8, device="cuda")
    all_cnt = torch.full((n,), -9, dtype=torch.int32, device="cuda")
    comm = _comm(od)
    s = stream(torch)
    comm.gather_keypoints(d_rec.data_ptr(), d_cnt.data_ptr(), n, ctx.cap, all_rec.data_ptr(), all_cnt.data_ptr(), 0,
                          exact, s)
    t = comm.ticket()
    assert t == 1
    comm.wait_ticket(t, s)
    torch.cuda.synchronize()
    comm.sync()
    np.testing.assert_array_equal(all_cnt.cpu().numpy(), cnt)
    got = all_rec.cpu().numpy()
    if not exact:
        assert got.tobytes() == rec.tobytes()
    else:
        dense = np.concatenate([rec[f, :cnt[f]] for f in range(n)])
        assert got[:dense.nbytes].tobytes() == dense.tobytes()
        assert (got[dense.nbytes:] == 0xEE).all(), "nothing beyond sum(counts) records may be written"
    # host reductions and the barrier are the identity at world size 1
    assert comm.host_allreduce([1.5, -2.0], "max") == [1.5, -2.0]
    comm.barrier()
    comm.close()


def test_rccl_allreduce_keys_world1_and_c5_flow(gpu, oracle_mod):
    """Tile-sharded detection merged through the C ABI's export -> all-reduce(MAX) -> import (the
    reduction itself is the identity with one rank; the shards are merged by maximum on the device,
    which is what RCCL's ncclMax does across ranks)."""
    torch, orbfe = gpu
    from orbfe import dist as od
    w, h, shards = 640, 480, 4
    img = synth.frame(w, h, 6, "rects", **synth.DENSE)
    cfg = dict(levels=6, cell=16, min_arc=9, max_features=500)
    ctx = orbfe.Context(w, h, max_batch=1, **cfg)
    s = stream(torch)
    d_in = dev(torch, img)
    comm = _comm(od)
    merged = torch.zeros(ctx.K, dtype=torch.int32, device="cuda")
    part = torch.zeros(ctx.K, dtype=torch.int32, device="cuda")
    ctx.build_pyramid(d_in.data_ptr(), w, w * h, 1, s)
    for i in range(shards):
        ctx.detect_batch_shard(1, i, shards, s)
        ctx.export_cell_keys(1, part.data_ptr(), s)
        comm.allreduce_max_keys(part.data_ptr(), ctx.K, s)
        comm.wait(s)
        merged = torch.maximum(merged, part)
    ctx.import_cell_keys(1, merged.data_ptr(), s)
    rec = torch.zeros(ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.describe_batch(1, rec.data_ptr(), cnt.data_ptr(), None, s)
    torch.cuda.synchronize()
    ref = oracle_mod.extract_frame(img, oracle_mod.make_config(w, h, **cfg))
    assert int(cnt.cpu()[0]) == ref["count"] > 100
    assert rec.cpu().numpy().view(orbfe.KEYPOINT_DTYPE)[:ref["count"]].tobytes() == ref["records"].tobytes()
    comm.close()


def test_detect_after_eager_build_inside_a_capture_clears_keys(gpu, oracle_mod):
    """the pyramid kernel clears the cell keys as a by-product and detect_batch skips its
    memset -- valid only when both are issued in the same mode on one stream.  Build eagerly, capture
    ONLY detect + describe, replay twice: the graph must contain its own clear, or the second replay
    would atomicMax onto the first one's keys."""
    torch, orbfe = gpu
    w, h = 320, 240
    imgs = [synth.frame(w, h, 9, "rects", n_rects=200, min_size=6, max_size=32)]
    cfg = dict(levels=4, cell=16, min_arc=9)
    ctx = orbfe.Context(w, h, max_batch=1, **cfg)
    rec = torch.zeros(ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    d_in = [dev(torch, i) for i in imgs]
    with torch.cuda.stream(side):
        s = side.cuda_stream
        ctx.build_pyramid(d_in[0].data_ptr(), w, w * h, 1, s)  # eager
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            ctx.detect_batch(1, side.cuda_stream)
            ctx.describe_batch(1, rec.data_ptr(), cnt.data_ptr(), None, side.cuda_stream)
        g.replay()
        side.synchronize()
        ref0 = oracle_mod.extract_frame(imgs[0], oracle_mod.make_config(w, h, **cfg))
        assert int(cnt.cpu()[0]) == ref0["count"] > 20
        # dirty the cell keys behind the graph's back (every cell claims a score-4000 corner): a graph
        # without its own clear would atomicMax onto them and resurrect 4000-score keypoints
        junk = torch.full((ctx.K,), (4000 << 15) | (7 << 12) | 4095, dtype=torch.int32, device="cuda")
        ctx.import_cell_keys(1, junk.data_ptr(), s)
        side.synchronize()
        g.replay()
        side.synchronize()
        assert int(cnt.cpu()[0]) == ref0["count"]
        assert rec.cpu().numpy().view(orbfe.KEYPOINT_DTYPE)[:ref0["count"]].tobytes() == ref0["records"].tobytes()


def test_calls_leave_the_current_device_alone(gpu):
    torch, orbfe = gpu
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    cur = ctypes.c_int(-1)
    ctx = orbfe.Context(64, 64, max_batch=1)
    buf = torch.zeros(64 * 64, dtype=torch.uint8, device="cuda")
    rec = torch.zeros(ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.extract(buf.data_ptr(), 64, 64 * 64, 1, rec.data_ptr(), cnt.data_ptr(), None, stream(torch))
    assert hip.hipGetDevice(ctypes.byref(cur)) == 0 and cur.value == torch.cuda.current_device()
