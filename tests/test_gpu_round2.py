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


# ------------------------------------------------------------------ a11 / a12 output contract
@pytest.mark.parametrize("n_prev,n_curr,with_points", [(0, 5, True), (7, 0, True), (300, 300, True), (405, 380, False),
                                                        (1000, 900, True), (5000, 4000, False)])
def test_match_compact(gpu, oracle_mod, n_prev, n_curr, with_points):
    """previous_matched_points / current_matched_points / d_pos_frame of kernel_match_keypoints
    (post_processing.cu:176-198), in prev order."""
    torch, orbfe = gpu
    rng = np.random.default_rng(n_prev * 7 + n_curr)
    idx = np.where(rng.random(n_prev) < 0.6, rng.integers(0, max(n_curr, 1), n_prev), -1).astype(np.int32)
    if n_curr == 0:
        idx[:] = -1
    pos_curr = (rng.random((max(n_curr, 1), 2)) * [847, 479]).astype(np.float32)
    pts_prev = rng.normal(size=(max(n_prev, 1), 3)) * 1000
    pts_curr = rng.normal(size=(max(n_curr, 1), 3)) * 1000
    d_idx, d_pos = dev(torch, idx if n_prev else np.zeros(1, np.int32)), dev(torch, pos_curr)
    d_pp, d_pc = dev(torch, pts_prev), dev(torch, pts_curr)
    m = max(n_prev, 1)
    kx = torch.full((m,), 0xEEEE - 65536, dtype=torch.int16, device="cuda")
    ky = torch.full((m,), 0xEEEE - 65536, dtype=torch.int16, device="cuda")
    pm = torch.full((m, 3), -7.0, dtype=torch.float64, device="cuda")
    cm = torch.full((m, 3), -7.0, dtype=torch.float64, device="cuda")
    cnt = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    orbfe.check(orbfe.lib().orbfe_match_compact(d_idx.data_ptr(), n_prev, d_pp.data_ptr() if with_points else None,
                                                d_pc.data_ptr() if with_points else None, d_pos.data_ptr(),
                                                pm.data_ptr() if with_points else None,
                                                cm.data_ptr() if with_points else None, kx.data_ptr(), ky.data_ptr(),
                                                cnt.data_ptr(), stream(torch)))
    rkx, rky, rpm, rcm = oracle_mod.match_compact(idx[:n_prev], pos_curr, pts_prev if with_points else None,
                                                  pts_curr if with_points else None)
    n = int(cnt.cpu()[0])
    assert n == len(rkx) == int((idx[:n_prev] >= 0).sum())
    np.testing.assert_array_equal(kx.cpu().numpy().view(np.uint16)[:n], rkx)
    np.testing.assert_array_equal(ky.cpu().numpy().view(np.uint16)[:n], rky)
    assert (kx.cpu().numpy().view(np.uint16)[n:] == 0xEEEE).all()
    if with_points:
        np.testing.assert_array_equal(pm.cpu().numpy()[:n].view(np.uint64), rpm.view(np.uint64))
        np.testing.assert_array_equal(cm.cpu().numpy()[:n].view(np.uint64), rcm.view(np.uint64))
        assert (pm.cpu().numpy()[n:] == -7.0).all()
    # an unbalanced pointer pair is an argument error
    assert orbfe.lib().orbfe_match_compact(d_idx.data_ptr(), n_prev, d_pp.data_ptr(), None, d_pos.data_ptr(), None, None,
                                           kx.data_ptr(), ky.data_ptr(), cnt.data_ptr(), stream(torch)) == orbfe.ERR_INVALID_ARG


@pytest.mark.parametrize("model", [0, 1, 2, 4])
def test_reproject_points(gpu, oracle_mod, model):
    """kernel_reproject_prev_points (post_processing.cu:72-90): pose * point, then the float projection.  Models 2 and 4
    carry coefficients the projection never reads (post_processing.cu:15-38: only model 1 and 3 have a branch)."""
    torch, orbfe = gpu
    rng = np.random.default_rng(5 + model)
    n = 777
    pts = np.stack([rng.normal(size=n) * 400, rng.normal(size=n) * 300, rng.uniform(300, 5000, n)], 1)
    a = 0.03
    T = np.array([[np.cos(a), 0, np.sin(a), 12.5], [0, 1, 0, -3.25], [-np.sin(a), 0, np.cos(a), 40.0], [0, 0, 0, 1]])
    coeffs = (0.0, 0.0, 0.0, 0.0, 0.0) if model == 0 else (0.11, -0.23, 0.0007, -0.0004, 0.09)
    if model in (2, 4):  # the same pixels as model 0: the coefficients are not read
        plain = oracle_mod.Intrinsics(848, 480, 421.5, 237.25, 615.5, 615.25, 0, (C.c_float * 5)())
    intr = orbfe.Intrinsics(848, 480, 421.5, 237.25, 615.5, 615.25, model, (C.c_float * 5)(*coeffs))
    ointr = oracle_mod.Intrinsics(848, 480, 421.5, 237.25, 615.5, 615.25, model, (C.c_float * 5)(*coeffs))
    d_pts = dev(torch, pts)
    out = torch.full((n, 2), -1.0, dtype=torch.float32, device="cuda")
    Tc = (C.c_double * 16)(*np.ascontiguousarray(T.T).reshape(-1))  # column-major, as Eigen stores it
    orbfe.check(orbfe.lib().orbfe_reproject_points(out.data_ptr(), d_pts.data_ptr(), n, Tc, C.byref(intr), stream(torch)))
    ref = oracle_mod.reproject_points(pts, T, ointr)
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    assert np.isfinite(ref).all() and ref[:, 0].std() > 10
    if model in (2, 4):
        np.testing.assert_array_equal(ref.view(np.uint32), oracle_mod.reproject_points(pts, T, plain).view(np.uint32))
    # (model 3, f-theta: tests/test_gpu_round5.py)


def test_cpp_match_port(gpu, oracle_mod, tmp_path):
    """examples/match_port.cpp: two frames through the reference's call sequence incl. align_depth_to_other on its own
    stream, keypoint_pixel_to_point on the aligned depth and match_keypoints(current, previous, 2, 4, T, ...) with
    slam_frame_t (buildStream.cpp:376-556) via compat/jetracer_compat.hpp; outputs must equal the oracle's."""
    import os
    import subprocess
    from test_align_oracle import extr, intr
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "match_port")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    torch, orbfe = gpu
    w, h = 848, 480
    a, b = synth.shifted_pair(w, h, 77, dx=1, dy=0, **synth.DENSE)
    rgbs = [np.stack([g, g, g], -1) for g in (a, b)]
    depth = synth.depth_frame(w, h, 3)
    d, o, e, scale = synth.rig("d435", w, h)
    paths = [str(tmp_path / n) for n in ("a.bin", "b.bin", "depth.bin", "rig.bin", "out.bin")]
    rgbs[0].tofile(paths[0]); rgbs[1].tofile(paths[1]); depth.tofile(paths[2])
    with open(paths[3], "wb") as f:  # orbfe_intrinsics x 2, orbfe_extrinsics, float: as the C structs lie in memory
        f.write(bytes(intr(orbfe, d)) + bytes(intr(orbfe, o)) + bytes(extr(orbfe, e)) + np.float32(scale).tobytes())
    subprocess.check_call([exe, str(w), str(h), *paths])
    raw = np.fromfile(paths[4], np.uint8)
    nprev, ncurr, n = raw[:12].view(np.int32)
    kx = raw[12:12 + 2 * n].view(np.uint16)
    ky = raw[12 + 2 * n:12 + 4 * n].view(np.uint16)
    pm = raw[12 + 4 * n:12 + 4 * n + 24 * n].view(np.float64).reshape(n, 3)
    cm = raw[12 + 28 * n:12 + 52 * n].view(np.float64).reshape(n, 3)
    # the oracle, call by call
    oi = intr(oracle_mod, o)
    aligned, _ = oracle_mod.align_depth_to_other(depth, scale, w, h, intr(oracle_mod, d), oi, extr(oracle_mod, e))
    assert (aligned > 1).mean() > 0.5
    frames = []
    for rgb in rgbs:
        gray = oracle_mod.rgb_to_grayscale(rgb)
        ref = oracle_mod.extract_frame(gray, oracle_mod.make_config(w, h, levels=1))
        pos, pts, d32, _ = oracle_mod.keypoint_pixel_to_point(aligned, oi, ref["pos"], ref["score"], ref["desc32"], 0)
        frames.append((pos, pts, d32))
    (ppos, ppts, pd32), (cpos, cpts, cd32) = frames
    assert (nprev, ncurr) == (len(ppos), len(cpos))
    pos_tmp = oracle_mod.reproject_points(ppts, np.eye(4), oi)
    ridx, rn = oracle_mod.match_keypoints(pos_tmp, pd32, cpos, cd32, 2, 4)
    rkx, rky, rpm, rcm = oracle_mod.match_compact(ridx, cpos, ppts, cpts)
    assert n == rn == len(rkx) and n > 20
    np.testing.assert_array_equal(kx, rkx)
    np.testing.assert_array_equal(ky, rky)
    np.testing.assert_array_equal(pm.view(np.uint64), rpm.view(np.uint64))
    np.testing.assert_array_equal(cm.view(np.uint64), rcm.view(np.uint64))


# ------------------------------------------------------------------ stage API detect: fused path, float thresholds
def _detect_case(torch, orbfe, oracle_mod, w, h, levels, arc, thr, build_lut_on_device, pitch_extra, with_resp=True, seed=9,
                 lut_np=None, make_d_lut=None):
    """lut_np: the table the oracle uses (default: the arc table); make_d_lut(lut_np) -> device tensor holding it,
    however it got there (default: orbfe_fast_calculate_lut on the device, or a plain upload)."""
    from test_gpu_parity import bits, pitched
    lut = oracle_mod.fast_lut(arc) if lut_np is None else lut_np
    imgs = [oracle_mod.gaussian_blur_3x3(synth.frame(w, h, seed, "rects", **synth.DENSE))]
    for _ in range(1, levels):
        imgs.append(oracle_mod.halfsample(imgs[-1]))
    resps = [oracle_mod.fast_response(i, lut, thr) if min(i.shape) > 0 else np.zeros(i.shape, np.float32) for i in imgs]
    rpos, rscore, rlevel = oracle_mod.grid_nms(resps, 32)
    k = oracle_mod.num_cells(w, h)
    pitch = [((w >> l) + 3) // 4 * 4 + pitch_extra for l in range(levels)]  # a multiple of 4 iff pitch_extra is
    d_img = [pitched(torch, imgs[l], pitch[l]) for l in range(levels)]
    d_res = [torch.full((max(h >> l, 1), (w >> l) + 5), -1.0, dtype=torch.float32, device="cuda") for l in range(levels)]
    lv = orbfe.make_levels([(d_img[l].data_ptr(), w >> l, h >> l, pitch[l]) for l in range(levels)],
                           [(d_res[l].data_ptr() if with_resp else 0, w >> l, h >> l, ((w >> l) + 5) * 4) for l in range(levels)])
    if make_d_lut is not None:
        d_lut = make_d_lut(lut)
    elif build_lut_on_device:
        d_lut = torch.zeros(65536, dtype=torch.uint8, device="cuda")
        orbfe.check(orbfe.lib().orbfe_fast_calculate_lut(d_lut.data_ptr(), arc, stream(torch)))
    else:
        d_lut = dev(torch, lut)
    grid = torch.full((4 * k,), -3.0, dtype=torch.float32, device="cuda")
    b = grid.data_ptr()
    orbfe.check(orbfe.lib().orbfe_detect(lv, levels, d_lut.data_ptr(), thr, b, b + 8 * k, b + 12 * k, stream(torch)))
    g = grid.cpu().numpy()
    np.testing.assert_array_equal(bits(g[2 * k:3 * k]), bits(rscore))
    np.testing.assert_array_equal(bits(g[:2 * k].reshape(k, 2)), bits(rpos))
    np.testing.assert_array_equal(g[3 * k:].view(np.int32), rlevel)
    if with_resp:
        for l in range(levels):
            got = d_res[l].cpu().numpy()
            np.testing.assert_array_equal(bits(got[:h >> l, :w >> l]), bits(resps[l]), err_msg="response level %d" % l)
            assert (got[:, w >> l:] == -1.0).all(), "response written outside the level width"
    return int((rscore > 0).sum())


@pytest.mark.parametrize("w,h,levels,arc,thr,extra", [
    (640, 480, 1, 12, 13.0, 0),    # the reference's live configuration (PYRAMID_LEVELS 1)
    (640, 480, 6, 12, 13.0, 0),
    (848, 480, 6, 9, 13.0, 4),     # odd level widths from level 4 on, padded pitch
    (100, 70, 5, 10, 20.0, 8),     # levels 4 is 6x4: no tiles, response zeroed by memset
    (1280, 720, 4, 11, 5.0, 0),
])
def test_detect_stage_fused_path(gpu, oracle_mod, w, h, levels, arc, thr, extra):
    """LUT built by orbfe_fast_calculate_lut + integer threshold + dword-aligned levels: orbfe_detect runs the
    fused tile kernel (one launch) and must leave the same feature grid AND response maps as the oracle."""
    torch, orbfe = gpu
    n = _detect_case(torch, orbfe, oracle_mod, w, h, levels, arc, thr, True, extra)
    assert n > 10
    _detect_case(torch, orbfe, oracle_mod, w, h, levels, arc, thr, True, extra, with_resp=False)


@pytest.mark.parametrize("thr,on_device,extra", [(7.5, True, 0), (13.0, False, 0), (13.0, True, 3), (0.5, True, 0)])
def test_detect_stage_float_threshold_and_unaligned(gpu, oracle_mod, thr, on_device, extra):
    """Non-integer thresholds (responses x.5), foreign LUT buffers and odd pitches take the unfused path;
    grid_nms must then compare float responses (it truncated them)."""
    torch, orbfe = gpu
    n = _detect_case(torch, orbfe, oracle_mod, 640, 480, 5, 12, thr, on_device, extra)
    assert n > 10


# ------------------------------------------------------------------ windowed 256-bit matching through the cell index
def _window_fuzz_cases():
    """ORBFE_FUZZ_WINDOW=N appends N seeded random (geometry, window, count, order) cases (one-off soak runs)."""
    import os
    n = int(os.environ.get("ORBFE_FUZZ_WINDOW", "0"))
    rng = np.random.default_rng(int(os.environ.get("ORBFE_FUZZ_SEED", "7")))
    out = []
    while len(out) < n:
        cell = int(rng.choice([8, 16, 32]))
        w, h = int(rng.integers(12, 60)) * 16, int(rng.integers(10, 40)) * 16
        k = ((w + cell - 1) // cell) * ((h + cell - 1) // cell)
        if k < 80:
            continue
        out.append((w, h, cell, int(rng.integers(0, 4 * cell)), int(rng.integers(60, min(k, 3000))), bool(rng.integers(0, 2)), 0))
    return out


@pytest.mark.parametrize("w,h,cell,window,n_rec,ordered,max_features", _window_fuzz_cases() + [
    (640, 480, 8, 16, 1500, False, 0), (640, 480, 8, 3, 2000, False, 0), (320, 240, 16, 40, 300, False, 0),
    (848, 480, 32, 7, 405, False, 0), (640, 480, 8, 0, 1000, False, 0),
    # records in cell order, as the extractor writes them: a workgroup's windows span a few cell rows and the
    # walk runs on the piece of the sorted list staged in LDS (unordered records overflow it: global path)
    (640, 480, 8, 16, 1500, True, 0), (848, 480, 8, 16, 2000, True, 0), (640, 480, 8, 40, 900, True, 0),
    # 129 600 cells: the bucket counters do not fit LDS (match_bucket_kernel<false>)
    (3840, 2160, 8, 16, 3000, True, 4000), (3840, 2160, 8, 24, 2500, False, 4000)])
def test_windowed_match_on_arbitrary_records(gpu, oracle_mod, w, h, cell, window, n_rec, ordered, max_features):
    """Records as a caller may hand them in: several per cell, non-integer positions, positions outside the
    image (bucket clamping), ragged counts.  The cell-indexed matcher must give the brute-force answer."""
    torch, orbfe = gpu
    ctx = orbfe.Context(w, h, cell=cell, min_arc=9, max_batch=4, max_features=max_features)
    cap = ctx.cap
    assert n_rec <= cap
    rng = np.random.default_rng(cell * 1000 + window)
    n = 4
    rec = np.zeros((n, cap), dtype=orbfe.KEYPOINT_DTYPE)
    cnt = np.array([n_rec, n_rec - 7, cap if cap < 3000 else n_rec, 1], np.int32)
    base_desc = rng.integers(0, 256, (64, 32)).astype(np.uint8)  # few distinct descriptors -> many distance ties
    for f in range(n):
        m = cnt[f]
        rec["x"][f, :m] = rng.uniform(-25, w + 25, m).astype(np.float32)
        rec["y"][f, :m] = rng.uniform(-25, h + 25, m).astype(np.float32)
        rec["x"][f, :m:3] = np.round(rec["x"][f, :m:3])
        d = base_desc[rng.integers(0, 64, m)].copy()
        flip = rng.random((m, 32)) < 0.05
        d[flip] ^= rng.integers(1, 256, int(flip.sum())).astype(np.uint8)
        rec["desc"][f, :m] = d
    rec["x"][1, :50] = rec["x"][0, :50]  # exact coincidences and window-edge cases
    rec["y"][1, :50] = rec["y"][0, :50] + window
    if ordered:
        for f in range(n):
            m = cnt[f]
            cx = np.clip(np.floor(rec["x"][f, :m] / cell), 0, (w + cell - 1) // cell - 1)
            cy = np.clip(np.floor(rec["y"][f, :m] / cell), 0, (h + cell - 1) // cell - 1)
            rec[f, :m] = rec[f, :m][np.lexsort((cx, cy))]
    d_rec, d_cnt = dev(torch, rec.view(np.uint8).reshape(-1)), dev(torch, cnt)
    d_idx = torch.full(((n - 1) * cap,), -7, dtype=torch.int32, device="cuda")
    d_dist = torch.full(((n - 1) * cap,), -7, dtype=torch.int32, device="cuda")
    ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), n, 1, window, 200, d_idx.data_ptr(), d_dist.data_ptr(), stream(torch))
    idx = d_idx.cpu().numpy().reshape(n - 1, cap)
    dist = d_dist.cpu().numpy().reshape(n - 1, cap)
    hits = 0
    for p in range(n - 1):
        A, B = rec[p, :cnt[p]], rec[p + 1, :cnt[p + 1]]
        ref_idx, ref_dist = _match_ref(oracle_mod, A, B, 1, window, 200)
        np.testing.assert_array_equal(idx[p, :cnt[p]], ref_idx, err_msg="pair %d" % p)
        np.testing.assert_array_equal(dist[p, :cnt[p]], ref_dist)
        assert (idx[p, cnt[p]:] == -1).all()
        hits += int((ref_idx >= 0).sum())
    assert hits > 20


# ------------------------------------------------------------------ BASELINE configs at their own sizes, vs the oracle
@pytest.mark.parametrize("w,h,cfg", [
    (1280, 720, dict(levels=8, cell=8, min_arc=9, max_features=2000)),    # C4's per-frame configuration
    (848, 480, dict(levels=8, cell=8, min_arc=9, max_features=2000)),     # C3's
    (640, 480, dict(levels=1, cell=16, min_arc=12, max_features=1000)),   # C1 "1000-feature" variant: cell 16, top-1000
    (3840, 2160, dict(levels=12, cell=16, min_arc=9, max_features=8000)), # C5's
    (3840, 2160, dict(levels=3, cell=8, min_arc=10, max_features=5000)),  # 129 600 cells: select's chunk loop past its register chunks
])
def test_ext_regime_at_baseline_sizes(gpu, oracle_mod, w, h, cfg):
    from test_gpu_parity import _check_extract
    torch, orbfe = gpu
    kw = dict(n_rects=800 * (w * h) // (640 * 480), min_size=6, max_size=32)
    frames = np.stack([synth.frame(w, h, 41, "rects", **kw), synth.frame(w, h, 42, "rects", n_rects=96)])
    ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, want_soa=(w < 3000), **cfg)
    total = _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg)
    assert cnt.max() <= cfg["max_features"]
    if cfg["min_arc"] == 9:
        assert cnt[0] == cfg["max_features"], "the dense scene fills the feature budget"
    assert total > 250


def _c5_rank(rank, world, port, out_path):
    """One rank of the C5 rehearsal: every rank on cuda:0 (RCCL refuses two ranks on one device, so the
    collective is gloo's), detection tiles sharded by rank, keys merged through orbfe.dist.merge_cell_keys."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "oracle"), os.path.join(root, "jetracer-orbslam2_amd")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import orbfe
    from orbfe import synth as sy
    from orbfe.dist import merge_cell_keys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h = 1280, 720
    cfg = dict(levels=8, cell=16, min_arc=9, max_features=1500)
    img = sy.frame(w, h, 12, "rects", n_rects=2400, min_size=6, max_size=32)
    ctx = orbfe.Context(w, h, max_batch=1, **cfg)
    s = torch.cuda.current_stream().cuda_stream
    d_in = torch.from_numpy(img).cuda()
    ctx.build_pyramid(d_in.data_ptr(), w, w * h, 1, s)
    ctx.detect_batch_shard(1, rank, world, s)
    keys = torch.zeros(ctx.K, dtype=torch.int32, device="cuda")
    ctx.export_cell_keys(1, keys.data_ptr(), s)
    torch.cuda.synchronize()
    mine = int((keys > 0).sum())
    hk = keys.cpu()
    merge_cell_keys(hk)  # all_reduce(MAX) over the ranks
    keys.copy_(hk)
    ctx.import_cell_keys(1, keys.data_ptr(), s)
    rec = torch.zeros(ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.describe_batch(1, rec.data_ptr(), cnt.data_ptr(), None, s)
    torch.cuda.synchronize()
    np.savez(out_path % rank, rec=rec.cpu().numpy(), cnt=cnt.cpu().numpy(), mine=mine)
    dist.barrier()
    dist.destroy_process_group()


def test_c5_two_rank_rehearsal_through_merge_cell_keys(gpu, oracle_mod, tmp_path):
    """The C5 flow with 2 real ranks (processes) sharing the test GPU: shard -> export -> merge_cell_keys
    (gloo all-reduce MAX) -> import -> describe; both ranks must end with the oracle's records."""
    import socket
    import torch.multiprocessing as mp
    torch, orbfe = gpu
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_c5_rank, args=(2, port, out), nprocs=2, join=True)
    w, h = 1280, 720
    cfg = dict(levels=8, cell=16, min_arc=9, max_features=1500)
    img = synth.frame(w, h, 12, "rects", n_rects=2400, min_size=6, max_size=32)
    ref = oracle_mod.extract_frame(img, oracle_mod.make_config(w, h, **cfg))
    partial = []
    for r in range(2):
        got = np.load(out % r)
        n = int(got["cnt"][0])
        assert n == ref["count"] == 1500
        assert got["rec"].view(orbfe.KEYPOINT_DTYPE)[:n].tobytes() == ref["records"].tobytes()
        partial.append(int(got["mine"]))
    assert all(0 < p for p in partial), "each rank detected a part of the cells"


# ------------------------------------------------------------------ both describe kernels, forced
@pytest.mark.parametrize("which", ["patch", "tile"])
@pytest.mark.parametrize("cfg", [dict(levels=6), dict(levels=8, cell=8, min_arc=9, max_features=2000),
                                 dict(levels=4, cell=64, min_arc=12), dict(levels=5, cell=16, min_arc=9, angle_in_radians=1)])
def test_both_describe_kernels_match_the_oracle(gpu, oracle_mod, monkeypatch, which, cfg):
    """The library picks the per-keypoint-patch or the tile describe kernel by keypoint density and call size;
    ORBFE_DESCRIBE forces one (read when the context is created).  Both must equal the oracle in every regime."""
    from test_gpu_parity import _check_extract, _mixed_frames
    torch, orbfe = gpu
    monkeypatch.setenv("ORBFE_DESCRIBE", which)
    frames = _mixed_frames(640, 480)
    ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, **cfg)
    total = _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg)
    assert total > 50


def test_reference_regime_at_bench_size_uses_the_patch_kernel_and_matches(gpu, oracle_mod):
    """256 frames in the reference regime: large sparse calls take the patch kernel (n_frames * cap >= 32768)."""
    torch, orbfe = gpu
    w, h, n = 640, 480, 256
    base = synth.frames(w, h, 4, 300, "rects", **synth.DENSE)
    frames = base[np.arange(n) % 4]
    cfg = dict(levels=6)
    ctx, rec, cnt, _ = _run_extract(torch, orbfe, frames, want_soa=False, **cfg)
    ocfg = oracle_mod.make_config(w, h, levels=6)
    for f in (0, 1, 2, 3, 255):
        ref = oracle_mod.extract_frame(frames[f], ocfg)
        assert cnt[f] == ref["count"] > 100
        assert rec[f, :cnt[f]].tobytes() == ref["records"].tobytes()


@pytest.mark.parametrize("exact", [False, True])
def test_cpp_multi_gpu_port(gpu, oracle_mod, tmp_path, exact):
    """examples/multi_gpu_port.cpp: C++ host, one thread + context + RCCL rank per GPU (the reference's
    thread-per-stream model, SlamGpuPipeline.cpp:43-50), frames sharded, records gathered on rank 0 -- run with
    the one GPU of the test box; rank 0's gathered records must be the oracle's, frame by frame."""
    import os
    import subprocess
    torch, orbfe = gpu
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "multi_gpu_port")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    w, h, n = 640, 480, 6
    frames = np.stack([synth.frame(w, h, 60 + i, "rects", n_rects=96 if i % 2 else 800, min_size=6, max_size=None if i % 2 else 32)
                       for i in range(n)])
    fin, fout = str(tmp_path / "frames.bin"), str(tmp_path / "out.bin")
    frames.tofile(fin)
    subprocess.check_call([exe, "1", str(w), str(h), str(n), fin, fout] + (["exact"] if exact else []))
    raw = np.fromfile(fout, np.uint8)
    nf, cap = raw[:8].view(np.int32)
    assert nf == n and cap == 2000
    counts = raw[8:8 + 4 * n].view(np.int32)
    rec = raw[8 + 4 * n:].view(orbfe.KEYPOINT_DTYPE).reshape(n, cap)
    ocfg = oracle_mod.make_config(w, h, levels=8, cell=8, min_arc=9, max_features=2000)
    for f in range(n):
        ref = oracle_mod.extract_frame(frames[f], ocfg)
        assert counts[f] == ref["count"]
        assert rec[f, :counts[f]].tobytes() == ref["records"].tobytes()
    assert counts.min() < 1500 < counts.max(), "ragged counts: the exact-length form ships fewer bytes"


@pytest.mark.parametrize("w,h,cell,window,maxham,n_rec", [(640, 480, 32, 2, 4, 300), (640, 480, 8, 5, 3, 1999), (320, 240, 16, 12, 33, 290),
                                                         (848, 480, 32, 0, 1, 405), (640, 480, 8, 40, 2, 777)])
def test_reference_matcher_through_the_cell_index(gpu, oracle_mod, w, h, cell, window, maxham, n_rec):
    """Reference semantics (first strictly smaller distance in the rotated tile-of-32 visiting order, partial-tile
    skip) computed as an order-free minimum of (distance, visiting rank) over the window's cells: must equal the
    oracle's literal restatement of kernel_match_keypoints on records with MANY equal distances, several
    keypoints per cell, positions outside the image and ragged counts (partial last tiles)."""
    torch, orbfe = gpu
    ctx = orbfe.Context(w, h, cell=cell, min_arc=9, max_batch=4, max_features=0)
    cap = ctx.cap
    assert n_rec <= cap
    rng = np.random.default_rng(cell * 100 + window)
    n = 4
    rec = np.zeros((n, cap), dtype=orbfe.KEYPOINT_DTYPE)
    cnt = np.array([n_rec, n_rec - 13, min(cap, n_rec + 1), 31], np.int32)
    centres = rng.uniform(0, [w, h], size=(40, 2))  # shared by the frames: clustered positions, many candidates per window
    base = rng.random((6, 32)) < 0.3
    for f in range(n):
        m = cnt[f]
        pos = centres[rng.integers(0, 40, m)] + rng.normal(size=(m, 2)) * max(window, 1) * 0.8
        rec["x"][f, :m] = np.round(pos[:, 0]).astype(np.float32)
        rec["y"][f, :m] = np.round(pos[:, 1]).astype(np.float32)
        # the 32-bit compression keeps "byte == 1" (orb.cu:156): six base words with one or two flips each, so
        # that distances are small and ties are everywhere
        ones = base[rng.integers(0, 6, m)] ^ (rng.random((m, 32)) < 0.03)
        d = rng.integers(2, 256, (m, 32)).astype(np.uint8)
        d[ones] = 1
        rec["desc"][f, :m] = d
    d_rec, d_cnt = dev(torch, rec.view(np.uint8).reshape(-1)), dev(torch, cnt)
    d_idx = torch.full(((n - 1) * cap,), -7, dtype=torch.int32, device="cuda")
    ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), n, 0, window, maxham, d_idx.data_ptr(), None, stream(torch))
    idx = d_idx.cpu().numpy().reshape(n - 1, cap)
    hits = 0
    for p in range(n - 1):
        A, B = rec[p, :cnt[p]], rec[p + 1, :cnt[p + 1]]
        ref_idx, _ = _match_ref(oracle_mod, A, B, 0, window, maxham)
        np.testing.assert_array_equal(idx[p, :cnt[p]], ref_idx, err_msg="pair %d" % p)
        assert (idx[p, cnt[p]:] == -1).all()
        hits += int((ref_idx >= 0).sum())
    assert hits > 20
