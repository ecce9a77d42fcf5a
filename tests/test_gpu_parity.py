"""GPU parity tests: every HIP entry point of include/orbfe.h against the CPU oracle on the
same seeded inputs, called through the C ABI (ctypes).  Bit-exact: integer / byte / index
outputs must be equal; float outputs (scores, positions, angles) are compared by their bit
patterns (tolerance 0), because the build owns the deterministic math on both sides
(include/orbfe_math.h).  The oracle itself is unpinned by the reference (it has no tests);
see oracle/orbfe_oracle.h.
"""
import ctypes as C

import numpy as np
import pytest

from orbfe import synth

pytestmark = pytest.mark.gpu


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def stream(torch):
    return torch.cuda.current_stream().cuda_stream


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def pitched(torch, img, pitch):
    h, w = img.shape
    buf = np.full((h, pitch), 0xA5, np.uint8)
    buf[:, :w] = img
    return dev(torch, buf)


FRAMES = {
    "rects": lambda w, h: synth.frame(w, h, 1, "rects"),
    "dense": lambda w, h: synth.frame(w, h, 2, "rects", **synth.DENSE),
    "uniform": lambda w, h: synth.frame(w, h, 3, "uniform"),
    "const": lambda w, h: synth.frame(w, h, 0, "const"),
    "checker": lambda w, h: synth.frame(w, h, 0, "checker"),
}


# ------------------------------------------------------------------ a2 blur
@pytest.mark.parametrize("w,h,extra", [(640, 480, 0), (848, 480, 0), (37, 19, 5), (33, 5, 3),
                                        (64, 3, 0), (8, 8, 1), (95, 40, 0), (1280, 720, 0)])
def test_blur_stage(gpu, oracle_mod, w, h, extra):
    torch, orbfe = gpu
    img = synth.frame(w, h, 7, "uniform")
    src = pitched(torch, img, w + extra)
    dst = torch.full((h, w + extra + 2), 0x5A, dtype=torch.uint8, device="cuda")
    orbfe.check(orbfe.lib().orbfe_gaussian_blur_3x3(dst.data_ptr(), w + extra + 2, src.data_ptr(),
                                                    w + extra, w, h, stream(torch)))
    got = dst.cpu().numpy()
    np.testing.assert_array_equal(got[:, :w], oracle_mod.gaussian_blur_3x3(img))
    assert (got[:, w:] == 0x5A).all(), "blur wrote outside the image width"


# ------------------------------------------------------------------ a3 pyramid
@pytest.mark.parametrize("w,h,levels", [(640, 480, 8), (848, 480, 8), (100, 70, 5), (9, 9, 3)])
def test_pyramid_stage(gpu, oracle_mod, w, h, levels):
    torch, orbfe = gpu
    img0 = synth.frame(w, h, 11, "uniform")
    ref = [img0]
    for _ in range(1, levels):
        ref.append(oracle_mod.halfsample(ref[-1]))
    bufs, descs = [], []
    for l in range(levels):
        lw, lh = w >> l, h >> l
        pitch = max(lw, 1) + 3
        t = torch.full((max(lh, 1), pitch), 0x77, dtype=torch.uint8, device="cuda")
        if l == 0:
            t[:, :lw] = dev(torch, img0)
        bufs.append(t)
        descs.append((t.data_ptr(), lw, lh, pitch))
    lv = orbfe.make_levels(descs)
    orbfe.check(orbfe.lib().orbfe_pyramid_create_levels(lv, levels, stream(torch)))
    for l in range(1, levels):
        lw, lh = w >> l, h >> l
        got = bufs[l].cpu().numpy()
        np.testing.assert_array_equal(got[:lh, :lw], ref[l])
        assert (got[:, lw:] == 0x77).all()


# ------------------------------------------------------------------ a4 LUT
@pytest.mark.parametrize("arc", [1, 9, 10, 11, 12, 16])
def test_lut(gpu, oracle_mod, arc):
    torch, orbfe = gpu
    lut = torch.zeros(65536, dtype=torch.uint8, device="cuda")
    orbfe.check(orbfe.lib().orbfe_fast_calculate_lut(lut.data_ptr(), arc, stream(torch)))
    np.testing.assert_array_equal(lut.cpu().numpy(), oracle_mod.fast_lut(arc))


# ------------------------------------------------------------------ a5 response
@pytest.mark.parametrize("kind", ["rects", "dense", "uniform", "const"])
@pytest.mark.parametrize("w,h,arc,thr", [(640, 480, 12, 13.0), (212, 120, 9, 13.0), (53, 30, 12, 7.5)])
def test_fast_response_stage(gpu, oracle_mod, kind, w, h, arc, thr):
    torch, orbfe = gpu
    img = oracle_mod.gaussian_blur_3x3(FRAMES[kind](w, h))
    lut_np = oracle_mod.fast_lut(arc)
    ref = oracle_mod.fast_response(img, lut_np, thr)
    d_img, d_lut = pitched(torch, img, w + 4), dev(torch, lut_np)
    resp = torch.full((h, w + 8), -3.0, dtype=torch.float32, device="cuda")
    orbfe.check(orbfe.lib().orbfe_fast_calc_corner_response(
        w, h, w + 4, d_img.data_ptr(), 3, 3, d_lut.data_ptr(), thr, arc,
        orbfe.SUM_OF_ABS_DIFF_ON_ARC, w + 8, resp.data_ptr(), stream(torch)))
    got = resp.cpu().numpy()
    np.testing.assert_array_equal(bits(got[:, :w]), bits(ref))
    if kind in ("dense", "uniform"):
        assert (ref > 0).sum() > 10


def _response_pyramid(oracle_mod, img0, levels, arc=12, thr=13.0):
    lut = oracle_mod.fast_lut(arc)
    imgs = [oracle_mod.gaussian_blur_3x3(img0)]
    for _ in range(1, levels):
        imgs.append(oracle_mod.halfsample(imgs[-1]))
    return imgs, [oracle_mod.fast_response(i, lut, thr) if min(i.shape) > 0 else
                  np.zeros(i.shape, np.float32) for i in imgs], lut


def _tie_responses(w, h, levels, seed):
    """Response maps with very many equal maxima to exercise the tie order (Q6)."""
    rng = np.random.default_rng(seed)
    out = []
    for l in range(levels):
        lw, lh = w >> l, h >> l
        r = np.zeros((lh, lw), np.float32)
        if lw > 6 and lh > 6:
            m = rng.random((lh, lw)) < 0.08
            r[m] = rng.integers(1, 4, size=int(m.sum())).astype(np.float32) * 10
            r[:3] = 0
            r[-3:] = 0
            r[:, :3] = 0
            r[:, -3:] = 0
        out.append(r)
    return out


# ------------------------------------------------------------------ a6 grid NMS
@pytest.mark.parametrize("w,h,levels,src", [(640, 480, 1, "img"), (640, 480, 6, "img"),
                                            (848, 480, 6, "img"), (640, 480, 6, "ties"),
                                            (848, 480, 5, "ties"), (100, 70, 4, "ties")])
def test_grid_nms_stage(gpu, oracle_mod, w, h, levels, src):
    torch, orbfe = gpu
    if src == "img":
        _, resps, _ = _response_pyramid(oracle_mod, synth.frame(w, h, 5, "rects", **synth.DENSE), levels)
    else:
        resps = _tie_responses(w, h, levels, 42)
    rpos, rscore, rlevel = oracle_mod.grid_nms(resps, 32)
    k = oracle_mod.num_cells(w, h, 32)
    d_res = [dev(torch, r if r.size else np.zeros((1, 1), np.float32)) for r in resps]
    descs_i = [(0, w >> l, h >> l, max(w >> l, 1)) for l in range(levels)]
    descs_r = [(d_res[l].data_ptr(), w >> l, h >> l, max(w >> l, 1) * 4) for l in range(levels)]
    lv = orbfe.make_levels(descs_i, descs_r)
    grid = torch.full((4 * k,), -1.0, dtype=torch.float32, device="cuda")
    base = grid.data_ptr()
    orbfe.check(orbfe.lib().orbfe_grid_nms(lv, levels, base, base + 8 * k, base + 12 * k, stream(torch)))
    g = grid.cpu().numpy()
    np.testing.assert_array_equal(bits(g[2 * k:3 * k]), bits(rscore))
    np.testing.assert_array_equal(bits(g[:2 * k].reshape(k, 2)), bits(rpos))
    np.testing.assert_array_equal(g[3 * k:].view(np.int32), rlevel)
    if src == "ties":
        assert (rscore > 0).sum() > k // 2


# ------------------------------------------------------------------ a7 detect
def test_detect_stage(gpu, oracle_mod):
    torch, orbfe = gpu
    w, h, levels = 640, 480, 5
    imgs, resps, lut = _response_pyramid(oracle_mod, synth.frame(w, h, 9, "rects", **synth.DENSE), levels)
    rpos, rscore, rlevel = oracle_mod.grid_nms(resps, 32)
    k = oracle_mod.num_cells(w, h)
    d_img = [dev(torch, i) for i in imgs]
    d_res = [torch.full(i.shape, -1.0, dtype=torch.float32, device="cuda") for i in imgs]
    lv = orbfe.make_levels([(d_img[l].data_ptr(), w >> l, h >> l, w >> l) for l in range(levels)],
                           [(d_res[l].data_ptr(), w >> l, h >> l, (w >> l) * 4) for l in range(levels)])
    d_lut = dev(torch, lut)
    grid = torch.zeros(4 * k, dtype=torch.float32, device="cuda")
    b = grid.data_ptr()
    orbfe.check(orbfe.lib().orbfe_detect(lv, levels, d_lut.data_ptr(), 13.0, b, b + 8 * k, b + 12 * k,
                                         stream(torch)))
    g = grid.cpu().numpy()
    np.testing.assert_array_equal(bits(g[2 * k:3 * k]), bits(rscore))
    np.testing.assert_array_equal(bits(g[:2 * k].reshape(k, 2)), bits(rpos))
    np.testing.assert_array_equal(g[3 * k:].view(np.int32), rlevel)
    for l in range(levels):
        np.testing.assert_array_equal(bits(d_res[l].cpu().numpy()), bits(resps[l]))


# ------------------------------------------------------------------ a8 / a9 / a10
def _keypoint_positions(w, h, n, seed):
    rng = np.random.default_rng(seed)
    pos = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], 1).astype(np.float32)
    # corners, borders and the descriptor guard band (17 px) explicitly
    edge = np.array([[0, 0], [w - 1, h - 1], [3, 3], [w - 4, h - 4], [16, 16], [17, 17],
                     [w - 17, h - 17], [w - 16, h - 16], [17, h - 17], [w - 17, 17],
                     [15, h // 2], [w // 2, 15], [w - 1, h // 2], [w // 2, h - 1]], np.float32)
    return np.concatenate([edge, pos])  # all inside the image: the oracle (like the reference)
    # reads out of bounds for positions outside it


@pytest.mark.parametrize("w,h,kind", [(640, 480, "dense"), (848, 480, "uniform"), (100, 70, "rects")])
def test_angle_and_orb_stage(gpu, oracle_mod, w, h, kind):
    torch, orbfe = gpu
    img = oracle_mod.gaussian_blur_3x3(FRAMES[kind](w, h))
    pos = _keypoint_positions(w, h, 500, 3)
    n = pos.shape[0]
    ref_angle = oracle_mod.compute_fast_angle(pos, None, img)
    ref_desc, ref_d32 = oracle_mod.calc_orb(ref_angle, pos, img)
    d_img = pitched(torch, img, w + 7)
    d_pos = dev(torch, pos)
    d_angle = torch.full((n,), 9.0, dtype=torch.float32, device="cuda")
    orbfe.check(orbfe.lib().orbfe_compute_fast_angle(d_angle.data_ptr(), d_pos.data_ptr(),
                                                     d_img.data_ptr(), w + 7, w, h, n, stream(torch)))
    np.testing.assert_array_equal(bits(d_angle.cpu().numpy()), bits(ref_angle))
    d_desc = torch.full((n, 32), 0xEE, dtype=torch.uint8, device="cuda")
    d_d32 = torch.zeros(n, dtype=torch.int32, device="cuda")
    orbfe.check(orbfe.lib().orbfe_calc_orb(d_angle.data_ptr(), d_pos.data_ptr(), d_desc.data_ptr(),
                                           d_d32.data_ptr(), d_img.data_ptr(), w + 7, w, h, n,
                                           stream(torch)))
    np.testing.assert_array_equal(d_desc.cpu().numpy(), ref_desc)
    np.testing.assert_array_equal(d_d32.cpu().numpy().view(np.uint32), ref_d32)
    assert (ref_desc.sum(1) > 0).sum() > n // 4
    assert np.unique(bits(ref_angle)).size > n // 2


# ------------------------------------------------------------------ a11 matchers
def _match_inputs(n_prev, n_curr, seed):
    rng = np.random.default_rng(seed)
    # few distinct positions and sparse 32-bit words -> many in-window candidates and ties
    pp = rng.integers(0, 12, (n_prev, 2)).astype(np.float32)
    pc = rng.integers(0, 12, (n_curr, 2)).astype(np.float32)
    dp = (rng.integers(0, 2, (n_prev, 32)) * (rng.random((n_prev, 32)) < 0.15)).astype(np.uint32)
    dc = (rng.integers(0, 2, (n_curr, 32)) * (rng.random((n_curr, 32)) < 0.15)).astype(np.uint32)
    sh = np.arange(32, dtype=np.uint32)
    return pp, (dp << sh).sum(1).astype(np.uint32), pc, (dc << sh).sum(1).astype(np.uint32)


@pytest.mark.parametrize("n_prev,n_curr", [(0, 10), (10, 0), (1, 1), (31, 33), (32, 32), (33, 31),
                                           (300, 300), (405, 397), (1000, 70), (70, 1000), (257, 64)])
def test_match_keypoints_stage(gpu, oracle_mod, n_prev, n_curr):
    torch, orbfe = gpu
    pp, dp, pc, dc = _match_inputs(n_prev, n_curr, n_prev * 1000 + n_curr)
    ref_idx, ref_n = oracle_mod.match_keypoints(pp, dp, pc, dc, 2, 4)
    mk = lambda a, dt: dev(torch, a.view(dt) if a.size else np.zeros(2, dt))
    d_pp, d_pc = mk(pp, np.float32), mk(pc, np.float32)
    d_dp, d_dc = mk(dp, np.int32), mk(dc, np.int32)
    d_idx = torch.full((max(n_prev, 1),), -5, dtype=torch.int32, device="cuda")
    d_n = torch.full((1,), -5, dtype=torch.int32, device="cuda")
    orbfe.check(orbfe.lib().orbfe_match_keypoints(d_pp.data_ptr(), d_dp.data_ptr(), n_prev,
                                                  d_pc.data_ptr(), d_dc.data_ptr(), n_curr, 2, 4,
                                                  d_idx.data_ptr(), d_n.data_ptr(), stream(torch)))
    assert int(d_n.cpu()[0]) == ref_n
    np.testing.assert_array_equal(d_idx.cpu().numpy()[:n_prev], ref_idx)
    if min(n_prev, n_curr) >= 300:
        assert ref_n > 20


@pytest.mark.parametrize("na,nb,window,maxd", [(0, 5, -1, 256), (5, 0, -1, 256), (1, 1, -1, 256),
                                               (405, 405, -1, 256), (2000, 1999, -1, 256),
                                               (300, 700, 3, 256), (513, 255, -1, 100)])
def test_match256_stage(gpu, oracle_mod, na, nb, window, maxd):
    torch, orbfe = gpu
    da, db = synth.descriptors(max(na, 1), 5)[:na], synth.descriptors(max(nb, 1), 6)[:nb]
    if nb > 10 and na > 10:
        db[7] = db[3]            # exact duplicates in B: tie -> lower index
        da[5] = db[3]
        da[6] = db[9]
    rng = np.random.default_rng(na + nb)
    pa = rng.integers(0, 12, (na, 2)).astype(np.float32)
    pb = rng.integers(0, 12, (nb, 2)).astype(np.float32)
    ref_idx, ref_dist = oracle_mod.match256(da, db, pa, pb, window, maxd)
    z = lambda a, dt: dev(torch, a if a.size else np.zeros((1, 32), dt))
    d_a, d_b = z(da, np.uint8), z(db, np.uint8)
    d_pa, d_pb = z(pa, np.float32), z(pb, np.float32)
    d_idx = torch.full((max(na, 1),), -5, dtype=torch.int32, device="cuda")
    d_dist = torch.full((max(na, 1),), -5, dtype=torch.int32, device="cuda")
    orbfe.check(orbfe.lib().orbfe_match256(d_a.data_ptr(), d_pa.data_ptr(), na, d_b.data_ptr(),
                                           d_pb.data_ptr(), nb, window, maxd, d_idx.data_ptr(),
                                           d_dist.data_ptr(), stream(torch)))
    np.testing.assert_array_equal(d_idx.cpu().numpy()[:na], ref_idx)
    np.testing.assert_array_equal(d_dist.cpu().numpy()[:na], ref_dist)
    if na > 10 and nb > 10 and window < 0 and maxd == 256:
        assert ref_idx[5] == 3 and ref_dist[5] == 0


# ------------------------------------------------------------------ batch extract
def _run_extract(torch, orbfe, frames, want_soa=True, **cfg):
    n, h, w = frames.shape
    ctx = orbfe.Context(w, h, max_batch=n, **cfg)
    d_in = dev(torch, frames)
    rec = torch.zeros(n * ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    K = ctx.K
    soa_t = None
    soa = None
    if want_soa:
        soa_t = dict(pos=torch.full((n, K, 2), -1.0, device="cuda"), score=torch.full((n, K), -1.0, device="cuda"),
                     level=torch.full((n, K), -1, dtype=torch.int32, device="cuda"),
                     angle=torch.full((n, K), -1.0, device="cuda"),
                     desc=torch.full((n, K, 32), 0xEE, dtype=torch.uint8, device="cuda"),
                     desc32=torch.full((n, K), -1, dtype=torch.int32, device="cuda"))
        soa = orbfe.Soa(soa_t["pos"].data_ptr(), soa_t["score"].data_ptr(), soa_t["level"].data_ptr(),
                        soa_t["angle"].data_ptr(), soa_t["desc"].data_ptr(), soa_t["desc32"].data_ptr())
    ctx.extract(d_in.data_ptr(), w, w * h, n, rec.data_ptr(), cnt.data_ptr(), soa, stream(torch))
    torch.cuda.synchronize()
    records = rec.cpu().numpy().view(orbfe.KEYPOINT_DTYPE).reshape(n, ctx.cap)
    counts = cnt.cpu().numpy()
    soa_np = {k: v.cpu().numpy() for k, v in soa_t.items()} if want_soa else None
    return ctx, records, counts, soa_np


def _check_extract(oracle_mod, ctx, frames, records, counts, soa_np, **cfg):
    n, h, w = frames.shape
    ocfg = oracle_mod.make_config(w, h, levels=cfg.get("levels", 1), cell=cfg.get("cell", 32),
                                  fast_threshold=float(cfg.get("fast_threshold", 13)),
                                  min_arc=cfg.get("min_arc", 12),
                                  max_features=cfg.get("max_features", 0),
                                  angle_in_radians=cfg.get("angle_in_radians", 0),
                                  descriptor_level=cfg.get("descriptor_level", 0))
    total = 0
    for f in range(n):
        ref = oracle_mod.extract_frame(frames[f], ocfg, want_pyramid=True)
        for l in range(ocfg.levels):
            np.testing.assert_array_equal(ctx.read_level(l, f), ref["pyramid"][l],
                                          err_msg="pyramid level %d frame %d" % (l, f))
        assert counts[f] == ref["count"], "frame %d count" % f
        got = records[f, :counts[f]]
        assert got.tobytes() == ref["records"].tobytes(), "frame %d records differ" % f
        if soa_np is not None:
            np.testing.assert_array_equal(bits(soa_np["score"][f]), bits(ref["score"]))
            np.testing.assert_array_equal(bits(soa_np["pos"][f]), bits(ref["pos"]))
            np.testing.assert_array_equal(soa_np["level"][f], ref["level"])
            np.testing.assert_array_equal(bits(soa_np["angle"][f]), bits(ref["angle"]))
            np.testing.assert_array_equal(soa_np["desc"][f], ref["desc"])
            np.testing.assert_array_equal(soa_np["desc32"][f].view(np.uint32), ref["desc32"])
        total += ref["count"]
    return total


def _mixed_frames(w, h):
    return np.stack([FRAMES[k](w, h) for k in ("rects", "dense", "uniform", "const", "checker")])


@pytest.mark.parametrize("w,h,levels", [(640, 480, 1), (640, 480, 6), (848, 480, 6), (100, 70, 3),
                                        (640, 480, 10), (1280, 720, 9), (636, 476, 2), (130, 258, 8),
                                        (1030, 770, 11)])  # odd width: unfused levels 1..7, then the one-launch tail
def test_extract_reference_mode(gpu, oracle_mod, w, h, levels):
    """Reference-parity configuration: 32-px cells, FAST-12, t = 13, level-0 description."""
    torch, orbfe = gpu
    frames = _mixed_frames(w, h)
    cfg = dict(levels=levels)
    ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, **cfg)
    total = _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg)
    assert cnt[3] == 0, "constant frame must give no keypoints"
    assert total > 20


@pytest.mark.parametrize("cfg", [
    dict(levels=8, cell=8, min_arc=9, max_features=2000),   # the bench configuration (C2)
    dict(levels=8, cell=8, min_arc=9, max_features=0),
    dict(levels=8, cell=16, min_arc=10, max_features=300, fast_threshold=20),
    dict(levels=7, cell=64, min_arc=12, max_features=0),
    dict(levels=4, cell=32, min_arc=11, max_features=17, angle_in_radians=1),
    dict(levels=8, cell=8, min_arc=9, max_features=2000, angle_in_radians=1),
])
def test_extract_ext_modes(gpu, oracle_mod, cfg):
    torch, orbfe = gpu
    frames = _mixed_frames(640, 480)
    ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, **cfg)
    total = _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg)
    assert total > 50
    if cfg["max_features"]:
        assert cnt.max() <= cfg["max_features"]
        if cfg["cell"] == 8:
            assert cnt[2] == cfg["max_features"], "uniform noise fills the feature budget"


def test_extract_unaligned_input_pitch(gpu, oracle_mod):
    torch, orbfe = gpu
    w, h = 333, 97
    img = synth.frame(w, h, 4, "rects", **synth.DENSE)
    ctx = orbfe.Context(w, h, levels=4, max_batch=1)
    d_in = pitched(torch, img, w + 1)
    rec = torch.zeros(ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.extract(d_in.data_ptr(), w + 1, (w + 1) * h, 1, rec.data_ptr(), cnt.data_ptr(), None, stream(torch))
    ref = oracle_mod.extract_frame(img, oracle_mod.make_config(w, h, levels=4))
    assert int(cnt.cpu()[0]) == ref["count"] > 0
    assert rec.cpu().numpy().view(orbfe.KEYPOINT_DTYPE)[:ref["count"]].tobytes() == ref["records"].tobytes()


def test_error_paths(gpu):
    torch, orbfe = gpu
    with pytest.raises(orbfe.OrbfeError) as e:
        orbfe.Context(640, 480, cell=24)
    assert e.value.code == orbfe.ERR_UNSUPPORTED
    with pytest.raises(orbfe.OrbfeError) as e:
        orbfe.Context(640, 480, min_arc=8)
    assert e.value.code == orbfe.ERR_UNSUPPORTED
    ctx = orbfe.Context(64, 64, max_batch=2)
    buf = torch.zeros(64 * 64 * 3, dtype=torch.uint8, device="cuda")
    rec = torch.zeros(3 * ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(3, dtype=torch.int32, device="cuda")
    with pytest.raises(orbfe.OrbfeError) as e:
        ctx.extract(buf.data_ptr(), 64, 64 * 64, 3, rec.data_ptr(), cnt.data_ptr(), None, stream(torch))
    assert e.value.code == orbfe.ERR_CAPACITY
    assert orbfe.lib().orbfe_gaussian_blur_3x3(0, 0, 0, 0, 0, 0, 0) == orbfe.ERR_INVALID_ARG
    lv = orbfe.make_levels([(1, 640 >> l, 480 >> l, 640 >> l) for l in range(7)],
                           [(1, 640 >> l, 480 >> l, (640 >> l) * 4) for l in range(7)])
    assert orbfe.lib().orbfe_grid_nms(lv, 7, 1, 1, 1, 0) == orbfe.ERR_UNSUPPORTED


# ------------------------------------------------------------------ batch matcher
@pytest.mark.parametrize("mode,window,maxd", [(0, 2, 4), (0, 6, 9), (1, -1, 256), (1, 8, 64)])
def test_match_batch(gpu, oracle_mod, mode, window, maxd):
    torch, orbfe = gpu
    w, h = 640, 480
    a, b = synth.shifted_pair(w, h, 3, dx=1, dy=0, **synth.DENSE)
    c = synth.frame(w, h, 8, "rects", **synth.DENSE)
    frames = np.stack([a, b, c, c])
    cfg = dict(levels=4, cell=16, min_arc=9)
    ctx, rec, cnt, _ = _run_extract(torch, orbfe, frames, want_soa=False, **cfg)
    n = frames.shape[0]
    d_rec = dev(torch, rec.view(np.uint8).reshape(-1))
    d_cnt = dev(torch, cnt)
    d_idx = torch.full(((n - 1) * ctx.cap,), -7, dtype=torch.int32, device="cuda")
    d_dist = torch.full(((n - 1) * ctx.cap,), -7, dtype=torch.int32, device="cuda")
    ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), n, mode, window, maxd, d_idx.data_ptr(),
                    d_dist.data_ptr(), stream(torch))
    idx = d_idx.cpu().numpy().reshape(n - 1, ctx.cap)
    dist = d_dist.cpu().numpy().reshape(n - 1, ctx.cap)
    n_matched = 0
    for p in range(n - 1):
        A, B = rec[p, :cnt[p]], rec[p + 1, :cnt[p + 1]]
        pa = np.stack([A["x"], A["y"]], 1)
        pb = np.stack([B["x"], B["y"]], 1)
        if mode == 0:
            comp = lambda d: ((d == 1).astype(np.uint32) << np.arange(32, dtype=np.uint32)).sum(1).astype(np.uint32)
            ref_idx, _ = oracle_mod.match_keypoints(pa, comp(A["desc"]), pb, comp(B["desc"]), window, maxd)
        else:
            ref_idx, ref_dist = oracle_mod.match256(A["desc"], B["desc"], pa, pb, window, maxd)
            np.testing.assert_array_equal(dist[p, :cnt[p]], ref_dist)
        np.testing.assert_array_equal(idx[p, :cnt[p]], ref_idx)
        assert (idx[p, cnt[p]:] == -1).all()
        n_matched += int((ref_idx >= 0).sum())
    if mode == 1:
        # frames 2 and 3 are identical: every keypoint finds a distance-0 partner, which is
        # itself unless an earlier keypoint holds the same descriptor (e.g. the all-zero
        # descriptors of the 17-px guard band): ties go to the lower index
        own = np.arange(cnt[2])
        assert (dist[2, :cnt[2]] == 0).all()
        assert (idx[2, :cnt[2]] <= own).all() and (idx[2, :cnt[2]] == own).mean() > 0.8
        np.testing.assert_array_equal(rec[3, idx[2, :cnt[2]]]["desc"], rec[2, :cnt[2]]["desc"])
    assert n_matched > 10


@pytest.mark.parametrize("w,h,kw", [(100, 70, dict(cell=8)),                       # cap 117: not a multiple of 16
                                    (640, 480, dict(cell=32)),                     # cap 300
                                    (640, 480, dict(cell=8, max_features=2000)),   # the bench regime
                                    (640, 480, dict(cell=8))])                     # cap 4800: 38 row blocks
@pytest.mark.parametrize("maxd", [256, 90, 0])
def test_match_batch_256_crafted_records(gpu, oracle_mod, w, h, kw, maxd):
    """The all-candidates 256-bit matcher (matrix-core path) on hand-made records: random,
    duplicated, all-zero and all-one descriptors, ragged counts including 0 and 1, ties (the
    lower index must win), every output slot beyond the count = -1."""
    torch, orbfe = gpu
    ctx = orbfe.Context(w, h, max_batch=8, **kw)
    cap = ctx.cap
    rng = np.random.default_rng(cap * 1000 + maxd)
    counts = np.array([cap, 0, 5, cap, 1, cap - 1, min(cap, 37), cap], np.int32)
    n = len(counts)
    rec = np.zeros((n, cap), orbfe.KEYPOINT_DTYPE)
    rec["desc"] = rng.integers(0, 256, (n, cap, 32), dtype=np.uint8)
    rec["x"] = rng.uniform(0, w, (n, cap)).astype(np.float32)
    rec["y"] = rng.uniform(0, h, (n, cap)).astype(np.float32)
    # near-duplicates across consecutive frames so small distances exist, plus exact ties
    for f in range(1, n):
        m = min(counts[f - 1], counts[f])
        if m < 4:
            continue
        src = rng.integers(0, counts[f - 1], m // 2)
        dst = rng.integers(0, counts[f], m // 2)
        d = rec["desc"][f - 1, src].copy()
        flips = rng.integers(0, 40, len(src))
        for k, nf in enumerate(flips):
            for bit in rng.integers(0, 256, nf):
                d[k, bit >> 3] ^= np.uint8(1 << (bit & 7))
        rec["desc"][f, dst] = d
        rec["desc"][f, dst[: len(dst) // 4]] = rec["desc"][f, dst[0]]  # ties: same descriptor at many indices
    rec["desc"][3, : min(cap, 3)] = 0
    rec["desc"][3, min(cap, 3): min(cap, 6)] = 255
    rec["desc"][7, cap // 2] = 0
    rec["desc"][7, cap - 1] = 255
    d_rec = dev(torch, rec.view(np.uint8).reshape(-1))
    d_cnt = dev(torch, counts)
    d_idx = torch.full(((n - 1) * cap,), -7, dtype=torch.int32, device="cuda")
    d_dist = torch.full(((n - 1) * cap,), -7, dtype=torch.int32, device="cuda")
    ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), n, 1, -1, maxd, d_idx.data_ptr(), d_dist.data_ptr(),
                    stream(torch))
    idx = d_idx.cpu().numpy().reshape(n - 1, cap)
    dist = d_dist.cpu().numpy().reshape(n - 1, cap)
    for p in range(n - 1):
        A, B = rec[p, :counts[p]], rec[p + 1, :counts[p + 1]]
        ref_idx, ref_dist = oracle_mod.match256(A["desc"], B["desc"], None, None, -1, maxd)
        np.testing.assert_array_equal(idx[p, :counts[p]], ref_idx)
        np.testing.assert_array_equal(dist[p, :counts[p]], ref_dist)
        assert (idx[p, counts[p]:] == -1).all() and (dist[p, counts[p]:] == -1).all()
    # d_dist may be NULL
    d_idx2 = torch.full(((n - 1) * cap,), -7, dtype=torch.int32, device="cuda")
    ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), n, 1, -1, maxd, d_idx2.data_ptr(), None, stream(torch))
    assert torch.equal(d_idx, d_idx2)


@pytest.mark.parametrize("w,h,kw", [(100, 70, dict(cell=8)), (640, 480, dict(cell=16)),
                                    (640, 480, dict(cell=8, max_features=2000))])
def test_match_batch_256_ragged_counts_fuzz(gpu, oracle_mod, w, h, kw):
    """Random keypoint counts per frame (0 .. cap, many small ones): every wave of the matrix-core
    matcher sees 0, 1, 2, ... candidate blocks, so all tails of its 4-slot LDS ring are exercised;
    descriptors are drawn from a small pool so that exact ties are frequent."""
    torch, orbfe = gpu
    n = 12
    ctx = orbfe.Context(w, h, max_batch=n, **kw)
    cap = ctx.cap
    rng = np.random.default_rng(cap + 7)
    for trial in range(6):
        small = rng.integers(0, min(cap, 200) + 1, n)
        big = rng.integers(0, cap + 1, n)
        counts = np.where(rng.random(n) < 0.5, small, big).astype(np.int32)
        counts[rng.integers(0, n)] = cap
        pool = rng.integers(0, 256, (64, 32), dtype=np.uint8)
        rec = np.zeros((n, cap), orbfe.KEYPOINT_DTYPE)
        rec["desc"] = rng.integers(0, 256, (n, cap, 32), dtype=np.uint8)
        dup = rng.random((n, cap)) < 0.3
        rec["desc"][dup] = pool[rng.integers(0, 64, int(dup.sum()))]
        maxd = int(rng.choice([256, 120, 40, 0]))
        d_rec = dev(torch, rec.view(np.uint8).reshape(-1))
        d_cnt = dev(torch, counts)
        d_idx = torch.full(((n - 1) * cap,), -7, dtype=torch.int32, device="cuda")
        d_dist = torch.full(((n - 1) * cap,), -7, dtype=torch.int32, device="cuda")
        ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), n, 1, -1, maxd, d_idx.data_ptr(), d_dist.data_ptr(),
                        stream(torch))
        idx = d_idx.cpu().numpy().reshape(n - 1, cap)
        dist = d_dist.cpu().numpy().reshape(n - 1, cap)
        for p in range(n - 1):
            A, B = rec[p, :counts[p]], rec[p + 1, :counts[p + 1]]
            ref_idx, ref_dist = oracle_mod.match256(A["desc"], B["desc"], None, None, -1, maxd)
            np.testing.assert_array_equal(idx[p, :counts[p]], ref_idx, err_msg="trial %d pair %d" % (trial, p))
            np.testing.assert_array_equal(dist[p, :counts[p]], ref_dist)
            assert (idx[p, counts[p]:] == -1).all()


@pytest.mark.parametrize("maxf,path", [(16384, "matrix cores, index uses all 14 bits"),
                                       (16400, "VALU kernel: more than 16384 keypoints per frame")])
def test_match_batch_256_at_the_key_packing_limit(gpu, oracle_mod, maxf, path):
    """The matrix-core matcher packs (distance, index) into one f32-exact integer with a 14-bit
    index: 16384 keypoints per frame is its limit and the last index must still win / tie-break
    correctly; one keypoint more takes the VALU kernel.  Same results either way."""
    torch, orbfe = gpu
    ctx = orbfe.Context(2048, 1024, max_batch=2, cell=8, max_features=maxf)  # K = 32768 cells
    cap = ctx.cap
    assert cap == maxf
    rng = np.random.default_rng(maxf)
    rec = np.zeros((2, cap), orbfe.KEYPOINT_DTYPE)
    rec["desc"] = rng.integers(0, 256, (2, cap, 32), dtype=np.uint8)
    # query 0's only exact partner is the LAST candidate; queries 1 and 2 have two exact partners
    # (cap - 3 and cap - 2): the lower index wins
    rec["desc"][1, cap - 1] = rec["desc"][0, 0]
    rec["desc"][1, cap - 2] = rec["desc"][0, 1]
    rec["desc"][1, cap - 3] = rec["desc"][0, 1]
    rec["desc"][0, 2] = rec["desc"][0, 1]
    # extreme popcounts: distance 256 and 0 against all-ones / all-zeros
    rec["desc"][0, 3] = 0
    rec["desc"][0, 4] = 255
    rec["desc"][1, 7] = 255
    rec["desc"][1, 9] = 0
    counts = np.array([cap, cap], np.int32)
    d_rec = dev(torch, rec.view(np.uint8).reshape(-1))
    d_cnt = dev(torch, counts)
    d_idx = torch.full((cap,), -7, dtype=torch.int32, device="cuda")
    d_dist = torch.full((cap,), -7, dtype=torch.int32, device="cuda")
    ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), 2, 1, -1, 256, d_idx.data_ptr(), d_dist.data_ptr(),
                    stream(torch))
    idx, dist = d_idx.cpu().numpy(), d_dist.cpu().numpy()
    ref_idx, ref_dist = oracle_mod.match256(rec["desc"][0], rec["desc"][1], None, None, -1, 256)
    np.testing.assert_array_equal(idx, ref_idx)
    np.testing.assert_array_equal(dist, ref_dist)
    assert idx[0] == cap - 1 and dist[0] == 0
    assert idx[1] == cap - 3 and idx[2] == cap - 3 and dist[1] == 0
    assert dist[3] == 0 and idx[3] == 9 and dist[4] == 0 and idx[4] == 7


# ------------------------------------------------------------------ full-size properties
def test_full_size_batch_properties(gpu, oracle_mod):
    """BASELINE configs[1] at bench size (batch 256): size-independent properties.
    (i) a frame's result does not depend on its position in the batch or on its neighbours;
    (ii) two runs are bit-identical (no atomics-order dependence); (iii) counts respect the
    feature budget; (iv) sampled frames equal the oracle."""
    torch, orbfe = gpu
    w, h, n = 640, 480, 256
    base = synth.frames(w, h, 8, 100, "rects", **synth.DENSE)
    frames = base[np.arange(n) % 8]
    cfg = dict(levels=8, cell=8, min_arc=9, max_features=2000)
    ctx, rec, cnt, _ = _run_extract(torch, orbfe, frames, want_soa=False, **cfg)
    _, rec2, cnt2, _ = _run_extract(torch, orbfe, frames, want_soa=False, **cfg)
    assert rec.tobytes() == rec2.tobytes() and (cnt == cnt2).all()
    assert cnt.max() <= 2000 and cnt.min() > 500
    for f in range(8, n):
        assert cnt[f] == cnt[f % 8]
        assert rec[f, :cnt[f]].tobytes() == rec[f % 8, :cnt[f]].tobytes()
    ocfg = oracle_mod.make_config(w, h, levels=8, cell=8, min_arc=9, max_features=2000)
    for f in (0, 5):
        ref = oracle_mod.extract_frame(frames[f], ocfg)
        assert cnt[f] == ref["count"]
        assert rec[f, :cnt[f]].tobytes() == ref["records"].tobytes()


# ------------------------------------------------------------------ tile-sharded detection (C5)
@pytest.mark.parametrize("w,h,levels,shards", [(3840, 2160, 12, 8), (640, 480, 6, 3)])
def test_tile_sharded_detection_merges_exactly(gpu, oracle_mod, w, h, levels, shards):
    """BASELINE configs[4]: one 4K frame, 12 levels built, K = 8160 cells, detection tiles
    sharded 8 ways.  Each shard produces partial cell keys; their element-wise maximum (what an
    all-reduce(MAX) over RCCL computes) followed by describe must equal the unsharded result and
    the oracle, bit for bit.  Shards run one after another on the single test GPU."""
    torch, orbfe = gpu
    img = synth.frame(w, h, 12, "rects", n_rects=800 * (w * h) // (640 * 480), min_size=6, max_size=32)
    ctx = orbfe.Context(w, h, levels=levels, max_batch=1)
    d_in = dev(torch, img)
    s = stream(torch)
    ctx.build_pyramid(d_in.data_ptr(), w, w * h, 1, s)
    merged = torch.zeros(ctx.K, dtype=torch.int32, device="cuda")
    part = torch.zeros(ctx.K, dtype=torch.int32, device="cuda")
    nonempty = []
    for i in range(shards):
        ctx.detect_batch_shard(1, i, shards, s)
        ctx.export_cell_keys(1, part.data_ptr(), s)
        nonempty.append(int((part > 0).sum()))
        merged = torch.maximum(merged, part)          # == dist.all_reduce(op=MAX)
    ctx.import_cell_keys(1, merged.data_ptr(), s)
    rec = torch.zeros(ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.describe_batch(1, rec.data_ptr(), cnt.data_ptr(), None, s)
    torch.cuda.synchronize()
    ref = oracle_mod.extract_frame(img, oracle_mod.make_config(w, h, levels=levels))
    n = int(cnt.cpu()[0])
    assert n == ref["count"] and n > ctx.K // 2
    assert rec.cpu().numpy().view(orbfe.KEYPOINT_DTYPE)[:n].tobytes() == ref["records"].tobytes()
    assert min(nonempty) > 0 and max(nonempty) < n, "every shard contributes, none sees everything"


# ------------------------------------------------------------------ f1: RGB8 -> gray
def _rgb_frame(w, h, seed):
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    # make sure the exact-tie triples 7B + 72G + 21R == 50 (mod 100) are well represented
    rgb[0, : min(w, 64)] = [[r, 50, (50 - 21 * r - 72 * 50) * 43 % 100] for r in range(min(w, 64))]
    return rgb


@pytest.mark.parametrize("w,h,extra", [(640, 480, 0), (37, 19, 5), (848, 480, 4)])
def test_rgb_to_grayscale_stage(gpu, oracle_mod, w, h, extra):
    torch, orbfe = gpu
    rgb = _rgb_frame(w, h, 5)
    ref = oracle_mod.rgb_to_grayscale(rgb)
    buf = np.full((h, 3 * w + extra), 0x33, np.uint8)
    buf[:, :3 * w] = rgb.reshape(h, 3 * w)
    d_src = dev(torch, buf)
    d_dst = torch.full((h, w + 3), 0x44, dtype=torch.uint8, device="cuda")
    orbfe.check(orbfe.lib().orbfe_rgb_to_grayscale(d_dst.data_ptr(), d_src.data_ptr(), w, h, w + 3,
                                                   3 * w + extra, stream(torch)))
    got = d_dst.cpu().numpy()
    np.testing.assert_array_equal(got[:, :w], ref)
    assert (got[:, w:] == 0x44).all()


@pytest.mark.parametrize("w,h,levels", [(640, 480, 8), (100, 72, 3)])
def test_extract_rgb_fused(gpu, oracle_mod, w, h, levels):
    """RGB input: the conversion is fused into the pyramid kernel; result == oracle gray -> extract."""
    torch, orbfe = gpu
    n = 3
    gray_scene = [synth.frame(w, h, 20 + i, "rects", **synth.DENSE) for i in range(n)]
    rng = np.random.default_rng(9)
    rgb = np.stack([np.stack([g, np.roll(g, 1, 1), 255 - g], -1) for g in gray_scene])  # three unlike channels
    rgb = (rgb.astype(np.int16) + rng.integers(-2, 3, rgb.shape)).clip(0, 255).astype(np.uint8)
    cfg = dict(levels=levels, cell=8, min_arc=9, max_features=500)
    ctx = orbfe.Context(w, h, max_batch=n, **cfg)
    d_in = dev(torch, rgb)
    rec = torch.zeros(n * ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.extract_rgb(d_in.data_ptr(), 3 * w, 3 * w * h, n, rec.data_ptr(), cnt.data_ptr(), None, stream(torch))
    torch.cuda.synchronize()
    records = rec.cpu().numpy().view(orbfe.KEYPOINT_DTYPE).reshape(n, ctx.cap)
    ocfg = oracle_mod.make_config(w, h, **cfg)
    for f in range(n):
        gray = oracle_mod.rgb_to_grayscale(rgb[f])
        ref = oracle_mod.extract_frame(gray, ocfg, want_pyramid=True)
        np.testing.assert_array_equal(ctx.read_level(0, f), ref["pyramid"][0])
        assert int(cnt[f]) == ref["count"] > 50
        assert records[f, :ref["count"]].tobytes() == ref["records"].tobytes()


# ------------------------------------------------------------------ f2: keypoint filter + deprojection
@pytest.mark.parametrize("model,fix", [(0, 0), (2, 0), (4, 1), (2, 1)])
@pytest.mark.parametrize("n", [0, 1, 300, 405, 1000])
def test_keypoint_pixel_to_point(gpu, oracle_mod, model, fix, n):
    torch, orbfe = gpu
    w, h = 848, 480
    rng = np.random.default_rng(n * 10 + model + fix)
    depth = rng.integers(0, 5000, (h, w)).astype(np.uint32)
    depth[rng.random((h, w)) < 0.3] = 0            # holes
    depth[rng.random((h, w)) < 0.05] = 1           # depth == 1 is rejected too (depth > 1)
    pos = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], 1).astype(np.float32)
    score = rng.choice([0.0, 1.0, 2.0, 57.0], n).astype(np.float32)
    desc = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    coeffs = (C.c_float * 5)(0.12, -0.25, 0.001, -0.0007, 0.09)
    ok = oracle_mod.Intrinsics(w, h, 423.6, 241.3, 610.5, 609.8, model, coeffs)
    gk = orbfe.Intrinsics(w, h, 423.6, 241.3, 610.5, 609.8, model, coeffs)
    rpos, rpts, rdesc, rcnt = oracle_mod.keypoint_pixel_to_point(depth, ok, pos, score, desc, fix)
    z = lambda a, dt: dev(torch, a if a.size else np.zeros(4, dt))
    d_depth, d_pos, d_score, d_desc = dev(torch, depth.view(np.int32)), z(pos, np.float32), z(score, np.float32), z(desc.view(np.int32), np.int32)
    o_pos = torch.full((max(n, 1), 2), -1.0, device="cuda")
    o_pts = torch.full((max(n, 1), 3), -1.0, dtype=torch.float64, device="cuda")
    o_desc = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
    o_n = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    orbfe.check(orbfe.lib().orbfe_keypoint_pixel_to_point(
        d_depth.data_ptr(), C.byref(gk), w, h, o_pos.data_ptr(), d_pos.data_ptr(), d_score.data_ptr(),
        o_pts.data_ptr(), o_desc.data_ptr(), d_desc.data_ptr(), n, o_n.data_ptr(), fix, stream(torch)))
    cnt = int(o_n.cpu()[0])
    assert cnt == rcnt
    np.testing.assert_array_equal(o_pos.cpu().numpy()[:cnt], rpos)
    np.testing.assert_array_equal(o_pts.cpu().numpy()[:cnt].view(np.uint64), rpts.view(np.uint64))  # f64 bit patterns
    np.testing.assert_array_equal(o_desc.cpu().numpy()[:cnt].view(np.uint32), rdesc)
    if n >= 300:
        assert 0 < cnt < n


def test_keypoint_pixel_to_point_rejects_forward_distortion(gpu):
    torch, orbfe = gpu
    k = orbfe.Intrinsics(64, 48, 32, 24, 50, 50, 1, (C.c_float * 5)())
    one = torch.zeros(8, dtype=torch.int32, device="cuda")
    rc = orbfe.lib().orbfe_keypoint_pixel_to_point(one.data_ptr(), C.byref(k), 64, 48, one.data_ptr(), one.data_ptr(),
                                                   one.data_ptr(), one.data_ptr(), one.data_ptr(), one.data_ptr(), 1,
                                                   one.data_ptr(), 1, 0)
    assert rc == orbfe.ERR_UNSUPPORTED


# ------------------------------------------------------------------ randomised configurations
def test_fuzz_random_configurations(gpu, oracle_mod):
    """40 seeded random (size, levels, cell, arc, threshold, budget, angle mode, scene) draws: the
    batch path must equal the oracle record for record, whatever the geometry (odd sizes take the
    unfused pyramid kernels, W % 4 == 0 the fused one; tiles, cells and levels rarely align)."""
    torch, orbfe = gpu
    import os
    rng = np.random.default_rng(int(os.environ.get("ORBFE_FUZZ_SEED", "20261004")))
    kinds = ["rects", "dense", "uniform", "checker"]
    total = 0
    for trial in range(int(os.environ.get("ORBFE_FUZZ_TRIALS", "40"))):
        w = int(rng.integers(40, 400))
        h = int(rng.integers(40, 300))
        if trial % 3 == 0:
            w -= w % 4
        cfg = dict(levels=int(rng.integers(1, 9)), cell=int(rng.choice([8, 16, 32, 64])),
                   min_arc=int(rng.integers(9, 13)), fast_threshold=int(rng.integers(3, 40)),
                   max_features=int(rng.choice([0, 0, 7, 50, 400])), angle_in_radians=int(rng.integers(0, 2)))
        n = int(rng.integers(1, 4))
        frames = np.stack([synth.frame(w, h, int(rng.integers(0, 10 ** 6)), "uniform") if k == "uniform" else
                           FRAMES[k](w, h) for k in rng.choice(kinds, n)])
        ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, **cfg)
        try:
            total += _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg)
        except AssertionError as e:
            raise AssertionError("trial %d: %dx%d %r: %s" % (trial, w, h, cfg, e))
        ctx.close()
    assert total > 2000


# ------------------------------------------------------------------ the C++ port of buildStream
def test_cpp_buildstream_port(gpu, oracle_mod, tmp_path):
    """examples/buildstream_port.cpp issues the reference's per-frame call sequence
    (buildStream.cpp:399-466) through compat/jetracer_compat.hpp, i.e. with the reference's own
    function names; its feature grid, angles and descriptors must equal the oracle's."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "buildstream_port")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    w, h, levels = 848, 480, 5
    g = synth.frame(w, h, 31, "rects", **synth.DENSE)
    rgb = np.stack([g, np.roll(g, 2, 0), np.roll(g, 3, 1)], -1)
    fin, fout = str(tmp_path / "rgb.bin"), str(tmp_path / "out.bin")
    rgb.tofile(fin)
    subprocess.check_call([exe, str(w), str(h), str(levels), fin, fout])
    k = oracle_mod.num_cells(w, h)
    raw = np.fromfile(fout, np.uint8)
    assert raw.size == k * (16 + 4 + 32 + 4)
    grid = raw[:16 * k].view(np.float32)
    angle = raw[16 * k:20 * k].view(np.float32)
    desc = raw[20 * k:52 * k].reshape(k, 32)
    d32 = raw[52 * k:].view(np.uint32)
    # oracle, stage by stage as the reference calls them (all K cells get an angle: the stage
    # API has no score input, exactly like compute_fast_angle)
    gray = oracle_mod.rgb_to_grayscale(rgb)
    imgs, resps, _ = _response_pyramid(oracle_mod, gray, levels)
    rpos, rscore, rlevel = oracle_mod.grid_nms(resps, 32)
    rangle = oracle_mod.compute_fast_angle(rpos, None, imgs[0])
    rdesc, rd32 = oracle_mod.calc_orb(rangle, rpos, imgs[0])
    np.testing.assert_array_equal(bits(grid[:2 * k].reshape(k, 2)), bits(rpos))
    np.testing.assert_array_equal(bits(grid[2 * k:3 * k]), bits(rscore))
    np.testing.assert_array_equal(grid[3 * k:].view(np.int32), rlevel)
    np.testing.assert_array_equal(bits(angle), bits(rangle))
    np.testing.assert_array_equal(desc, rdesc)
    np.testing.assert_array_equal(d32, rd32)
    assert (rscore > 0).sum() > k // 2


# ------------------------------------------------------------------ stream capture
def test_calls_are_graph_capturable(gpu):
    """No entry point allocates or synchronises, so a whole step can be captured into a HIP graph
    (the caller may do that; the bench does not: replay measured slower than eager issue here)."""
    torch, orbfe = gpu
    w, h, n = 640, 480, 4
    frames = dev(torch, synth.frames(w, h, n, 50, "rects", **synth.DENSE))
    ctx = orbfe.Context(w, h, levels=8, cell=8, min_arc=9, max_features=2000, max_batch=n)
    rec = torch.zeros(n * ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    idx = torch.zeros((n - 1) * ctx.cap, dtype=torch.int32, device="cuda")

    def step(s):
        ctx.extract(frames.data_ptr(), w, w * h, n, rec.data_ptr(), cnt.data_ptr(), None, s)
        ctx.match_batch(rec.data_ptr(), cnt.data_ptr(), n, 1, -1, 256, idx.data_ptr(), None, s)

    step(stream(torch))
    torch.cuda.synchronize()
    want = (rec.clone(), cnt.clone(), idx.clone())
    rec.zero_(), cnt.zero_(), idx.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step(stream(torch))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(rec, want[0]) and torch.equal(cnt, want[1]) and torch.equal(idx, want[2])
    assert int(cnt.sum()) == n * 2000


# ------------------------------------------------------------------ BASELINE configs[2]
@pytest.mark.parametrize("mode,cfg,window,maxd", [
    (0, dict(levels=1), 4, 4),                                      # the reference's live regime (K = 405)
    (0, dict(levels=6), 4, 8),
    (1, dict(levels=8, cell=8, min_arc=9, max_features=2000), -1, 256),
    (1, dict(levels=8, cell=8, min_arc=9, max_features=2000, angle_in_radians=1), 6, 80),
])
def test_stereo_pair_848x480_extract_and_match(gpu, oracle_mod, mode, cfg, window, maxd):
    """848x480 pair (right = left shifted by 3 px + noise): extract both frames, match left -> right;
    records, match indices and distances equal the oracle's."""
    torch, orbfe = gpu
    w, h = 848, 480
    left, right = synth.shifted_pair(w, h, 77, dx=3, dy=0, **synth.DENSE)
    frames = np.stack([left, right])
    ctx, rec, cnt, _ = _run_extract(torch, orbfe, frames, want_soa=False, **cfg)
    ocfg = oracle_mod.make_config(w, h, levels=cfg.get("levels", 1), cell=cfg.get("cell", 32),
                                  min_arc=cfg.get("min_arc", 12), max_features=cfg.get("max_features", 0),
                                  angle_in_radians=cfg.get("angle_in_radians", 0))
    refs = [oracle_mod.extract_frame(frames[f], ocfg)["records"] for f in range(2)]
    for f in range(2):
        assert cnt[f] == len(refs[f]) and rec[f, :cnt[f]].tobytes() == refs[f].tobytes()
    d_rec, d_cnt = dev(torch, rec.view(np.uint8).reshape(-1)), dev(torch, cnt)
    d_idx = torch.full((ctx.cap,), -7, dtype=torch.int32, device="cuda")
    d_dist = torch.full((ctx.cap,), -7, dtype=torch.int32, device="cuda")
    ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), 2, mode, window, maxd, d_idx.data_ptr(), d_dist.data_ptr(),
                    stream(torch))
    a, b = refs
    pa, pb = np.stack([a["x"], a["y"]], 1), np.stack([b["x"], b["y"]], 1)
    if mode == 0:
        comp = lambda d: ((d == 1).astype(np.uint32) << np.arange(32, dtype=np.uint32)).sum(1).astype(np.uint32)
        ref_idx, _ = oracle_mod.match_keypoints(pa, comp(a["desc"]), pb, comp(b["desc"]), window, maxd)
    else:
        ref_idx, ref_dist = oracle_mod.match256(a["desc"], b["desc"], pa, pb, window, maxd)
        np.testing.assert_array_equal(d_dist.cpu().numpy()[:cnt[0]], ref_dist)
    np.testing.assert_array_equal(d_idx.cpu().numpy()[:cnt[0]], ref_idx)
    assert (ref_idx >= 0).sum() > 5
    if mode == 1 and window < 0:
        # a 3-px shift: most left keypoints find their right counterpart 3 px to the right
        m = ref_idx >= 0
        dx = pb[ref_idx[m], 0] - pa[m, 0]
        assert (np.abs(dx - 3) <= 1).mean() > 0.3
