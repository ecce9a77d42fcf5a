"""Round-5 GPU parity tests (HIP through the C ABI vs the CPU oracle, bit for bit):
  * f1, the staging half (include/orbfe_ingest.h; buildStream.cpp:376-381, :399-406, :462-466, :483-487): frames that go
    host (pinned ring) -> device -> orbfe_extract / orbfe_extract_rgb (+ orbfe_match_batch) -> host must give the records
    the device-resident entry points give and the oracle gives, through ring wrap-around, partial slots, pageable
    sources with a pitch, and the slot state machine's refusals;
  * 8e, the branches of liborbfe_dist.so under `world > 1` (grouped ncclSend / ncclRecv, root placement, exact-length
    packing, the all-reduce), executed on ONE GPU by 2 and 3 ranks over a test-only loopback transport that stands in for
    librccl.so.1 (tests/fake_rccl): worker processes without torch, and examples/multi_gpu_port with both threads on device 0;
  * the stage boundary's holes: SUM_OF_ABS_DIFF_ALL and MAX_THRESHOLD of orbfe_fast_calc_corner_response (fast.cu:233-241,
    :256-283), the f-theta projection (cuda-align.cu:44-50, post_processing.cu:32-38), camera structs passed as DEVICE
    pointers as the reference does (buildStream.cpp:391-393);
  * a11, EXT brute force: BOTH forms of the matrix-core matcher (ORBFE_MATCH=stream: match_expand_kernel + match_mfma_kernel;
    tile: match_tile_kernel, which expands its own operands -- what calls with >= 256 (pair, 512-query tile) items take
    by themselves) through every brute-force matcher test of the earlier rounds, plus a call large enough for the
    size rule to pick the tile form on its own.
The oracle is unpinned by the reference (it holds no tests); see oracle/orbfe_oracle.h."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from orbfe import synth
from test_align_oracle import extr, intr
from test_gpu_parity import FRAMES, bits, dev, pitched, stream

pytestmark = pytest.mark.gpu

EXT = dict(levels=8, cell=8, min_arc=9, max_features=500)


def _resident(torch, orbfe, ctx, frames, rgb, match):
    """The device-resident path on the caller's stream: (records [n, cap], counts [n], idx, dist)."""
    n = frames.shape[0]
    h, w = frames.shape[1], frames.shape[2]
    d_in = dev(torch, frames)
    rec = torch.zeros(n * ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    if rgb:
        ctx.extract_rgb(d_in.data_ptr(), 3 * w, 3 * w * h, n, rec.data_ptr(), cnt.data_ptr(), None, stream(torch))
    else:
        ctx.extract(d_in.data_ptr(), w, w * h, n, rec.data_ptr(), cnt.data_ptr(), None, stream(torch))
    idx = dist = None
    if match and n >= 2:
        idx = torch.zeros((n - 1) * ctx.cap, dtype=torch.int32, device="cuda")
        dist = torch.zeros((n - 1) * ctx.cap, dtype=torch.int32, device="cuda")
        ctx.match_batch(rec.data_ptr(), cnt.data_ptr(), n, 1, -1, 256, idx.data_ptr(), dist.data_ptr(), stream(torch))
    torch.cuda.synchronize()
    out = [rec.cpu().numpy().view(orbfe.KEYPOINT_DTYPE).reshape(n, ctx.cap), cnt.cpu().numpy()]
    out += [idx.cpu().numpy().reshape(n - 1, ctx.cap), dist.cpu().numpy().reshape(n - 1, ctx.cap)] if idx is not None else [None, None]
    return out


def _valid_equal(rec_a, cnt_a, rec_b, cnt_b):
    np.testing.assert_array_equal(cnt_a, cnt_b)
    for f in range(len(cnt_a)):
        assert rec_a[f, :cnt_a[f]].tobytes() == rec_b[f, :cnt_b[f]].tobytes(), "records of frame %d differ" % f


@pytest.mark.parametrize("rgb", [False, True])
def test_ingest_ring_equals_resident_path_and_oracle(gpu, oracle_mod, rgb):
    """Seven batches through a 3-slot ring (every slot is reused, two stay in flight while the third is refilled), the
    last one partial.  Each batch: staged records == device-resident records (valid prefix, counts, matcher outputs)
    and, for the first and the last batch, == the oracle's."""
    torch, orbfe = gpu
    w, h, F, n_batches = 640, 480, 3, 7
    ctx = orbfe.Context(w, h, max_batch=F, **EXT)
    ref_ctx = orbfe.Context(w, h, max_batch=F, **EXT)  # the resident path runs on its own context while slots are in flight
    ing = orbfe.Ingest(ctx, F, slots=3, channels=3 if rgb else 1, match_mode=1, match_window=-1, match_max_distance=256,
                       download_matches=2)
    assert ing.slots == 3 and ing.frame_bytes == w * h * (3 if rgb else 1)
    rng = np.random.default_rng(5)

    def batch(b):
        n = F if b < n_batches - 1 else 2
        g = np.stack([synth.frame(w, h, 300 + 10 * b + i, "rects", **synth.DENSE) for i in range(n)])
        if not rgb:
            return g
        c = np.stack([g, np.roll(g, 1, 2), 255 - g], -1)
        return (c.astype(np.int16) + rng.integers(-2, 3, c.shape)).clip(0, 255).astype(np.uint8)

    batches = [batch(b) for b in range(n_batches)]
    results = {}

    def collect(b):
        rec, cnt, idx, dst = ing.wait(b % 3, batches[b].shape[0])
        results[b] = (rec.copy(), cnt.copy(), None if idx is None else idx.copy(), None if dst is None else dst.copy())
        t = ing.timing(b % 3)
        assert t["upload_bytes"] == batches[b].nbytes and t["upload_ms"] > 0 and t["compute_ms"] > 0 and t["download_ms"] > 0
        assert t["download_bytes"] == batches[b].shape[0] * (ctx.cap * 52 + 4) + 2 * (batches[b].shape[0] - 1) * ctx.cap * 4

    for b in range(n_batches):
        s = b % 3
        if b >= 3:
            collect(b - 3)
        n = batches[b].shape[0]
        ing.host_frames(s)[:n] = batches[b]
        ing.submit(s, n)
    for b in range(max(n_batches - 3, 0), n_batches):
        collect(b)
    assert all(ing.ready(s) for s in range(3))
    ocfg = oracle_mod.make_config(w, h, **EXT)
    for b in range(n_batches):
        n = batches[b].shape[0]
        rec, cnt, idx, dst = results[b]
        r_rec, r_cnt, r_idx, r_dst = _resident(torch, orbfe, ref_ctx, batches[b], rgb, True)
        _valid_equal(rec, cnt, r_rec, r_cnt)
        assert cnt.min() > 100
        for f in range(n - 1):
            np.testing.assert_array_equal(idx[f, :cnt[f]], r_idx[f, :cnt[f]])
            np.testing.assert_array_equal(dst[f, :cnt[f]], r_dst[f, :cnt[f]])
        if b in (0, n_batches - 1):
            refs = []
            for f in range(n):
                gray = oracle_mod.rgb_to_grayscale(batches[b][f]) if rgb else batches[b][f]
                refs.append(oracle_mod.extract_frame(gray, ocfg))
                assert cnt[f] == refs[f]["count"]
                assert rec[f, :cnt[f]].tobytes() == refs[f]["records"].tobytes()
            for f in range(n - 1):
                o_idx, o_dst = oracle_mod.match256(refs[f]["records"]["desc"], refs[f + 1]["records"]["desc"])
                np.testing.assert_array_equal(idx[f, :cnt[f]], o_idx)
                np.testing.assert_array_equal(dst[f, :cnt[f]], o_dst)
    ing.close()
    ctx.close()
    ref_ctx.close()


def test_ingest_pageable_source_with_a_pitch_and_device_side_consumer(gpu, oracle_mod):
    """orbfe_ingest_submit_from: rows at a pitch, frames at a stride, in ordinary memory (what rgbd_frame->rgb_image is
    in the reference, buildStream.cpp:399-406).  A consumer enqueued on the ring's compute stream reads the slot's device
    records (here: a copy of them), as orbfe_keypoint_pixel_to_point would."""
    torch, orbfe = gpu
    w, h, F = 100, 72, 4
    cfg = dict(levels=3, cell=8, min_arc=9, max_features=200)
    ctx = orbfe.Context(w, h, max_batch=F, **cfg)
    ing = orbfe.Ingest(ctx, F, slots=2)
    pitch, stride = w + 12, (w + 12) * h + 40
    raw = np.full(F * stride, 0xEE, np.uint8)
    frames = [synth.frame(w, h, 70 + i, "rects", n_rects=60, min_size=4, max_size=20) for i in range(F)]
    for f in range(F):
        v = raw[f * stride:f * stride + pitch * h].reshape(h, pitch)
        v[:, :w] = frames[f]
    lib = orbfe.lib()
    assert lib.orbfe_ingest_submit_from(ing.handle, 1, F, raw.ctypes.data, pitch, stride) == orbfe.OK
    _, d_rec, d_cnt, _, _ = ing.device_buffers(1)
    # ordered behind the slot's extraction only by being enqueued on the ring's own compute stream
    on_dev = np.zeros((F, ctx.cap), dtype=orbfe.KEYPOINT_DTYPE)
    orbfe.check(lib.orbfe_memcpy_d2h(on_dev.ctypes.data, d_rec, F * ctx.cap * 52, ing.compute_stream()))
    rec, cnt, idx, dst = ing.wait(1)
    assert idx is None and dst is None
    ocfg = oracle_mod.make_config(w, h, **cfg)
    for f in range(F):
        ref = oracle_mod.extract_frame(frames[f], ocfg)
        assert cnt[f] == ref["count"] > 20
        assert rec[f, :cnt[f]].tobytes() == ref["records"].tobytes()
        assert on_dev[f, :cnt[f]].tobytes() == ref["records"].tobytes()
    # bad pitch / overlapping frames are refused before anything is copied
    assert lib.orbfe_ingest_submit_from(ing.handle, 0, F, raw.ctypes.data, w - 1, stride) == orbfe.ERR_INVALID_ARG
    assert lib.orbfe_ingest_submit_from(ing.handle, 0, F, raw.ctypes.data, pitch, pitch) == orbfe.ERR_INVALID_ARG
    assert ing.ready(0)
    ing.close()
    ctx.close()


@pytest.mark.parametrize("rgb", [False, True])
def test_cpp_ingest_port(gpu, oracle_mod, tmp_path, rgb):
    """examples/ingest_port.cpp: the host side of the reference's frame loop (pageable camera frames in, host results out;
    buildStream.cpp:376-381, :399-406, :462-466, :483-487) as a C++ program over include/orbfe_ingest.h only -- 11 frames in
    batches of 3 through 3 pinned slots (a partial last batch, every slot reused).  Its output file must hold the oracle's
    records frame by frame and the brute-force matches of consecutive frames inside a batch."""
    torch, orbfe = gpu
    exe = os.path.join(ROOT, "examples", "ingest_port")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    w, h, n, per = 640, 480, 11, 3
    g = np.stack([synth.frame(w, h, 400 + i, "rects", n_rects=96 if i % 3 else 800, min_size=6, max_size=None if i % 3 else 32) for i in range(n)])
    frames = np.stack([g, np.roll(g, 2, 2), 255 - g], -1) if rgb else g
    fin, fout = str(tmp_path / "frames.bin"), str(tmp_path / "out.bin")
    np.ascontiguousarray(frames).tofile(fin)
    r = subprocess.run([exe, str(w), str(h), str(n), str(per), fin, fout] + (["rgb"] if rgb else []), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=240)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    assert b"4 batches" in r.stdout and b"3 pinned slots" in r.stdout
    raw = np.fromfile(fout, np.uint8)
    nf, cap = raw[:8].view(np.int32)
    assert nf == n and cap == 2000
    counts = raw[8:8 + 4 * n].view(np.int32)
    o = 8 + 4 * n
    rec = raw[o:o + n * cap * 52].view(orbfe.KEYPOINT_DTYPE).reshape(n, cap)
    idx = raw[o + n * cap * 52:].view(np.int32).reshape(n, cap)
    ocfg = oracle_mod.make_config(w, h, levels=8, cell=8, min_arc=9, max_features=2000)
    refs = [oracle_mod.extract_frame(oracle_mod.rgb_to_grayscale(frames[f]) if rgb else frames[f], ocfg) for f in range(n)]
    for f in range(n):
        assert counts[f] == refs[f]["count"] > 100
        assert rec[f, :counts[f]].tobytes() == refs[f]["records"].tobytes()
        if f % per != per - 1 and f != n - 1:  # a pair inside a batch
            want, _ = oracle_mod.match256(refs[f]["records"]["desc"], refs[f + 1]["records"]["desc"])
            np.testing.assert_array_equal(idx[f, :counts[f]], want)
        else:
            assert (idx[f] == -1).all()


def test_ingest_slot_state_machine(gpu):
    """A slot is free or in flight: a second submit without a wait is ORBFE_ERR_CAPACITY (the ring is full), a wait on a
    free slot and a timing query before any pass are ORBFE_ERR_INVALID_ARG, sizes beyond the slot or the context are
    refused at creation / submit."""
    torch, orbfe = gpu
    w, h = 64, 48
    ctx = orbfe.Context(w, h, max_batch=4, levels=1, cell=8, min_arc=9, max_features=0)
    lib = orbfe.lib()
    with pytest.raises(orbfe.OrbfeError) as e:
        orbfe.Ingest(ctx, 5)
    assert e.value.code == orbfe.ERR_CAPACITY
    with pytest.raises(orbfe.OrbfeError) as e:
        orbfe.Ingest(ctx, 2, slots=1)
    assert e.value.code == orbfe.ERR_INVALID_ARG
    with pytest.raises(orbfe.OrbfeError) as e:
        orbfe.Ingest(ctx, 2, download_matches=1)  # nothing to download without a matcher
    assert e.value.code == orbfe.ERR_INVALID_ARG
    ing = orbfe.Ingest(ctx, 2, slots=2)
    assert lib.orbfe_ingest_wait(ing.handle, 0, None, None, None, None) == orbfe.ERR_INVALID_ARG
    assert b"not in flight" in lib.orbfe_ingest_last_error(ing.handle)
    assert lib.orbfe_ingest_timing(ing.handle, 0, None, None, None, None, None) == orbfe.ERR_INVALID_ARG
    assert lib.orbfe_ingest_submit(ing.handle, 0, 3) == orbfe.ERR_CAPACITY
    assert lib.orbfe_ingest_submit(ing.handle, 2, 1) == orbfe.ERR_INVALID_ARG
    ing.host_frames(0)[:] = synth.frames(w, h, 2, first_index=3, kind="uniform")
    ing.submit(0, 2)
    assert lib.orbfe_ingest_submit(ing.handle, 0, 2) == orbfe.ERR_CAPACITY
    assert b"in flight" in lib.orbfe_ingest_last_error(ing.handle)
    rec, cnt, _, _ = ing.wait(0)
    assert cnt.shape == (2,) and (cnt > 0).all()
    assert ing.ready(0)
    ing.submit(0, 1)  # free again
    ing.wait(0, 1)
    ing.close()
    ctx.close()


# ------------------------------------------------------------------ 8e: world > 1 over the loopback transport
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")


def _loopback_env():
    """Environment of a child whose liborbfe_dist.so resolves librccl.so.1 to the loopback transport (built on demand;
    __graft_entry__.build() builds it too).  The product library is the shipped one: only the loader's search path changes."""
    out = os.path.join(FAKE_DIR, "_build", "librccl.so.1")
    src = os.path.join(FAKE_DIR, "fake_rccl.cpp")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", FAKE_DIR])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(FAKE_DIR, "_build") + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


@pytest.mark.parametrize("world", [2, 3])
def test_dist_world_gt1_over_the_loopback_transport(gpu, oracle_mod, tmp_path, world):
    """`world` worker processes (tests/fake_rccl/world2_worker.py), all on device 0, drive every entry point of
    include/orbfe_dist.h whose body sits under `world > 1`; what the root received must be the oracle's records in frame
    order: fixed stride (root extracting in place), exact length (+ the offset table), a non-zero root, the C5
    all-reduce(MAX) flow on every rank, and the host reductions."""
    torch, orbfe = gpu
    sys.path.insert(0, FAKE_DIR)
    import world2_worker as ww
    env = _loopback_env()
    scratch = str(tmp_path)
    procs = [subprocess.Popen([sys.executable, os.path.join(FAKE_DIR, "world2_worker.py"), str(r), str(world), scratch], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
    n_total, n = ww.N_TOTAL, ww.N_TOTAL // world
    ocfg = oracle_mod.make_config(ww.W, ww.H, **ww.CFG)
    refs = [oracle_mod.extract_frame(ww.scene(i), ocfg) for i in range(n_total)]
    ref_counts = np.array([r["count"] for r in refs], np.int32)
    assert ref_counts.min() < 200 < ref_counts.max(), "ragged counts: the exact form must ship fewer bytes than the fixed one"
    for name in ("fixed", "root1"):
        z = np.load(os.path.join(scratch, name + ".npz"))
        cap = int(z["cap"])
        rec = z["records"].view(orbfe.KEYPOINT_DTYPE).reshape(n_total, cap)
        np.testing.assert_array_equal(z["counts"], ref_counts)
        for f in range(n_total):
            assert rec[f, :ref_counts[f]].tobytes() == refs[f]["records"].tobytes(), (name, f)
    z = np.load(os.path.join(scratch, "exact.npz"))
    cap = int(z["cap"])
    np.testing.assert_array_equal(z["counts"], ref_counts)
    rec = z["records"].view(orbfe.KEYPOINT_DTYPE)
    want_off = np.array([(f // n) * n * cap + ref_counts[(f // n) * n:f].sum() for f in range(n_total)], np.int64)
    np.testing.assert_array_equal(z["offsets"], want_off)
    for f in range(n_total):
        got = rec[want_off[f]:want_off[f] + ref_counts[f]]
        assert got.tobytes() == refs[f]["records"].tobytes(), ("exact", f)
    for r in range(1, world):  # the bytes behind a rank's dense block were never shipped: still the root's fill pattern
        used = int(ref_counts[r * n:(r + 1) * n].sum())
        tail = z["records"][52 * (r * n * cap + used):52 * (r + 1) * n * cap]
        assert tail.size > 0 and (tail == 0xEE).all()
    # C5: every rank ends with the full frame's records; the partial keys differ per rank and their maximum is the merge
    o5 = oracle_mod.make_config(ww.C5["width"], ww.C5["height"], **ww.C5["cfg"])
    ref5 = oracle_mod.extract_frame(ww.c5_scene(), o5)
    parts = []
    for r in range(world):
        z = np.load(os.path.join(scratch, "c5_%d.npz" % r))
        assert int(z["count"][0]) == ref5["count"] > 300
        got = z["records"].view(orbfe.KEYPOINT_DTYPE)[:ref5["count"]]
        assert got.tobytes() == ref5["records"].tobytes(), "rank %d" % r
        parts.append(z["partial_keys"])
        np.testing.assert_array_equal(z["merged_keys"], np.load(os.path.join(scratch, "c5_0.npz"))["merged_keys"])
    assert all((parts[0] != parts[r]).any() for r in range(1, world)), "the shards saw different tiles"
    np.testing.assert_array_equal(np.maximum.reduce(parts), np.load(os.path.join(scratch, "c5_0.npz"))["merged_keys"])
    for r in range(world):
        hst = json.load(open(os.path.join(scratch, "host_%d.json" % r)))
        assert hst["max"] == [float(world), 10.0 * (world - 1), -3.5]
        assert hst["sum"] == [world * (world + 1) / 2.0, 10.0 * world * (world - 1) / 2.0, -3.5 * world]
        assert all("fake_rccl/_build" in p for p in hst["librccl"])


@pytest.mark.parametrize("exact", [False, True])
def test_cpp_multi_gpu_port_two_ranks_on_one_device(gpu, oracle_mod, tmp_path, exact):
    """examples/multi_gpu_port (the reference's thread-per-stream model, one C++ thread + context + communicator rank per
    GPU) with TWO ranks, both on device 0 (`devices=0,0`), over the loopback transport: rank 0's gathered file must be the
    oracle's records frame by frame.  With real RCCL the same command needs two GPUs."""
    torch, orbfe = gpu
    exe = os.path.join(ROOT, "examples", "multi_gpu_port")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    w, h, n = 640, 480, 6
    frames = np.stack([synth.frame(w, h, 160 + i, "rects", n_rects=96 if i % 2 else 800, min_size=6, max_size=None if i % 2 else 32)
                       for i in range(n)])
    fin, fout = str(tmp_path / "frames.bin"), str(tmp_path / "out.bin")
    frames.tofile(fin)
    cmd = [exe, "2", str(w), str(h), str(n), fin, fout] + (["exact"] if exact else []) + ["devices=0,0"]
    r = subprocess.run(cmd, env=_loopback_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    assert b"2 GPU(s)" in r.stdout
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


# ------------------------------------------------------------------ stage boundary: the other fast_score values
@pytest.mark.parametrize("score", [0, 2])
@pytest.mark.parametrize("kind,w,h,arc,thr", [("dense", 640, 480, 9, 13.0), ("rects", 848, 480, 12, 13.0),
                                              ("uniform", 100, 72, 10, 7.5), ("checker", 96, 64, 9, 40.0),
                                              ("uniform", 67, 35, 12, 0.0)])
def test_fast_score_modes_stage(gpu, oracle_mod, kind, w, h, arc, thr, score):
    """orbfe_fast_calc_corner_response with the reference's other two score arguments (enum fast_score, fast.cuh:18-23):
    0 = SUM_OF_ABS_DIFF_ALL (fast.cu:233-241), 2 = MAX_THRESHOLD (:256-283, bisection).  Bit patterns against the
    oracle, fractional and zero thresholds included, borders and pitch padding untouched."""
    torch, orbfe = gpu
    img = oracle_mod.gaussian_blur_3x3(FRAMES[kind](w, h)) if kind != "checker" else FRAMES[kind](w, h)
    lut_np = oracle_mod.fast_lut(arc)
    ref = oracle_mod.fast_response(img, lut_np, thr, score=score)
    d_img, d_lut = pitched(torch, img, w + 4), dev(torch, lut_np)
    resp = torch.full((h, w + 8), -3.0, dtype=torch.float32, device="cuda")
    orbfe.check(orbfe.lib().orbfe_fast_calc_corner_response(w, h, w + 4, d_img.data_ptr(), 3, 3, d_lut.data_ptr(), thr, arc, score,
                                                            w + 8, resp.data_ptr(), stream(torch)))
    got = resp.cpu().numpy()
    np.testing.assert_array_equal(bits(got[:, :w]), bits(ref))
    assert (got[:, w:] == -3.0).all()
    if kind in ("dense", "uniform"):
        assert (ref > 0).sum() > 10
    if score == 2 and (ref > 0).any():
        assert ref[ref > 0].min() >= np.floor(thr)  # the bisection ends at (an integer >= threshold + 1) - 1


def test_fast_score_argument_is_validated(gpu):
    torch, orbfe = gpu
    one = torch.zeros(65536, dtype=torch.uint8, device="cuda")
    resp = torch.zeros(64 * 64, dtype=torch.float32, device="cuda")
    call = lambda score, thr: orbfe.lib().orbfe_fast_calc_corner_response(64, 64, 64, one.data_ptr(), 3, 3, one.data_ptr(), thr, 9,
                                                                          score, 64, resp.data_ptr(), stream(torch))
    assert call(3, 13.0) == orbfe.ERR_INVALID_ARG and call(-1, 13.0) == orbfe.ERR_INVALID_ARG
    assert call(2, float("nan")) == orbfe.ERR_INVALID_ARG and call(2, float("inf")) == orbfe.ERR_INVALID_ARG
    assert call(2, 13.0) == orbfe.OK and call(0, 13.0) == orbfe.OK


# ------------------------------------------------------------------ f-theta (rs2 distortion model 3) on the projecting side
FTHETA = [0.92, 0.0, 0.0, 0.0, 0.0]  # coeffs[0]: the lens' field-of-view parameter, radians


@pytest.mark.parametrize("size", [(848, 480, 848, 480), (101, 67, 80, 60), (424, 240, 848, 480)])
@pytest.mark.parametrize("kind", ["d435", "distorted"])
def test_align_depth_ftheta_other_camera(gpu, oracle_mod, size, kind):
    """orbfe_align_depth_to_other / _batch with other.model = 3 (RS2_DISTORTION_FTHETA, cuda-align.cu:44-50), bit for bit
    against the oracle ("distorted": the depth side carries the inverse Brown-Conrady polynomial at the same time)."""
    torch, orbfe = gpu
    dw, dh, ow, oh = size
    d, o, e, scale = synth.rig(kind, dw, dh, ow, oh)
    o = list(o)
    o[6], o[7] = 3, FTHETA
    depth = synth.depth_frames(dw, dh, 3, first_index=40 + dw)
    iw, ih = max(dw, ow), max(dh, oh)
    want = [oracle_mod.align_depth_to_other(depth[f], scale, iw, ih, intr(oracle_mod, d), intr(oracle_mod, o), extr(oracle_mod, e))[0]
            for f in range(3)]
    d_depth = dev(torch, depth.view(np.int16))
    d_out = torch.full((3, oh, ow), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    orbfe.check(orbfe.lib().orbfe_align_depth_to_other(d_out[0].data_ptr(), d_depth[0].data_ptr(), None, scale, iw, ih,
                                                       C.byref(intr(orbfe, d)), C.byref(intr(orbfe, o)), C.byref(extr(orbfe, e)),
                                                       stream(torch)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d_out[0].cpu().numpy().view(np.uint32), want[0])
    assert (want[0] != 0).mean() > 0.2
    d_out.fill_(0x5A5A5A5A)
    orbfe.check(orbfe.lib().orbfe_align_depth_batch(d_out.data_ptr(), ow * oh, d_depth.data_ptr(), dw * dh, 3, scale,
                                                    C.byref(intr(orbfe, d)), C.byref(intr(orbfe, o)), C.byref(extr(orbfe, e)),
                                                    stream(torch)))
    torch.cuda.synchronize()
    for f in range(3):
        np.testing.assert_array_equal(d_out[f].cpu().numpy().view(np.uint32), want[f])


def test_reproject_points_ftheta(gpu, oracle_mod):
    """orbfe_reproject_points with model 3 (post_processing.cu:32-38), the origin (r = 0: 0 / 0 as in the reference) included."""
    torch, orbfe = gpu
    rng = np.random.default_rng(33)
    n = 1000
    pts = np.stack([rng.normal(size=n) * 400, rng.normal(size=n) * 300, rng.uniform(300, 5000, n)], 1)
    pts[0] = (0.0, 0.0, 1000.0)
    T = np.eye(4)
    T[:3, 3] = (5.0, -2.0, 11.0)
    T[0, 3] = 0.0
    T[1, 3] = 0.0
    k = (848, 480, 421.5, 237.25, 615.5, 615.25, 3, FTHETA)
    d_pts = dev(torch, pts)
    out = torch.full((n, 2), -1.0, dtype=torch.float32, device="cuda")
    Tc = (C.c_double * 16)(*np.ascontiguousarray(T.T).reshape(-1))
    orbfe.check(orbfe.lib().orbfe_reproject_points(out.data_ptr(), d_pts.data_ptr(), n, Tc, C.byref(intr(orbfe, k)), stream(torch)))
    ref = oracle_mod.reproject_points(pts, T, intr(oracle_mod, k))
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    assert np.isnan(ref[0]).all() and np.isfinite(ref[1:]).all() and ref[1:, 0].std() > 10
    # against libm in float64: a few ulp of float
    x, y = pts[1:, 0] / (pts[1:, 2] + 11.0), pts[1:, 1] / (pts[1:, 2] + 11.0)
    r = np.sqrt(x * x + y * y)
    rd = np.arctan(2 * r * np.tan(FTHETA[0] / 2)) / FTHETA[0]
    np.testing.assert_allclose(ref[1:, 0], x * rd / r * k[4] + k[2], rtol=0, atol=2e-3)
    np.testing.assert_allclose(ref[1:, 1], y * rd / r * k[5] + k[3], rtol=0, atol=2e-3)


# ------------------------------------------------------------------ camera structs handed over as DEVICE pointers
def _on_device(torch, struct):
    return torch.from_numpy(np.frombuffer(bytes(struct), dtype=np.uint8).copy()).cuda()


def test_camera_structs_as_device_pointers(gpu, oracle_mod):
    """The reference passes DEVICE copies of rs2_intrinsics / rs2_extrinsics (_d_depth_intrinsics, _d_rgb_intrinsics,
    _d_depth_rgb_extrinsics: SlamGpuPipeline.cpp:53-55, buildStream.cpp:391-393, :469, post_processing.cuh:45).  The three
    entry points that take them accept either kind of pointer: same results as with host structs, i.e. the oracle's."""
    torch, orbfe = gpu
    L = orbfe.lib()
    dw, dh = 424, 240
    d, o, e, scale = synth.rig("d435", dw, dh)
    depth = synth.depth_frame(dw, dh, 77)
    want, _ = oracle_mod.align_depth_to_other(depth, scale, dw, dh, intr(oracle_mod, d), intr(oracle_mod, o), extr(oracle_mod, e))
    d_di, d_oi, d_ex = _on_device(torch, intr(orbfe, d)), _on_device(torch, intr(orbfe, o)), _on_device(torch, extr(orbfe, e))
    d_depth = dev(torch, depth.view(np.int16))
    d_al = torch.zeros((dh, dw), dtype=torch.int32, device="cuda")
    as_i = lambda t: C.cast(C.c_void_p(t.data_ptr()), C.POINTER(orbfe.Intrinsics))
    as_e = lambda t: C.cast(C.c_void_p(t.data_ptr()), C.POINTER(orbfe.Extrinsics))
    orbfe.check(L.orbfe_align_depth_to_other(d_al.data_ptr(), d_depth.data_ptr(), None, scale, dw, dh, as_i(d_di), as_i(d_oi),
                                             as_e(d_ex), stream(torch)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d_al.cpu().numpy().view(np.uint32), want)
    d_al2 = torch.zeros((2, dh, dw), dtype=torch.int32, device="cuda")
    d_depth2 = torch.stack([d_depth, d_depth])
    orbfe.check(L.orbfe_align_depth_batch(d_al2.data_ptr(), dw * dh, d_depth2.data_ptr(), dw * dh, 2, scale, as_i(d_di), as_i(d_oi),
                                          as_e(d_ex), stream(torch)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d_al2[1].cpu().numpy().view(np.uint32), want)
    # keypoint_pixel_to_point with the device intrinsics on that aligned depth
    rng = np.random.default_rng(8)
    n = 300
    pos = np.stack([rng.integers(0, dw, n), rng.integers(0, dh, n)], 1).astype(np.float32)
    score = rng.choice([0.0, 2.0, 57.0], n).astype(np.float32)
    desc = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    _, ref_pts, _, ref_n = oracle_mod.keypoint_pixel_to_point(want, intr(oracle_mod, o), pos, score, desc, fix_depth_index=1)
    d_pos, d_score, d_desc = dev(torch, pos), dev(torch, score), dev(torch, desc.view(np.int32))
    o_pos = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
    o_pts = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    o_desc = torch.zeros(n, dtype=torch.int32, device="cuda")
    o_n = torch.zeros(1, dtype=torch.int32, device="cuda")
    orbfe.check(L.orbfe_keypoint_pixel_to_point(d_al.data_ptr(), as_i(d_oi), dw, dh, o_pos.data_ptr(), d_pos.data_ptr(),
                                                d_score.data_ptr(), o_pts.data_ptr(), o_desc.data_ptr(), d_desc.data_ptr(), n,
                                                o_n.data_ptr(), 1, stream(torch)))
    m = int(o_n.cpu()[0])
    assert m == ref_n > 20
    np.testing.assert_array_equal(o_pts.cpu().numpy()[:m].view(np.uint64), ref_pts.view(np.uint64))
    # reproject_points with the device intrinsics
    out = torch.zeros((m, 2), dtype=torch.float32, device="cuda")
    T = np.eye(4)
    Tc = (C.c_double * 16)(*T.T.reshape(-1))
    orbfe.check(L.orbfe_reproject_points(out.data_ptr(), o_pts.data_ptr(), m, Tc, as_i(d_oi), stream(torch)))
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), oracle_mod.reproject_points(ref_pts, T, intr(oracle_mod, o)).view(np.uint32))


# ------------------------------------------------------------------ a11 EXT: both forms of the matrix-core matcher
@pytest.mark.parametrize("form", ["stream", "tile"])
def test_matrix_core_matcher_forms_through_the_earlier_rounds_tests(gpu, oracle_mod, monkeypatch, form):
    """ORBFE_MATCH (read when a context is created) forces one form of the 256-bit brute-force matcher.  Every brute-force
    test of rounds 1-2 is run on each: extraction-made records with identical frames (ties to the lower index), crafted
    records (random, duplicated, all-zero, all-one descriptors; counts 0 and 1; caps that are not multiples of 16; 38 row
    blocks; three distance limits), ragged counts (every tail of the candidate ring), 16384 keypoints per frame (the
    14-bit index), strided pair lists.  The default (by call size) picks the stream forms for all of these calls."""
    import test_gpu_parity as t1
    import test_gpu_round2 as t2
    torch, orbfe = gpu
    monkeypatch.setenv("ORBFE_MATCH", form)
    probe = orbfe.Context(640, 480, cell=8, min_arc=9, max_features=2000, max_batch=4)
    want = "match_tile_kernel" if form == "tile" else "match_expand_kernel+match_mfma_kernel"
    assert probe.dispatch_info(4, 1, -1)["match"] == want
    probe.close()
    t1.test_match_batch(gpu, oracle_mod, 1, -1, 256)
    for w, h, kw in [(100, 70, dict(cell=8)), (640, 480, dict(cell=32)), (640, 480, dict(cell=8, max_features=2000)), (640, 480, dict(cell=8))]:
        for maxd in (256, 90, 0):
            t1.test_match_batch_256_crafted_records(gpu, oracle_mod, w, h, kw, maxd)
    for w, h, kw in [(100, 70, dict(cell=8)), (640, 480, dict(cell=16)), (640, 480, dict(cell=8, max_features=2000))]:
        t1.test_match_batch_256_ragged_counts_fuzz(gpu, oracle_mod, w, h, kw)
    t1.test_match_batch_256_at_the_key_packing_limit(gpu, oracle_mod, 16384, "matrix cores, index uses all 14 bits")
    for first, stride in [(0, 2), (1, 2), (0, 3), (2, 1)]:
        t2.test_match_pairs_strided(gpu, oracle_mod, 1, -1, 256, first, stride)


def test_tile_matcher_is_what_a_large_call_runs(gpu, oracle_mod):
    """600 frames x 500 keypoints: 599 (pair, tile) items >= 256, so the size rule itself takes match_tile_kernel; sampled
    pairs against the oracle, the whole call against the stream form (forced on a second context)."""
    torch, orbfe = gpu
    B, n = 600, 500
    rng = np.random.default_rng(77)
    ctx = orbfe.Context(640, 480, cell=16, min_arc=9, max_features=n, max_batch=B)
    assert ctx.cap == n and ctx.dispatch_info(B, 1, -1)["match"] == "match_tile_kernel"
    pool = rng.integers(0, 256, (300, 32), dtype=np.uint8)  # a small pool: exact ties are frequent
    rec = np.zeros((B, n), dtype=orbfe.KEYPOINT_DTYPE)
    rec["desc"] = pool[rng.integers(0, 300, (B, n))]
    flip = rng.random((B, n, 32)) < 0.02
    rec["desc"] ^= (flip * rng.integers(1, 256, (B, n, 32))).astype(np.uint8)
    rec["score"] = 20
    cnt = rng.integers(0, n + 1, B).astype(np.int32)
    cnt[:8] = [n, 0, n, 1, 17, 16, 15, n]
    d_rec, d_cnt = dev(torch, rec.view(np.uint8).reshape(-1)), dev(torch, cnt)
    out = {}
    for form in ("auto", "stream"):
        c = ctx
        if form == "stream":
            import os
            os.environ["ORBFE_MATCH"] = "stream"
            try:
                c = orbfe.Context(640, 480, cell=16, min_arc=9, max_features=n, max_batch=B)
            finally:
                del os.environ["ORBFE_MATCH"]
            assert c.dispatch_info(B, 1, -1)["match"] == "match_expand_kernel+match_mfma_kernel"
        d_idx = torch.full(((B - 1) * n,), -7, dtype=torch.int32, device="cuda")
        d_dst = torch.full(((B - 1) * n,), -7, dtype=torch.int32, device="cuda")
        c.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), B, 1, -1, 100, d_idx.data_ptr(), d_dst.data_ptr(), stream(torch))
        torch.cuda.synchronize()
        out[form] = (d_idx.cpu().numpy().reshape(B - 1, n), d_dst.cpu().numpy().reshape(B - 1, n))
    np.testing.assert_array_equal(out["auto"][0], out["stream"][0])
    np.testing.assert_array_equal(out["auto"][1], out["stream"][1])
    idx, dst = out["auto"]
    for p in list(range(8)) + [100, 333, 598]:
        A, Bf = rec[p, :cnt[p]], rec[p + 1, :cnt[p + 1]]
        ref_idx, ref_dst = oracle_mod.match256(A["desc"], Bf["desc"], None, None, -1, 100)
        np.testing.assert_array_equal(idx[p, :cnt[p]], ref_idx)
        np.testing.assert_array_equal(dst[p, :cnt[p]], ref_dst)
        assert (idx[p, cnt[p]:] == -1).all()


def test_tile_matcher_fuzz(gpu, oracle_mod, monkeypatch):
    """Random record capacities (not multiples of 16 or 512, 1 .. ~5000), random counts (0, 1, 15, 16, 17, cap), a small
    descriptor pool (exact ties: the lower index must win), random distance limits -- match_tile_kernel forced, against the
    oracle.  ORBFE_FUZZ_TRIALS / ORBFE_FUZZ_SEED as in the other fuzzes."""
    torch, orbfe = gpu
    monkeypatch.setenv("ORBFE_MATCH", "tile")
    trials = int(os.environ.get("ORBFE_FUZZ_TRIALS", "10"))
    rng = np.random.default_rng(int(os.environ.get("ORBFE_FUZZ_SEED", "505")))
    for trial in range(trials):
        cell = int(rng.choice([8, 16, 32]))
        w, h = int(rng.integers(5, 40)) * cell + int(rng.integers(0, cell)), int(rng.integers(4, 30)) * cell + int(rng.integers(0, cell))
        K = ((w + cell - 1) // cell) * ((h + cell - 1) // cell)
        maxf = int(rng.choice([0, 0, max(1, K // 3), max(1, K - 1)]))
        n = int(rng.integers(2, 6))
        ctx = orbfe.Context(w, h, cell=cell, min_arc=9, max_features=maxf, max_batch=n)
        cap = ctx.cap
        assert ctx.dispatch_info(n, 1, -1)["match"] == "match_tile_kernel"
        pool = rng.integers(0, 256, (max(4, cap // 5), 32), dtype=np.uint8)
        pool[0], pool[1] = 0, 255
        rec = np.zeros((n, cap), dtype=orbfe.KEYPOINT_DTYPE)
        rec["desc"] = pool[rng.integers(0, len(pool), (n, cap))]
        rec["desc"] ^= ((rng.random((n, cap, 32)) < 0.03) * rng.integers(1, 256, (n, cap, 32))).astype(np.uint8)
        cnt = np.array([int(rng.choice([0, 1, min(15, cap), min(16, cap), min(17, cap), cap, int(rng.integers(0, cap + 1))])) for _ in range(n)], np.int32)
        maxd = int(rng.choice([256, 0, 40, 128]))
        d_rec, d_cnt = dev(torch, rec.view(np.uint8).reshape(-1)), dev(torch, cnt)
        d_idx = torch.full(((n - 1) * cap,), -7, dtype=torch.int32, device="cuda")
        d_dst = torch.full(((n - 1) * cap,), -7, dtype=torch.int32, device="cuda")
        ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), n, 1, -1, maxd, d_idx.data_ptr(), d_dst.data_ptr(), stream(torch))
        idx, dst = d_idx.cpu().numpy().reshape(n - 1, cap), d_dst.cpu().numpy().reshape(n - 1, cap)
        for p in range(n - 1):
            ri, rd = oracle_mod.match256(rec[p, :cnt[p]]["desc"], rec[p + 1, :cnt[p + 1]]["desc"], None, None, -1, maxd)
            np.testing.assert_array_equal(idx[p, :cnt[p]], ri, err_msg="trial %d pair %d cap %d counts %s" % (trial, p, cap, cnt))
            np.testing.assert_array_equal(dst[p, :cnt[p]], rd)
            assert (idx[p, cnt[p]:] == -1).all() and (dst[p, cnt[p]:] == -1).all()
        ctx.close()


# ------------------------------------------------------------------ a5-a7: detect_tile_kernel's tile groups
@pytest.mark.parametrize("groups", ["multi", "single"])
def test_detect_tile_groups_through_the_earlier_rounds_tests(gpu, oracle_mod, monkeypatch, groups):
    """Launches of many rounds of workgroups run detect_tile_kernel with four consecutive tile-list entries per workgroup (the
    next tile's pixels prefetched under the current tile's phases); everything the tests launch is far too small for the size
    rule to pick that form, so ORBFE_DETECT_GROUPS (read when a context is created) forces it -- and its opposite -- through
    the extraction tests of round 1: every reference-mode size (tile lists of 1 .. 2700 entries, most not multiples of four,
    levels mixed inside a group), the EXT modes, RGB input, tile-sharded detection (tile_step 3 and 8: a group's entries are
    then 3 or 8 apart in the list) and the fuzz over random configurations."""
    import test_gpu_parity as t1
    torch, orbfe = gpu
    monkeypatch.setenv("ORBFE_DETECT_GROUPS", groups)
    probe = orbfe.Context(640, 480, levels=8, cell=8, min_arc=9, max_features=2000, max_batch=4)
    assert ("groups of 4 tiles" in probe.dispatch_info(4, 1, -1)["detect"]) == (groups == "multi")
    probe.close()
    for w, h, levels in [(640, 480, 1), (640, 480, 6), (100, 70, 3), (640, 480, 10), (1280, 720, 9), (636, 476, 2), (130, 258, 8)]:
        t1.test_extract_reference_mode(gpu, oracle_mod, w, h, levels)
    for cfg in [dict(levels=8, cell=8, min_arc=9, max_features=2000), dict(levels=8, cell=16, min_arc=10, max_features=300, fast_threshold=20),
                dict(levels=7, cell=64, min_arc=12, max_features=0), dict(levels=4, cell=32, min_arc=11, max_features=17, angle_in_radians=1)]:
        t1.test_extract_ext_modes(gpu, oracle_mod, cfg)
    t1.test_extract_rgb_fused(gpu, oracle_mod, 640, 480, 6)
    t1.test_tile_sharded_detection_merges_exactly(gpu, oracle_mod, 3840, 2160, 12, 8)
    t1.test_tile_sharded_detection_merges_exactly(gpu, oracle_mod, 640, 480, 6, 3)
    t1.test_fuzz_random_configurations(gpu, oracle_mod)


def test_detect_tile_groups_are_what_a_large_launch_runs(gpu, oracle_mod):
    """The size rule itself: 4096 frames of the bench configuration dispatch to the grouped form, 256 do not (no compute here
    beyond the context: the 4096-frame step is bench.py's, and the soak's c2 case checks its results)."""
    torch, orbfe = gpu
    ctx = orbfe.Context(640, 480, levels=8, cell=8, min_arc=9, max_features=2000, max_batch=2048)
    assert "groups of 4 tiles" in ctx.dispatch_info(2048, 1, -1)["detect"]
    assert "groups" not in ctx.dispatch_info(256, 1, -1)["detect"]
    ctx.close()


def test_a_bench_sized_step_against_the_oracle(gpu, oracle_mod):
    """One extract + match over 2048 frames of the bench configuration -- the launch sizes at which the size rules pick detect's
    tile groups and the tile matcher, which nothing else in the suite reaches by itself.  The batch is eight distinct frames
    (dense scenes, a survey scene, noise, a constant frame) repeated in a fixed order, so eight oracle extractions and eight
    oracle matches check every one of the 2048 record blocks and 2047 match lists, byte for byte."""
    torch, orbfe = gpu
    w, h, B, n = 640, 480, 2048, 2000
    cfg = dict(levels=8, cell=8, min_arc=9, max_features=n)
    base = [synth.frame(w, h, 900 + i, "rects", **synth.DENSE) for i in range(5)]
    base += [synth.frame(w, h, 906, "rects", n_rects=96, min_size=6), FRAMES["uniform"](w, h), FRAMES["const"](w, h)]
    base = np.stack(base)
    ctx = orbfe.Context(w, h, max_batch=B, **cfg)
    info = ctx.dispatch_info(B, 1, -1)
    assert "groups of 4 tiles" in info["detect"] and info["match"] == "match_tile_kernel"
    d_in = dev(torch, base)[torch.arange(B, device="cuda") % 8].contiguous()  # frame f = base[f % 8]
    rec = torch.zeros(B * n * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    idx = torch.full(((B - 1) * n,), -7, dtype=torch.int32, device="cuda")
    dst = torch.full(((B - 1) * n,), -7, dtype=torch.int32, device="cuda")
    s = stream(torch)
    ctx.extract(d_in.data_ptr(), w, w * h, B, rec.data_ptr(), cnt.data_ptr(), None, s)
    ctx.match_batch(rec.data_ptr(), cnt.data_ptr(), B, 1, -1, 64, idx.data_ptr(), dst.data_ptr(), s)
    torch.cuda.synchronize()
    ocfg = oracle_mod.make_config(w, h, levels=8, cell=8, fast_threshold=13.0, min_arc=9, max_features=n,
                                  angle_in_radians=0, descriptor_level=0)
    refs = [oracle_mod.extract_frame(base[i], ocfg) for i in range(8)]
    counts = cnt.cpu().numpy()
    records = rec.cpu().numpy().reshape(B, n * 52)
    for i in range(8):
        want = np.zeros(n * 52, dtype=np.uint8)
        rb = np.frombuffer(refs[i]["records"].tobytes(), dtype=np.uint8)
        want[:rb.size] = rb
        assert (counts[i::8] == refs[i]["count"]).all(), "counts of the copies of frame %d" % i
        np.testing.assert_array_equal(records[i::8][:, :rb.size], np.broadcast_to(rb, (len(records[i::8]), rb.size)),
                                      err_msg="records of the copies of frame %d" % i)
    assert counts[7] == 0 and counts[0] == n
    got_i, got_d = idx.cpu().numpy().reshape(B - 1, n), dst.cpu().numpy().reshape(B - 1, n)
    for i in range(8):  # pair (f, f + 1) with f % 8 = i
        a, b = refs[i], refs[(i + 1) % 8]
        if a["count"] and b["count"]:
            ri, rd = oracle_mod.match256(a["records"]["desc"], b["records"]["desc"], window=-1, max_dist=64)
        else:
            ri = rd = np.full(a["count"], -1, np.int32)
        k = a["count"]
        np.testing.assert_array_equal(got_i[i::8][:, :k], np.broadcast_to(ri, (len(got_i[i::8]), k)), err_msg="match indices, pair class %d" % i)
        np.testing.assert_array_equal(got_d[i::8][:, :k], np.broadcast_to(rd, (len(got_d[i::8]), k)), err_msg="match distances, pair class %d" % i)
        assert (got_i[i::8][:, k:] == -1).all()
