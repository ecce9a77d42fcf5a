"""Round-5 GPU parity tests (HIP through the C ABI vs the CPU oracle, bit for bit):
  * f1, the staging half (include/orbfe_ingest.h; buildStream.cpp:376-381, :399-406, :462-466, :483-487): frames that go
    host (pinned ring) -> device -> orbfe_extract / orbfe_extract_rgb (+ orbfe_match_batch) -> host must give the records
    the device-resident entry points give and the oracle gives, through ring wrap-around, partial slots, pageable
    sources with a pitch, and the slot state machine's refusals;
  * 8e, the branches of liborbfe_dist.so under `world > 1` (grouped ncclSend / ncclRecv, root placement, exact-length
    packing, the all-reduce), executed on ONE GPU by 2 and 3 ranks over a test-only loopback transport that stands in for
    librccl.so.1 (tests/fake_rccl): worker processes without torch, and examples/multi_gpu_port with both threads on device 0.
The oracle is unpinned by the reference (it holds no tests); see oracle/orbfe_oracle.h."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from orbfe import synth
from test_gpu_parity import dev, stream

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
