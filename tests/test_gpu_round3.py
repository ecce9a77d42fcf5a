"""Round-3 GPU parity tests (HIP through the C ABI vs the CPU oracle, bit for bit):
  * orbfe_detect treats the corner table as the opaque buffer the reference defines (fast.cuh:25-26, :42-48):
    overwritten, re-used and arbitrary tables (VERDICT r2 item 1 -- the process-global pointer -> arc registry is gone);
  * EXT iv descriptor_level: orientation + descriptor on the keypoint's own pyramid level;
  * frames with more than 65534 records (the 16-bit cell -> slot map).
The oracle is unpinned by the reference (it holds no tests); see oracle/orbfe_oracle.h."""
import numpy as np
import pytest

from orbfe import synth
from test_gpu_parity import _check_extract, _mixed_frames, _run_extract, dev, stream
from test_gpu_round2 import _detect_case

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ the corner table is opaque
def _built_then_overwritten(torch, orbfe, built_arc):
    """A buffer orbfe_fast_calculate_lut filled for `built_arc`, then overwritten by a plain device copy."""
    def make(lut_np):
        d_lut = torch.zeros(65536, dtype=torch.uint8, device="cuda")
        orbfe.check(orbfe.lib().orbfe_fast_calculate_lut(d_lut.data_ptr(), built_arc, stream(torch)))
        torch.cuda.synchronize()
        d_lut.copy_(torch.from_numpy(lut_np).cuda())  # hipMemcpy D2D: nothing tells the library
        return d_lut
    return make


@pytest.mark.parametrize("built,used", [(12, 9), (9, 12), (12, 10), (10, 11)])
def test_detect_reads_the_table_not_its_history(gpu, oracle_mod, built, used):
    """Build an arc-`built` table with orbfe_fast_calculate_lut, overwrite the SAME buffer with the arc-`used` table:
    orbfe_detect (fused path: integer threshold, aligned levels) must give the oracle's arc-`used` grid."""
    torch, orbfe = gpu
    n = _detect_case(torch, orbfe, oracle_mod, 640, 480, 4, used, 13.0, True, 0,
                     make_d_lut=_built_then_overwritten(torch, orbfe, built))
    assert n > 10


def test_detect_with_a_caller_filled_table_at_a_reused_address(gpu, oracle_mod):
    """Free a buffer the library built a table in, get the same address back from the allocator, fill it from the
    host with another table: the result must follow the contents (the registry of round 2 kept the stale arc)."""
    torch, orbfe = gpu
    seen = {}

    def make(lut_np):
        a = torch.zeros(65536, dtype=torch.uint8, device="cuda")
        orbfe.check(orbfe.lib().orbfe_fast_calculate_lut(a.data_ptr(), 12, stream(torch)))
        torch.cuda.synchronize()
        seen["old"] = a.data_ptr()
        del a
        b = torch.empty(65536, dtype=torch.uint8, device="cuda")  # the caching allocator hands the block back
        seen["new"] = b.data_ptr()
        b.copy_(torch.from_numpy(lut_np))
        return b

    n = _detect_case(torch, orbfe, oracle_mod, 640, 480, 3, 9, 13.0, True, 0, make_d_lut=make)
    assert n > 10
    assert seen["old"] == seen["new"], "the allocator did not reuse the address: the case is not exercised"


@pytest.mark.parametrize("kind", ["arc7", "random", "checker_masks", "all", "none"])
def test_detect_with_an_arbitrary_table(gpu, oracle_mod, kind):
    """The reference's kernel accepts whatever the table says after its own prechecks (fast.cu:98-124, :230-231).
    Tables that are no arc table at all -- arc 7 (the prechecks then DO reject corners: both sides must agree),
    random bits, every mask, no mask."""
    torch, orbfe = gpu
    rng = np.random.default_rng(5)
    lut = {"arc7": lambda: oracle_mod.fast_lut(7),
           "random": lambda: (rng.random(65536) < 0.05).astype(np.uint8) * rng.integers(1, 255, 65536).astype(np.uint8),
           "checker_masks": lambda: (np.arange(65536) % 3 == 0).astype(np.uint8),
           "all": lambda: np.ones(65536, np.uint8),
           "none": lambda: np.zeros(65536, np.uint8)}[kind]()
    n = _detect_case(torch, orbfe, oracle_mod, 424, 240, 3, 0, 13.0, False, 0, lut_np=lut)
    assert (n > 10) == (kind != "none")


# ------------------------------------------------------------------ EXT iv: descriptor_level
DL_CONFIGS = [
    (640, 480, dict(levels=8, cell=8, min_arc=9, max_features=2000)),       # C2
    (848, 480, dict(levels=8, cell=8, min_arc=9, max_features=2000)),       # C3 (odd level widths from level 4 on)
    (1280, 720, dict(levels=8, cell=8, min_arc=9, max_features=2000)),      # C4
    (640, 480, dict(levels=6)),                                             # the reference's regime, 6 levels
    (640, 480, dict(levels=8, cell=16, min_arc=10, max_features=300, fast_threshold=20)),
    (640, 480, dict(levels=7, cell=64, min_arc=12)),
    (640, 480, dict(levels=8, cell=8, min_arc=9, max_features=0, angle_in_radians=1)),
    (333, 97, dict(levels=5, cell=8, min_arc=9)),                           # unfused pyramid kernels, ragged tiles
]


@pytest.mark.parametrize("which", ["auto", "patch", "tile"])
@pytest.mark.parametrize("w,h,cfg", DL_CONFIGS)
def test_descriptor_level_matches_the_oracle(gpu, oracle_mod, monkeypatch, which, w, h, cfg):
    """descriptor_level = 1: angle and descriptor of a keypoint come from the pyramid level that won its cell
    (at position >> level).  Both describe kernels, forced and as the library picks them."""
    torch, orbfe = gpu
    if which != "auto":
        monkeypatch.setenv("ORBFE_DESCRIBE", which)
    cfg = dict(cfg, descriptor_level=1)
    frames = _mixed_frames(w, h)
    ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, **cfg)
    total = _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg)
    assert total > 50
    # it is a different result from level-0 description exactly for the keypoints of coarser levels
    cfg0 = dict(cfg, descriptor_level=0)
    ctx0, rec0, cnt0, _ = _run_extract(torch, orbfe, frames, want_soa=False, **cfg0)
    assert (cnt == cnt0).all()
    coarse_differs = 0
    for f in range(len(frames)):
        a, b = rec[f, :cnt[f]], rec0[f, :cnt[f]]
        for k in ("x", "y", "score", "level"):
            assert (a[k] == b[k]).all(), "detection must not depend on descriptor_level"
        l0 = a["level"] == 0
        assert a[l0].tobytes() == b[l0].tobytes(), "level-0 keypoints are described as before"
        coarse_differs += int((a["desc"][~l0] != b["desc"][~l0]).any(axis=1).sum())
    if cfg.get("levels", 1) > 1:
        assert coarse_differs > 5, "coarse-level keypoints must get their own level's descriptor"


def test_descriptor_level_at_4k_and_many_levels(gpu, oracle_mod):
    """C5's shape: 3840x2160, 12 levels built, 16-px cells (5 detection levels), 8000 features."""
    torch, orbfe = gpu
    w, h = 3840, 2160
    cfg = dict(levels=12, cell=16, min_arc=9, max_features=8000, descriptor_level=1)
    kw = dict(n_rects=800 * (w * h) // (640 * 480), min_size=6, max_size=32)
    frames = np.stack([synth.frame(w, h, 41, "rects", **kw)])
    ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, want_soa=False, **cfg)
    total = _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg)
    assert total == 8000
    assert len(np.unique(rec[0, :cnt[0]]["level"])) >= 3


def _smooth_field(w, h, seed, factor):
    """Uniform noise at 1 / factor of the resolution, bilinearly enlarged: weak corners on level 0, strong ones on the
    coarse levels, so that the coarse levels win most cells (rectangles and pixel noise make level 0 win nearly all)."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    small = rng.integers(0, 256, (h // factor + 2, w // factor + 2)).astype(np.float32)
    return np.clip(ndimage.zoom(small, factor, order=1)[:h, :w], 0, 255).astype(np.uint8)


@pytest.mark.parametrize("which", ["tile", "patch"])
def test_descriptor_level_dense_coarse_tiles_take_several_passes(gpu, oracle_mod, monkeypatch, which):
    """With 8-px cells a level-1 tile spans 256 cells, a level-2 tile 1024 and a level-3 tile up to 4096, so the tile
    kernel's keypoint list (64 entries) is gathered in several passes when a coarse level wins many cells."""
    torch, orbfe = gpu
    monkeypatch.setenv("ORBFE_DESCRIBE", which)
    w, h = 640, 480
    cfg = dict(levels=5, cell=8, min_arc=9, max_features=0, fast_threshold=5, descriptor_level=1)
    frames = np.stack([_smooth_field(w, h, 7, 8), _smooth_field(w, h, 8, 4), synth.frame(w, h, 77, "uniform")])
    ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, **cfg)
    total = _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg)
    per_level = np.bincount(rec[0, :cnt[0]]["level"], minlength=4)
    # 6 level-2 tiles and 2 level-3 tiles per frame: more than 64 keypoints per tile on each coarse level
    assert total > 10000 and per_level[1] > 1500 and per_level[2] > 6 * 64 and per_level[3] > 2 * 64, per_level


def test_fuzz_descriptor_level(gpu, oracle_mod):
    """Seeded random geometries with descriptor_level = 1 (the same draw as test_fuzz_random_configurations)."""
    torch, orbfe = gpu
    import os
    from test_gpu_parity import FRAMES
    rng = np.random.default_rng(int(os.environ.get("ORBFE_FUZZ_SEED", "20261005")))
    kinds = ["rects", "dense", "uniform", "checker"]
    total = 0
    for trial in range(int(os.environ.get("ORBFE_FUZZ_TRIALS", "30"))):
        w = int(rng.integers(40, 400))
        h = int(rng.integers(40, 300))
        if trial % 3 == 0:
            w -= w % 4
        cfg = dict(levels=int(rng.integers(1, 9)), cell=int(rng.choice([8, 16, 32, 64])),
                   min_arc=int(rng.integers(9, 13)), fast_threshold=int(rng.integers(3, 40)),
                   max_features=int(rng.choice([0, 0, 7, 50, 400])), angle_in_radians=int(rng.integers(0, 2)),
                   descriptor_level=1)
        n = int(rng.integers(1, 4))
        frames = np.stack([synth.frame(w, h, int(rng.integers(0, 10 ** 6)), "uniform") if k == "uniform" else
                           FRAMES[k](w, h) for k in rng.choice(kinds, n)])
        ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, **cfg)
        try:
            total += _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg)
        except AssertionError as e:
            raise AssertionError("trial %d: %dx%d %r: %s" % (trial, w, h, cfg, e))
        ctx.close()
    assert total > 1500


# ------------------------------------------------------------------ more than 65534 records per frame
@pytest.mark.parametrize("dl", [0, 1])
def test_frames_with_more_than_65534_keypoints(gpu, oracle_mod, dl):
    """3840x2160 with 8-px cells and no feature budget: cap = K = 129 600 cells.  The tile describe kernel addresses
    records through a 16-bit cell -> slot map (0xFFFF = none), so such contexts must take the patch kernel (ADVICE r2:
    slots wrapped modulo 65536 and records overwrote each other while the counts looked right)."""
    torch, orbfe = gpu
    w, h = 3840, 2160
    cfg = dict(levels=3, cell=8, min_arc=9, max_features=0, descriptor_level=dl)
    frames = np.stack([synth.frame(w, h, 3, "uniform")])
    ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, want_soa=False, **cfg)
    assert ctx.cap == 129600
    assert cnt[0] > 65535, "the scene must exceed the 16-bit slot range to exercise the case"
    _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg)


# ------------------------------------------------------------------ exact-length gather: the root's offset table
def test_rccl_exact_gather_offsets_world1(gpu, oracle_mod):
    """orbfe_dist_exact_offsets: a consumer of the exact-length layout gets every frame's first record index from
    the library instead of re-deriving the dense packing (world = 1: the only size this box can run; the counts
    include empty and full frames and more than one 256-frame chunk of the kernel's running prefix)."""
    torch, orbfe = gpu
    from orbfe import dist as od
    w, h, n = 160, 120, 300
    base = np.stack([synth.frame(w, h, 3, "rects", n_rects=60, min_size=6, max_size=30), synth.frame(w, h, 0, "const"),
                     synth.frame(w, h, 4, "uniform"), synth.frame(w, h, 5, "rects", n_rects=20, min_size=6, max_size=40)])
    frames = base[np.arange(n) % 4]
    cfg = dict(levels=3, cell=16, min_arc=9, max_features=40)
    ctx, rec, cnt, _ = _run_extract(torch, orbfe, frames, want_soa=False, **cfg)
    d_rec, d_cnt = dev(torch, rec.view(np.uint8).reshape(-1)), dev(torch, cnt)
    all_rec = torch.zeros(n * ctx.cap * 52, dtype=torch.uint8, device="cuda")
    all_cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    offs = torch.full((n,), -1, dtype=torch.int64, device="cuda")
    comm = od.RcclComm(0, 1, 0, lambda ident: ident)
    s = stream(torch)
    comm.gather_keypoints(d_rec.data_ptr(), d_cnt.data_ptr(), n, ctx.cap, all_rec.data_ptr(), all_cnt.data_ptr(), 0, 1, s)
    comm.wait(s)
    comm.exact_offsets(all_cnt.data_ptr(), n, ctx.cap, offs.data_ptr(), s)
    torch.cuda.synchronize()
    comm.sync()
    want = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64)
    np.testing.assert_array_equal(offs.cpu().numpy(), want)
    got = all_rec.cpu().numpy().view(orbfe.KEYPOINT_DTYPE)
    o = offs.cpu().numpy()
    for f in (0, 1, 2, 3, 255, 256, 257, 299):
        assert got[o[f]:o[f] + cnt[f]].tobytes() == rec[f, :cnt[f]].tobytes()
    comm.close()


# ------------------------------------------------------------------ round-3 kernel changes
@pytest.mark.parametrize("n", [1, 7, 8, 9, 11, 16, 23])
def test_frame_counts_around_the_eight_frame_grid_rows(gpu, oracle_mod, n):
    """pyramid / detect / describe place a block by (blockIdx.x & 7, blockIdx.y) in rows of 8 frames (frame_item,
    device_common.hpp); the last row of a batch that is no multiple of 8 holds blocks without a frame.  Every frame
    of every batch size must still equal the oracle, the SoA view included."""
    torch, orbfe = gpu
    w, h = 320, 240
    base = synth.frames(w, h, 5, 700 + n, "rects", **synth.DENSE)
    frames = base[np.arange(n) % 5]
    cfg = dict(levels=4, cell=8, min_arc=9, max_features=300)
    ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, **cfg)
    assert _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg) > 50 * n


def test_context_reuse_clears_every_cell_key(gpu, oracle_mod):
    """The pyramid kernel clears the frame's cell keys in slices, one per tile (it takes the number of tiles from the
    grid: with the 8-frames-per-row grid a wrong divisor left most keys of a REUSED context uncleared).  Busy frames
    first, then near-empty ones through the same context: the second result must not inherit a key."""
    torch, orbfe = gpu
    w, h, n = 640, 480, 16
    busy = synth.frames(w, h, 4, 41, "rects", **synth.DENSE)[np.arange(n) % 4]
    quiet = np.full((n, h, w), 128, np.uint8)
    quiet[:, 200:232, 300:340] = 30  # one dark rectangle: a handful of corners
    cfg = dict(levels=6)  # reference regime: cell 32, levels 0..5
    ctx = orbfe.Context(w, h, max_batch=n, **cfg)
    rec = torch.zeros(n * ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    ocfg = oracle_mod.make_config(w, h, levels=6)
    for frames in (busy, quiet, busy):
        d_in = dev(torch, frames)
        ctx.extract(d_in.data_ptr(), w, w * h, n, rec.data_ptr(), cnt.data_ptr(), None, stream(torch))
        torch.cuda.synchronize()
        counts = cnt.cpu().numpy()
        records = rec.cpu().numpy().view(orbfe.KEYPOINT_DTYPE).reshape(n, ctx.cap)
        for f in (0, 1, 2, 3, n - 1):
            ref = oracle_mod.extract_frame(frames[f], ocfg)
            assert counts[f] == ref["count"]
            assert records[f, :counts[f]].tobytes() == ref["records"].tobytes()
    ctx.close()


@pytest.mark.parametrize("thr", [1, 100, 127, 128, 200, 254])
@pytest.mark.parametrize("arc", [9, 12])
def test_compass_pretest_at_every_threshold_range(gpu, oracle_mod, thr, arc):
    """The byte-wise compass pre-test (compass4, batch_kernels.hip) forms c + t and c - t modulo 256 with explicit
    overflow / underflow masks; thresholds with the top bit set and sums that wrap are its corner cases.  Frames of
    uniform noise and of saturated black / white blocks (c + t > 255, c - t < 0 on most pixels)."""
    torch, orbfe = gpu
    w, h = 256, 192
    rng = np.random.default_rng(1000 + thr)
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    blocks = (rng.integers(0, 2, (h // 8, w // 8), dtype=np.uint8) * 255).repeat(8, 0).repeat(8, 1)
    blocks[::5, ::7] ^= 0x80
    edge = np.where(rng.random((h, w)) < 0.5, 255 - rng.integers(0, 3, (h, w)), rng.integers(0, 3, (h, w))).astype(np.uint8)
    frames = np.stack([noise, blocks, edge])
    cfg = dict(levels=3, cell=8, min_arc=arc, fast_threshold=thr)
    ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, **cfg)
    _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg)


@pytest.mark.parametrize("thr", [1.0, 127.0, 128.0, 200.0])
def test_stage_detect_prechecks_at_every_threshold_range(gpu, oracle_mod, thr):
    """orbfe_detect's fused path runs the reference's own prechecks (fast.cu:98-124) in the byte-wise form
    (compass4<2>): not similar = brighter or darker, each polarity masked by its own overflow / underflow."""
    torch, orbfe = gpu
    n = _detect_case(torch, orbfe, oracle_mod, 424, 240, 3, 9, thr, True, 0)
    assert n > 0 or thr >= 127.0


def test_steering_table_equals_the_arithmetic_for_every_orientation(gpu):
    """The tile describe kernel looks the rotated rBRIEF sample offsets up by orientation (steer_table.cpp; generated
    at build time with the oracle's float arithmetic).  Exhaustive proof on the device: every float in [-pi, pi],
    all 512 pattern points, table == orb.cu:12-14 / :42-46 as evaluated everywhere else."""
    torch, orbfe = gpu
    ctx = orbfe.Context(640, 480, max_batch=1, levels=1)
    n, bad = ctx.selfcheck_steer_table()
    assert n == 2 * (0x40490FDB + 1) and bad == 0, (n, bad)
    ctx.close()
    rad = orbfe.Context(640, 480, max_batch=1, levels=1, cell=8, min_arc=9, angle_in_radians=1)
    assert rad.selfcheck_steer_table() == (0, 0)  # no table in that regime: the offsets are computed
    rad.close()


@pytest.mark.parametrize("n", [7, 9, 16, 19])
@pytest.mark.parametrize("which", ["patch", "tile"])
def test_frame_counts_with_each_describe_kernel(gpu, oracle_mod, monkeypatch, which, n):
    """Both describe kernels place their blocks with frame_item (rows of 8 frames) and both take the rotated pattern from
    the orientation table in this regime: frame counts around a multiple of 8, reference regime and EXT regime."""
    torch, orbfe = gpu
    monkeypatch.setenv("ORBFE_DESCRIBE", which)
    w, h = 320, 240
    base = synth.frames(w, h, 5, 900 + n, "rects", **synth.DENSE)
    frames = base[np.arange(n) % 5]
    for cfg in (dict(levels=4), dict(levels=4, cell=16, min_arc=10, max_features=150)):
        ctx, rec, cnt, soa = _run_extract(torch, orbfe, frames, **cfg)
        assert _check_extract(oracle_mod, ctx, frames, rec, cnt, soa, **cfg) > 10 * n
