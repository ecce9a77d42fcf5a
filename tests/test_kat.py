"""Known-answer tests that pin the CPU oracle (CPU only, no GPU).

The reference ships no tests or golden vectors for this path, so the
oracle is pinned by the answers that can be derived from the reference's source alone
(SURVEY.md Appendix B), by properties of the algorithms, and by the committed digests in
tests/golden/ (test_golden.py).
"""
import hashlib
import struct

import numpy as np
import pytest

from orbfe import synth


def test_pattern_sha256_and_shape(oracle_mod):
    pat = oracle_mod.pattern().astype(np.int32)
    assert pat.shape == (1024,)
    assert list(pat[:4]) == [8, -3, 9, 5] and list(pat[-4:]) == [-1, -6, 0, -11]
    assert pat.sum() == -406 and pat.min() == -13 and pat.max() == 12
    sha = hashlib.sha256(struct.pack("<1024i", *pat)).hexdigest()
    assert sha == "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"
    # max sample radius 18.38 px
    r = np.sqrt((pat.reshape(-1, 2).astype(np.float64) ** 2).sum(1)).max()
    assert abs(r - np.sqrt(13 ** 2 + 13 ** 2)) < 1e-9


def test_lut_population_and_naive_run_length(oracle_mod):
    assert oracle_mod.fast_lut(12).sum() == 129
    assert oracle_mod.fast_lut(9).sum() == 1025

    def naive(m, arc):
        bits = [(m >> i) & 1 for i in range(16)] * 2
        run = best = 0
        for b in bits:
            run = run + 1 if b else 0
            best = max(best, run)
        return int(min(best, 16) >= arc)

    rng = np.random.default_rng(1)
    for arc in (9, 10, 11, 12):
        lut = oracle_mod.fast_lut(arc)
        for m in list(rng.integers(0, 65536, 3000)) + [0, 0xFFFF, 0x0FFF, 0xF00F, 0x01FF, 0xFF80]:
            assert lut[m] == naive(int(m), arc)


@pytest.mark.parametrize("arc", [9, 10, 11, 12])
def test_closed_form_arc_test_equals_lut(oracle_mod, arc):
    """include/orbfe_math.h orbfe_has_arc (used by the fused HIP kernel) == the literal
    restatement of fast_gpu_is_corner, for all 65536 masks."""
    lut = oracle_mod.fast_lut(arc)
    got = np.array([oracle_mod.has_arc(m, arc) for m in range(65536)], np.uint8)
    np.testing.assert_array_equal(got, lut)


def test_orientation_half_widths():
    u = [int(np.floor(np.sqrt(np.float32(225 - dy * dy)) + 0.5)) for dy in range(16)]
    assert u == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 7, 5, 0]


def test_cell_counts_and_halving_chain(oracle_mod):
    assert oracle_mod.num_cells(640, 480) == 300
    assert oracle_mod.num_cells(848, 480) == 405
    assert oracle_mod.num_cells(1280, 720) == 920
    assert oracle_mod.num_cells(3840, 2160) == 8160
    assert [oracle_mod.level_dims(848, 480, l) for l in range(1, 8)] == [
        (424, 240), (212, 120), (106, 60), (53, 30), (26, 15), (13, 7), (6, 3)]


def test_blur_constant_impulse_and_unwritten_rows(oracle_mod):
    c = np.full((20, 70), 93, np.uint8)
    out = oracle_mod.gaussian_blur_3x3(c)
    assert (out[1:-2] == 93).all() and (out[0] == 0).all() and (out[-2:] == 0).all()
    imp = np.zeros((20, 70), np.uint8)
    imp[10, 40] = 255  # 40 % 32 not in {0, 31}
    out = oracle_mod.gaussian_blur_3x3(imp)
    np.testing.assert_array_equal(out[9:12, 39:42], [[16, 32, 16], [32, 64, 32], [16, 32, 16]])
    assert out.sum() == 256
    # seam (Q2): an impulse in column 32 does not bleed into column 31 and vice versa, and the
    # seam pixel takes its own value in place of the missing neighbour
    imp = np.zeros((20, 70), np.uint8)
    imp[10, 32] = 255
    out = oracle_mod.gaussian_blur_3x3(imp)
    assert (out[:, 31] == 0).all()
    np.testing.assert_array_equal(out[9:12, 32], [48, 96, 48])  # (1+2)/16, (2+4)/16 of 255
    np.testing.assert_array_equal(out[9:12, 33], [16, 32, 16])


def test_halving_truncates(oracle_mod):
    a = np.array([[1, 2, 9], [3, 5, 9], [7, 7, 7]], np.uint8)
    np.testing.assert_array_equal(oracle_mod.halfsample(a), [[2]])  # (1+2+3+5) >> 2 = 2
    assert (oracle_mod.halfsample(np.full((6, 10), 200, np.uint8)) == 200).all()


def test_fast_threshold_and_score_range(oracle_mod):
    lut = oracle_mod.fast_lut(12)

    def ring_image(center, ring):
        img = np.full((9, 9), center, np.uint8)
        offs = [(0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3), (0, -3),
                (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3)]
        for (dx, dy), v in zip(offs, ring):
            img[4 + dy, 4 + dx] = v
        return img

    # brighter iff p >= c + 14, darker iff p <= c - 14
    assert oracle_mod.fast_response(ring_image(100, [114] * 16), lut)[4, 4] == 16 * 1
    assert oracle_mod.fast_response(ring_image(100, [113] * 16), lut)[4, 4] == 0
    assert oracle_mod.fast_response(ring_image(100, [86] * 16), lut)[4, 4] == 16 * 1
    assert oracle_mod.fast_response(ring_image(100, [87] * 16), lut)[4, 4] == 0
    assert oracle_mod.fast_response(ring_image(0, [255] * 16), lut)[4, 4] == 3872  # max
    # exactly 12 contiguous bright pixels, the other 4 similar: minimum score 12
    assert oracle_mod.fast_response(ring_image(100, [114] * 12 + [100] * 4), lut)[4, 4] == 12
    # 11 contiguous: not a FAST-12 corner, but a FAST-9 one
    img11 = ring_image(100, [114] * 11 + [100] * 5)
    assert oracle_mod.fast_response(img11, lut)[4, 4] == 0
    assert oracle_mod.fast_response(img11, oracle_mod.fast_lut(9))[4, 4] == 11
    # the score sums ALL labelled pixels, not only the arc (Q4): an isolated 13th bright pixel
    img = ring_image(100, [114] * 12 + [100, 120, 100, 100])
    assert oracle_mod.fast_response(img, lut)[4, 4] == 12 + 7
    # wrap-around arc: positions 10..15 and 0..5
    ring = [114] * 6 + [100] * 4 + [114] * 6
    assert oracle_mod.fast_response(ring_image(100, ring), lut)[4, 4] == 12
    # border: nothing within 3 px of the edge
    r = oracle_mod.fast_response(synth.frame(40, 30, 0, "uniform"), oracle_mod.fast_lut(9))
    assert (r[:3] == 0).all() and (r[-3:] == 0).all() and (r[:, :3] == 0).all() and (r[:, -3:] == 0).all()
    assert (r > 0).any()


def test_grid_nms_tie_order_and_levels(oracle_mod):
    """Q6: equal maxima -> lower level, then lower warp, then lower bit-reversed lane."""
    r0 = np.zeros((480, 640), np.float32)
    # two equal isolated maxima in cell (0,0) of level 0 (block 32x4: warp id = row & 3):
    # (x=10,y=5) -> row offset 2 -> thread row 2 ; (x=20,y=8) -> row offset 5 -> thread row 1
    r0[5, 10] = 50
    r0[8, 20] = 50
    pos, score, level = oracle_mod.grid_nms([r0], 32)
    assert score[0] == 50 and tuple(pos[0]) == (20.0, 8.0) and level[0] == 0
    # same warp (thread row 2): lanes 10 and 20 -> bitrev5(10)=10, bitrev5(20)=5 -> lane 20 wins
    r0[:] = 0
    r0[5, 10] = 50
    r0[9, 20] = 50
    pos, score, level = oracle_mod.grid_nms([r0], 32)
    assert tuple(pos[0]) == (20.0, 9.0)
    # same thread (column 10, rows 5 and 9): the earlier row iteration wins
    r0[:] = 0
    r0[5, 10] = 50
    r0[9, 10] = 50
    pos, _, _ = oracle_mod.grid_nms([r0], 32)
    assert tuple(pos[0]) == (10.0, 5.0)
    # a non-strict maximum (equal neighbour) is suppressed
    r0[:] = 0
    r0[5, 10] = 50
    r0[5, 11] = 50
    _, score, _ = oracle_mod.grid_nms([r0], 32)
    assert score[0] == 0
    # across levels: an equal score on level 1 does not replace level 0; a larger one does,
    # and its position is scaled by 2
    r0[:] = 0
    r0[5, 10] = 50
    r1 = np.zeros((240, 320), np.float32)
    r1[6, 7] = 50
    pos, score, level = oracle_mod.grid_nms([r0, r1], 32)
    assert level[0] == 0 and tuple(pos[0]) == (10.0, 5.0)
    r1[6, 7] = 51
    pos, score, level = oracle_mod.grid_nms([r0, r1], 32)
    assert level[0] == 1 and tuple(pos[0]) == (14.0, 12.0) and score[0] == 51
    # empty cells read score 0, pos (0,0), level 0 (Q5)
    assert score[1] == 0 and tuple(pos[1]) == (0.0, 0.0) and level[1] == 0


def test_constant_frame_gives_nothing(oracle_mod):
    cfg = oracle_mod.make_config(640, 480, levels=6)
    out = oracle_mod.extract_frame(synth.frame(640, 480, 0, "const"), cfg)
    assert out["count"] == 0 and not out["desc"].any() and not out["score"].any()
    idx, n = oracle_mod.match_keypoints(out["pos"], out["desc32"], out["pos"], out["desc32"], 2, 4)
    # empty cells all sit at (0,0) with descriptor 0: they trivially match each other, which is
    # why only score > 0 cells are emitted as records
    assert n == out["pos"].shape[0]
    idx, n = oracle_mod.match_keypoints(np.zeros((0, 2)), np.zeros(0), np.zeros((0, 2)), np.zeros(0))
    assert n == 0


def test_deterministic_math_accuracy_and_known_bits(oracle_mod):
    assert oracle_mod.atan2f(0, 0) == 0 and oracle_mod.atan2f(0, -5) == np.float32(np.pi)
    assert oracle_mod.atan2f(3, 0) == np.float32(np.pi / 2) and oracle_mod.atan2f(-3, 0) == np.float32(-np.pi / 2)
    rng = np.random.default_rng(0)
    ys = rng.integers(-1200000, 1200000, 4000)
    xs = rng.integers(-1200000, 1200000, 4000)
    worst = 0.0
    for y, x in zip(ys, xs):
        got = float(oracle_mod.atan2f(y, x))
        ref = np.arctan2(float(y), float(x))
        worst = max(worst, abs(got - ref) / float(np.spacing(np.float32(abs(ref)) + np.float32(1e-30))))
    assert worst <= 4.0, worst
    worst = 0.0
    for t in np.linspace(-3.2, 3.2, 4001):
        s, c = oracle_mod.sincosf(t)
        t32 = float(np.float32(t))
        worst = max(worst, abs(float(s) - np.sin(t32)), abs(float(c) - np.cos(t32)))
    assert worst <= 2.5e-7, worst
    # frozen bit patterns (a compiler that contracts a*b+c into an fma changes these)
    known = {(1.0, 1.0): 0x3F490FDB, (123456.0, -7890.0): None}
    assert np.float32(oracle_mod.atan2f(1.0, 1.0)).view(np.uint32) == known[(1.0, 1.0)]


def test_descriptor_known_structure(oracle_mod):
    """On a horizontal ramp with angle 0 every test reduces to P.x < Q.x on the pattern."""
    w, h = 64, 64
    img = np.tile(np.arange(w, dtype=np.uint8) * 3, (h, 1))
    pos = np.array([[32, 32]], np.float32)
    desc, d32 = oracle_mod.calc_orb(np.zeros(1, np.float32), pos, img)
    pat = oracle_mod.pattern().reshape(256, 4)
    expect = np.packbits((pat[:, 0] < pat[:, 2]).astype(np.uint8).reshape(32, 8), axis=1, bitorder="little")[:, 0]
    np.testing.assert_array_equal(desc[0], expect)
    assert d32[0] == sum(int(expect[i] == 1) << i for i in range(32))
    # guard band: x < 17 or x > w - 17 gives a zero descriptor (orb.cu:34-38)
    d, _ = oracle_mod.calc_orb(np.zeros(2, np.float32), np.array([[16, 32], [48, 32]], np.float32), img)
    assert not d.any()
    d, _ = oracle_mod.calc_orb(np.zeros(2, np.float32), np.array([[17, 17], [47, 47]], np.float32), img)
    assert d.any(axis=1).all()


def test_orientation_known_answers(oracle_mod):
    w = h = 64
    pos = np.array([[32, 32]], np.float32)
    ramp_x = np.tile(np.arange(w, dtype=np.uint8), (h, 1))
    assert oracle_mod.compute_fast_angle(pos, None, ramp_x)[0] == 0.0          # m01 = 0, m10 > 0
    assert oracle_mod.compute_fast_angle(pos, None, ramp_x.T.copy())[0] == np.float32(np.pi / 2)
    assert oracle_mod.compute_fast_angle(pos, None, (63 - ramp_x))[0] == np.float32(np.pi)
    assert oracle_mod.compute_fast_angle(pos, None, np.full((h, w), 9, np.uint8))[0] == 0.0
    # the quirk of Q7: the descriptor uses angle * pi/180.  A stored angle of pi/2 steers the
    # pattern by 1.57 degrees: every rotated offset moves by < 0.5 px (13 * 0.0274 = 0.36), so
    # the descriptor is IDENTICAL to the unsteered one; only near |angle| = pi do a few of
    # the outermost samples move by one pixel.
    img = synth.frame(w, h, 1, "uniform")
    d0, _ = oracle_mod.calc_orb(np.zeros(1, np.float32), pos, img)
    d1, _ = oracle_mod.calc_orb(np.full(1, np.pi / 2, np.float32), pos, img)
    assert (d0 == d1).all()
    d2, _ = oracle_mod.calc_orb(np.full(1, 3.1, np.float32), pos, img)
    assert 0 < np.unpackbits(d0 ^ d2).sum() < 100
    # with the EXT fix (angle_in_radians) a quarter turn changes about half of the bits
    d3, _ = oracle_mod.calc_orb(np.full(1, np.pi / 2, np.float32), pos, img, angle_in_radians=1)
    assert np.unpackbits(d0 ^ d3).sum() > 60


def test_matcher_partial_tile_quirk(oracle_mod):
    """Q8: in a last tile of m < 32 entries, prev keypoints with (i % 32) >= m never see it."""
    n_curr = 5
    pc = np.zeros((n_curr, 2), np.float32)
    dc = np.zeros(n_curr, np.uint32)
    pp = np.zeros((40, 2), np.float32)
    dp = np.zeros(40, np.uint32)
    idx, n = oracle_mod.match_keypoints(pp, dp, pc, dc, 2, 4)
    expect = np.array([i % 32 if (i % 32) < n_curr else -1 for i in range(40)])
    # thread tid starts at j = tid, distance 0 cannot be improved -> matches curr index tid
    np.testing.assert_array_equal(idx, expect)
    assert n == (expect >= 0).sum()


def test_match256_brute_force_against_numpy(oracle_mod):
    a, b = synth.descriptors(200, 1), synth.descriptors(300, 2)
    b[17] = a[5]
    b[40] = a[5]
    idx, dist = oracle_mod.match256(a, b)
    d = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)
    np.testing.assert_array_equal(idx, d.argmin(1))      # argmin = lowest index among ties
    np.testing.assert_array_equal(dist, d.min(1))
    assert idx[5] == 17 and dist[5] == 0


def test_top_n_selection_rule(oracle_mod):
    img = synth.frame(640, 480, 2, "rects", **synth.DENSE)
    full = oracle_mod.extract_frame(img, oracle_mod.make_config(640, 480, levels=8, cell=8, min_arc=9))
    top = oracle_mod.extract_frame(img, oracle_mod.make_config(640, 480, levels=8, cell=8, min_arc=9,
                                                              max_features=500))
    assert full["count"] > 2000 and top["count"] == 500
    order = np.lexsort((np.arange(full["score"].size), -full["score"]))  # score desc, cell asc
    keep = np.sort(order[:500])
    np.testing.assert_array_equal(np.flatnonzero(top["angle"] != 0) <= keep.max(), True)
    np.testing.assert_array_equal(top["records"]["score"], full["score"][keep])
    np.testing.assert_array_equal(top["records"]["x"], full["pos"][keep, 0])


def test_rgb_to_grayscale_literal_double(oracle_mod):
    """f1: floor((B*0.07 + G*0.72 + R*0.21) + 0.5) in double, left to right, no contraction.
    It equals the exactly rounded value (7B + 72G + 21R + 50) // 100 except on some exact ties
    (7B + 72G + 21R == 50 mod 100), where the binary constants decide -- that is the reference's
    arithmetic, not a bug of the restatement."""
    r, g, b = np.meshgrid(np.arange(256), np.arange(0, 256, 5), np.arange(256), indexing="ij")
    rgb = np.stack([r, g, b], -1).reshape(256, -1, 3).astype(np.uint8)
    got = oracle_mod.rgb_to_grayscale(rgb).astype(np.int64)
    r, g, b = [rgb[..., i].astype(np.int64) for i in range(3)]
    s = 7 * b + 72 * g + 21 * r
    exact = (s + 50) // 100
    diff = got != exact
    assert (np.abs(got - exact) <= 1).all()
    assert (s[diff] % 100 == 50).all(), "only exact ties may differ from round-half-up"
    assert 0 < diff.sum() < (s % 100 == 50).sum()
    # numpy evaluates the same IEEE double expression: bit-for-bit agreement
    ref = np.floor((b.astype(np.float64) * 0.07 + g.astype(np.float64) * 0.72) + r.astype(np.float64) * 0.21 + 0.5)
    np.testing.assert_array_equal(got, ref.astype(np.int64))
    assert oracle_mod.rgb_to_grayscale(np.full((2, 2, 3), 255, np.uint8)).max() == 255


def test_deprojection_known_answers(oracle_mod):
    """f2: pinhole deprojection point = depth * ((px - ppx) / fx, (py - ppy) / fy, 1); the filter
    keeps depth > 1 and score > 1; the reference's depth lookup uses y for row AND column."""
    import ctypes as C
    w, h = 64, 48
    depth = np.zeros((h, w), np.uint32)
    depth[10, 10] = 2000      # what the reference reads for a keypoint at y = 10 (any x)
    depth[10, 40] = 3000      # what a correct lookup reads for (x, y) = (40, 10)
    k = oracle_mod.Intrinsics(w, h, 32.0, 24.0, 50.0, 40.0, 0, (C.c_float * 5)())
    pos = np.array([[40, 10], [5, 5], [40, 10]], np.float32)
    score = np.array([9, 9, 1], np.float32)            # third: score == 1 is rejected
    desc = np.array([11, 22, 33], np.uint32)
    p, pts, d, n = oracle_mod.keypoint_pixel_to_point(depth, k, pos, score, desc, 0)
    assert n == 1 and d[0] == 11
    np.testing.assert_allclose(pts[0], [2000 * (40 - 32) / 50, 2000 * (10 - 24) / 40, 2000], rtol=1e-6)
    p, pts, d, n = oracle_mod.keypoint_pixel_to_point(depth, k, pos, score, desc, 1)
    assert n == 1 and pts[0][2] == 3000


def test_blur_and_pyramid_against_scipy_and_numpy(oracle_mod):
    """Independent restatements of C.1 / C.2 with library primitives: away from the reference's
    32-column shuffle seams and its unwritten rows (Q1, Q2) the level-0 image is scipy.ndimage's 3x3 binomial
    convolution rounded as floor(s / 16 + 0.5) (gaussian_blur_3x3.cu:15-53); every further level is the truncated mean of
    2x2 blocks (pyramid.cu:6-29), odd trailing row / column dropped."""
    ndimage = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(3)
    for w, h in ((640, 480), (848, 480), (100, 70)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        got = oracle_mod.gaussian_blur_3x3(img)
        k = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], np.int64)
        s = ndimage.convolve(img.astype(np.int64), k, mode="nearest")
        want = np.floor(s / 16.0 + 0.5).astype(np.uint8)
        x = np.arange(w)
        cols = (x % 32 != 0) & (x % 32 != 31) & (x != w - 1)  # columns whose left and right taps are the true neighbours
        np.testing.assert_array_equal(got[1:h - 2][:, cols], want[1:h - 2][:, cols])
        assert not got[0].any() and not got[h - 2:].any()      # Q1: rows 0, H-2, H-1 stay 0
        lvl = got
        for _ in range(4):
            nxt = oracle_mod.halfsample(lvl)
            hh, ww = lvl.shape[0] // 2, lvl.shape[1] // 2
            blocks = lvl[:2 * hh, :2 * ww].astype(np.int64).reshape(hh, 2, ww, 2).sum(axis=(1, 3)) >> 2
            np.testing.assert_array_equal(nxt, blocks.astype(np.uint8))
            lvl = nxt


@pytest.mark.parametrize("arc", [9, 10, 12])
def test_fast_response_against_a_definition_level_numpy_restatement(oracle_mod, arc):
    """The oracle restates fast.cu statement by statement (bit masks, LUT, prechecks).  Here is FAST as its DEFINITION,
    vectorised over the image, with nothing shared with that code: label the 16 ring pixels brighter / darker than the
    centre by more than t, a corner iff some `arc` cyclically consecutive ring pixels carry the same label, score =
    max(sum of (p - c - t) over ALL brighter, sum of (c - p - t) over ALL darker) (fast.cu:196-255)."""
    offs = [(0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3), (0, -3),
            (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3)]
    t = 13
    for kind, w, h, seed in (("rects", 160, 120, 1), ("uniform", 96, 64, 2), ("checker", 80, 48, 3)):
        img = synth.frame(w, h, seed, kind)
        I = img.astype(np.int32)
        c = I[3:h - 3, 3:w - 3]
        ring = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in offs])  # [16, h-6, w-6]
        br, dk = ring > c + t, ring < c - t

        def has_run(lab):  # some `arc` cyclically consecutive positions all labelled
            run = np.ones_like(lab)
            for k in range(arc):
                run &= np.roll(lab, -k, axis=0)
            return run.any(axis=0)

        corner = has_run(br) | has_run(dk)
        sb = np.where(br, ring - c - t, 0).sum(axis=0)
        sd = np.where(dk, c - ring - t, 0).sum(axis=0)
        want = np.zeros((h, w), np.float32)
        want[3:h - 3, 3:w - 3] = np.where(corner, np.maximum(sb, sd), 0)
        got = oracle_mod.fast_response(img, oracle_mod.fast_lut(arc), float(t))
        np.testing.assert_array_equal(got, want)
        assert (want > 0).any() or kind == "checker"


@pytest.mark.parametrize("arc", [9, 12])
def test_fast_score_modes_against_their_definitions(oracle_mod, arc):
    """The other two values of the reference's enum fast_score (fast.cuh:18-23) as definitions, nothing shared with the
    restatement: SUM_OF_ABS_DIFF_ALL = sum of |p - c| over the 16 ring pixels of an accepted pixel (fast.cu:233-241);
    MAX_THRESHOLD = the largest integer threshold in (t, 255] at which the pixel is still a corner, t itself when there
    is none (:256-283 finds it by bisection; being a corner is monotone in the threshold, so a linear scan is the same)."""
    offs = [(0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3), (0, -3),
            (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3)]
    t = 13
    lut = oracle_mod.fast_lut(arc)
    for kind, w, h, seed in (("rects", 120, 90, 4), ("uniform", 64, 48, 5)):
        img = synth.frame(w, h, seed, kind)
        I = img.astype(np.int32)
        c = I[3:h - 3, 3:w - 3]
        ring = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in offs])

        def corner_at(thr):
            def has_run(lab):
                run = np.ones_like(lab)
                for k in range(arc):
                    run &= np.roll(lab, -k, axis=0)
                return run.any(axis=0)
            return has_run(ring > c + thr) | has_run(ring < c - thr)

        is_corner = corner_at(t)
        assert is_corner.sum() > 20
        want = np.zeros((h, w), np.float32)
        want[3:h - 3, 3:w - 3] = np.where(is_corner, np.abs(ring - c).sum(axis=0), 0)
        np.testing.assert_array_equal(oracle_mod.fast_response(img, lut, float(t), score=0), want)
        best = np.full(c.shape, t, np.int32)
        for thr in range(t + 1, 256):
            best = np.where(corner_at(thr), thr, best)
        want = np.zeros((h, w), np.float32)
        want[3:h - 3, 3:w - 3] = np.where(is_corner, best, 0)
        got = oracle_mod.fast_response(img, lut, float(t), score=2)
        np.testing.assert_array_equal(got, want)
        assert got.max() > t + 20
        # the live score is untouched by the new argument
        np.testing.assert_array_equal(oracle_mod.fast_response(img, lut, float(t), score=1), oracle_mod.fast_response(img, lut, float(t)))


@pytest.mark.parametrize("radians", [0, 1])
def test_orientation_and_rbrief_against_definition_level_numpy(oracle_mod, radians):
    """C.6 / C.7 of SURVEY.md Appendix C written down directly in numpy (nothing shared with oracle/orbfe_oracle.c but
    the two transcendental routines, which the build has to own: include/orbfe_math.h): intensity-centroid moments over
    the radius-15 disc with the reference's per-sample bounds (orb.cu:77-134), angle = ATAN2F(m01, m10); descriptor bit
    i = I(P_i rotated) < I(Q_i rotated) with single-precision products, ONE single-precision add / subtract and
    round-half-even, no contraction (orb.cu:12-14, :42-75), zero inside the guard band."""
    u = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 7, 5, 0]
    w, h = 96, 80
    img = synth.frame(w, h, 5, "uniform")
    I = img.astype(np.int64)
    rng = np.random.default_rng(8)
    pos = np.stack([rng.integers(3, w - 3, 300), rng.integers(3, h - 3, 300)], axis=1).astype(np.float32)
    pos[:8] = [[3, 3], [w - 4, h - 4], [16, 40], [17, 40], [w - 17, 40], [w - 16, 40], [40, 17], [40, h - 17]]
    ang = oracle_mod.compute_fast_angle(pos, None, img)
    pat = oracle_mod.pattern().reshape(256, 4).astype(np.float32)
    desc, _ = oracle_mod.calc_orb(ang, pos, img, angle_in_radians=radians)
    guard = 19 if radians else 17
    for k in range(len(pos)):
        kx, ky = int(pos[k, 0]), int(pos[k, 1])
        m10 = m01 = 0
        for dy in range(-15, 16):
            y = ky + dy
            if dy != 0 and not (0 < y < h):  # rows above / below: ky - dy > 0, ky + dy < h; the centre row has no test
                continue
            for dx in range(-u[abs(dy)], u[abs(dy)] + 1):
                x = kx + dx
                if 0 < x < w:
                    m10 += dx * I[y, x]
                    m01 += dy * I[y, x]
        want = oracle_mod.atan2f(np.float32(m01), np.float32(m10))
        assert ang[k].tobytes() == np.float32(want).tobytes(), (k, kx, ky)
        lx, ly = kx, ky
        if lx < guard or lx > w - guard - (1 if radians else 0) or ly < guard or ly > h - guard - (1 if radians else 0):
            assert not desc[k].any()
            continue
        theta = ang[k] if radians else np.float32(ang[k] * np.float32(np.float32(3.141592654) / np.float32(180.0)))
        b, a = oracle_mod.sincosf(theta)
        def sample(px, py):  # one float32 product per term, one float32 add, round half to even
            col = np.rint(np.float32(np.float32(px * a) - np.float32(py * b))).astype(int)
            row = np.rint(np.float32(np.float32(px * b) + np.float32(py * a))).astype(int)
            return I[ly + row, lx + col]
        bits = (sample(pat[:, 0], pat[:, 1]) < sample(pat[:, 2], pat[:, 3])).astype(np.uint8)
        np.testing.assert_array_equal(desc[k], np.packbits(bits.reshape(32, 8), axis=1, bitorder="little")[:, 0], err_msg=str(k))
