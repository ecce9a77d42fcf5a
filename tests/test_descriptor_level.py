"""EXT iv (descriptor_level) in the CPU oracle: the whole-frame oracle must equal the composition of its own
per-stage functions on the keypoint's level -- compute_fast_angle (orb.cu:77-142) and calc_orb (orb.cu:17-75) run on
pyramid[level] at position / 2^level -- i.e. exactly what a caller of the reference's stage functions would do to get
scale-aware ORB.  No reference behaviour exists for it (buildStream.cpp:442-460 always passes pyramid[0]); parity
with the reference is unpinned.  CPU only."""
import numpy as np
import pytest

from orbfe import synth


@pytest.mark.parametrize("cfg", [dict(levels=8, cell=8, min_arc=9, max_features=2000),
                                 dict(levels=6),
                                 dict(levels=5, cell=16, min_arc=10, angle_in_radians=1)])
def test_oracle_descriptor_level_is_the_stage_functions_on_the_keypoints_level(oracle_mod, cfg):
    w, h = 640, 480
    img = synth.frame(w, h, 21, "rects", **synth.DENSE)
    o0 = oracle_mod.extract_frame(img, oracle_mod.make_config(w, h, **cfg), want_pyramid=True)
    o1 = oracle_mod.extract_frame(img, oracle_mod.make_config(w, h, descriptor_level=1, **cfg), want_pyramid=True)
    for k in ("pos", "score", "level"):
        np.testing.assert_array_equal(o0[k], o1[k])  # detection does not depend on it
    r0, r1 = o0["records"], o1["records"]
    assert len(r0) == len(r1) > 100
    lv = r1["level"]
    assert (lv > 0).sum() > 10
    assert r0[lv == 0].tobytes() == r1[lv == 0].tobytes()
    rad = cfg.get("angle_in_radians", 0)
    for l in np.unique(lv):
        m = lv == l
        pos = np.stack([r1["x"][m], r1["y"][m]], axis=1) / np.float32(1 << l)
        assert (pos == np.floor(pos)).all()
        lvl_img = o1["pyramid"][l]
        ang = oracle_mod.compute_fast_angle(pos, None, lvl_img)
        desc, _ = oracle_mod.calc_orb(ang, pos, lvl_img, angle_in_radians=rad)
        np.testing.assert_array_equal(ang.view(np.uint32), r1["angle"][m].view(np.uint32))
        np.testing.assert_array_equal(desc, r1["desc"][m])
    # and it matters: coarse keypoints get another descriptor than the level-0 patch at 2^l times the scale
    assert (r0["desc"][lv > 0] != r1["desc"][lv > 0]).any(axis=1).mean() > 0.9


def test_oracle_descriptor_level_guard_band_is_in_level_coordinates(oracle_mod):
    """A keypoint won by level l is zero-described iff its LEVEL position is within 17 px of the LEVEL border."""
    w, h = 320, 240
    img = synth.frame(w, h, 5, "uniform")
    cfg = oracle_mod.make_config(w, h, levels=4, cell=8, min_arc=9, fast_threshold=5, descriptor_level=1)
    r = oracle_mod.extract_frame(img, cfg)["records"]
    seen_border = 0
    for rec in r:
        l = int(rec["level"])
        x, y, wl, hl = int(rec["x"]) >> l, int(rec["y"]) >> l, w >> l, h >> l
        border = x < 17 or x > wl - 17 or y < 17 or y > hl - 17
        assert border == (not rec["desc"].any()) or not border, (rec, wl, hl)
        if border:
            assert not rec["desc"].any()
            seen_border += l > 0
    assert seen_border > 5
