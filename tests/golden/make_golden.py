#!/usr/bin/env python3
"""Generate tests/golden/oracle_digests.json: SHA-256 of every stage output of the CPU
oracle on the deterministic synthetic frames (orbfe/synth.py).

The reference (dsvua/jetracer-orbslam2) ships no fixtures and its CUDA path cannot be built
here, so these are NOT outputs of the reference: they freeze the oracle against regressions
(an accidental change of the restatement, of the deterministic math, or of a compiler flag
such as -ffp-contract).  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))

CASES = {
    # name: (width, height, frame kind/kwargs, oracle config)
    "vga_ref_L1_rects": (640, 480, dict(index=0, kind="rects"), dict(levels=1)),
    "vga_ref_L6_dense": (640, 480, dict(index=1, kind="rects", n_rects=800, min_size=6, max_size=32), dict(levels=6)),
    "vga_ref_L6_uniform": (640, 480, dict(index=2, kind="uniform"), dict(levels=6)),
    "wvga_ref_L6_dense": (848, 480, dict(index=3, kind="rects", n_rects=800, min_size=6, max_size=32), dict(levels=6)),
    "vga_c2_top2000": (640, 480, dict(index=4, kind="rects", n_rects=800, min_size=6, max_size=32),
                       dict(levels=8, cell=8, min_arc=9, max_features=2000)),
    "vga_cell16_arc10_radians": (640, 480, dict(index=5, kind="rects", n_rects=800, min_size=6, max_size=32),
                                 dict(levels=5, cell=16, min_arc=10, angle_in_radians=1)),
    "small_64x64": (64, 64, dict(index=6, kind="uniform"), dict(levels=2)),
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def digests(oracle, synth):
    out = {}
    for name, (w, h, fk, ck) in CASES.items():
        img = synth.frame(w, h, **fk)
        cfg = oracle.make_config(w, h, **ck)
        r = oracle.extract_frame(img, cfg, want_pyramid=True)
        d = {"input": sha(img), "count": int(r["count"])}
        for l, p in enumerate(r["pyramid"]):
            d["pyramid_%d" % l] = sha(p)
        for k in ("pos", "score", "level", "angle", "desc", "desc32", "records"):
            d[k] = sha(r[k])
        # matcher digests on the frame against a 1-px shifted copy of itself
        img2 = np.roll(img, 1, axis=1)
        r2 = oracle.extract_frame(img2, cfg)
        a, b = r["records"], r2["records"]
        pa, pb = np.stack([a["x"], a["y"]], 1), np.stack([b["x"], b["y"]], 1)
        comp = lambda x: ((x == 1).astype(np.uint32) << np.arange(32, dtype=np.uint32)).sum(1).astype(np.uint32)
        idx, n = oracle.match_keypoints(pa, comp(a["desc"]), pb, comp(b["desc"]), 2, 4)
        d["match_ref"] = sha(idx)
        d["match_ref_count"] = int(n)
        idx256, dist = oracle.match256(a["desc"], b["desc"])
        d["match256_idx"] = sha(idx256)
        d["match256_dist"] = sha(dist)
        # round 2: the windowed 256-bit matcher, the matcher's compacted outputs and the reprojection
        widx, wdist = oracle.match256(a["desc"], b["desc"], pa, pb, 6, 80)
        d["match256_window_idx"] = sha(widx)
        d["match256_window_dist"] = sha(wdist)
        rng = np.random.default_rng(len(a) + 7)
        pts_a = rng.normal(size=(max(len(a), 1), 3)) * [300, 200, 900] + [0, 0, 2500]
        pts_b = rng.normal(size=(max(len(b), 1), 3)) * [300, 200, 900] + [0, 0, 2500]
        kx, ky, pm, cm = oracle.match_compact(idx, pb, pts_a[:len(a)], pts_b[:len(b)])
        d["compact_xy"] = sha(np.concatenate([kx, ky]))
        d["compact_points"] = sha(np.concatenate([pm.reshape(-1), cm.reshape(-1)]))
        intr = oracle.Intrinsics(w, h, w * 0.5 - 3.25, h * 0.5 + 1.5, 615.5, 615.25, 1,
                                 (oracle.C.c_float * 5)(0.11, -0.23, 0.0007, -0.0004, 0.09))
        T = np.array([[0.9995, 0.0, 0.0316, 12.5], [0.0, 1.0, 0.0, -3.25], [-0.0316, 0.0, 0.9995, 40.0], [0, 0, 0, 1.0]])
        d["reproject"] = sha(oracle.reproject_points(pts_a[:len(a)], T, intr))
        out[name] = d
    # round 4: align_depth_to_other (cuda-align.cu:366-399) on the synthetic depth frame 0, four rigs, 848x480 and a ragged pair
    import ctypes as C
    al = {}
    for kind in ("identity", "d435", "distorted", "wild"):
        for (dw, dh, ow, oh) in ((848, 480, 848, 480), (101, 67, 80, 60)):
            dk, ok, ek, scale = synth.rig(kind, dw, dh, ow, oh)
            mk = lambda t: oracle.Intrinsics(t[0], t[1], t[2], t[3], t[4], t[5], t[6], (C.c_float * 5)(*t[7]))
            ex = oracle.Extrinsics((C.c_float * 9)(*ek[0]), (C.c_float * 3)(*ek[1]))
            depth = synth.depth_frame(dw, dh, 0)
            aligned, pm = oracle.align_depth_to_other(depth, scale, max(dw, ow), max(dh, oh), mk(dk), mk(ok), ex, want_map=True)
            al["%s_%dx%d_to_%dx%d" % (kind, dw, dh, ow, oh)] = {"depth": sha(depth), "aligned": sha(aligned), "map": sha(pm),
                                                              "covered": int((aligned != 0).sum())}
    out["align_depth"] = al
    return out


if __name__ == "__main__":
    import oracle
    from orbfe import synth
    path = os.path.join(ROOT, "tests", "golden", "oracle_digests.json")
    json.dump(digests(oracle, synth), open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
