#!/usr/bin/env python3
"""Determinism soak: run extraction + matching on the same resident batch many times and require every
output byte to repeat (a race in a queue, an atomic or a barrier shows up as a changed checksum).
usage: soak.py [iters]   -- C2, C3 (windowed), reference and C5 shapes, small batches"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
CASES = [
    ("c2", 640, 480, 32, dict(levels=8, cell=8, min_arc=9, max_features=2000), dict(mode=1, window=-1, md=64, stride=1)),
    # 71 pairs x 4 query tiles >= 256 work items: the size rule takes match_tile_kernel (LDS ring, one barrier per step)
    ("c2_tile", 640, 480, 72, dict(levels=8, cell=8, min_arc=9, max_features=2000), dict(mode=1, window=-1, md=64, stride=1)),
    # 2048 frames (32 distinct ones, repeated): the launch sizes at which detect runs in groups of four tiles per workgroup
    # and the tile matcher covers the chip for 16 rounds of workgroups; a quarter of the iterations
    ("c2_bench", 640, 480, 2048, dict(levels=8, cell=8, min_arc=9, max_features=2000), dict(mode=1, window=-1, md=64, stride=1, div=4, distinct=32)),
    ("c3", 848, 480, 32, dict(levels=8, cell=8, min_arc=9, max_features=2000), dict(mode=1, window=16, md=80, stride=2)),
    ("ref", 640, 480, 32, dict(), dict(mode=0, window=32, md=8, stride=1)),
    ("c5", 3840, 2160, 2, dict(levels=12, cell=16, min_arc=9, max_features=8000), dict(mode=1, window=-1, md=64, stride=1)),
    # the corrected EXT modes (arithmetic descriptor loop, description on the keypoint's own level)
    ("c2_fixed", 640, 480, 32, dict(levels=8, cell=8, min_arc=9, max_features=2000, angle_in_radians=1, descriptor_level=1),
     dict(mode=1, window=-1, md=64, stride=1)),
]
s = torch.cuda.current_stream().cuda_stream
bad = 0
for name, w, h, B, cfg, mm in CASES:
    ctx = orbfe.Context(w, h, max_batch=B, **cfg)
    cap = ctx.cap
    scale = (w * h) // (640 * 480)
    nd = mm.get("distinct", B)
    frames = np.stack([synth.frame(w, h, 100 + i, "rects", n_rects=800 * scale, min_size=6, max_size=32) for i in range(nd)])
    d_in = torch.from_numpy(frames).cuda()[torch.arange(B, device="cuda") % nd].reshape(-1).contiguous()
    n_it = max(iters // mm.get("div", 1), 2)
    rec = torch.zeros(B * cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    n_pairs = (B - 1) if mm["stride"] == 1 else B // 2
    idx = torch.zeros(n_pairs * cap, dtype=torch.int32, device="cuda")
    dist = torch.zeros(n_pairs * cap, dtype=torch.int32, device="cuda")
    first = None
    wts = torch.arange(1, rec.numel() + 1, device="cuda", dtype=torch.int64) % 65521
    for it in range(n_it):
        rec.zero_()
        ctx.extract(d_in.data_ptr(), w, w * h, B, rec.data_ptr(), cnt.data_ptr(), None, s)
        ctx.match_pairs(rec.data_ptr(), cnt.data_ptr(), B, 0, mm["stride"], mm["mode"], mm["window"], mm["md"],
                        idx.data_ptr(), dist.data_ptr(), s)
        sig = (int((rec.to(torch.int64) * wts).sum()), int(cnt.sum()), int((idx.to(torch.int64) * 31 + dist).sum()))
        if first is None:
            first = sig
        elif sig != first:
            bad += 1
            print("MISMATCH", name, it, sig, first)
            break
    print("%s: %d iterations, keypoints %d, matches %d, signature %s" % (name, n_it, first[1], int((idx >= 0).sum()), "repeats" if bad == 0 else "CHANGED"))
# align_depth_to_other (r4): LDS-window + global atomicMin splat, both output protocols, pipelined launches
import ctypes as C  # noqa: E402
w, h, B = 848, 480, 24
d, o, e, scale = synth.rig("d435", w, h)
mk = lambda t: orbfe.Intrinsics(t[0], t[1], t[2], t[3], t[4], t[5], t[6], (C.c_float * 5)(*t[7]))
di, oi, ex = mk(d), mk(o), orbfe.Extrinsics((C.c_float * 9)(*e[0]), (C.c_float * 3)(*e[1]))
src = torch.from_numpy(synth.depth_frames(w, h, B, first_index=900).view(np.int16)).cuda()
out = torch.zeros((B, h * w), dtype=torch.int32, device="cuda")
for proto in ("literal", "zero"):
    os.environ["ORBFE_ALIGN_PROTOCOL"] = proto
    os.environ["ORBFE_ALIGN_CHUNK"] = "8"
    first = None
    for it in range(iters):
        out.fill_(-1)
        orbfe.check(orbfe.lib().orbfe_align_depth_batch(out.data_ptr(), w * h, src.data_ptr(), w * h, B, scale, C.byref(di),
                                                        C.byref(oi), C.byref(ex), s))
        wts = torch.arange(1, out.numel() + 1, device="cuda", dtype=torch.int64).reshape(out.shape) % 65521
        sig = int((out.to(torch.int64) * wts).sum())
        if first is None:
            first = sig
        elif sig != first:
            bad += 1
            print("MISMATCH align", proto, it, sig, first)
            break
    print("align (%s protocol): %d iterations of %d frames, covered %.3f, signature %s" % (
        proto, iters, B, float((out != 0).float().mean()), "repeats" if sig == first else "CHANGED"))
sys.exit(1 if bad else 0)
