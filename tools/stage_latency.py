#!/usr/bin/env python3
"""Latency of the stage API for one frame (the call sequence of buildStream.cpp:424-460) on cuda:0.
usage: stage_latency.py [width height levels]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import torch
import orbfe
from orbfe import synth

w, h, levels = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (640, 480, 1)
L = orbfe.lib()
s = torch.cuda.current_stream().cuda_stream
k = ((w + 31) // 32) * ((h + 31) // 32)
gray = torch.from_numpy(synth.frame(w, h, 1, "rects", **synth.DENSE)).cuda()
imgs = [torch.zeros((h >> l, w >> l), dtype=torch.uint8, device="cuda") for l in range(levels)]
resp = [torch.zeros((h >> l, w >> l), dtype=torch.float32, device="cuda") for l in range(levels)]
lv = orbfe.make_levels([(imgs[l].data_ptr(), w >> l, h >> l, w >> l) for l in range(levels)],
                       [(resp[l].data_ptr(), w >> l, h >> l, (w >> l) * 4) for l in range(levels)])
lut = torch.zeros(65536, dtype=torch.uint8, device="cuda")
grid = torch.zeros(4 * k, dtype=torch.float32, device="cuda")
angle = torch.zeros(k, dtype=torch.float32, device="cuda")
desc = torch.zeros(k * 32, dtype=torch.uint8, device="cuda")
d32 = torch.zeros(k, dtype=torch.int32, device="cuda")
L.orbfe_fast_calculate_lut(lut.data_ptr(), 12, s)
b = grid.data_ptr()


def frame():
    L.orbfe_gaussian_blur_3x3(imgs[0].data_ptr(), w, gray.data_ptr(), w, w, h, s)
    L.orbfe_pyramid_create_levels(lv, levels, s)
    L.orbfe_detect(lv, levels, lut.data_ptr(), 13.0, b, b + 8 * k, b + 12 * k, s)
    L.orbfe_compute_fast_angle(angle.data_ptr(), b, imgs[0].data_ptr(), w, w, h, k, s)
    L.orbfe_calc_orb(angle.data_ptr(), b, desc.data_ptr(), d32.data_ptr(), imgs[0].data_ptr(), w, w, h, k, s)


for _ in range(20):
    frame()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
e0.record()
for _ in range(n):
    frame()
e1.record()
torch.cuda.synchronize()
print("stage API, %dx%d, %d level(s), K = %d: %.1f us per frame (%d keypoints)"
      % (w, h, levels, k, e0.elapsed_time(e1) / n * 1e3, int((grid[2 * k:3 * k] > 0).sum())))

# the same frame through the batch API with a batch of one (orbfe_extract = 4 launches), and the
# 8-level / 2000-feature regime of the bench
for tag, cfg in (("reference regime", dict(levels=levels, cell=32, min_arc=12)),
                 ("bench regime", dict(levels=8, cell=8, min_arc=9, max_features=2000))):
    ctx = orbfe.Context(w, h, max_batch=1, **cfg)
    rec = torch.zeros(ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    for _ in range(20):
        ctx.extract(gray.data_ptr(), w, w * h, 1, rec.data_ptr(), cnt.data_ptr(), None, s)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        ctx.extract(gray.data_ptr(), w, w * h, 1, rec.data_ptr(), cnt.data_ptr(), None, s)
    e1.record()
    torch.cuda.synchronize()
    print("batch API, batch of 1, %s: %.1f us per frame (%d keypoints)"
          % (tag, e0.elapsed_time(e1) / n * 1e3, int(cnt.item())))

# r4: one depth frame 848x480 through the stage entry of align_depth_to_other (what buildStream.cpp:385 issues per frame),
# under both output protocols (zero-init = 2 launches, the stage entry's default; literal = 3)
import ctypes as C  # noqa: E402
import os  # noqa: E402
dw, dh = 848, 480
d, o, e, scale = synth.rig("d435", dw, dh)
mk = lambda t: orbfe.Intrinsics(t[0], t[1], t[2], t[3], t[4], t[5], t[6], (C.c_float * 5)(*t[7]))
di, oi, ex = mk(d), mk(o), orbfe.Extrinsics((C.c_float * 9)(*e[0]), (C.c_float * 3)(*e[1]))
d_depth = torch.from_numpy(synth.depth_frame(dw, dh, 3).view(np.int16)).cuda()
d_al = torch.zeros((dh, dw), dtype=torch.int32, device="cuda")
for proto in ("zero", "literal"):
    os.environ["ORBFE_ALIGN_PROTOCOL"] = proto
    call = lambda: orbfe.check(orbfe.lib().orbfe_align_depth_to_other(d_al.data_ptr(), d_depth.data_ptr(), None, scale, dw, dh,
                                                                      C.byref(di), C.byref(oi), C.byref(ex), s))
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    print("align_depth_to_other, one %dx%d depth frame, %s protocol: %.1f us per frame" % (dw, dh, proto, e0.elapsed_time(e1) / n * 1e3))
os.environ.pop("ORBFE_ALIGN_PROTOCOL", None)
